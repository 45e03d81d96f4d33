#!/usr/bin/env python3
"""Verification rate against row length for generic CSR systems (acx_r1cs_load): rows of <= 6 entries take the SELL kernel,
7 .. 48 entries eight lanes per row, longer ones a wave per row.  python tools/rowlen_probe.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
import kbench

ctx = acx.Context("bn254", 0)
stream = torch.cuda.ExternalStream(ctx.stream)
n, m = 1 << 16, 1 << 16
rs = np.random.RandomState(1)
w = synth.random_fr(m, 9, 1)
dw = kbench.to_dev(ctx, w)
res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
for L in (3, 6, 7, 12, 13, 24, 25, 48, 49, 100, 257):
    rowptr = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
    mats = []
    for k in range(3):
        col = np.sort((rs.randint(0, m - L, size=(n, 1)) + np.arange(L)[None, :] * 1).astype(np.uint32), axis=1).reshape(-1)
        mats.append((rowptr, col, synth.random_fr(n * L, 10 + k, L)))
    r = acx.R1CS.load(ctx, n, m, *mats)
    torch.cuda.synchronize()
    us = kbench.time_stream(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr()), 20)
    print(f"rows of {L:3d} entries (x3 matrices), n = 2^16: {us:9.1f} us = {n / us * 1e6:.3e} rows/s, {3 * n * L / us * 1e6:.3e} entries/s  format {r.format()}")
    r.close()
