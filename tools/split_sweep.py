#!/usr/bin/env python3
"""verify_dev latency of ONE device-resident system at n = 2^10 ... 2^19, for the wave-specialised kernel
(k_r1cs_sell_split, two waves per slice) against the one-wave-per-slice kernel: run twice, ACX_SELL_SPLIT=1 and =0;
LNS=10,16,21 selects the sizes.  python tools/split_sweep.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
import kbench

ctx = acx.Context("bn254", 0)
stream = torch.cuda.ExternalStream(ctx.stream)
out = []
for ln in [int(x) for x in os.environ.get("LNS", "10,11,12,13,14,15,16,17,18,19").split(",")]:
    n = 1 << ln
    s = synth.mulgraph(n, n_in=64 if ln < 12 else 1024)
    r = s.circuit.to_r1cs(ctx)
    dw = kbench.to_dev(ctx, s.witness())
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    us = kbench.time_stream(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr()), 300)
    assert int(res[0]) == 0
    out.append(f"2^{ln}: {us:6.1f}")
print("ACX_SELL_SPLIT=%s  verify_dev us:  " % os.environ.get("ACX_SELL_SPLIT", "1") + "  ".join(out))
