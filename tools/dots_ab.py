#!/usr/bin/env python3
"""Residual launch with and without the three dot-product vectors (the first step of h(x)), one 2^logn-row mulgraph system:
python tools/dots_ab.py [--logn 20]   (ACX_LIB selects another build of the library)"""
import argparse, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")


def timed(stream, fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=20)
a = ap.parse_args()
ctx = acx.Context("bn254", 0)
stream = torch.cuda.ExternalStream(ctx.stream)
n = 1 << a.logn
s = synth.mulgraph(n)
r = s.circuit.to_r1cs(ctx)
w = s.witness()
dw = torch.from_numpy(w.view(np.int64).copy()).cuda()
torch.cuda.synchronize()
ctx.dev_from_canonical(w.shape[0], dw.data_ptr(), dw.data_ptr())
dots = torch.zeros((3 * n, 4), dtype=torch.int64, device="cuda")
res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
out = []
for rnd in range(3):
    t0 = timed(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr()))
    t1 = timed(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr(), d_dots=dots.data_ptr()))
    out.append(f"{t0:.1f}/{t1:.1f}")
print(os.environ.get("ACX_LIB", "default"), f"2^{a.logn} rows: residual only / with dots (us):", "  ".join(out))
