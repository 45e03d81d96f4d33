#!/usr/bin/env python3
"""Latency of the entry points on SMALL systems (the sizes the reference's own tests and Example.hs use):
device-resident h(x), host-buffer h(x) and host-buffer verify at n = 2^4 ... 2^14.  python tools/small_probe.py"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
import kbench

ctx = acx.Context("bn254", 0)
stream = torch.cuda.ExternalStream(ctx.stream)
for ln in (4, 8, 10, 12, 14, 16):
    n = 1 << ln
    s = synth.mulgraph(n, n_in=16 if ln < 10 else 64)
    r = s.circuit.to_r1cs(ctx)
    w = s.witness()
    dw = kbench.to_dev(ctx, w)
    dh = torch.zeros((n + 1, 4), dtype=torch.int64, device="cuda")
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    us_dev = kbench.time_stream(stream, lambda: r.qap_h_dev(dw.data_ptr(), dh.data_ptr(), res.data_ptr()), 200)
    us_ver = kbench.time_stream(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr()), 200)
    def wall(fn, reps=200):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e6
    print(f"n=2^{ln:2d} m={r.m:6d}: device-resident h(x) {us_dev:7.1f} us, verify_dev {us_ver:6.1f} us | host-buffer qap_h {wall(lambda: r.qap_h(w)):7.1f} us, "
          f"verify {wall(lambda: r.verify(w)):6.1f} us")
