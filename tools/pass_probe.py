#!/usr/bin/env python3
"""One-pass transforms (one k_ntt_r4 launch each: 2^LP points per thread group) on batches that fit the Infinity Cache and on
batches that stream from HBM: us per launch, ns per element, and per butterfly product (sum over stages of (1 - 2^-s)/2 per
element).  Separates the arithmetic of a pass from its exposed load / store phases.  python tools/pass_probe.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
import kbench
ctx = acx.Context("bn254", 0)
stream = torch.cuda.ExternalStream(ctx.stream)
for lp in (8, 10, 12):
    prods = sum((1 - 2.0 ** -s) / 2 for s in range(lp))
    for total_log in (18, 20, 24):
        batch = 1 << (total_log - lp)
        x = kbench.to_dev(ctx, synth.random_fr(1 << total_log, 5, 1))
        us = kbench.time_stream(stream, lambda: ctx.ntt_dev(x.data_ptr(), lp, batch, inverse=False), 30, warmup=10)
        n = 1 << total_log
        print(f"LP={lp:2d} batch=2^{total_log - lp:2d} ({n * 32 >> 20:4d} MB): {us:9.2f} us  {us * 1e3 / n:6.3f} ns/element  "
              f"{us * 1e3 / n / prods:6.3f} ns per element-product ({prods:.2f} products/element)")
        del x
