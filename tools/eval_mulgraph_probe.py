#!/usr/bin/env python3
"""acx_r1cs_eval on mulgraph(2^20): launches per evaluation and their duration (run under tools/prof.py)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
ctx = acx.Context("bn254", 0)
s = synth.mulgraph(1 << 20)
r = s.circuit.to_r1cs(ctx)
r.eval_witness(s.inputs, download=False)
t0 = time.perf_counter()
for _ in range(5):
    r.eval_witness(s.inputs, download=False)
ctx.sync()
print(f"eval 2^20 gates: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
