#!/usr/bin/env python3
"""Thread scaling of the CPU restatement (oracle/acx_oracle.c) on this host: verifyAssignment of one 2^16-constraint system,
constraints/s for 1 .. os.cpu_count() threads (what bench.py's cpu_baseline should run with).  python tools/cpu_scaling.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.c_oracle import COracle
synth = importlib.import_module("arithmetic-circuits_amd.synth")
o = COracle("bn254")
n = 1 << 16
s = synth.mulgraph(n, seed=1, field="bn254")
mats, w = s.rows(), s.witness()
m = w.shape[0]
print("os.cpu_count()", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
th = 1
while th <= (os.cpu_count() or 1):
    rep = max(4, 8 * th // 4)
    o.r1cs_residuals(n, m, *mats, w, want_residuals=False, nthreads=th, repeat=2)
    t = time.time(); o.r1cs_residuals(n, m, *mats, w, want_residuals=False, nthreads=th, repeat=rep); dt = time.time() - t
    print(f"threads {th:4d} repeat {rep:4d}: {n * rep / dt:.3e} constraints/s, {n * rep / dt / th:.3e} per thread ({dt:.2f} s)")
    th *= 2
