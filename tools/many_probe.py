#!/usr/bin/env python3
"""Host-buffer throughput of `all (verifyAssignment qap) assignments`: acx_r1cs_verify_many against a loop of
acx_r1cs_verify calls (one thread) on systems of 2^10 ... 2^16 constraints.  python tools/many_probe.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")

ctx = acx.Context("bn254", 0)
for ln, count in ((10, 2000), (13, 500), (16, 100)):
    s = synth.mulgraph(1 << ln, n_in=64 if ln == 10 else 1024)
    r = s.circuit.to_r1cs(ctx)
    w = s.witness()
    W = np.repeat(w[None], count, axis=0).copy()
    W[count // 2, 7, 0] ^= np.uint64(1)
    r.verify_many(W[:4]); r.verify(w)
    t0 = time.perf_counter(); ok, _, _ = r.verify_many(W); t_many = time.perf_counter() - t0
    assert ok.sum() == count - 1
    t0 = time.perf_counter()
    for k in range(count):
        r.verify(W[k])
    t_loop = time.perf_counter() - t0
    print(f"n = 2^{ln}, m = {r.m}, {count} witnesses ({W.nbytes / 2**20:.0f} MiB): verify_many {t_many * 1e3:8.2f} ms = {count * (1 << ln) / t_many:.3e} constraints/s"
          f" | loop of verify {t_loop * 1e3:8.2f} ms = {count * (1 << ln) / t_loop:.3e} constraints/s | x{t_loop / t_many:.1f}")
