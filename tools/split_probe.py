#!/usr/bin/env python3
"""How much of a verification goes to the rows the SELL layout does not hold (the 2^j row of every Split gate, 257 entries):
a circuit in the reference's own gate mix (Mul : Equal : Split = 50 : 10 : 1, test/Test/Circuit/Arithmetic.hs:136) with 256-bit
splits.  python tools/split_probe.py [gates]   (run under tools/prof.py for the per-kernel split)"""
import importlib, os, random, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
acx = importlib.import_module("arithmetic-circuits_amd")
from tests import helpers as H
import kbench

size = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
ctx = acx.Context("bn254", 0)
stream = torch.cuda.ExternalStream(ctx.stream)
rnd = random.Random(5)
gates = H.arb_arith_circuit(rnd, ctx.p, 6, size, dist=(50, 10, 1), split_bits=256)
circ = H.to_acx_circuit(acx, gates).marshal("bn254")
r = circ.to_r1cs(ctx)
w, _ = circ.eval(acx.ints_to_fr([rnd.randrange(ctx.p) for _ in range(6)]))
assert r.verify(w)[0]
dw = kbench.to_dev(ctx, w)
res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
us = kbench.time_stream(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr()), 200)
nnz = sum(int(m[0][-1]) for m in circ.rows())
print(f"{len(gates)} gates -> n = {r.n} rows, nnz = {nnz}, format {r.format()}: verify_dev {us:.1f} us")
import time
inp = acx.ints_to_fr([rnd.randrange(ctx.p) for _ in range(6)])
r.eval_witness(inp)
t0 = time.perf_counter()
for _ in range(5):
    wg, _ = r.eval_witness(inp)
t_gpu = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5):
    wh, _ = circ.eval(inp)
t_host = (time.perf_counter() - t0) / 5
assert np.array_equal(wg, wh)
print(f"generateAssignment: GPU (acx_r1cs_eval, incl. D2H of the witness) {t_gpu * 1e3:.2f} ms, host (acx_circuit_eval) {t_host * 1e3:.2f} ms; levels: see ACX trace")
h, ok = r.qap_h(wh)
t0 = time.perf_counter()
for _ in range(5):
    r.qap_h(wh)
print(f"qap_h (host buffers): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms, ok={ok}")
