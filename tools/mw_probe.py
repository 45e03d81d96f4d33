#!/usr/bin/env python3
"""Probe: ONE system verified against W different witnesses in one batched launch (the usage shape of
test/Test/Circuit/Arithmetic.hs:200-209: build once, verify many), against W different systems."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
import kbench

def main():
    ln = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    ctx = acx.Context("bn254", 0)
    stream = torch.cuda.ExternalStream(ctx.stream)
    n = 1 << ln
    s = synth.mulgraph(n)
    r = s.circuit.to_r1cs(ctx)
    ws = []
    for i in range(W):
        inp = synth.random_fr(s.n_in, 77 + i, 12)
        w, _ = s.circuit.eval(inp)
        ws.append(kbench.to_dev(ctx, w))
    res = torch.zeros((W, 2), dtype=torch.int64, device="cuda"); res[:, 1] = -1
    torch.cuda.synchronize()
    b = acx.Batch(ctx, [r] * W, [w.data_ptr() for w in ws], res.data_ptr(), per_system=True)
    us = kbench.time_stream(stream, b.verify_dev, 50)
    ctx.sync()
    assert int(res[:, 0].sum()) == 0
    print(f"same system 2^{ln} x {W} witnesses: {us:.2f} us/launch = {us / W:.2f} us per witness, {n * W / us * 1e6:.3e} constraint-checks/s")
    # corrupt one witness: exactly that member must fail
    bad = ws[W // 2].clone(); bad[1000, 0] ^= 1
    b2 = acx.Batch(ctx, [r] * W, [(bad if i == W // 2 else ws[i]).data_ptr() for i in range(W)], res.data_ptr(), per_system=True)
    res[:, 0] = 0; res[:, 1] = -1; torch.cuda.synchronize()
    b2.verify_dev(); ctx.sync()
    nz = [i for i in range(W) if int(res[i, 0]) != 0]
    print("corrupted member detected:", nz)
    if ln <= 17:
        systems = []
        for c in range(W):
            sc = synth.mulgraph(n, seed=0xAC355 + c)
            systems.append((sc.circuit.to_r1cs(ctx), kbench.to_dev(ctx, sc.witness())))
        res[:, 0] = 0; res[:, 1] = -1; torch.cuda.synchronize()
        b3 = acx.Batch(ctx, [x[0] for x in systems], [x[1].data_ptr() for x in systems], res.data_ptr(), per_system=True)
        us3 = kbench.time_stream(stream, b3.verify_dev, 50)
        print(f"{W} different systems 2^{ln}: {us3:.2f} us/launch = {us3 / W:.2f} us per system")

main()
