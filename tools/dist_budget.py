#!/usr/bin/env python3
"""Per-rank time budget of the distributed configs[3] job, measured on ONE GPU: the local work of rank 0 of a
W-rank job (W = 8 by default) is exactly reproducible without the other ranks -- the local steps of the
four-step transform take (world, rank) as plain arguments, the rank's block-cyclic rows are marshalled from the
row source, and the all-to-all moves a known number of bytes.  Prints microseconds per local step (HIP events on
libacx's stream) and the exchange volume, from which the 8-GPU time of one distributed h(x) follows:

    t = residual_dots_h + 3 * inv0t + 2 * inv1c + inv1 + 2 * (fwd0 + fwd1) + (inv0m + inv1ca) + 6 * exchange + all-reduce
(round 4: the coset factor g^i of L and R rides on the closing multiplication of their INVERSE transforms -- inv1c instead of inv1 --
so their forward transforms are plain, fwd0 instead of fwd0c; round 3's sequence, 3 * (inv0t + inv1) + 2 * (fwd0c + fwd1), is printed beside it.)
(six transforms: O(x) stays in coefficient form; the rank's rows are loaded in ascending order, so the three
inverse transforms of the dots start from the transposed ROWS block: inv0t; 1/z and -1/z ride on the stored dots, the last
transform takes the product L * R on the way in (inv0m) and adds -O/z on the way out (inv1ca): no elementwise pass is left.
Round 2's sequence -- residual_dots, pointwise, inv0 + inv1c, sub_o -- is printed beside it.)

python tools/dist_budget.py [--world 8] [--logn 24]   (run under rocprofv3 --kernel-trace for the kernel view)"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
par = importlib.import_module("arithmetic-circuits_amd.parallel")


def timed(stream, fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--logn", type=int, default=24)
    ap.add_argument("--field", default="bn254")
    a = ap.parse_args()
    W, ln, lr = a.world, a.logn, a.logn // 2
    ctx = acx.Context(a.field, 0)
    stream = torch.cuda.ExternalStream(ctx.stream)
    g = 5 if a.field == "bn254" else 7
    L = (1 << ln) // W
    x = torch.from_numpy(synth.random_fr(L, 1, 1, a.field).view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.dev_from_canonical(L, x.data_ptr(), x.data_ptr())
    y = torch.empty_like(x)
    t = {}
    x2 = x.clone()
    for name, inv, step, shift, rows_t, mul, add in (("fwd0", False, 0, None, False, 0, 0), ("fwd0c", False, 0, g, False, 0, 0), ("fwd1", False, 1, None, False, 0, 0),
                                                     ("inv0", True, 0, None, False, 0, 0), ("inv0t", True, 0, None, True, 0, 0), ("inv1", True, 1, None, False, 0, 0),
                                                     ("inv1c", True, 1, g, False, 0, 0), ("inv0m", True, 0, g, False, x2.data_ptr(), 0),
                                                     ("inv1ca", True, 1, g, False, 0, x2.data_ptr())):
        t[name] = timed(stream, lambda: ctx.ntt_dist_step_dev(x.data_ptr(), y.data_ptr(), ln, lr, W, 0, inv, step, shift, rows_t, d_mul=mul, d_add=add))
    # rank 0's rows of the 2^logn-constraint block system, block-cyclic ownership
    bs = synth.BlockSystem(synth.mulgraph(1 << 16, seed=0xAC4, field=a.field), (1 << ln) >> 16)
    rows = par.cyclic_rows(ln, lr, W, 0, ascending=True)          # ascending order: the dots come out as the transposed ROWS block
    r = acx.R1CS.load(ctx, rows.shape[0], bs.m, *bs.rows_of(rows))
    w = bs.witness()
    dw = torch.from_numpy(w.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.dev_from_canonical(w.shape[0], dw.data_ptr(), dw.data_ptr())
    dots = torch.zeros((3 * L, 4), dtype=torch.int64, device="cuda")
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    t["residual_dots"] = timed(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr(), d_dots=dots.data_ptr()))
    t["residual_dots_h"] = timed(stream, lambda: r.dots_h_dev(dw.data_ptr(), res.data_ptr(), dots.data_ptr(), ln))
    t["residual_only"] = timed(stream, lambda: r.verify_dev(dw.data_ptr(), res.data_ptr()))
    assert int(res[0]) == 0
    t["pointwise"] = timed(stream, lambda: ctx.qap_pointwise_dev(dots.data_ptr(), dots[L:].data_ptr(), None, y.data_ptr(), L, ln, g))
    t["sub_o"] = timed(stream, lambda: ctx.qap_sub_o_dev(y.data_ptr(), dots[2 * L:].data_ptr(), L, ln, g))
    xbytes = L * 32 * (W - 1) // W
    local_r2 = t["residual_dots"] + 3 * (t["inv0t"] + t["inv1"]) + 2 * (t["fwd0c"] + t["fwd1"]) + t["pointwise"] + t["inv0"] + t["inv1c"] + t["sub_o"]
    local_r3 = t["residual_dots_h"] + 3 * (t["inv0t"] + t["inv1"]) + 2 * (t["fwd0c"] + t["fwd1"]) + t["inv0m"] + t["inv1ca"]
    local = t["residual_dots_h"] + 3 * t["inv0t"] + 2 * t["inv1c"] + t["inv1"] + 2 * (t["fwd0"] + t["fwd1"]) + t["inv0m"] + t["inv1ca"]
    print(f"rank-local budget of a {W}-rank job, N = 2^{ln} = 2^{lr} x 2^{ln - lr}, {a.field} Fr, {L} elements per rank (us):")
    for k, v in t.items():
        print(f"  {k:14s} {v:10.1f}")
    print(f"  all-to-all: {xbytes / 2**20:.1f} MiB out per rank per transform ({xbytes // (W - 1) / 2**20:.2f} MiB per peer); 6 per h(x)")
    for bw in (50e9, 100e9, 153e9):
        ex = xbytes / (7 * bw) * 1e6 if W > 1 else 0.0
        print(f"  h(x) per rank: local {local:9.1f} us + 6 exchanges at {bw / 1e9:.0f} GB/s/link x 7 links {6 * ex:8.1f} us = {local + 6 * ex:9.1f} us"
              f" -> {(1 << ln) / (local + 6 * ex) * 1e6:.3e} constraints/s over {W} GPUs")
    print(f"  (round 3's sequence, coset factor on the forward transforms' load: local {local_r3:9.1f} us; round 2's, with the two elementwise kernels: {local_r2:9.1f} us)")
    dnt = t["fwd0"] + t["fwd1"]
    print(f"  one forward transform per rank: {dnt:.1f} us local (+ exchange)")


if __name__ == "__main__":
    main()
