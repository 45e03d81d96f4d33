#!/usr/bin/env python3
"""A/B harness for the NTT kernel families and their tunables (development tool, runs on an MI355X):
every variant = a set of ACX_NTT_* environment values read at context creation; each is checked
bit-exactly against the C oracle first (forward / inverse / coset, batched, both fields on request)
and then timed.   python tools/ntt_ab.py [--check-only] [--sizes 16 20 22] [--variants name=K=V,K=V ...]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
from oracle.c_oracle import COracle

DEFAULT_VARIANTS = [
    "tile=ACX_NTT_IMPL=tile",
    "r4=ACX_NTT_IMPL=r4",
]
KEYS = ["ACX_NTT_IMPL", "ACX_NTT_TILE_LOG", "ACX_NTT_DIRECT_TW", "ACX_NTT_DIGITS"]


def make_ctx(spec, field):
    for k in KEYS:
        os.environ.pop(k, None)
    name, _, kv = spec.partition("=")
    for item in kv.split(","):
        if item:
            k, _, v = item.partition("=")
            os.environ[k] = v.replace("+", ",")
    return name, acx.Context(field, 0)


def to_dev(ctx, arr):
    t = torch.from_numpy(arr.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.dev_from_canonical(arr.shape[0], t.data_ptr(), t.data_ptr())
    ctx.sync()
    return t


def check(ctx, orc, field, sizes, seed=7):
    bad = 0
    cases = []
    for ln in sizes:
        for batch in ((1, 2, 3) if ln <= 14 else (1,)):
            for inverse in (False, True):
                for shifted in (False, True):
                    cases.append((ln, batch, inverse, shifted))
    for i, (ln, batch, inverse, shifted) in enumerate(cases):
        n = 1 << ln
        shift = (12345 + 17 * i) if shifted else None
        x = synth.random_fr(n * batch, seed + i, ln, field)
        got = ctx.ntt(x, ln, inverse=inverse, shift=shift)
        want = np.concatenate([orc.ntt(x[b * n:(b + 1) * n], ln, inverse=inverse, shift=shift, nthreads=64) for b in range(batch)])
        if not np.array_equal(got, want):
            bad += 1
            nz = int((got != want).any(axis=1).sum())
            print(f"   MISMATCH log_n={ln} batch={batch} inverse={inverse} coset={shifted}: {nz} of {n * batch} elements differ")
    return bad, len(cases)


def time_ntt(ctx, stream, ln, batch, inverse=True, reps=30, prewarm=0.2):
    x = to_dev(ctx, synth.random_fr((1 << ln) * batch, 5, 1, ctx.field))
    fn = lambda: ctx.ntt_dev(x.data_ptr(), ln, batch, inverse=inverse)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm:
        for _ in range(4):
            fn()
        stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="*", default=DEFAULT_VARIANTS)
    ap.add_argument("--check-sizes", type=int, nargs="*", default=[10, 11, 12, 13, 14, 15, 16, 17, 18, 20])
    ap.add_argument("--sizes", type=int, nargs="*", default=[16, 20, 22])
    ap.add_argument("--batch-sizes", type=int, nargs="*", default=[20])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--fields", nargs="*", default=["bn254"])
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    for field in a.fields:
        orc = COracle(field)
        for spec in a.variants:
            name, ctx = make_ctx(spec, field)
            stream = torch.cuda.ExternalStream(ctx.stream)
            line = f"[{field}] {name:28s}"
            if not a.no_check:
                bad, total = check(ctx, orc, field, a.check_sizes)
                line += f" parity {total - bad}/{total}"
                if bad:
                    print(line + "  ** FAILED, not timed **")
                    continue
            if not a.check_only:
                for ln in a.sizes:
                    us = time_ntt(ctx, stream, ln, 1)
                    line += f" | 2^{ln}: {us:8.1f} us ({(1.5 * ln + 1) * (1 << ln) / us * 1e6:.2e} op/s)"
                for ln in a.batch_sizes:
                    us = time_ntt(ctx, stream, ln, a.batch, reps=5)
                    line += f" | 2^{ln}x{a.batch}: {us:8.1f} us/transform"
            print(line, flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
