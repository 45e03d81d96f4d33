#!/usr/bin/env python3
"""Host side of acx_mgpu_* on the ONE GPU of this box (review item: "bound the host side of acx_mgpu before the first 8-GPU
run"): a device list that repeats ordinal 0 puts W shards on one device, so every API call the N-GPU job makes is made here
too, against W times the per-GPU device work.  For every call:
    issue  = entry -> everything enqueued (acx_mgpu_debug_times: the issuing threads' API calls)
    total  = entry -> results on the host
    device = total of the same call at W = 1 on the same system (one shard: what the device needs when the host is not in the way)
Prints one table per W.   python tools/mgpu_host.py [--logn 21] [--w 1 2 4 8] [--reps 10]"""
import argparse
import ctypes as C
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


def times(mg):
    out = (C.c_double * 2)()
    mg.lib.acx_mgpu_debug_times.argtypes = [C.c_void_p, C.POINTER(C.c_double * 2)]
    mg.lib.acx_mgpu_debug_times(mg._h, C.byref(out))
    return out[0], out[1]


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=21)
    ap.add_argument("--w", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--field", default="bn254")
    a = ap.parse_args()
    blocks = 1 << (a.logn - 16)
    bs = synth.BlockSystem(synth.mulgraph(1 << 16, seed=0xAC355, field=a.field), blocks)
    mats, w = bs.full_rows(), bs.witness()
    free0 = torch.cuda.mem_get_info()[0]
    print(f"system: 2^{a.logn} constraints, m = {bs.m}, witness {bs.m * 32 / 1e6:.1f} MB, transport per W below; us per call (median of {a.reps})")
    print(f"{'W':>2} {'transport':>9} | {'verify_enqueue issue':>20} {'/ device':>9} | {'qap_h_resident issue':>20} {'total':>8} | {'r1cs_verify issue':>17} {'total':>8} | "
          f"{'witness_upload':>14} | {'qap_columns first':>17} {'mem +MB':>8}")
    for W in a.w:
        mg = acx.MultiGpu(a.field, [0] * W)
        mr = mg.load(bs.n, bs.m, *mats)
        mr.upload_witness(w)
        # verify_enqueue: issue time per call; device time from the whole burst
        enq, dev = [], []
        for _ in range(a.reps):
            mg.sync()
            t0 = time.perf_counter()
            for k in range(16):
                mr.verify_enqueue(k)
            t1 = time.perf_counter()
            assert not mr.verdicts(0, 16).any()
            t2 = time.perf_counter()
            enq.append((t1 - t0) / 16 * 1e6)
            dev.append((t2 - t0) / 16 * 1e6)
        hi, ht = [], []
        ok = mr.qap_h_resident()
        for _ in range(a.reps):
            ok = mr.qap_h_resident() and ok
            i, t = times(mg)
            hi.append(i * 1e6); ht.append(t * 1e6)
        assert ok
        vi, vt = [], []
        for _ in range(max(3, a.reps // 2)):
            assert mr.verify(w, want_first=False)[0]
            i, t = times(mg)
            vi.append(i * 1e6); vt.append(t * 1e6)
        up = []
        for _ in range(max(3, a.reps // 2)):
            t0 = time.perf_counter()
            mr.upload_witness(w)
            up.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
        f0 = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter()
        mr.qap_columns(0, 1 + 1024 + (1 << (a.logn - 1)), 8)
        tc = (time.perf_counter() - t0) * 1e6
        f1 = torch.cuda.mem_get_info()[0]
        print(f"{W:>2} {mg.transport:>9} | {med(enq):>20.1f} {med(dev):>9.1f} | {med(hi):>20.1f} {med(ht):>8.1f} | {med(vi):>17.1f} {med(vt):>8.1f} | "
              f"{med(up):>14.1f} | {tc:>17.0f} {(f0 - f1) / 1e6:>8.0f}", flush=True)
        mr.close()
        mg.close()


if __name__ == "__main__":
    main()
