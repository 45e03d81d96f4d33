#!/usr/bin/env python3
"""Kernel micro-benchmarks on one GPU (development tool; bench.py is the judged harness).
usage: python tools/kbench.py [r1cs|ntt|all] [--logn 16 20] [--reps 20]"""
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


def alg_bytes(mats, n):
    nnz = sum(int(m[1].shape[0]) for m in mats)
    m_ref = np.unique(np.concatenate([m[1] for m in mats])).shape[0]
    return 36 * nnz + 12 * (n + 1) + 32 * m_ref + 8, nnz, m_ref


PREWARM_S = float(os.environ.get("ACX_KBENCH_PREWARM", "0.25"))


def time_stream(stream, fn, reps, warmup=3):
    """Average launch time in us.  From idle the GPU needs ~35 ms of work to reach its sustained clock
    (tools/microbench/ramp.py) and measures ~15 % slower until then, so the same call is repeated for
    PREWARM_S seconds first (ACX_KBENCH_PREWARM=0 gives the from-idle figure older notes quote)."""
    import time as _time
    t0 = _time.perf_counter()
    while _time.perf_counter() - t0 < PREWARM_S:
        for _ in range(8):
            fn()
        stream.synchronize()
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps   # us


def to_dev(ctx, arr):
    t = torch.from_numpy(arr.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.dev_from_canonical(arr.shape[0], t.data_ptr(), t.data_ptr())
    ctx.sync()
    return t


def bench_r1cs(ctx, stream, log_n, reps, copies):
    n = 1 << log_n
    systems = []
    for c in range(copies):
        s = synth.mulgraph(n, seed=0xAC355 + c, n_in=N_IN, window=WINDOW, coeff=COEFF)
        mats = s.rows()
        r = s.circuit.to_r1cs(ctx)
        w = to_dev(ctx, s.witness())
        systems.append((r, w, mats))
    b, nnz, m_ref = alg_bytes(systems[0][2], n)
    res = torch.zeros(2, dtype=torch.int64, device="cuda")
    res[1] = -1
    ctx.sync()
    i = [0]

    def fn():
        r, w, _ = systems[i[0] % copies]
        i[0] += 1
        r.verify_dev(w.data_ptr(), res.data_ptr())

    us = time_stream(stream, fn, reps * copies)
    ctx.sync()
    assert int(res[0]) == 0 or os.environ.get("ACX_ABLATION"), "witness must verify"
    print(f"r1cs[{COEFF}, format {systems[0][0].format()}] n=2^{log_n} copies={copies}: {us:9.2f} us/launch  {n / us * 1e6:.3e} constraints/s  "
          f"alg {b / 1e6:.2f} MB -> {b / us * 1e-3:.1f} GB/s ({b / us * 1e-3 / 8000 * 100:.1f}% of 8 TB/s)  nnz={nnz} m_ref={m_ref}")


def bench_ntt(ctx, stream, log_n, reps, batch=1):
    n = 1 << log_n
    x = to_dev(ctx, synth.random_fr(n * batch, 5, 1))
    us = time_stream(stream, lambda: ctx.ntt_dev(x.data_ptr(), log_n, batch, inverse=True), reps)
    ops = (1.5 * n * log_n + n) * batch
    print(f"intt N=2^{log_n} batch={batch}: {us:9.2f} us  {ops / us * 1e6:.3e} field-ops/s  "
          f"alg(128N) {128 * n * batch / us * 1e-3:.1f} GB/s")


def bench_h(ctx, log_n):
    """verificationWitness end to end (host witness in, host h out): residual dots + 7 NTT-sized passes."""
    n = 1 << log_n
    s = synth.mulgraph(n)
    r = s.circuit.to_r1cs(ctx)
    w = s.witness()
    r.qap_h(w)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        h, ok = r.qap_h(w)
    dt = (time.perf_counter() - t0) / reps
    assert ok
    print(f"qap_h n=2^{log_n}: {dt * 1e3:9.2f} ms per call (host in/out, includes H2D of w and D2H of h)")


def bench_cols(ctx, stream, log_n, wires=64):
    """createPolynomialsFFT throughput (src/QAP.hs:512-525): per-wire interpolation of `wires` columns of A at a time --
    device-resident (acx_qap_columns_dev: scatter, batched iNTT, stripped lengths) and through host buffers
    (acx_qap_columns: the same plus the D2H of every coefficient, double-buffered against the next batch's kernels)."""
    n = 1 << log_n
    s = synth.mulgraph(n)
    r = s.circuit.to_r1cs(ctx)
    out = torch.empty((wires * n, 4), dtype=torch.int64, device="cuda")
    lens = torch.zeros(wires, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    # wires 1 .. 1024 are the circuit's inputs (hundreds of appearances each: batched inverse NTT); the intermediate wires
    # behind them appear once in C and once or twice in A / B (direct interpolation, k_col_direct)
    mid0 = 1 + N_IN + n // 2
    for what, w0, mat in (("input wires, A", 1, 0), ("intermediate wires, A", mid0, 0), ("intermediate wires, C", mid0, 2)):
        us = time_stream(stream, lambda: r.qap_columns_dev(mat, w0, wires, out.data_ptr(), lens.data_ptr()), 5)
        print(f"qap_columns_dev N=2^{log_n}, {wires} {what} per call: {us:9.1f} us = {us / wires:7.1f} us per column, {wires / us * 1e6:.0f} columns/s")
    total = 4 * wires
    r.qap_columns(0, 1, wires)
    t0 = time.perf_counter()
    r.qap_columns(0, 1, total)
    dt = time.perf_counter() - t0
    print(f"qap_columns (host buffers) N=2^{log_n}, {total} wires: {dt * 1e3:9.1f} ms = {dt / total * 1e6:7.1f} us per column, "
          f"{total * n * 32 / dt / 1e9:.1f} GB/s of coefficients to the host")


def bench_cols_by_entries(ctx, stream, log_n, wires=64, kmax=14):
    """createPolynomialsFFT of columns with EXACTLY k entries, k = 0 .. kmax (a synthetic A matrix: column w of block k holds k
    random entries on random rows): what a column costs by its entry count -- k_col_direct (k <= 4), k_col_direct_mid (5 .. 8),
    scatter + batched inverse transform beyond.  Every block is checked against the C oracle on its first column."""
    from oracle.c_oracle import COracle
    orc = COracle("bn254")
    n, m = 1 << log_n, 1 + (kmax + 1) * wires
    rng = np.random.RandomState(7)
    rows, cols = [], []
    for k in range(kmax + 1):
        for w in range(wires):
            rr = rng.choice(n, size=k, replace=False)
            rows.append(rr)
            cols.append(np.full(k, 1 + k * wires + w))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowptr = np.zeros(n + 1, dtype=np.uint32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.uint32)
    A = (rowptr, cols.astype(np.uint32), synth.random_fr(cols.shape[0], 11, 1))
    empty = (np.zeros(n + 1, dtype=np.uint32), np.zeros(0, dtype=np.uint32), np.zeros((0, 4), dtype=np.uint64))
    r = acx.R1CS.load(ctx, n, m, A, empty, empty)
    out = torch.empty((wires * n, 4), dtype=torch.int64, device="cuda")
    lens = torch.zeros(wires, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for k in range(kmax + 1):
        w0 = 1 + k * wires
        r.qap_columns_dev(0, w0, wires, out.data_ptr(), lens.data_ptr())
        ctx.sync()
        got = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.dev_to_canonical(n, out.data_ptr(), got.data_ptr())
        ctx.sync()
        want = orc.qap_columns(n, log_n, A, w0, 1, nthreads=16)[0]
        assert np.array_equal(got.cpu().numpy().view(np.uint64), want), f"k = {k}: column differs from the oracle's"
        us = time_stream(stream, lambda: r.qap_columns_dev(0, w0, wires, out.data_ptr(), lens.data_ptr()), 5)
        print(f"qap_columns_dev N=2^{log_n}, {wires} columns of exactly {k:2d} entries: {us / wires:7.1f} us per column ({32 * n / (us / wires) * 1e-6:6.2f} TB/s of coefficients)")


N_IN, WINDOW, COEFF = 1024, 4096, "random"


def main():
    global N_IN, WINDOW, COEFF
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-in", type=int, default=1024)
    ap.add_argument("--window", type=int, default=4096)
    ap.add_argument("--coeff", default="random", choices=["random", "small"])
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--logn", type=int, nargs="*", default=[16, 20])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--copies", type=int, default=4)
    a = ap.parse_args()
    N_IN, WINDOW, COEFF = a.n_in, a.window, a.coeff
    ctx = acx.Context("bn254", 0)
    stream = torch.cuda.ExternalStream(ctx.stream)
    for ln in a.logn:
        if a.what in ("r1cs", "all"):
            bench_r1cs(ctx, stream, ln, a.reps, a.copies if ln <= 18 else 1)
        if a.what in ("ntt", "all"):
            bench_ntt(ctx, stream, ln, a.reps)
        if a.what == "nttbatch":
            bench_ntt(ctx, stream, ln, a.reps, batch=64 if ln <= 20 else 8)
        if a.what == "cols":
            bench_cols(ctx, stream, ln)
        if a.what == "colsk":
            bench_cols_by_entries(ctx, stream, ln)
        if a.what == "h":
            bench_h(ctx, ln)


if __name__ == "__main__":
    main()
