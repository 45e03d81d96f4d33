#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X (contract: see the round brief).

Workload (BASELINE.json configs[1]): verifyAssignment over seeded random R1CS of
2^16 constraints each (mulgraph k=2, n_in=1024, window=4096, BN254 Fr).  One STEP = one batched
launch that checks `--copies` (default 32) independent 2^16-constraint systems against their
device-resident witnesses: 2^21 constraints per GPU per step, ~520 MB of constraint data per GPU
(> the 256 MiB Infinity Cache, so the stream comes from HBM).  With N GPUs every rank holds its own
32 systems (rows sharded with no data-path collective; N=8 is the 2^24-constraint job of
configs[3]); the violated-row counts are combined by ONE RCCL all-reduce per 8 steps, issued
asynchronously on a double-buffered ring of result slots.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def algorithmic_bytes(mats, n):
    """36*nnz + 12*(n+1) + 32*m_ref + 8 per verification of one system."""
    nnz = sum(int(m[1].shape[0]) for m in mats)
    m_ref = int(np.unique(np.concatenate([m[1] for m in mats])).shape[0])
    return 36 * nnz + 12 * (n + 1) + 32 * m_ref + 8, nnz, m_ref


def to_dev(ctx, arr):
    t = torch.from_numpy(arr.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.dev_from_canonical(arr.shape[0], t.data_ptr(), t.data_ptr())
    return t


def effective_cpus():
    """CPUs this process may really use: the affinity mask, cut by the cgroup's CPU quota (cpu.max = "quota period") when there is
    one -- on the GPU boxes 256 hardware threads are visible under a quota of 16 CPUs, and 256 busy threads are then throttled
    below what 16 deliver (tools/cpu_scaling.py)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def measure_counter_live(a, counter, timeout_s=120):
    """One hardware counter of the headline kernel, per launch, measured NOW: this script re-runs its batched launches (only
    those: --only-steps) as a child under `rocprofv3 --pmc <counter>` -- a counter pass of its own, no tracing beside it, as
    MI355X_MICROARCH.md prescribes -- and reads the per-dispatch values from the profiler's database.  Returns (average value
    per launch, launches averaged) or None when the profiler is missing or fails."""
    import glob, shutil, sqlite3, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = tempfile.mkdtemp(prefix="acx_pmc_", dir="/tmp")
    cmd = [sys.executable, os.path.abspath(__file__), "--only-steps", "--no-cpu", "--no-ntt", "--no-pmc", "--sustain", "0", "--steps", "20",
           "--warmup", "2", "--prewarm", "0.05", "--field", a.field, "--copies", str(a.copies), "--logn", str(a.logn)]
    try:
        subprocess.call(["rocprofv3", "--pmc", counter, "-d", out, "-o", "pass", "--"] + cmd, stdout=subprocess.DEVNULL,
                        stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=timeout_s)
        best = None
        for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
            cur = sqlite3.connect(db).cursor()
            for name, v, cnt in cur.execute("select kernel_name, avg(value), count(*) from counters_collection "
                                            "where counter_name = ? group by kernel_name", (counter,)):
                if "k_r1cs_sell_split" in name and (best is None or cnt > best[1]):
                    best = (float(v), cnt)
        return best
    except Exception:                                  # a profiler problem must not cost the line
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_traffic_live(a):
    """HBM bytes the headline kernel fetches per launch: FETCH_SIZE counts KiB and, on gfx950, half of what wide coalesced
    reads move (calibrated on a copy kernel, tools/prof.py): bytes = value * 1024 * 2."""
    r = measure_counter_live(a, "FETCH_SIZE")
    return None if r is None else (r[0] * 1024.0 * 2.0, r[1])


def cpu_baseline(sample, field="bn254", budget_s=12.0):
    """The CPU restatement (oracle/acx_oracle.c, "port") timed on this host's cores on a bounded
    sample of the same workload: one 2^16-constraint system verified `repeat` times per call
    (threads persist across the repeats of a call), calls repeated for ~budget_s seconds."""
    from oracle.c_oracle import COracle
    orc = COracle(field)
    mats, w, n, m = sample
    threads = effective_cpus()
    repeat = 64
    orc.r1cs_residuals(n, m, *mats, w, want_residuals=False, nthreads=threads, repeat=2)   # warm-up
    calls, t0 = 0, time.perf_counter()
    while True:
        _, nbad, _ = orc.r1cs_residuals(n, m, *mats, w, want_residuals=False, nthreads=threads, repeat=repeat)
        assert nbad == 0
        calls += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    out = {"value": n * repeat * calls / dt, "unit": "constraints/s", "cores": threads, "kind": "port",
           "sample": f"{calls * repeat} x verifyAssignment of one 2^{n.bit_length() - 1}-constraint system "
                     f"(oracle/acx_oracle.c, {threads} pthreads = the CPUs this process may use: {os.cpu_count()} hardware threads visible, "
                     f"cgroup quota applied; {dt:.1f} s)"}
    out["reference_algorithm"] = cpu_reference_algorithm(orc, field)
    return out


def cpu_reference_algorithm(orc, field="bn254", log_n=10):
    """BASELINE.md section 2 / SURVEY.md 8(d), configs[0] (the reference's CPU-runnable case, 2^10 gates): what the Haskell
    ALGORITHM costs -- createPolynomialsFFT into dense per-wire polynomials (src/QAP.hs:512-525: 3 m interpolations) and
    verifyAssignment in the polynomial domain (src/QAP.hs:276-327: m scalar x polynomial sums per matrix, dense product,
    long division by x^N - 1), single threaded like the reference.  A C restatement (oracle/acx_oracle.c orc_qap_columns,
    orc_ref_verify), NOT GHC: boxed Naturals and lazy lists cost more than this."""
    n = 1 << log_n
    s = synth.mulgraph(n, n_in=64, seed=0xAC1, field=field)
    mats, w = s.rows(), s.witness()
    m = w.shape[0]
    t0 = time.perf_counter()
    cols = np.stack([orc.qap_columns(n, log_n, mats[k], 0, m, nthreads=1) for k in range(3)])
    t_create = time.perf_counter() - t0
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 1.0:
        q, ok = orc.ref_verify(m, log_n, cols, w)
        assert ok
        reps += 1
    t_verify = (time.perf_counter() - t0) / reps
    h, ok2 = orc.qap_h(n, m, log_n, *mats, w)
    assert ok2 and np.array_equal(h[:n], q) and not h[n:].any(), "polynomial-domain quotient differs from the evaluation-domain h(x)"
    return {"config": f"configs[0]: 2^{log_n}-gate mulgraph circuit, m = {m} wires ({field} Fr)", "cores": 1,
            "create_qap_s": t_create, "verify_s": t_verify, "constraints_per_s": n / t_verify,
            "note": "C restatement of the reference's polynomial-domain algorithm (3 m dense interpolations; m scalar x polynomial "
                    "sums per matrix, dense product, long division), not GHC; quotient checked against the evaluation-domain h(x)"}


def _timed(stream, fn, reps, prewarm):
    """Average us per call of fn over `reps` back-to-back calls (HIP events on libacx's stream), after the
    clock-ramp pre-run."""
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < prewarm:
        for _ in range(8):
            fn()
        stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def _from_dev(ctx, t, count):
    out = torch.empty_like(t)
    ctx.dev_to_canonical(count, t.data_ptr(), out.data_ptr())
    ctx.sync()
    return out.cpu().numpy().view(np.uint64).reshape(-1, 4)[:count]


def _valu_issue(key, count_field, us, workload):
    """VALU-issue fraction of a kernel: SQ_INSTS_VALU (wave instructions per launch, from the committed rocprofv3 --pmc
    pass of the same workload) / 1024 SIMDs / the measured multiplier issue rate, over the live launch time."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
        if tr[key]["workload"] != workload:
            return None
        rate = tr["valu_rate"]
        bound_us = tr[key][count_field] / rate["simds"] / rate["wave_insts_per_s_per_simd"] * 1e6
        return {"wave_insts": tr[key][count_field], "issue_bound_us": bound_us, "frac": bound_us / us,
                "source": "profiles/r03_traffic.json, profiles/r01_valu_rates.txt", "measured_in_run": False}
    except (OSError, KeyError, ValueError):
        return None


def _hbm(alg_bytes, us, **extra):
    d = {"bound": "hbm", "achieved": alg_bytes / us * 1e-3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": alg_bytes / us * 1e-3 / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes}
    d.update(extra)
    return d


def bench_ntt(ctx, stream, field="bn254", log_n=20, reps=40, prewarm=0.25, batch=64):
    """Secondary metrics of configs[2]: one 2^20-point transform (= FFT.interpolate of one QAP
    column) and a batch of 64 of them.  Parity gate first: the inverse transform of a fixed random vector is
    compared with the C oracle's, element by element.  The timed loop alternates inverse and forward on that
    vector, so every launch works on the same data (inverse then forward is the identity)."""
    from oracle.c_oracle import COracle
    orc = COracle(field)
    n = 1 << log_n
    x_host = synth.random_fr(n, 5, 1, field)
    x = to_dev(ctx, x_host)
    ctx.ntt_dev(x.data_ptr(), log_n, 1, inverse=True)
    parity = bool(np.array_equal(_from_dev(ctx, x, n), orc.ntt(x_host, log_n, inverse=True, nthreads=effective_cpus())))
    ctx.ntt_dev(x.data_ptr(), log_n, 1, inverse=False)
    parity = parity and bool(np.array_equal(_from_dev(ctx, x, n), x_host))
    flip = [False]

    def one():
        flip[0] = not flip[0]
        ctx.ntt_dev(x.data_ptr(), log_n, 1, inverse=flip[0])

    us = _timed(stream, one, reps, prewarm)
    ops = 1.5 * n * log_n + n / 2          # butterflies * 3, + the 1/N scaling of the inverse half of the launches
    xb = to_dev(ctx, synth.random_fr(n * batch, 6, 1, field))
    flipb = [False]

    def many():
        flipb[0] = not flipb[0]
        ctx.ntt_dev(xb.data_ptr(), log_n, batch, inverse=flipb[0])

    us_b = _timed(stream, many, 6, prewarm) / batch
    alg = 128 * n      # two reads + two writes of every 32-byte element (two-pass four-step)
    note = ("VALU-bound: ~10 Montgomery products (171 v_mad_u64_u32 each) per element; the HBM fraction is what "
            "SURVEY.md 8(d) asks to be quoted, valu_issue is the fraction of the integer-issue bound (profiles/r02_ntt.txt)")
    return {"workload": f"NTT N=2^{log_n} ({field} Fr), alternating inverse/forward on a fixed vector, acx::k_ntt_r4",
            "parity_vs_oracle": parity, "us": us, "field_ops_per_s": ops / us * 1e6,
            "roofline": _hbm(alg, us, note=note, valu_issue=_valu_issue("acx::k_ntt_r4", "valu_wave_insts_per_transform", us,
                                                                        {"field": field, "logn": log_n})),
            "batch": {"transforms": batch, "us_per_transform": us_b, "field_ops_per_s": ops / us_b * 1e6,
                      "roofline": _hbm(alg, us_b)}}


def bench_qap_h(ctx, stream, field="bn254", log_n=20, reps=10, prewarm=0.25):
    """configs[2]'s third C3 metric: the h(x) pipeline of verificationWitness (src/QAP.hs:309-327) on a
    2^20-constraint mulgraph system, device resident (witness in, N+1 coefficients out): residual dots,
    3 iNTT, 2 coset NTT (L, R), pointwise, coset iNTT, minus O / z in coefficient form -- six transforms.
    Parity gate: every coefficient against the C oracle (which runs the textbook seven)."""
    from oracle.c_oracle import COracle
    orc = COracle(field)
    n = 1 << log_n
    s = synth.mulgraph(n, seed=0xAC3, field=field)
    mats, w = s.rows(), s.witness()
    r = s.circuit.to_r1cs(ctx)
    dw = to_dev(ctx, w)
    dh = torch.zeros((n + 1, 4), dtype=torch.int64, device="cuda")
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    r.qap_h_dev(dw.data_ptr(), dh.data_ptr(), res.data_ptr())
    ctx.sync()
    got = _from_dev(ctx, dh, n + 1)
    want, ok = orc.qap_h(n, r.m, log_n, *mats, w, nthreads=effective_cpus())
    parity = bool(ok and int(res[0]) == 0 and np.array_equal(got, want))
    us = _timed(stream, lambda: r.qap_h_dev(dw.data_ptr(), dh.data_ptr(), res.data_ptr()), reps, prewarm)
    b_r1cs, nnz, _ = algorithmic_bytes(mats, n)
    alg = 7 * 128 * n + (b_r1cs + 3 * 32 * n) + 5 * 32 * n     # SURVEY.md 8(d)'s definition of the job: 7 NTTs + residual-style dots (written) + pointwise
    ops = 6 * (1.5 * n * log_n) + 4 * n + nnz + 2 * n           # butterflies of the SIX transforms actually run, scalings, dot-product MACs, pointwise + O / z
    return {"workload": f"verificationWitness h(x), 2^{log_n}-constraint mulgraph ({field} Fr), device resident: residual dots (1/z, -1/z riding on them) + 6 NTTs (O stays in coefficient form; the last one takes L*R on load and adds -O/z on store)", "transforms": 6,
            "parity_vs_oracle": parity, "us": us, "field_ops_per_s": ops / us * 1e6, "roofline": _hbm(alg, us)}


def bench_small_coeff(ctx, stream, field="bn254", copies=32, log_n=16, reps=50, prewarm=0.25):
    """The same batched verification on systems of a compiled program's shape (src/Circuit/Expr.hs:256-305: coefficients
    +-c with c <= 2^16 instead of uniform field elements): libacx stores such matrices as {coefficient, column} pairs and
    the dot products need no 256-bit products.  Parity gate: the first system's residual vector under a corrupted witness
    against the C oracle, every satisfying witness accepted."""
    from oracle.c_oracle import COracle
    orc = COracle(field)
    n = 1 << log_n
    systems, witnesses, alg, fmt = [], [], 0, None
    for c in range(copies):
        s = synth.mulgraph(n, seed=0x5AC355 + c, field=field, coeff="small")
        r = s.circuit.to_r1cs(ctx)
        w = s.witness()
        if c == 0:
            mats = s.rows()
            w2 = w.copy()
            w2[[3, 1500, r.m - 2], 0] ^= np.uint64(1)
            want, nbad, first = orc.r1cs_residuals(n, r.m, *mats, w2, nthreads=effective_cpus())
            parity = bool(np.array_equal(r.residuals(w2), want) and r.verify(w2) == (False, nbad, first) and nbad > 0)
            fmt = r.format()
            alg = algorithmic_bytes(mats, n)[0] * copies
        systems.append(r)
        witnesses.append(to_dev(ctx, w))
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    batch = acx.Batch(ctx, systems, [w.data_ptr() for w in witnesses], res.data_ptr())
    us = _timed(stream, batch.verify_dev, reps, prewarm)
    ctx.sync()
    parity = parity and int(res[0]) == 0
    return {"workload": f"r1cs_verify: {copies} independent 2^{log_n}-constraint mulgraph systems per launch with coefficients +-c, c <= 2^16 "
                        f"(compiled-program shape; {field} Fr)", "parity_vs_oracle": parity, "us_per_launch": us,
            "constraints_per_s": copies * n / us * 1e6, "small_coefficient_matrices": fmt[0], "unit_c": fmt[1],
            "algorithmic_bytes_as_8d": alg, "note": "8 bytes per entry are streamed instead of 40; not an HBM-roofline figure"}


def bench_distributed(ctx, a, world, rank, dist):
    """configs[3] beside the headline (N > 1, or --force-dist on one GPU): the distributed four-step NTT at
    N = 2^24 (one all-to-all per transform) and the distributed h(x) pipeline on a 2^24-constraint block system
    (256 x 2^16 mulgraph blocks, rows marshalled per rank in block-cyclic ownership; 6 all-to-alls + 1 all-reduce: O(x) stays in coefficient form).
    Times are max over ranks.  Parity gates: transform round trip at 2^24 plus a full oracle comparison of the same
    code path at 2^16; the pipeline must accept the satisfying witness and reject a corrupted one."""
    par = importlib.import_module("arithmetic-circuits_amd.parallel")
    from oracle.c_oracle import COracle
    orc = COracle(a.field)
    ops = par.HipOps(ctx)
    coll = None                                      # the product's collectives: torch.distributed over RCCL
    if a.backend == "gloo":                          # test mode (several ranks sharing one GPU): host-staged exchange
        from tests.helpers_dist import HostStagedCollectives
        coll = HostStagedCollectives()
    dev = torch.device("cuda", torch.cuda.current_device())

    def tmax(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def wall(fn, reps):
        fn()
        torch.cuda.synchronize(); ctx.sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(); ctx.sync()
        return tmax((time.perf_counter() - t0) / reps)

    out = {}
    # --- parity of the distributed transform against the oracle at 2^16 (same kernels, same exchange)
    ln = 16
    force = world == 1 and a.backend == "nccl"      # one rank: still issue the RCCL all-to-all (its stream ordering is what is tested)
    d16 = par.DistributedNTT(ln, ops, log_r=8, force_collective=force, collectives=coll)
    x16 = synth.random_fr(1 << ln, 31, 1, a.field)
    got = _from_dev(ctx, d16.forward(to_dev(ctx, x16[d16.cols_indices()])), d16.local)
    parity = bool(np.array_equal(got, orc.ntt(x16, ln, nthreads=effective_cpus())[d16.rows_indices()]))
    # --- 2^24 transform
    ln, lr = a.dist_logn, a.dist_logn // 2
    d = par.DistributedNTT(ln, ops, log_r=lr, force_collective=force, collectives=coll)
    x = to_dev(ctx, synth.random_fr(d.local, 32 + rank, 1, a.field))
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    d.forward(x, out=y)
    d.inverse(y, out=z)
    torch.cuda.synchronize(); ctx.sync()
    parity = parity and bool(np.array_equal(_from_dev(ctx, z, d.local), _from_dev(ctx, x, d.local)))
    flip = [False]

    def one():
        flip[0] = not flip[0]
        if flip[0]:
            d.forward(x, out=y)
        else:
            d.inverse(y, out=z)

    sec = wall(one, 10)
    n = 1 << ln
    out["dist_ntt"] = {"workload": f"distributed four-step NTT N=2^{ln} = 2^{lr} x 2^{ln - lr} over {world} rank(s), alternating forward/inverse",
                       "parity_vs_oracle": parity, "us": sec * 1e6, "field_ops_per_s": 1.5 * n * ln / sec,
                       "all_to_all_bytes_per_rank": (n // world) * 32 * (world - 1) // world,
                       "roofline": _hbm(128 * n // world, sec * 1e6, note="per rank: 128*N/W algorithmic bytes")}
    del x, y, z
    # --- h(x) on the 2^24-constraint block system
    blocks = n >> 16
    bs = synth.BlockSystem(synth.mulgraph(1 << 16, seed=0xAC4, field=a.field), blocks)
    sh = par.ShardedR1CS.from_cyclic(bs.rows_of, bs.n, bs.m, ln, lr, ctx=ctx, collectives=coll)
    qh = par.DistributedQapH(sh, d, orc.generator)
    w = bs.witness()
    dw = to_dev(ctx, w)
    _, ok = qh.run(dw)
    wb = w.copy()
    wb[bs.wire(blocks - 1, 99), 0] ^= np.uint64(1)
    _, ok_bad = qh.run(to_dev(ctx, wb))
    sec = wall(lambda: qh.run(dw), 3)
    out["dist_qap_h"] = {"workload": f"distributed verificationWitness h(x): {blocks} x 2^16-constraint mulgraph blocks = 2^{ln} constraints over "
                                     f"{world} rank(s), block-cyclic rows, 6 all-to-alls (5 of them issued asynchronously under the next vector's local steps) + 1 all-reduce",
                         "exchange_overlapped": bool(d.overlapped(dw)),
                         "accepts_valid_rejects_corrupt": bool(ok and not ok_bad), "us": sec * 1e6,
                         "constraints_per_s": bs.n / sec}
    return out


def main_mgpu(a, devices):
    """The single-process launcher: ONE process drives all GPUs through acx_mgpu_* (include/acx.h) -- the shape of the
    reference's callers -- instead of one rank per GPU under torch.distributed.run.  Same workload per GPU and step as the
    multi-process path (`--copies` 2^logn-constraint mulgraph blocks per GPU: one block-diagonal system of
    copies * W * 2^logn constraints, rows sharded by the library), same accounting: a STEP is one verification of the
    device-resident witness over all GPUs (acx_mgpu_r1cs_verify_enqueue), the verdicts of 8 steps are combined by ONE
    RCCL all-reduce (acx_mgpu_r1cs_verdicts).  `devices` may repeat an ordinal (several shards on one GPU: peer-copy
    transport) to exercise the W > 1 path on a one-GPU box; the line says so."""
    from oracle.c_oracle import COracle
    W = len(devices)
    mg = acx.MultiGpu(a.field, devices)
    n0 = 1 << a.logn
    base = synth.mulgraph(n0, seed=0xAC355, field=a.field)
    bs = synth.BlockSystem(base, a.copies * W)
    mats = bs.full_rows()
    w = bs.witness()
    t_load = time.perf_counter()
    mr = mg.load(bs.n, bs.m, *mats)
    t_load = time.perf_counter() - t_load
    assert mr.n_shards == W
    b0, nnz0, _ = algorithmic_bytes(bs.mats, n0)
    bytes_per_launch = b0 * a.copies                     # per GPU
    mr.upload_witness(w)
    stream0 = torch.cuda.ExternalStream(mg.stream(0), device=torch.device("cuda", devices[0]))
    ring = 8

    def step(i):
        k = i % (2 * ring)
        mr.verify_enqueue(k)
        if k % ring == ring - 1:
            bad = mr.verdicts((k // ring) * ring, ring)
            assert not bad.any(), "a satisfying witness was rejected"

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < a.prewarm:
        for i in range(2 * ring):
            step(i)
    for i in range(a.warmup):
        step(i)
    assert not mr.verdicts(0, 2 * ring).any()
    mg.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream0)
    for i in range(a.steps):
        step(i)
    e1.record(stream0)
    assert not mr.verdicts(0, 2 * ring).any(), "a satisfying witness was rejected"     # ONE collective + wait: the clock stops after it
    mg.sync()
    dt = time.perf_counter() - t0
    kernel_us = e0.elapsed_time(e1) * 1e3 / a.steps
    # parity gates outside the timed region: the corrupted witness must be caught with the oracle's count and first row
    orc = COracle(a.field)
    wb = w.copy()
    wb[bs.wire(a.copies * W - 1, 77), 0] ^= np.uint64(1)
    _, nbad0, first0 = orc.r1cs_residuals(n0, bs.m0, *bs.mats, np.concatenate([wb[:1], wb[-(bs.m0 - 1):]]), nthreads=effective_cpus())
    got = mr.verify(wb)
    parity = bool(nbad0 > 0 and got == (False, nbad0, (a.copies * W - 1) * n0 + first0))
    out = {
        "metric": "R1CS constraints/sec (verifyAssignment over %s Fr, bit-exact vs oracle)" % ("BN254" if a.field == "bn254" else "BLS12-381"),
        "value": bs.n * a.steps / dt, "unit": "constraints/s", "n_gpus": len(set(devices)), "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u256 (9x29-bit limbs, Montgomery)", "data": "synthetic",
        "launcher": f"single process, acx_mgpu_* over {W} shard(s) on devices {devices} ({mg.transport})",
        "config": {"workload": f"r1cs_verify: ONE block-diagonal system of {a.copies} x {W} mulgraph blocks of 2^{a.logn} constraints "
                               f"(k=2, n_in=1024, window=4096, seed 0xAC355), rows sharded by libacx in nnz-balanced slabs; a step = one "
                               f"verification of the resident witness on every GPU, 1 verdict all-reduce per {ring} steps",
                   "constraints_per_step_per_gpu": a.copies * n0, "field": a.field + "_fr", "load_s": t_load,
                   "parallelism": f"{W} shards x {a.copies} blocks (weak scaling)"},
        "parity_vs_oracle": parity,
        "roofline": _hbm(bytes_per_launch, kernel_us, kernel="acx::k_r1cs_sell_split (shard 0's stream)", kernel_us=kernel_us,
                         traffic=None, traffic_measured_in_run=False),
    }
    ln = (bs.n - 1).bit_length()
    if not a.no_dist_pipeline and mr.n == 1 << ln and ln <= 24:
        mr.upload_witness(w)                           # the parity gate above left the corrupted witness resident
        ok = mr.qap_h_resident()                       # first call: buffers, tables
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ok = mr.qap_h_resident() and ok
        mg.sync()
        sec = (time.perf_counter() - t1) / reps
        mr.upload_witness(wb)
        ok_bad = mr.qap_h_resident()
        out["mgpu_qap_h"] = {"workload": f"acx_mgpu_qap_h_resident: verificationWitness h(x) of the same 2^{ln}-constraint system over {W} shard(s): "
                                         f"block-cyclic rows, 6 all-to-alls + 1 all-reduce per call, blocking call",
                             "accepts_valid_rejects_corrupt": bool(ok and not ok_bad), "us": sec * 1e6, "constraints_per_s": bs.n / sec}
    mr.close()
    mg.close()
    sys.stdout.flush()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--copies", type=int, default=32)
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--field", default="bn254", choices=["bn254", "bls12_381"])
    ap.add_argument("--force-dist", action="store_true",
                    help="run the collective code path even with one rank (RCCL smoke test on a 1-GPU box)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (with ACX_BENCH_ONE_DEVICE=1: every rank on cuda:0) lets the "
                         "multi-rank control flow be tested on a 1-GPU box")
    ap.add_argument("--prewarm", type=float, default=0.25, help="seconds of untimed launches before the warmup steps (clock ramp)")
    ap.add_argument("--sustain", type=float, default=0.5, help="seconds of extra K-step blocks for the median/min per-step figures")
    ap.add_argument("--dist-logn", type=int, default=24, help="size of the distributed NTT / h(x) job timed when N > 1 or --force-dist")
    ap.add_argument("--no-dist-pipeline", action="store_true", help="skip the distributed NTT / h(x) measurements of a multi-rank run")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc FETCH_SIZE pass (roofline.traffic then comes from profiles/r03_traffic.json, flagged)")
    ap.add_argument("--only-steps", action="store_true", help="internal: nothing but the batched launches (the child of the PMC pass)")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--launcher", default="auto", choices=["auto", "ranks", "mgpu"],
                    help="ranks: one process per GPU (torch.distributed.run, what the driver uses); mgpu: ONE process, all GPUs "
                         "through acx_mgpu_*; auto: mgpu when --gpus N > 1 is asked for outside torch.distributed.run")
    ap.add_argument("--mgpu-devices", default=None,
                    help="device ordinals of the mgpu launcher, e.g. 0,1,2,3 or 0,0 (repeats = several shards on one GPU); default 0..N-1")
    a = ap.parse_args()
    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.launcher == "mgpu" or (a.launcher == "auto" and not in_torchrun and (a.gpus > 1 or a.mgpu_devices)):
        devices = [int(x) for x in a.mgpu_devices.split(",")] if a.mgpu_devices else list(range(a.gpus))
        return main_mgpu(a, devices)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    if os.environ.get("ACX_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    ctx = acx.Context(a.field, local_rank)
    stream = torch.cuda.ExternalStream(ctx.stream)
    n = 1 << a.logn
    systems, witnesses, bytes_per_launch, nnz_total = [], [], 0, 0
    sample = None
    for c in range(a.copies):
        s = synth.mulgraph(n, seed=0xAC355 + 1000 * rank + c, field=a.field)
        mats = s.rows()
        w = s.witness()
        b, nnz, _ = algorithmic_bytes(mats, n)
        bytes_per_launch += b
        nnz_total += nnz
        systems.append(s.circuit.to_r1cs(ctx))
        witnesses.append(to_dev(ctx, w))
        if c == 0:
            sample = (mats, w, n, s.circuit.m)
    # Result slots {n_bad, first_bad}: a satisfied system never touches its slot (the kernel issues
    # atomics only for violated rows), so slots need no per-step reset.  Two half-rings of `ring`
    # slots: while the verdicts of one half are being all-reduced, the steps write the other half.
    ring = 8
    init = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    results = init.repeat(2 * ring, 1).contiguous()                  # [2*ring, 2]
    batches = [acx.Batch(ctx, systems, [w.data_ptr() for w in witnesses], results[k].data_ptr()) for k in range(2 * ring)]
    ctx.sync()
    torch.cuda.synchronize()

    pending = [None, None]

    def step(i):
        k = i % (2 * ring)
        half = k // ring
        if use_dist and k % ring == 0 and pending[half] is not None:
            pending[half].wait()                 # stream-level: the half's previous reduction is done
            pending[half] = None
        batches[k].verify_dev()
        if use_dist and k % ring == ring - 1:
            # ONE collective per `ring` verifications: SUM of the violated-row counts over the row
            # shards (the first_bad words ride along and are not meaningful after a SUM; a MIN
            # reduction for them is issued only on request, parallel.ShardedR1CS.verify)
            pending[half] = dist.all_reduce(results[half * ring:(half + 1) * ring], op=dist.ReduceOp.SUM, async_op=True)

    def drain():
        for h in (0, 1):
            if pending[h] is not None:
                pending[h].wait()
                pending[h] = None

    with torch.cuda.stream(stream):
        if use_dist:        # first use of the collectives (communicator set-up, kernel load) stays out of the timed region
            dist.barrier()
            for _ in range(2):
                dist.all_reduce(results, op=dist.ReduceOp.SUM, async_op=True).wait()
            torch.cuda.synchronize()
        # Clock ramp: from idle the GPU needs ~250 launches (35 ms) to reach its sustained clock
        # (tools/microbench/ramp.py: 150 us per step falling to 130), and it falls back within
        # milliseconds of idling.  A fixed untimed pre-run brings it there whatever --warmup the caller
        # chose; the W warmup steps, a (by now cheap) barrier and the K timed steps follow immediately.
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < a.prewarm:
            for _ in range(32):
                batches[0].verify_dev()
            ctx.sync()
        for i in range(a.warmup):
            step(i)
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for i in range(a.steps):
            step(i)
        e1.record(stream)
        if use_dist:        # verdicts of the last (possibly partial) revolution
            tail = dist.all_reduce(results, op=dist.ReduceOp.SUM, async_op=True)
            tail.wait()
        drain()
        torch.cuda.synchronize()
        # the clock stops here: the tail all-reduce above completes only when every rank has contributed,
        # i.e. finished its K steps, and the MAX over ranks below is the job's time
        dt = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
    kernel_us = e0.elapsed_time(e1) * 1e3 / a.steps
    # Sustained figure (SURVEY.md 8d "median and min reported"): blocks of K steps, each bracketed by HIP events, for
    # at least --sustain seconds.  `value` stays the wall-clock of the exactly-K-step region above; these are
    # reported beside it so that a DVFS hiccup in a few-millisecond region is visible.
    block_us = []
    if not use_dist and a.sustain > 0:
        with torch.cuda.stream(stream):
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < a.sustain:
                evs = []
                for _ in range(16):
                    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    b0.record(stream)
                    for i in range(a.steps):
                        step(i)
                    b1.record(stream)
                    evs.append((b0, b1))
                stream.synchronize()
                block_us += [x.elapsed_time(y) * 1e3 / a.steps for x, y in evs]
        assert int(results[:, 0].abs().sum()) == 0
    assert int(results[:, 0].abs().sum()) == 0, "a satisfying witness was rejected"     # parity gate of the timed config

    # negative control outside the timed region: one flipped witness limb must be caught.  A full
    # batched launch like the timed ones, so that every k_r1cs_sell call in a profile of this command
    # is the same workload (the rocprofv3 average in profiles/ is comparable with kernel_us).
    neg = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    wbad = witnesses[0].clone()
    wbad[77, 0] ^= 1          # flip the lowest bit of limb 0 of one witness entry
    neg_batch = acx.Batch(ctx, systems, [wbad.data_ptr()] + [w.data_ptr() for w in witnesses[1:]], neg.data_ptr())
    neg_batch.verify_dev()
    ctx.sync()
    assert int(neg[0]) > 0, "a corrupted witness was accepted"
    # ... and not merely "something was caught": system 0's whole residual vector, violated-row count and first violated
    # row under that corrupted witness against the CPU oracle, and the batched launch's count against the oracle's
    parity_residuals = None
    if rank == 0 and not a.only_steps:
        from oracle.c_oracle import COracle
        mats0, w0, n0, m0 = sample
        w0b = w0.copy()
        w0b[77, 0] ^= np.uint64(1)
        want_res, want_bad, want_first = COracle(a.field).r1cs_residuals(n0, m0, *mats0, w0b, nthreads=effective_cpus())
        parity_residuals = bool(want_bad > 0 and int(neg[0]) == want_bad and np.array_equal(systems[0].residuals(w0b), want_res)
                                and systems[0].verify(w0b) == (False, want_bad, want_first))
        assert parity_residuals, "residual vector of the corrupted system differs from the oracle's"

    # for reference: ONE 2^16-constraint system per launch (configs[1] taken literally; cache resident)
    single_us = None
    if rank == 0 and world == 1 and not a.only_steps:
        with torch.cuda.stream(stream):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                systems[0].verify_dev(witnesses[0].data_ptr(), neg.data_ptr())
            s0.record(stream)
            for _ in range(50):
                systems[0].verify_dev(witnesses[0].data_ptr(), neg.data_ptr())
            s1.record(stream)
            s1.synchronize()
            single_us = s0.elapsed_time(s1) * 1e3 / 50
    # "also report the cache-resident number, labelled" -- the SAME system `copies` times per launch (each
    # against its own witness): its 16 MB of constraint stream stay in L2 / Infinity Cache, so this is NOT an HBM figure
    resident_us = None
    if rank == 0 and world == 1 and not a.only_steps:
        res_r = torch.tensor([0, -1], dtype=torch.int64, device="cuda")          # one {n_bad, first_bad} slot per batch
        rb = acx.Batch(ctx, [systems[0]] * a.copies, [witnesses[0].data_ptr()] * a.copies, res_r.data_ptr())
        with torch.cuda.stream(stream):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                rb.verify_dev()
            s0.record(stream)
            for _ in range(50):
                rb.verify_dev()
            s1.record(stream)
            s1.synchronize()
            resident_us = s0.elapsed_time(s1) * 1e3 / 50
        assert int(res_r[0]) == 0, "cache-resident batch rejected a valid witness"
        del rb

    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    dist_extra = {}
    if use_dist and not a.no_dist_pipeline:
        del batches, neg_batch
        systems.clear(); witnesses.clear()
        dist_extra = bench_distributed(ctx, a, world, rank, dist)
    if rank == 0:
        total = world * a.copies * n * a.steps
        value = total / dt
        achieved = bytes_per_launch / kernel_us * 1e-3   # GB/s, algorithmic bytes / launch duration
        out = {
            "metric": "R1CS constraints/sec (verifyAssignment over %s Fr, bit-exact vs oracle)" % ("BN254" if a.field == "bn254" else "BLS12-381"),
            "value": value, "unit": "constraints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (9x29-bit limbs, Montgomery)", "data": "synthetic", "parity_vs_oracle": parity_residuals,
            "launcher": "one process per GPU (torch.distributed.run), collectives through torch.distributed/RCCL" if use_dist else "single process, one GPU",
            "config": {"workload": f"r1cs_verify: {a.copies} independent 2^{a.logn}-constraint mulgraph systems per GPU per "
                                   f"step, one batched launch (k=2, n_in=1024, window=4096, seeds 0xAC355+1000*rank+c)",
                       "constraints_per_step_per_gpu": a.copies * n, "nnz_per_step_per_gpu": nnz_total,
                       "single_system_launch_us": single_us,
                       "field": a.field + "_fr", "parallelism": f"{world} rank(s) x {a.copies} independent systems (weak scaling), 1 verdict all-reduce per {ring} steps" if use_dist else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "acx::k_r1cs_sell_split", "kernel_us": kernel_us,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if resident_us:
            out["cache_resident"] = {"us_per_launch": resident_us, "constraints_per_s": a.copies * n / resident_us * 1e6,
                                     "single_system_us_per_launch": single_us,
                                     "note": f"labelled, NOT the metric: the same 2^{a.logn}-constraint system {a.copies} times per launch (and once per launch): "
                                             "its constraint stream stays in L2 / Infinity Cache; the headline rotates over independent systems from HBM"}
        out.update(dist_extra)
        if block_us:
            bs = sorted(block_us)
            out["sustained"] = {"blocks": len(bs), "steps_per_block": a.steps, "seconds": sum(bs) * a.steps * 1e-6,
                                "us_per_step_median": bs[len(bs) // 2], "us_per_step_min": bs[0], "us_per_step_max": bs[-1],
                                "frac_median": bytes_per_launch / bs[len(bs) // 2] * 1e-3 / HBM_PEAK_GBS,
                                "note": "kernel times at the sustained clock (0.25 s untimed pre-run before the timed region)"}
        live = None
        if world == 1 and not use_dist and not a.no_pmc and not a.only_steps:
            live = measure_traffic_live(a)
        if live is not None:
            # HBM bytes of the headline kernel measured in THIS run: a rocprofv3 --pmc FETCH_SIZE pass of its own over the
            # same batched launches in a child process (counters need the profiler around the process)
            out["roofline"]["traffic"] = live[0]
            out["roofline"]["traffic_measured_in_run"] = True
            out["roofline"]["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE around a child run of this script's batched launches ({live[1]} launches "
                                                 "averaged); bytes = FETCH_SIZE (KiB) * 1024 * 2 (gfx950 reports half of wide coalesced reads)")
            out["roofline"]["achieved_traffic"] = live[0] / kernel_us * 1e-3
            out["roofline"]["frac_traffic"] = out["roofline"]["achieved_traffic"] / HBM_PEAK_GBS
        live_valu = measure_counter_live(a, "SQ_INSTS_VALU") if live is not None else None      # second pass, its own run
        try:    # PMC-measured HBM bytes per launch, recorded from a separate rocprofv3 --pmc pass
            tr = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))["acx::k_r1cs_sell"]
            if live is not None and tr["workload"] == {"field": a.field, "copies": a.copies, "logn": a.logn}:
                out["roofline"]["traffic_committed_pass"] = tr["traffic_bytes_per_launch"]       # the earlier pass, for comparison
                out["roofline"]["valu_issue"] = _valu_issue("acx::k_r1cs_sell", "valu_wave_insts_per_launch", kernel_us, tr["workload"])
                if live_valu is not None and out["roofline"]["valu_issue"]:
                    # SQ_INSTS_VALU of this run instead of the committed pass (the issue RATE stays the microbenchmark's)
                    vi = out["roofline"]["valu_issue"]
                    vi["issue_bound_us"] *= live_valu[0] / vi["wave_insts"]
                    vi["wave_insts"] = live_valu[0]
                    vi["frac"] = vi["issue_bound_us"] / kernel_us
                    vi["measured_in_run"] = True
                    vi["source"] = "SQ_INSTS_VALU from a rocprofv3 --pmc pass of this run; issue rate from profiles/r01_valu_rates.txt"
            elif tr["workload"] == {"field": a.field, "copies": a.copies, "logn": a.logn}:
                # NOT measured in this run: a constant from the committed rocprofv3 --pmc pass of the same command (PMC
                # counters need the profiler around the process).  The live quantities of this line are the times.
                out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                out["roofline"]["traffic_measured_in_run"] = False
                out["roofline"]["traffic_source"] = tr["source"]
                out["roofline"]["traffic_measured_at_commit"] = tr.get("measured_at_commit")
                # the same launch time against the bytes the kernel really moves (the algorithmic figure counts 36 bytes
                # per C entry that the unit-C path never reads, and 36 instead of 40 per A / B entry)
                out["roofline"]["achieved_traffic"] = tr["traffic_bytes_per_launch"] / kernel_us * 1e-3
                out["roofline"]["frac_traffic"] = out["roofline"]["achieved_traffic"] / HBM_PEAK_GBS
                out["roofline"]["valu_issue"] = _valu_issue("acx::k_r1cs_sell", "valu_wave_insts_per_launch", kernel_us, tr["workload"])
        except (OSError, KeyError, ValueError):
            pass
        if world == 1 and not a.no_ntt:
            out["ntt"] = bench_ntt(ctx, stream, a.field, prewarm=a.prewarm)
            out["qap_h"] = bench_qap_h(ctx, stream, a.field, prewarm=a.prewarm)
            out["r1cs_small_coeff"] = bench_small_coeff(ctx, stream, a.field, copies=a.copies, log_n=a.logn, prewarm=a.prewarm)
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(sample, a.field)
        line = json.dumps(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes its version banner through C stdio, which a pipe buffers until exit: flush it first so that the JSON
    # line is the LAST line of this job's stdout whatever the collective library prints
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        if world > 1:
            time.sleep(0.5)     # the other ranks have nothing left to do but flush and exit
        print(line, flush=True)


if __name__ == "__main__":
    main()
