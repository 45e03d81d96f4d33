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

Beside the headline, rank 0 measures one parity-gated object per config of BASELINE.json -- `reference_bench` (configs[0]: the
reference's own four criterion benchmarks of bench/Circuit.hs:26-36 through the C ABI), `ntt` / `qap_h` / `load` / `qap_columns`
(configs[2]), `e2e` (the host-buffer boundary, PCIe included), `bls12_381` (configs[4]), `r1cs_small_coeff`, `gate_mix` (the
reference's own generator shape: Equal / Split gates, 257-entry rows), `cpu_baseline` --
each under guard(): a failure becomes an entry of "errors", the line is always printed.  --only / --skip select them.

Prints ONE JSON line on rank 0."""
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
T_START = time.perf_counter()

# Every secondary measurement runs under guard(): a failure becomes an entry of the line's "errors" list instead of costing
# the line (the headline `value` is measured before any of them).
ERRORS = []
PHASES = {}


def guard(where, fn, *args, **kw):
    t0 = time.perf_counter()
    try:
        if os.environ.get("ACX_BENCH_FAIL") == where:       # test hook: what a failing secondary measurement does to the line
            raise RuntimeError("injected failure (ACX_BENCH_FAIL)")
        return fn(*args, **kw)
    except BaseException as e:                      # AssertionError of a parity gate included
        if isinstance(e, (KeyboardInterrupt, SystemExit)):
            raise
        ERRORS.append({"where": where, "error": (type(e).__name__ + ": " + str(e))[:400]})
        return None
    finally:
        PHASES[where] = round(PHASES.get(where, 0.0) + time.perf_counter() - t0, 3)


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


def measure_counter_live(a, counter, kernel="k_r1cs_sell_split", child=("--only-steps",), timeout_s=120):
    """One hardware counter of one kernel, per launch, measured NOW: this script re-runs its launches of that kernel (only
    those: --only-steps / --only-ntt) as a child under `rocprofv3 --pmc <counter>` -- a counter pass of its own, no tracing
    beside it, as MI355X_MICROARCH.md prescribes -- and reads the per-dispatch values from the profiler's database.  Returns
    (average value per launch, launches averaged) or None when the profiler is missing or fails."""
    import glob, shutil, sqlite3, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = tempfile.mkdtemp(prefix="acx_pmc_", dir="/tmp")
    cmd = [sys.executable, os.path.abspath(__file__)] + list(child) + ["--skip", "all", "--sustain", "0", "--steps", "20",
           "--warmup", "2", "--prewarm", "0.05", "--field", a.field, "--copies", str(a.copies), "--logn", str(a.logn)]
    try:
        subprocess.call(["rocprofv3", "--pmc", counter, "-d", out, "-o", "pass", "--"] + cmd, stdout=subprocess.DEVNULL,
                        stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=timeout_s)
        best = None
        for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
            cur = sqlite3.connect(db).cursor()
            for name, v, cnt in cur.execute("select kernel_name, sum(value), count(*) from counters_collection "
                                            "where counter_name = ? group by kernel_name", (counter,)):
                if kernel in name:
                    best = (float(v) + (best[0] if best else 0.0), cnt + (best[1] if best else 0))     # every instance of the kernel
        return best
    except Exception:                                  # a profiler problem must not cost the line
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_traffic_live(a):
    """HBM bytes the headline kernel fetches per launch: FETCH_SIZE counts KiB and, on gfx950, half of what wide coalesced
    reads move (calibrated on a copy kernel, tools/prof.py): bytes = value * 1024 * 2."""
    r = measure_counter_live(a, "FETCH_SIZE")
    return None if r is None else (r[0] / r[1] * 1024.0 * 2.0, r[1])


def cpu_baseline(sample, field="bn254", budget_s=12.0, with_reference_algorithm=True):
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
    if with_reference_algorithm:
        out["reference_algorithm"] = cpu_reference_algorithm(orc, field)
    return out


_REF_CASE = {}


def ref_case(field="bn254", log_n=10):
    """configs[0] (the reference's CPU-runnable case scaled to 2^10 gates): the circuit, its rows and witness,
    and the ORACLE's createPolynomialsFFT of it (3 m dense interpolations, single threaded like the reference; timed).  Built
    once per run: the CPU baseline's reference-algorithm leg and the GPU-side `reference_bench` parity gate share it."""
    if field not in _REF_CASE:
        from oracle.c_oracle import COracle
        orc = COracle(field)
        n = 1 << log_n
        s = synth.mulgraph(n, n_in=64, seed=0xAC1, field=field)
        mats, w = s.rows(), s.witness()
        m = w.shape[0]
        t0 = time.perf_counter()
        cols = np.stack([orc.qap_columns(n, log_n, mats[k], 0, m, nthreads=1) for k in range(3)])
        _REF_CASE[field] = {"s": s, "mats": mats, "w": w, "m": m, "n": n, "log_n": log_n, "cols": cols, "t_create": time.perf_counter() - t0, "orc": orc}
    return _REF_CASE[field]


def cpu_reference_algorithm(orc, field="bn254", log_n=10):
    """BASELINE.md section 2 / SURVEY.md 8(d), configs[0] (the reference's CPU-runnable case, 2^10 gates): what the Haskell
    ALGORITHM costs -- createPolynomialsFFT into dense per-wire polynomials (src/QAP.hs:512-525: 3 m interpolations) and
    verifyAssignment in the polynomial domain (src/QAP.hs:276-327: m scalar x polynomial sums per matrix, dense product,
    long division by x^N - 1), single threaded like the reference.  A C restatement (oracle/acx_oracle.c orc_qap_columns,
    orc_ref_verify), NOT GHC: boxed Naturals and lazy lists cost more than this."""
    rc = ref_case(field, log_n)
    n, m, mats, w, cols = rc["n"], rc["m"], rc["mats"], rc["w"], rc["cols"]
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 1.0:
        q, ok = orc.ref_verify(m, log_n, cols, w)
        assert ok
        reps += 1
    t_verify = (time.perf_counter() - t0) / reps
    h, ok2 = orc.qap_h(n, m, log_n, *mats, w)
    assert ok2 and np.array_equal(h[:n], q) and not h[n:].any(), "polynomial-domain quotient differs from the evaluation-domain h(x)"
    return {"config": f"configs[0]: 2^{log_n}-gate mulgraph circuit, m = {m} wires ({field} Fr)", "cores": 1,
            "create_qap_s": rc["t_create"], "verify_s": t_verify, "constraints_per_s": n / t_verify,
            "note": "C restatement of the reference's polynomial-domain algorithm (3 m dense interpolations; m scalar x polynomial "
                    "sums per matrix, dense product, long division), not GHC; quotient checked against the evaluation-domain h(x)"}


def reference_bench(ctx, field="bn254"):
    """The reference's OWN benchmarked operations (bench/Circuit.hs:26-36: criterion on `evalArithCircuit`,
    `arithCircuitToGenQAP` "no interpolation", `arithCircuitToQAPFFT` "fast interpolation", `arithCircuitToQAP` "slow") through
    the C ABI on configs[0]'s 2^10-gate circuit -- wall clock per operation as a host would see it, blocking calls, host
    buffers in and (where the reference returns a value the host reads) out.  Every result is parity-gated: the evaluated
    witness equals the host fold's and satisfies the ORACLE's check; all 3 m FFT-path polynomials equal the oracle's
    coefficient for coefficient; the naive-path target vanishes on every root and a naive column takes its matrix's
    values on every root (degree < n: that determines it)."""
    rc = ref_case(field)
    s, mats, w, m, n, log_n, orc = rc["s"], rc["mats"], rc["w"], rc["m"], rc["n"], rc["log_n"], rc["orc"]
    p = acx.engine.FIELDS[field][1]
    c = s.circuit

    def wall(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    out = {"config": f"configs[0]: 2^{log_n}-gate mulgraph circuit, m = {m} wires, N = {n} ({field} Fr); the four criterion benchmarks of bench/Circuit.hs:28-35"}
    # 1. evalArithCircuit / generateAssignment: acx_r1cs_eval (level-parallel on the GPU), acx_circuit_eval (host fold) beside it
    r = c.to_r1cs(ctx)
    gw, _ = r.eval_witness(s.inputs)
    _, nbad, _ = orc.r1cs_residuals(n, m, *mats, gw, want_residuals=False)
    ok_eval = bool(np.array_equal(gw, w) and nbad == 0)
    out["evalArithCircuit"] = {"gpu_acx_r1cs_eval_s": wall(lambda: r.eval_witness(s.inputs), 20),
                               "gpu_resident_no_download_s": wall(lambda: r.eval_witness(s.inputs, download=False), 20),
                               "host_acx_circuit_eval_s": wall(lambda: c.eval(s.inputs), 20), "parity_vs_oracle": ok_eval,
                               "note": "23 dependency levels of at most 88 gates: ONE launch of one workgroup (k_eval_levels_fused, ~3 us of dependent latency per level) + input upload and result fetch; latency bound, on a par with the host fold at this size"}

    # 2. arithCircuitToGenQAP: acx_circuit_create + acx_circuit_to_r1cs
    def gen_qap():
        c2 = acx.Circuit(field, c._gate_list, c._keep)
        r2 = c2.to_r1cs(ctx)
        ctx.sync()
        return c2, r2

    def gen_qap_drop():
        c2, r2 = gen_qap()
        r2.close(); c2.close()

    rp, col, val = r.export(0)
    out["arithCircuitToGenQAP"] = {"s": wall(gen_qap_drop, 10), "calls": "acx_circuit_create + acx_circuit_to_r1cs",
                                   "parity_vs_oracle": bool(np.array_equal(rp, mats[0][0]) and np.array_equal(col, mats[0][1]) and np.array_equal(val, mats[0][2]))}
    # 3. arithCircuitToQAPFFT: the same + all 3 m per-wire polynomials (createPolynomialsFFT), device resident and to the host
    bufs = [torch.empty((m * n, 4), dtype=torch.int64, device="cuda") for _ in range(3)]
    lens = torch.zeros((3, m), dtype=torch.int64, device="cuda")

    def qap_fft():
        c2, r2 = gen_qap()
        for k in range(3):
            r2.qap_columns_dev(k, 0, m, bufs[k].data_ptr(), lens[k].data_ptr())
        ctx.sync()
        r2.close(); c2.close()

    t_fft = wall(qap_fft, 5)
    ok_cols = True
    for k in range(3):
        got = _from_dev(ctx, bufs[k], m * n).reshape(m, n, 4)
        ok_cols = ok_cols and bool(np.array_equal(got, rc["cols"][k]))

    def qap_fft_host():
        c2, r2 = gen_qap()
        for k in range(3):
            r2.qap_columns(k, 0, m)
        r2.close(); c2.close()

    out["arithCircuitToQAPFFT"] = {"s": t_fft, "to_host_buffers_s": wall(qap_fft_host, 2), "polynomials": 3 * m, "coefficients": 3 * m * n,
                                   "calls": "acx_circuit_create + acx_circuit_to_r1cs + 3 x acx_qap_columns_dev (all m wires); to_host_buffers: acx_qap_columns (D2H of every coefficient)",
                                   "parity_vs_oracle": ok_cols, "cpu_restatement_s": rc["t_create"]}
    # 4. arithCircuitToQAP ("slow": Lagrange on the roots 0 .. n-1, target prod (x - r)): acx_naive_create + target + 3 x columns
    roots = list(range(n))
    keep = {}

    def qap_naive():
        c2, r2 = gen_qap()
        nv = acx.Naive(r2, roots)
        keep["target"] = nv.target()
        keep["cols"] = [nv.columns(k, 0, m) for k in range(3)]
        nv.close(); r2.close(); c2.close()

    t_naive = wall(qap_naive, 2)

    def horner(coeffs, x):
        acc = 0
        for cf in reversed(coeffs):
            acc = (acc * x + cf) % p
        return acc
    tgt = acx.fr_to_ints(keep["target"])
    ok_naive = tgt[-1] == 1 and len(tgt) == n + 1 and all(horner(tgt, x) == 0 for x in roots)
    wire = int(mats[0][1][mats[0][0][n // 2]])                       # a wire row n/2 of A mentions
    poly = acx.fr_to_ints(keep["cols"][0][0][wire])
    dense = [0] * n
    rp, col, val = mats[0]
    for i in range(n):
        for e in range(int(rp[i]), int(rp[i + 1])):
            if int(col[e]) == wire:
                dense[i] = (dense[i] + acx.fr_to_ints(val[e:e + 1])[0]) % p
    ok_naive = bool(ok_naive and all(horner(poly, x) == dense[x] for x in roots))
    out["arithCircuitToQAP"] = {"s": t_naive, "calls": "acx_circuit_create + acx_circuit_to_r1cs + acx_naive_create + acx_naive_target + 3 x acx_naive_columns (all m wires, host buffers)",
                                "polynomials": 3 * m, "parity_vs_interpolation_conditions": ok_naive}
    r.close()
    return out


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


def _valu_rate():
    """measured integer-multiplier issue rate (tools/microbench/valu_rates.hip -> profiles/r01_valu_rates.txt), via the committed table"""
    tr = json.load(open(os.path.join(ROOT, "profiles", "r04_traffic.json")))
    return tr["valu_rate"]["simds"], tr["valu_rate"]["wave_insts_per_s_per_simd"], tr


def _valu_issue(key, count_field, us, workload, live_count=None):
    """VALU-issue fraction of a kernel: SQ_INSTS_VALU (wave instructions per launch: `live_count` from a rocprofv3 --pmc pass of
    THIS run, else the committed pass of the same workload) / 1024 SIMDs / the measured multiplier issue rate, over the live
    launch time."""
    try:
        simds, rate, tr = _valu_rate()
        if live_count is None:
            if tr[key]["workload"] != workload:
                return None
            count, measured, src = tr[key][count_field], False, "profiles/r04_traffic.json, profiles/r01_valu_rates.txt"
        else:
            count, measured, src = live_count, True, "SQ_INSTS_VALU from a rocprofv3 --pmc pass of this run; issue rate from profiles/r01_valu_rates.txt"
        bound_us = count / simds / rate * 1e6
        return {"wave_insts": count, "issue_bound_us": bound_us, "frac": bound_us / us, "source": src, "measured_in_run": measured}
    except (OSError, KeyError, ValueError):
        return None


def _hbm(alg_bytes, us, **extra):
    d = {"bound": "hbm", "achieved": alg_bytes / us * 1e-3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": alg_bytes / us * 1e-3 / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes}
    d.update(extra)
    return d


def bench_ntt(ctx, stream, field="bn254", log_n=20, reps=40, prewarm=0.25, batch=64, live_valu=None):
    """Secondary metrics of configs[2]: one 2^20-point transform (= FFT.interpolate of one QAP
    column) and a batch of 64 of them.  Parity gate first: the inverse transform of a fixed random vector is
    compared with the C oracle's, element by element.  The timed loop alternates inverse and forward on that
    vector, so every launch works on the same data (inverse then forward is the identity).  live_valu: SQ_INSTS_VALU per
    transform from this run's own counter pass (measure_ntt_valu_live)."""
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
    alg = 128 * n      # two reads + two writes of every 32-byte element (two-pass four-step)
    note = ("VALU-bound: ~10 Montgomery products (171 v_mad_u64_u32 each) per element; the HBM fraction is what "
            "SURVEY.md 8(d) asks to be quoted, valu_issue is the fraction of the integer-issue bound (profiles/r02_ntt.txt)")
    out = {"workload": f"NTT N=2^{log_n} ({field} Fr), alternating inverse/forward on a fixed vector, acx::k_ntt_r4",
           "parity_vs_oracle": parity, "us": us, "field_ops_per_s": ops / us * 1e6,
           "roofline": _hbm(alg, us, note=note, valu_issue=_valu_issue("acx::k_ntt_r4", "valu_wave_insts_per_transform", us,
                                                                       {"field": field, "logn": log_n}, live_valu))}
    if batch:
        xb = to_dev(ctx, synth.random_fr(n * batch, 6, 1, field))
        flipb = [False]

        def many():
            flipb[0] = not flipb[0]
            ctx.ntt_dev(xb.data_ptr(), log_n, batch, inverse=flipb[0])

        us_b = _timed(stream, many, 6, prewarm) / batch
        out["batch"] = {"transforms": batch, "us_per_transform": us_b, "field_ops_per_s": ops / us_b * 1e6, "roofline": _hbm(alg, us_b)}
    return out


def only_ntt(a):
    """child of the NTT counter pass (--only-ntt): nothing but `reps` single 2^20-point transforms on a resident vector"""
    ctx = acx.Context(a.field, 0)
    x = to_dev(ctx, synth.random_fr(1 << 20, 5, 1, a.field))
    for i in range(2 * 20):
        ctx.ntt_dev(x.data_ptr(), 20, 1, inverse=bool(i & 1))
    ctx.sync()
    print(json.dumps({"transforms": 40}))


def measure_ntt_valu_live(a):
    """SQ_INSTS_VALU of one 2^20-point transform measured in THIS run: the pass kernels' counts of a child that runs 40
    transforms and nothing else (its conversion kernel is not a k_ntt_r4), summed over the launches, / 40."""
    r = measure_counter_live(a, "SQ_INSTS_VALU", kernel="k_ntt_r4", child=("--only-ntt",))
    return None if r is None or r[1] == 0 else r[0] / 40.0


class C3System:
    """BASELINE.json configs[2] / configs[4]: ONE 2^20-gate mulgraph circuit (2^20 constraints), built once and shared by the
    `load`, `qap_h`, `qap_columns` and `e2e` objects.  prepare() is pure host work (numpy + acx_circuit_create: no device) and
    may run on a helper thread while the counter passes keep the GPU busy; load() puts it on the device and times that."""

    def __init__(self, field, log_n=20, seed=0xAC3):
        self.field, self.log_n, self.seed = field, log_n, seed
        self.s = self.mats = self.w = self.r = self.dw = None
        self.t = {}

    def prepare(self):
        t0 = time.perf_counter()
        self.s = synth.mulgraph(1 << self.log_n, seed=self.seed, field=self.field)
        self.t["synth_and_create_s"] = time.perf_counter() - t0
        self.mats, self.w = self.s.rows(), self.s.witness()
        return self

    def load(self, ctx):
        c = self.s.circuit
        t0 = time.perf_counter()
        again = acx.Circuit(self.field, c._gate_list, c._keep)      # acx_circuit_create alone, on the marshalled gate list
        self.t["circuit_create_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.r = again.to_r1cs(ctx)                                  # acx_circuit_to_r1cs: rows -> CSR + SELL-64 on the device
        ctx.sync()
        self.t["to_r1cs_s"] = time.perf_counter() - t0
        again.close()
        self.dw = to_dev(ctx, self.w)
        return self


def bench_load(c3):
    """configs[2]'s load path (`arithCircuitToGenQAP`, src/QAP.hs:530-539, at 2^20 gates): marshalled gate list ->
    acx_circuit_create (validation, Montgomery conversion, gateToGenQAP rows on the host's cores) -> acx_circuit_to_r1cs
    (device CSR + SELL-64).  Parity: the device system's rows exported again equal the host rows the oracle checks
    (first matrix, sampled rows) and the satisfying witness is accepted (bench_qap_h's gate covers every row through h)."""
    n = 1 << c3.log_n
    rp, col, val = c3.r.export(0)
    parity = bool(np.array_equal(rp, c3.mats[0][0]) and np.array_equal(col[:4096], c3.mats[0][1][:4096]) and np.array_equal(val[-4096:], c3.mats[0][2][-4096:]))
    nnz = int(sum(c3.r.nnz))
    return {"workload": f"arithCircuitToGenQAP at 2^{c3.log_n} gates ({c3.field} Fr): acx_circuit_create + acx_circuit_to_r1cs, m = {c3.r.m} wires, {nnz} entries",
            "circuit_create_s": c3.t["circuit_create_s"], "to_r1cs_s": c3.t["to_r1cs_s"],
            "constraints_per_s": n / (c3.t["circuit_create_s"] + c3.t["to_r1cs_s"]), "host_threads": effective_cpus(),
            "synthetic_generation_s": c3.t["synth_and_create_s"], "export_matches_host_rows": parity,
            "note": "wall clock of two C-ABI calls, host cores + one H2D of the rows; synthetic_generation_s (numpy, not product code) is outside"}


def bench_qap_h(ctx, stream, c3, reps=10, prewarm=0.25):
    """configs[2]'s third C3 metric: the h(x) pipeline of verificationWitness (src/QAP.hs:309-327) on a
    2^20-constraint mulgraph system, device resident (witness in, N+1 coefficients out): residual dots,
    3 iNTT, 2 coset NTT (L, R), pointwise, coset iNTT, minus O / z in coefficient form -- six transforms.
    Parity gate: every coefficient against the C oracle (which runs the textbook seven)."""
    from oracle.c_oracle import COracle
    field, log_n = c3.field, c3.log_n
    orc = COracle(field)
    n = 1 << log_n
    mats, w, r, dw = c3.mats, c3.w, c3.r, c3.dw
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
            "parity_vs_oracle": parity, "us": us, "field_ops_per_s": ops / us * 1e6, "constraints_per_s": n / us * 1e6, "roofline": _hbm(alg, us)}


def bench_qap_columns(ctx, stream, c3, prewarm=0.1):
    """`createPolynomialsFFT` (src/QAP.hs:512-525) at 2^20 constraints, results left on the device (acx_qap_columns_dev): sparse
    columns (intermediate wires: one entry in C, one or two in A -- interpolated directly, k_col_direct: k products per
    coefficient) and dense columns (input wires, hundreds of entries: scatter + batched inverse NTT).  The kernel only stores:
    bytes written / time against the HBM peak.  Parity gate: one sparse column of A, one of C and one dense column of A, every
    coefficient and the stripped length, against the C oracle's interpolation."""
    from oracle.c_oracle import COracle
    orc = COracle(c3.field)
    n, log_n, r = 1 << c3.log_n, c3.log_n, c3.r
    wires = 64
    out = torch.empty((wires * n, 4), dtype=torch.int64, device="cuda")
    lens = torch.zeros(wires, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    mid0 = 1 + c3.s.n_in + n // 2
    res, parity = {}, True
    for what, w0, mat, cnt in (("sparse_A", mid0, 0, wires), ("sparse_C", mid0, 2, wires), ("dense_A", 1, 0, 16)):
        r.qap_columns_dev(mat, w0, cnt, out.data_ptr(), lens.data_ptr())
        ctx.sync()
        want = orc.qap_columns(n, log_n, c3.mats[mat], w0 + 3, 1, nthreads=effective_cpus())[0]          # column 3 of the batch
        got = _from_dev(ctx, out[3 * n:4 * n], n)
        nz = np.nonzero(want.any(axis=1))[0]
        parity = parity and bool(np.array_equal(got, want) and int(lens[3]) == (int(nz[-1]) + 1 if nz.size else 0))
        us = _timed(stream, lambda: r.qap_columns_dev(mat, w0, cnt, out.data_ptr(), lens.data_ptr()), 5, prewarm)
        rp = c3.mats[mat][0]
        res[what] = {"columns_per_call": cnt, "us_per_column": us / cnt, "columns_per_s": cnt / us * 1e6,
                     "roofline": _hbm(32 * n, us / cnt, note="store-only: 32 N bytes written per column")}
    # entries per sparse column of the timed batches (what k_col_direct's cost follows)
    colc = np.bincount(c3.mats[0][1], minlength=r.m)[mid0:mid0 + wires]
    res["sparse_A"]["entries_per_column_mean"] = float(colc.mean())
    return {"workload": f"createPolynomialsFFT at N = 2^{log_n} ({c3.field} Fr), acx_qap_columns_dev, {wires} sparse / 16 dense columns per call, results device resident",
            "parity_vs_oracle": parity, **res,
            "note": "sparse columns are VALU-issue bound (one shared Montgomery reduction + 81 multiplier instructions per entry and coefficient: "
                    "~257 VALU instructions per 32 bytes at k = 1), not store bound: a plain store stream reaches 6.3-6.7 TB/s (tools/microbench/write_bw.hip)"}


def _hip_runtime():
    """the libamdhip64 this process already has mapped (torch's), for hipHostRegister on caller-owned numpy memory"""
    import ctypes
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    return ctypes.CDLL("libamdhip64.so")


class Pinned:
    """hipHostRegister on a numpy array for the life of a `with` block: what a host does once for a witness buffer it reuses"""

    def __init__(self, arr):
        self.arr, self.hip = arr, _hip_runtime()

    def __enter__(self):
        import ctypes
        rc = self.hip.hipHostRegister(ctypes.c_void_p(self.arr.ctypes.data), ctypes.c_size_t(self.arr.nbytes), ctypes.c_uint(0))
        if rc != 0:
            raise RuntimeError(f"hipHostRegister failed ({rc})")
        return self.arr

    def __exit__(self, *exc):
        import ctypes
        self.hip.hipHostUnregister(ctypes.c_void_p(self.arr.ctypes.data))


def bench_e2e(r, w, c3=None, many=48):
    """SURVEY.md 8(d) "end-to-end incl. witness H2D": the HOST-BUFFER boundary the reference's callers use
    (`verifyAssignment qap assignment`, src/QAP.hs:276-282) -- acx_r1cs_verify on a canonical witness in host memory: H2D copy,
    canonicity check + Montgomery conversion, the residual launch, the verdict back -- on one 2^16-constraint system of the
    headline workload: pageable and hipHostRegister-ed witness, one caller and four concurrent callers (four lanes per
    context), and acx_r1cs_verify_many (`all (verifyAssignment qap) inputs`: one copy + one batched launch per chunk).
    PCIe-inclusive constraints/s: never the headline `value`.  Every call must accept the satisfying witness."""
    import threading
    n, m = r.n, r.m
    out = {"workload": f"acx_r1cs_verify / acx_r1cs_verify_many on host buffers: one 2^{n.bit_length() - 1}-constraint system, witness {m * 32 / 1e6:.2f} MB per call"}

    def rate(rows, fn, reps, callers=1):
        for _ in range(3):
            fn()
        if callers == 1:
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return rows * reps / (time.perf_counter() - t0)
        bar = threading.Barrier(callers + 1)

        def body():
            bar.wait()
            for _ in range(reps):
                fn()
            bar.wait()
        th = [threading.Thread(target=body) for _ in range(callers)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        return rows * reps * callers / dt

    def one(sys_, wv):
        assert sys_.verify(wv)[0], "host-buffer verify rejected a satisfying witness"

    wp = w.copy()
    out["verify_pageable"] = {"constraints_per_s": rate(n, lambda: one(r, w), 100), "callers": 1}
    out["verify_pageable_4_callers"] = {"constraints_per_s": rate(n, lambda: one(r, w), 50, 4), "callers": 4}
    with Pinned(wp):
        out["verify_pinned"] = {"constraints_per_s": rate(n, lambda: one(r, wp), 100), "callers": 1}
        out["verify_pinned_4_callers"] = {"constraints_per_s": rate(n, lambda: one(r, wp), 50, 4), "callers": 4}
    W = np.ascontiguousarray(np.broadcast_to(w, (many,) + w.shape))

    def vm(buf):
        ok, _, _ = r.verify_many(buf)
        assert ok.all(), "verify_many rejected a satisfying witness"

    vm(W)
    t0 = time.perf_counter(); vm(W); vm(W); dt = (time.perf_counter() - t0) / 2
    out["verify_many_pageable"] = {"witnesses": many, "constraints_per_s": n * many / dt, "host_GB_per_s": W.nbytes / dt * 1e-9}
    with Pinned(W):
        vm(W)
        t0 = time.perf_counter(); vm(W); vm(W); dt = (time.perf_counter() - t0) / 2
        out["verify_many_pinned"] = {"witnesses": many, "constraints_per_s": n * many / dt, "host_GB_per_s": W.nbytes / dt * 1e-9}
    if c3 is not None:          # configs[2]'s size: 33 MB of witness per call
        big = {"workload": f"the same on the 2^{c3.log_n}-constraint system: witness {c3.r.m * 32 / 1e6:.1f} MB per call",
               "verify_pageable": {"constraints_per_s": rate(c3.r.n, lambda: one(c3.r, c3.w), 10)}}
        w2 = c3.w.copy()
        with Pinned(w2):
            big["verify_pinned"] = {"constraints_per_s": rate(c3.r.n, lambda: one(c3.r, w2), 10)}

        def hx():
            h, ok = c3.r.qap_h(c3.w)
            assert ok and h.shape[0] > 0, "host-buffer h(x) rejected a satisfying witness"
        # verificationWitness through host buffers: witness in, the quotient's N coefficients out (acx_qap_h)
        big["qap_h_host_buffers"] = {"constraints_per_s": rate(c3.r.n, hx, 3)}
        out["configs2"] = big
    out["note"] = "PCIe-inclusive wall clock per blocking C-ABI call (H2D + conversion + launch + verdict D2H); the device-resident launch is the headline"
    return out


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


def bench_field_subrun(field, device, copies, log_n, prewarm, c3_future):
    """BASELINE.json configs[4] (the field swap: BLS12-381 Fr) observed in the SAME run as the headline: the batched
    verification launch of the headline workload over that field (same generator, same seeds), one 2^20-point transform and
    h(x) of a 2^20-constraint system, each parity-gated against the C oracle like the headline's own objects."""
    from oracle.c_oracle import COracle
    orc = COracle(field)
    ctx = acx.Context(field, device)
    stream = torch.cuda.ExternalStream(ctx.stream)
    n = 1 << log_n
    systems, witnesses, alg = [], [], 0
    for c in range(copies):
        s = synth.mulgraph(n, seed=0xAC355 + c, field=field)
        w = s.witness()
        systems.append(s.circuit.to_r1cs(ctx))
        witnesses.append(to_dev(ctx, w))
        if c == 0:
            mats0, w0 = s.rows(), w
        alg += algorithmic_bytes(s.rows(), n)[0]
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    batch = acx.Batch(ctx, systems, [w.data_ptr() for w in witnesses], res.data_ptr())
    us = _timed(stream, batch.verify_dev, 30, prewarm)
    ctx.sync()
    accepted = int(res[0]) == 0
    wb = w0.copy()
    wb[77, 0] ^= np.uint64(1)
    want_res, want_bad, want_first = orc.r1cs_residuals(n, systems[0].m, *mats0, wb, nthreads=effective_cpus())
    parity = bool(accepted and want_bad > 0 and np.array_equal(systems[0].residuals(wb), want_res) and systems[0].verify(wb) == (False, want_bad, want_first))
    out = {"config": f"configs[4]: the field swap -- {field} Fr on the headline workload ({copies} x 2^{log_n}-constraint mulgraph systems per launch) and on configs[2]'s 2^20 sizes",
           "r1cs_verify": {"us_per_launch": us, "constraints_per_s": copies * n / us * 1e6, "parity_vs_oracle": parity,
                           "roofline": _hbm(alg, us, kernel="acx::k_r1cs_sell_split<Bls12381Fr>" if field == "bls12_381" else "acx::k_r1cs_sell_split<Bn254Fr>")}}
    del batch
    systems.clear(); witnesses.clear()
    out["ntt"] = bench_ntt(ctx, stream, field, prewarm=prewarm, batch=0)
    c3 = c3_future.result().load(ctx)
    out["qap_h"] = bench_qap_h(ctx, stream, c3, reps=5, prewarm=prewarm)
    return out


def run_with_deadline(fn, seconds, device):
    """fn() on a helper thread with a deadline: a collective that one rank never joins blocks inside the runtime and cannot be
    interrupted -- the line (already holding the headline value) must still be printed.  Returns (result, None) or
    (None, "timeout") / (None, error text); after a timeout the caller prints its line and leaves with os._exit."""
    import threading
    box = {}

    def body():
        try:
            torch.cuda.set_device(device)
            box["r"] = fn()
        except BaseException as e:
            box["e"] = (type(e).__name__ + ": " + str(e))[:400]
    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return None, "timeout"
    return box.get("r"), box.get("e")


def bench_gate_mix(ctx, field="bn254", n_gates=60000):
    """The reference's OWN circuit shape (test/Test/Circuit/Arithmetic.hs:69-136: Mul : Equal : Split = 50 : 10 : 1, 256-bit Split) at
    60 000 gates: `generateAssignment` on the GPU (level-parallel: Equal gates invert, Split gates extract bits) beside the host fold,
    and `verifyAssignment` with ~1000 rows of 257 entries on the long-row path.  Parity: the GPU witness equals the host fold's and
    satisfies the ORACLE's check; residual vector, violated-row count and first row under a corrupted witness equal the oracle's."""
    from oracle.c_oracle import COracle
    orc = COracle(field)
    s = synth.gatemix(n_gates, field=field)
    c = s.circuit
    mats = s.rows()
    r = c.to_r1cs(ctx)

    def wall(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    w_host, _ = c.eval(s.inputs)
    gw, _ = r.eval_witness(s.inputs)
    _, nbad, _ = orc.r1cs_residuals(r.n, r.m, *mats, gw, want_residuals=False, nthreads=effective_cpus())
    wb = gw.copy()
    wb[1 + c.n_inputs + 7, 0] ^= np.uint64(1)            # an early intermediate wire (a magic wire behind a zero input would go unnoticed)
    want_res, want_bad, want_first = orc.r1cs_residuals(r.n, r.m, *mats, wb, nthreads=effective_cpus())
    assert want_bad > 0
    parity = bool(np.array_equal(gw, w_host) and nbad == 0 and r.verify(gw)[0] and np.array_equal(r.residuals(wb), want_res)
                  and r.verify(wb) == (False, want_bad, want_first))
    fmt = r.format()
    t_verify = wall(lambda: r.verify_resident(), 20)
    return {"workload": f"{n_gates} gates in the reference's generator mix (Mul : Equal : Split = 50 : 10 : 1, 256-bit Split; {field} Fr): "
                        f"{r.n} constraints, m = {r.m} wires, {fmt[2]} rows on the long-row path",
            "parity_vs_oracle": parity,
            "generateAssignment": {"gpu_acx_r1cs_eval_s": wall(lambda: r.eval_witness(s.inputs, download=False), 5),
                                   "host_acx_circuit_eval_s": wall(lambda: c.eval(s.inputs), 3)},
            "verifyAssignment": {"resident_s": t_verify, "constraints_per_s": r.n / t_verify,
                                 "host_buffer_s": wall(lambda: r.verify(gw), 20)}}


def bench_distributed(ctx, a, world, rank, dist):
    """configs[3] beside the headline (N > 1, or --force-dist on one GPU): the distributed four-step NTT at
    N = 2^24 (one all-to-all per transform) and the distributed h(x) pipeline on a 2^24-constraint block system
    (256 x 2^16 mulgraph blocks, rows marshalled per rank in block-cyclic ownership; 6 all-to-alls + 1 all-reduce: O(x) stays in coefficient form).
    Times are max over ranks.  Parity gates: transform round trip at 2^24 plus a full oracle comparison of the same
    code path at 2^16; the pipeline must accept the satisfying witness and reject a corrupted one."""
    par = importlib.import_module("arithmetic-circuits_amd.parallel")
    from oracle.c_oracle import COracle
    if os.environ.get("ACX_BENCH_HANG_DIST") == "1" and rank == world - 1:     # test hook: one rank never joins the collectives
        time.sleep(1e6)
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

    def qap_h_resident():
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
        return {"workload": f"acx_mgpu_qap_h_resident: verificationWitness h(x) of the same 2^{ln}-constraint system over {W} shard(s): "
                            f"block-cyclic rows, 6 all-to-alls + 1 all-reduce per call, blocking call",
                "accepts_valid_rejects_corrupt": bool(ok and not ok_bad), "us": sec * 1e6, "constraints_per_s": bs.n / sec}

    if a.want("dist") and mr.n == 1 << ln and ln <= 24:
        r = guard("mgpu_qap_h", qap_h_resident)
        if r is not None:
            out["mgpu_qap_h"] = r
    if a.want("cpu"):           # the same CPU baseline as the one-process-per-GPU line: one block of this system on the host's cores
        r = guard("cpu", cpu_baseline, (bs.mats, np.concatenate([w[:1], w[-(bs.m0 - 1):]]), n0, bs.m0), a.field, 10.0, True)
        if r is not None:
            out["cpu_baseline"] = r
    if ERRORS:
        out["errors"] = ERRORS
    PHASES["total"] = round(time.perf_counter() - T_START, 3)
    out["phases_s"] = PHASES
    mr.close()
    mg.close()
    sys.stdout.flush()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500, help="timed steps: 500 launches = a 56 ms region at the sustained clock (the review of round 3: a 50-step region is 5.6 ms)")
    ap.add_argument("--warmup", type=int, default=20)
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
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc FETCH_SIZE pass (roofline.traffic then comes from profiles/r04_traffic.json, flagged)")
    ap.add_argument("--only-steps", action="store_true", help="internal: nothing but the batched launches (the child of the PMC pass)")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--only-ntt", action="store_true", help="internal: nothing but 40 single 2^20-point transforms (the child of the NTT counter pass)")
    ap.add_argument("--skip", default="", help="comma list of secondary objects to skip: pmc,ntt,qap_h,small,ref,gatemix,load,cols,e2e,field2,cpu,dist or all")
    ap.add_argument("--only", default="", help="comma list of secondary objects to run (the rest is skipped)")
    ap.add_argument("--dist-deadline", type=float, default=240.0, help="seconds the distributed extras of a multi-rank run may take before the line is printed without them")
    ap.add_argument("--launcher", default="auto", choices=["auto", "ranks", "mgpu"],
                    help="ranks: one process per GPU (torch.distributed.run, what the driver uses); mgpu: ONE process, all GPUs "
                         "through acx_mgpu_*; auto: mgpu when --gpus N > 1 is asked for outside torch.distributed.run")
    ap.add_argument("--mgpu-devices", default=None,
                    help="device ordinals of the mgpu launcher, e.g. 0,1,2,3 or 0,0 (repeats = several shards on one GPU); default 0..N-1")
    a = ap.parse_args()
    skip, only = set(filter(None, a.skip.split(","))), set(filter(None, a.only.split(",")))
    if a.no_cpu: skip.add("cpu")
    if a.no_pmc: skip.add("pmc")
    if a.no_ntt: skip |= {"ntt", "qap_h", "small", "load", "cols"}
    if a.no_dist_pipeline: skip.add("dist")
    a.want = lambda name: "all" not in skip and name not in skip and (not only or name in only)
    if a.only_ntt:
        return only_ntt(a)
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
    blk = min(a.steps, 50)          # block length of the sustained statistics (many short blocks: median / min / max mean something)
    if not use_dist and a.sustain > 0:
        with torch.cuda.stream(stream):
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < a.sustain:
                evs = []
                for _ in range(16):
                    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    b0.record(stream)
                    for i in range(blk):
                        step(i)
                    b1.record(stream)
                    evs.append((b0, b1))
                stream.synchronize()
                block_us += [x.elapsed_time(y) * 1e3 / blk for x, y in evs]
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

    PHASES["setup_and_headline"] = round(time.perf_counter() - T_START, 3)
    per_rank_us = [kernel_us]
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
        allk = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(allk, torch.tensor([kernel_us], dtype=torch.float64, device="cuda"))
        per_rank_us = [float(x[0]) for x in allk]
    dist_extra, timed_out = {}, False
    if use_dist and a.want("dist") and not a.only_steps:
        del batches, neg_batch
        systems.clear(); witnesses.clear()
        # under a deadline: a rank that fails inside a collective leaves the others blocked in the runtime
        t0 = time.perf_counter()
        got, err = run_with_deadline(lambda: bench_distributed(ctx, a, world, rank, dist), a.dist_deadline, local_rank)
        PHASES["dist"] = round(time.perf_counter() - t0, 3)
        dist_extra = got or {}
        if err:
            ERRORS.append({"where": f"bench_distributed (rank {rank})", "error": err})
            timed_out = err == "timeout"
    line = None
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
        if world > 1:       # every rank's own launch time (each rank streams its own 32 systems): the per-rank roofline fractions
            out["roofline"]["per_rank_kernel_us"] = per_rank_us
            out["roofline"]["per_rank_frac"] = [bytes_per_launch / u * 1e-3 / HBM_PEAK_GBS for u in per_rank_us]
        if resident_us:
            out["cache_resident"] = {"us_per_launch": resident_us, "constraints_per_s": a.copies * n / resident_us * 1e6,
                                     "single_system_us_per_launch": single_us,
                                     "note": f"labelled, NOT the metric: the same 2^{a.logn}-constraint system {a.copies} times per launch (and once per launch): "
                                             "its constraint stream stays in L2 / Infinity Cache; the headline rotates over independent systems from HBM"}
        out.update(dist_extra)
        if block_us:
            bs = sorted(block_us)
            out["sustained"] = {"blocks": len(bs), "steps_per_block": blk, "seconds": sum(bs) * blk * 1e-6,
                                "us_per_step_median": bs[len(bs) // 2], "us_per_step_min": bs[0], "us_per_step_max": bs[-1],
                                "frac_median": bytes_per_launch / bs[len(bs) // 2] * 1e-3 / HBM_PEAK_GBS,
                                "note": "kernel times at the sustained clock (0.25 s untimed pre-run before the timed region)"}
        if not a.only_steps and not timed_out:
            guard("secondary", secondary_objects, a, out, ctx, stream, world, use_dist, local_rank, systems, witnesses, sample, kernel_us)
        if ERRORS:
            out["errors"] = ERRORS
        PHASES["total"] = round(time.perf_counter() - T_START, 3)
        out["phases_s"] = PHASES
        line = json.dumps(out)
    if use_dist and not timed_out:
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
    if timed_out:               # a helper thread is still blocked inside a collective: no orderly teardown is possible
        sys.stdout.flush()
        os._exit(0)


def secondary_objects(a, out, ctx, stream, world, use_dist, device, systems, witnesses, sample, kernel_us):
    """Everything of the line beside the headline, rank 0 only, each object under guard().  One GPU: all of it.  Several
    ranks: the CPU baseline (short), the transform and the per-rank fractions -- the other ranks wait at the closing barrier."""
    from concurrent.futures import ThreadPoolExecutor
    single = world == 1 and not use_dist
    pool = ThreadPoolExecutor(2)
    # host-only preparation of the two 2^20-gate circuits (numpy + acx_circuit_create, ~7 s each) runs on helper threads
    # while the counter passes below keep this process waiting on its children
    need_c3 = single and any(a.want(x) for x in ("qap_h", "load", "cols", "e2e"))
    other = "bls12_381" if a.field == "bn254" else "bn254"
    c3_f = pool.submit(lambda: C3System(a.field).prepare()) if need_c3 else None
    c3o_f = pool.submit(lambda: C3System(other).prepare()) if single and a.want("field2") else None

    live = guard("pmc_fetch", measure_traffic_live, a) if single and a.want("pmc") else None
    if live is not None:
        # HBM bytes of the headline kernel measured in THIS run: a rocprofv3 --pmc FETCH_SIZE pass of its own over the
        # same batched launches in a child process (counters need the profiler around the process)
        out["roofline"]["traffic"] = live[0]
        out["roofline"]["traffic_measured_in_run"] = True
        out["roofline"]["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE around a child run of this script's batched launches ({live[1]} launches "
                                             "averaged); bytes = FETCH_SIZE (KiB) * 1024 * 2 (gfx950 reports half of wide coalesced reads)")
        out["roofline"]["achieved_traffic"] = live[0] / kernel_us * 1e-3
        out["roofline"]["frac_traffic"] = out["roofline"]["achieved_traffic"] / HBM_PEAK_GBS
    live_valu = guard("pmc_valu", measure_counter_live, a, "SQ_INSTS_VALU") if live is not None else None      # second pass, its own run
    ntt_valu = guard("pmc_ntt_valu", measure_ntt_valu_live, a) if live is not None and a.want("ntt") else None  # third: the transform's
    try:    # PMC-measured HBM bytes per launch, recorded from a separate rocprofv3 --pmc pass
        tr = json.load(open(os.path.join(ROOT, "profiles", "r04_traffic.json")))["acx::k_r1cs_sell"]
        same = tr["workload"] == {"field": a.field, "copies": a.copies, "logn": a.logn}
        if live is not None:
            if same:
                out["roofline"]["traffic_committed_pass"] = tr["traffic_bytes_per_launch"]       # the earlier pass, for comparison
            out["roofline"]["valu_issue"] = _valu_issue("acx::k_r1cs_sell", "valu_wave_insts_per_launch", kernel_us, {"field": a.field, "copies": a.copies, "logn": a.logn},
                                                        live_valu[0] / live_valu[1] if live_valu else None)
        elif same:
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

    def put(key, where, fn, *args, **kw):
        r = guard(where, fn, *args, **kw)
        if r is not None:
            out[key] = r
        return r

    if a.want("ntt"):
        put("ntt", "ntt", bench_ntt, ctx, stream, a.field, prewarm=a.prewarm, live_valu=ntt_valu, batch=64 if single else 0)
    if single and a.want("small"):
        put("r1cs_small_coeff", "small", bench_small_coeff, ctx, stream, a.field, copies=a.copies, log_n=a.logn, prewarm=a.prewarm)
    if single and a.want("ref"):
        put("reference_bench", "ref", reference_bench, ctx, a.field)
    if single and a.want("gatemix"):
        put("gate_mix", "gatemix", bench_gate_mix, ctx, a.field)
    c3 = None
    if need_c3:
        c3 = guard("c3_load", lambda: c3_f.result().load(ctx))
    if c3 is not None:
        if a.want("load"):
            put("load", "load", bench_load, c3)
        if a.want("qap_h"):
            put("qap_h", "qap_h", bench_qap_h, ctx, stream, c3, prewarm=a.prewarm)
        if a.want("cols"):
            put("qap_columns", "cols", bench_qap_columns, ctx, stream, c3)
    if single and a.want("e2e") and systems:
        put("e2e", "e2e", bench_e2e, systems[0], sample[1], c3)
    if c3 is not None:
        c3.r.close()
        c3 = None
    if single and a.want("field2"):
        put(other, "field2", bench_field_subrun, other, device, a.copies, a.logn, a.prewarm, c3o_f)
    if a.want("cpu"):
        put("cpu_baseline", "cpu", cpu_baseline, sample, a.field, 10.0 if single else 4.0, single)
    pool.shutdown(wait=False)


if __name__ == "__main__":
    main()
