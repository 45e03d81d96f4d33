#!/usr/bin/env python3
"""Differential fuzz of acx_qap_h (single GPU: every transform plan between 2^10 and 2^19, padding rows, both fields,
zero-knowledge shifts, corrupted witnesses) against the C oracle.  python tools/fuzz_h.py [cases] [first]"""
import importlib, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = acx.synth
from oracle.c_oracle import COracle
from oracle import ref_qap as R


def main(cases, first):
    ctxs = {f: acx.Context(f, 0) for f in ("bn254", "bls12_381")}
    orcs = {f: COracle(f) for f in ctxs}
    bad, t0 = 0, time.time()
    threads = min(64, os.cpu_count() or 8)
    for seed in range(first, first + cases):
        rnd = random.Random(51000 + seed)
        field = rnd.choice(list(ctxs))
        p = R.BN254.p if field == "bn254" else R.BLS12_381.p
        ln = rnd.randrange(10, 20)
        n = rnd.choice([1 << ln, (1 << ln) - rnd.randrange(1, 1 << (ln - 1)), (1 << (ln - 1)) + 1])
        s = synth.mulgraph(n, n_in=rnd.choice([3, 64, 1024]), window=rnd.choice([64, 4096]), seed=seed, field=field)
        mats, w = s.rows(), s.witness()
        r = s.circuit.to_r1cs(ctxs[field])
        delta = rnd.choice([None, None, [rnd.randrange(p) for _ in range(3)]])
        tag = f"seed {seed} {field} n={n} log_n={r.log_n} delta={'yes' if delta else 'no'}"
        try:
            h, ok = r.qap_h(w, delta=delta)
            want, want_ok = orcs[field].qap_h(n, r.m, r.log_n, *mats, w, delta=delta, nthreads=threads)
            assert ok and want_ok, "ok"
            assert np.array_equal(h, want[: h.shape[0]]) and not want[h.shape[0]:].any(), "h"
            wb = w.copy()
            wb[rnd.randrange(1, r.m), 0] ^= np.uint64(1)
            _, nbad, _ = orcs[field].r1cs_residuals(n, r.m, *mats, wb, want_residuals=False, nthreads=threads)
            hb, okb = r.qap_h(wb, delta=delta)
            assert okb == (nbad == 0) and (hb is None) == (nbad != 0), "corrupted"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", tag, e, flush=True)
        r.close()
    print(f"fuzz_h: {cases} cases from seed {first}, {bad} failures, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
