#!/usr/bin/env python3
"""Differential fuzz of the acx_mgpu_* entry points against the C oracle: random shard counts (repeated device 0: peer-copy
transport; one shard: RCCL), fields, system sizes around the shard threshold and the padding boundaries, random loads through
both loaders (circuit handle and raw CSR with arbitrary -- unsatisfiable -- matrices), corrupted witnesses, zero-knowledge
shifts, per-wire columns on random ranges, single transforms.  Run on an MI355X:  python tools/fuzz_mgpu.py [seeds] [first]"""
import importlib, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = acx.synth
from oracle.c_oracle import COracle
from oracle import ref_qap as R

U64 = 2**64 - 1


def random_csr(rs, rnd, n, m):
    mats = []
    for k in range(3):
        maxlen = min(m, rnd.choice([1, 2, 3, 6, 7, 9, 40]))
        lens = rs.randint(0, maxlen + 1, size=n)
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        col = np.concatenate([np.sort(rs.choice(m, size=l, replace=False)) for l in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.uint32)
        mats.append((rowptr, col, None))
    return mats


def main(seeds, first):
    orcs = {f: COracle(f) for f in ("bn254", "bls12_381")}
    mgs = {}
    bad = 0
    t0 = time.time()
    for seed in range(first, first + seeds):
        rnd, rs = random.Random(31000 + seed), np.random.RandomState(32000 + seed)
        field = rnd.choice(["bn254", "bls12_381"])
        W = rnd.choice([1, 2, 4, 8])
        key = (field, W)
        if key not in mgs:
            mgs[key] = acx.MultiGpu(field, [0] * W)
        mg, orc = mgs[key], orcs[field]
        p = R.BN254.p if field == "bn254" else R.BLS12_381.p
        thr = rnd.choice([10, 10, 11, 12, 14])
        mg.set_shard_threshold(thr)
        what = rnd.choice(["circuit", "circuit", "csr", "ntt"])
        tag = f"seed {seed} {field} W={W} thr={thr} {what}"
        try:
            if what == "ntt":
                ln = rnd.randrange(8, 17)
                x = synth.random_fr(1 << ln, seed, 1, field)
                inv = rnd.random() < 0.5
                sh = rnd.choice([None, 5 if field == "bn254" else 7, rnd.randrange(2, p)])
                got = mg.ntt(x, ln, inverse=inv, shift=sh)
                want = orc.ntt(x, ln, inverse=inv, shift=sh, nthreads=8)
                assert np.array_equal(got, want), "ntt"
                continue
            n = rnd.choice([1023, 1024, 1025, 2048, 4095, 4097, 8191, 8192, rnd.randrange(900, 20000), rnd.randrange(900, 5000)])
            if what == "circuit":
                s = synth.mulgraph(n, n_in=rnd.choice([1, 7, 64, 300]), window=rnd.choice([16, 200, 4096]), seed=seed, field=field)
                mats, w = s.rows(), s.witness()
                mr = mg.from_circuit(s.circuit, verify_only=rnd.random() < 0.15)
                m = s.circuit.m
                satisfiable = True
            else:
                m = rnd.choice([3, 64, 700, 5000])
                mats = random_csr(rs, rnd, n, m)
                mats = [(rp, cl, synth.random_fr(cl.shape[0], seed * 3 + k, 1, field)) for k, (rp, cl, _) in enumerate(mats)]
                w = synth.random_fr(m, seed + 77, 1, field)
                mr = mg.load(n, m, *mats, verify_only=rnd.random() < 0.15)
                satisfiable = False
            ln = mr.log_n
            res, nbad, firstbad = orc.r1cs_residuals(n, m, *mats, w, want_residuals=False, nthreads=8)
            assert mr.verify(w) == (nbad == 0, nbad, firstbad), "verify"
            wb = w.copy()
            wb[rnd.randrange(m), rnd.randrange(4)] ^= np.uint64(1 << rnd.randrange(20))
            if int(wb[:, 3].max()) >> 62:       # keep it canonical
                wb = w.copy(); wb[rnd.randrange(m), 0] ^= np.uint64(1)
            _, nbad2, first2 = orc.r1cs_residuals(n, m, *mats, wb, want_residuals=False, nthreads=8)
            assert mr.verify(wb) == (nbad2 == 0, nbad2, first2), "verify corrupted"
            oks, nbads = mr.verify_many(np.stack([w, wb, w]))
            assert oks.tolist() == [nbad == 0, nbad2 == 0, nbad == 0] and nbads.tolist() == [nbad, nbad2, nbad], "verify_many"
            can_h = True
            try:
                delta = rnd.choice([None, [rnd.randrange(p) for _ in range(3)]])
                h, ok = mr.qap_h(w, delta)
            except acx.AcxError as e:
                can_h = False
                assert "UNSUPPORTED" in str(e) or "unsupported" in str(e).lower() or "block-cyclic" in str(e), f"qap_h error {e}"
            if can_h:
                want_h, want_ok = orc.qap_h(n, m, ln, *mats, w, delta=delta, nthreads=8)
                assert ok == want_ok == satisfiable or (not satisfiable and ok == want_ok), "qap_h ok"
                if ok:
                    assert np.array_equal(h, want_h[: h.shape[0]]) and not want_h[h.shape[0]:].any(), "qap_h"
            k = rnd.randrange(3)
            cnt = rnd.choice([1, 2, 7, 33])
            w0 = rnd.randrange(0, max(1, m - cnt + 1))
            cnt = min(cnt, m - w0)
            cols, lens = mr.qap_columns(k, w0, cnt)
            assert np.array_equal(cols, orc.qap_columns(n, ln, mats[k], w0, cnt, nthreads=8)), "columns"
            mr.close()
        except AssertionError as e:
            bad += 1
            print("MISMATCH", tag, e, flush=True)
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e), flush=True)
    print(f"fuzz_mgpu: {seeds} cases from seed {first}, {bad} failures, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
