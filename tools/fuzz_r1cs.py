#!/usr/bin/env python3
"""One-off differential fuzz of acx_r1cs_load / residuals / verify against the C oracle on random sparse
systems of random shapes (run on an MI355X: python tools/fuzz_r1cs.py [seeds])."""
import importlib, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
from oracle.c_oracle import COracle

def main(seeds):
    bad = 0
    for field in ("bn254", "bls12_381"):
        ctx, orc = acx.Context(field, 0), COracle(field)
        p = ctx.p
        for seed in range(seeds):
            rs, rnd = np.random.RandomState(7000 + seed), random.Random(9000 + seed)
            n = rnd.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 257, 1000, 4095, 4096, 4097, rnd.randrange(1, 9000)])
            m = rnd.choice([1, 2, 5, 64, 300, 5000])
            unit_c = rnd.random() < 0.5
            mats = []
            for k in range(3):
                maxlen = min(m, rnd.choice([1, 2, 3, 6, 7, 8, 9, 13, 20, 30, 60, 300]))
                lens = rs.randint(0, maxlen + 1, size=n)
                if rnd.random() < 0.2:
                    lens[:] = 0 if rnd.random() < 0.5 else maxlen
                rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
                col = np.concatenate([np.sort(rs.choice(m, size=l, replace=False)) for l in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.uint32)
                nnz = int(rowptr[-1])
                small = rnd.random() < 0.4          # the small-coefficient form of a matrix (|c| <= 2^27), boundary included
                if k == 2 and unit_c:
                    vals = [1] * nnz
                elif small:
                    B = 1 << 27
                    vals = [(c if rnd.random() < 0.5 else (p - c) % p) for c in
                            (rnd.choice([0, 1, 2, B, B - 1, rnd.randrange(B), rnd.randrange(256)]) for _ in range(nnz))]
                    if rnd.random() < 0.3 and nnz:   # one entry just over the boundary: must fall back to the value stream
                        vals[rnd.randrange(nnz)] = rnd.choice([B + 1, p - B - 1])
                else:
                    vals = [rnd.choice([0, 1, p - 1, p - 2]) if rnd.random() < 0.2 else rnd.randrange(p) for _ in range(nnz)]
                mats.append((rowptr, col, acx.ints_to_fr(vals) if nnz else np.zeros((0, 4), dtype=np.uint64)))
            w = acx.ints_to_fr([1] + [rnd.randrange(p) for _ in range(m - 1)])
            r = acx.R1CS.load(ctx, n, m, *mats)
            want, nbad, first = orc.r1cs_residuals(n, m, *mats, w, nthreads=4)
            got = r.residuals(w)
            ok = np.array_equal(got, want) and r.verify(w) == (nbad == 0, nbad, first)
            if not ok:
                bad += 1
                print(f"MISMATCH field={field} seed={seed} n={n} m={m} unit_c={unit_c}")
    print("fuzz done, mismatches:", bad)
    return bad

if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 60) else 0)
