#!/usr/bin/env python3
"""One-off differential fuzz of acx_ntt (forward / inverse / coset, batched) against the C oracle
(run on an MI355X: python tools/fuzz_ntt.py [cases])."""
import importlib, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
from oracle.c_oracle import COracle

def main(cases):
    bad = 0
    for field in ("bn254", "bls12_381"):
        ctx, orc = acx.Context(field, 0), COracle(field)
        rnd = random.Random(4242)
        for i in range(cases):
            log_n = rnd.choice(list(range(0, 15)) + [16, 17, 18])
            batch = rnd.choice([1, 1, 2, 3, 5]) if log_n <= 14 else 1
            inverse = rnd.random() < 0.5
            shift = rnd.randrange(1, ctx.p) if rnd.random() < 0.4 else None
            x = synth.random_fr((1 << log_n) * batch, 100 + i, log_n, field)
            got = ctx.ntt(x, log_n, inverse=inverse, shift=shift)
            want = np.concatenate([orc.ntt(x[b << log_n:(b + 1) << log_n], log_n, inverse=inverse, shift=shift, nthreads=4)
                                   for b in range(batch)])
            if not np.array_equal(got, want):
                bad += 1
                print(f"MISMATCH field={field} case={i} log_n={log_n} batch={batch} inverse={inverse} shift={shift is not None}")
    print("ntt fuzz done, mismatches:", bad)
    return bad

if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 60) else 0)
