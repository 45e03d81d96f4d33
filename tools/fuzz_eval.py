#!/usr/bin/env python3
"""Differential fuzz of acx_r1cs_eval (`generateAssignment` on the GPU, level by level) against the reference's sequential
fold: the oracle's literal evalArithCircuit (oracle/ref_qap.py, small circuits) and the product's host fold acx_circuit_eval
(all sizes).  Circuits in the reference's generator shapes (test/Test/Circuit/Arithmetic.hs:69-126) with what its generator
never produces added: Split gates of every width (1 .. 300), Equal gates on zero, gates that READ an Equal gate's magic wire
(evalGate allows it, validArithCircuit does not: the inversions then stay inside the levels), absent inputs, and the same
circuits through the one-lane-per-gate kernel of wide levels (ACX_EVAL_LANES_BELOW=0 ACX_EVAL_FUSED=0 in a second process:
without the second variable runs of narrow levels still go to k_eval_levels_fused) and through the launch-per-level form of
the lanes kernel (ACX_EVAL_FUSED=0 alone).
    python tools/fuzz_eval.py [cases] [first]"""
import importlib, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
acx = importlib.import_module("arithmetic-circuits_amd")
synth = acx.synth
from oracle import ref_qap as R
import helpers as H


def arb_circuit(rnd, p, n_in, size):
    """incremental builder like arbArithCircuit, with free Split widths and optional reads of magic wires"""
    gates, readable, magic, nxt = [], [], [], 0
    read_magic = rnd.random() < 0.3
    for _ in range(size):
        pool = readable + (magic if read_magic else [])
        kinds = ["mul"] * 6 + (["equal"] * 3 + ["split"] if readable else [])
        k = rnd.choice(kinds)
        if k == "mul":
            gates.append(R.Mul(H.arb_affine_with_mids(rnd, p, n_in, pool, rnd.randrange(0, 3)),
                               H.arb_affine_with_mids(rnd, p, n_in, pool, rnd.randrange(0, 3)), R.IntermediateWire(nxt)))
            readable.append(nxt)
            nxt += 1
        elif k == "equal":
            gates.append(R.Equal(R.IntermediateWire(rnd.choice(readable)), R.IntermediateWire(nxt), R.IntermediateWire(nxt + 1)))
            magic.append(nxt)
            readable.append(nxt + 1)
            nxt += 2
        else:
            wd = rnd.choice([1, 2, 31, 32, 33, 64, 100, 255, 256, 256, 256, 257, 300])
            gates.append(R.Split(R.IntermediateWire(rnd.choice(readable)), [R.IntermediateWire(nxt + j) for j in range(wd)]))
            readable += list(range(nxt, nxt + min(wd, 4)))        # later gates read a few of the bits
            nxt += wd
    return gates


def main(cases, first):
    ctxs = {f: acx.Context(f, 0) for f in ("bn254", "bls12_381")}
    bad, t0 = 0, time.time()
    for seed in range(first, first + cases):
        rnd = random.Random(83000 + seed)
        field = rnd.choice(list(ctxs))
        ctx = ctxs[field]
        p = ctx.p
        tag = f"seed {seed} {field}"
        try:
            if seed % 8 == 7:
                # the reference's generator mix at size, flat arrays (host fold only: the oracle is Python)
                n = rnd.choice([2000, 20000, 60000])
                bits = rnd.choice([256, 64, 300])
                s = synth.gatemix(n, n_in=rnd.choice([1, 8, 64]), seed=seed, field=field, weights=rnd.choice([(50, 10, 1), (10, 10, 1), (50, 0, 5)]),
                                  split_bits=bits, window=rnd.choice([16, 4096]))
                tag += f" gatemix n={n}"
                r = s.circuit.to_r1cs(ctx)
                want, want_as = s.circuit.eval(s.inputs)
                got, got_as = r.eval_witness(s.inputs)
                assert np.array_equal(got, want) and np.array_equal(got_as, want_as), "witness"
                ok = r.verify_resident()[0]
                assert ok == r.verify(want)[0] and (ok or bits < 255), "verify"
                r.close()
                continue
            n_in = rnd.randrange(1, 6)
            gates = arb_circuit(rnd, p, n_in, rnd.randrange(1, 60))
            program = H.to_acx_circuit(acx, gates)
            circ = program.marshal(field)
            r = circ.to_r1cs(ctx)
            tag += f" gates={len(gates)}"
            for t in range(3):
                vals = [rnd.choice([0, 1, p - 1, rnd.randrange(p)]) for _ in range(n_in)]
                arr = acx.ints_to_fr(vals)
                pres = np.array([rnd.random() < 0.7 for _ in range(n_in)], dtype=np.uint8) if t == 2 else None
                want = status = None
                try:
                    want, want_as = circ.eval(arr, pres)
                except acx.AcxError as e:                          # an Equal / Split gate on an absent input: the reference panics
                    status = e.status
                if want is None:
                    try:
                        r.eval_witness(arr, pres)
                        raise AssertionError("the host fold failed, the device did not")
                    except acx.AcxError as e:
                        assert e.status == status, "error status"
                    continue
                got, got_as = r.eval_witness(arr, pres)
                assert np.array_equal(got, want) and np.array_equal(got_as, want_as), f"witness (inputs {t})"
                if pres is None:
                    oracle_w = H.qapset_to_flat(R.generate_assignment(gates, dict(enumerate(vals)), p), H.circuit_dims(gates), p)
                    assert acx.fr_to_ints(got) == oracle_w, "oracle"
                    # a Split gate narrower than its input's bit length evaluates (low bits) but does not satisfy its constraint
                    fits = all(len(g[2]) >= 255 for g in gates if g[0] == "split")
                    ok = r.verify_resident()[0]
                    assert ok == r.verify(want)[0] and (ok or not fits), "verify"
            r.close()
        except AssertionError as e:
            bad += 1
            print("MISMATCH", tag, e, flush=True)
    print(f"fuzz_eval: {cases} cases from seed {first}, {bad} failures, {time.time() - t0:.0f} s"
          + "".join(f" ({k}={os.environ[k]})" for k in ("ACX_EVAL_LANES_BELOW", "ACX_EVAL_FUSED") if k in os.environ))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
