#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the literal restatement oracle/ref_qap.py.

The reference is Haskell and cannot run here (no GHC), and its own tests pin only Bool
results, so these fixtures hold (1) the reference tests' KAT inputs with their expected Bools
(from /root/reference/test/Test/QAP.hs:48-90, Example.hs:10-38, bench/Circuit.hs:17-24) and
(2) DERIVED values (polynomial coefficients, h(x), NTT outputs) computed by the literal
restatement -- data only, no reference source text.   Run:  python tests/golden/gen_golden.py"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_qap as R  # noqa: E402
from tests import helpers as H   # noqa: E402


def hexs(xs):
    return [format(x, "x") for x in xs]


def affine_json(c):
    tag = c[0]
    if tag == "var":
        return {"var": [c[1].kind, c[1].index]}
    if tag == "const":
        return {"const": format(c[1], "x")}
    if tag == "smul":
        return {"smul": [format(c[1], "x"), affine_json(c[2])]}
    return {"add": [affine_json(c[1]), affine_json(c[2])]}


def gate_json(g):
    if g[0] == "mul":
        return {"mul": [affine_json(g[1]), affine_json(g[2]), [g[3].kind, g[3].index]]}
    if g[0] == "equal":
        return {"equal": [[w.kind, w.index] for w in g[1:4]]}
    return {"split": [[g[1].kind, g[1].index], [[w.kind, w.index] for w in g[2]]]}


def qapset_json(qs):
    return {"constant": format(qs.constant, "x"), "inputs": {str(k): format(v, "x") for k, v in qs.inputs.items()},
            "intermediates": {str(k): format(v, "x") for k, v in qs.intermediates.items()},
            "outputs": {str(k): format(v, "x") for k, v in qs.outputs.items()}}


def case(name, field, gates, roots, assignments, deltas=None):
    p = field.p
    gen = R.arith_circuit_to_gen_qap(roots, gates, p)
    qap = R.create_polynomials_fft(field.root_of_unity, gen, p)
    dims = H.circuit_dims(gates)
    out = {"name": name, "field": field.name, "gates": [gate_json(g) for g in gates],
           "roots": [hexs(r) for r in roots], "dims": list(dims), "target": hexs(qap.target), "assignments": []}
    polys = {}
    for mname, qs in (("A", qap.left), ("B", qap.right), ("C", qap.out)):
        d = {"0": hexs(qs.constant)}
        for kind, part in enumerate((qs.inputs, qs.intermediates, qs.outputs)):
            for idx, poly in part.items():
                d[str(H.flat_index(dims, R.Wire(kind, idx)))] = hexs(poly)
        polys[mname] = d
    out["polys"] = polys
    for a in assignments:
        h = R.verification_witness(qap, a, p)
        rec = {"assignment": qapset_json(a), "flat": hexs(H.qapset_to_flat(a, dims, p)),
               "valid": h is not None, "h": hexs(h) if h is not None else None}
        if deltas:
            hz = R.verification_witness_zk(*deltas, qap, a, p)
            rec["delta"] = hexs(deltas)
            rec["h_zk"] = hexs(hz) if hz is not None else None
        out["assignments"].append(rec)
    return out


def main():
    p = R.BN254.p
    cases = []
    kat = [R.Mul(R.Var(R.InputWire(0)), R.Var(R.InputWire(1)), R.IntermediateWire(0)),
           R.Mul(R.Var(R.InputWire(2)), R.Var(R.InputWire(3)), R.IntermediateWire(1)),
           R.Mul(R.Add(R.ConstGate(10), R.Var(R.IntermediateWire(0))), R.Var(R.IntermediateWire(1)), R.OutputWire(0))]
    good = R.generate_assignment(kat, {0: 2, 1: 3, 2: 4, 3: 5}, p)
    bad = R.QapSet(1, {0: 2, 1: 3, 2: 4, 3: 5}, {0: 7, 1: 20}, {0: 320})
    cases.append(case("test_qap_kat_fft", R.BN254, kat, [[1], [2], [3]], [good, bad], deltas=[3, 5, 7]))
    b = R.CircuitBuilder()
    i0, i1, i2 = ("var", b.input()), ("var", b.input()), ("var", b.input())
    b.ret(("mul", ("mul", i0, i1), ("add", i0, i2)))
    cases.append(case("example_hs", R.BN254, b.gates, R.fresh_roots(b.gates, 1),
                      [R.generate_assignment(b.gates, {0: 7, 1: 5, 2: 4}, p)]))
    bench = [R.Mul(R.Var(R.InputWire(0)), R.Var(R.InputWire(1)), R.IntermediateWire(0)),
             R.Mul(R.Var(R.IntermediateWire(0)), R.Add(R.Var(R.InputWire(0)), R.Var(R.InputWire(2))), R.OutputWire(0))]
    cases.append(case("bench_circuit", R.BN254, bench, R.fresh_roots(bench, 0),
                      [R.generate_assignment(bench, {0: 7, 1: 5, 2: 4}, p)]))
    for fld, seed in ((R.BN254, 11), (R.BN254, 12), (R.BLS12_381, 13)):
        rnd = random.Random(seed)
        nv = 3
        gates = H.arb_arith_circuit(rnd, fld.p, nv, 7, split_bits=4)
        asg = [R.generate_assignment(gates, H.arb_input_vector(rnd, fld.p, nv), fld.p) for _ in range(2)]
        broken = R.generate_assignment(gates, H.arb_input_vector(rnd, fld.p, nv), fld.p)
        k = sorted(broken.intermediates)[0]
        broken.intermediates[k] = (broken.intermediates[k] + 1) % fld.p
        cases.append(case(f"random_{fld.name}_{seed}", fld, gates, R.fresh_roots(gates, 1), asg + [broken],
                          deltas=[rnd.randrange(fld.p) for _ in range(3)]))
    json.dump(cases, open(os.path.join(HERE, "qap_cases.json"), "w"), indent=0)

    ntt = []
    for fld in (R.BN254, R.BLS12_381):
        rnd = random.Random(99)
        for log_n in (0, 1, 3, 5):
            xs = [rnd.randrange(fld.p) for _ in range(1 << log_n)]
            g = fld.generator
            w = fld.root_of_unity(log_n)
            ntt.append({"field": fld.name, "log_n": log_n, "in": hexs(xs),
                        "fft": hexs(R.fft(fld.root_of_unity, xs, fld.p)),
                        "interpolate": hexs(R.inverse_dft(fld.root_of_unity, xs, fld.p)),
                        "shift": format(g, "x"),
                        "coset_fft": hexs([R.poly_eval(xs, g * pow(w, i, fld.p) % fld.p, fld.p) for i in range(1 << log_n)])})
    json.dump(ntt, open(os.path.join(HERE, "ntt_cases.json"), "w"), indent=0)

    fld_cases = []
    for fld in (R.BN254, R.BLS12_381):
        rnd = random.Random(5)
        q = fld.p
        edge = [0, 1, 2, q - 1, q - 2, (q + 1) // 2, (1 << 256) % q, (1 << 261) % q, (1 << 253) - 1 if (1 << 253) - 1 < q else q - 3]
        vals = edge + [rnd.randrange(q) for _ in range(16)]
        fld_cases.append({"field": fld.name, "p": format(q, "x"), "values": hexs(vals),
                          "roots_of_unity": {str(k): format(fld.root_of_unity(k), "x") for k in (0, 1, 2, 3, 10, 20, fld.two_adicity)}})
    json.dump(fld_cases, open(os.path.join(HERE, "field_cases.json"), "w"), indent=0)
    print("wrote qap_cases.json, ntt_cases.json, field_cases.json")


if __name__ == "__main__":
    main()
