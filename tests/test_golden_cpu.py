"""Oracle (b) [C] and the product's host logic against the committed golden fixtures.  CPU."""
import numpy as np
import pytest

from oracle import ref_qap as R
from oracle.c_oracle import COracle, ints_to_limbs, limbs_to_ints
from tests import golden_util as G
from tests import helpers as H

FIELDS = {"bn254": R.BN254, "bls12_381": R.BLS12_381}


@pytest.mark.parametrize("case", G.load("qap_cases.json"), ids=lambda c: c["name"])
def test_qap_cases_c_oracle_and_host_rows(acx, case):
    fld = FIELDS[case["field"]]
    p = fld.p
    orc = COracle(case["field"])
    gates = G.gates_from_json(case["gates"])
    roots = [G.unhex(r) for r in case["roots"]]
    circ = H.to_acx_circuit(acx, gates).marshal(case["field"])
    assert [circ.n_inputs, circ.n_intermediates, circ.n_outputs] == case["dims"]
    mats = circ.rows(acx.ints_to_fr([r for rs in roots for r in rs]))
    n, m = circ.n_rows, circ.m
    log_n = max(0, (n - 1).bit_length())
    assert G.unhex(case["target"]) == [p - 1] + [0] * ((1 << log_n) - 1) + [1]
    for k, name in enumerate("ABC"):
        cols = orc.qap_columns(n, log_n, mats[k], 0, m)
        for w in range(m):
            want = G.unhex(case["polys"][name].get(str(w), []))
            assert R.to_poly(limbs_to_ints(cols[w]), p) == want
    for rec in case["assignments"]:
        w = ints_to_limbs(G.unhex(rec["flat"]))
        _, nbad, _ = orc.r1cs_residuals(n, m, *mats, w)
        assert (nbad == 0) == rec["valid"]
        h, ok = orc.qap_h(n, m, log_n, *mats, w)
        assert ok == rec["valid"]
        if ok:
            assert R.to_poly(limbs_to_ints(h), p) == G.unhex(rec["h"])
        if "delta" in rec:
            hz, okz = orc.qap_h(n, m, log_n, *mats, w, delta=G.unhex(rec["delta"]))
            assert okz == (rec["h_zk"] is not None)
            if okz:
                assert R.to_poly(limbs_to_ints(hz), p) == G.unhex(rec["h_zk"])


@pytest.mark.parametrize("case", G.load("ntt_cases.json"), ids=lambda c: f'{c["field"]}-{c["log_n"]}')
def test_ntt_cases_c_oracle(case):
    orc = COracle(case["field"])
    xs = ints_to_limbs(G.unhex(case["in"]))
    ln = case["log_n"]
    assert limbs_to_ints(orc.ntt(xs, ln)) == G.unhex(case["fft"])
    assert limbs_to_ints(orc.ntt(xs, ln, inverse=True)) == G.unhex(case["interpolate"])
    assert limbs_to_ints(orc.ntt(xs, ln, shift=int(case["shift"], 16))) == G.unhex(case["coset_fft"])


@pytest.mark.parametrize("case", G.load("field_cases.json"), ids=lambda c: c["field"])
def test_field_cases_c_oracle(case):
    orc = COracle(case["field"])
    p = int(case["p"], 16)
    vals = G.unhex(case["values"])
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert orc.op("mul", a, b) == a * b % p
        assert orc.op("add", a, b) == (a + b) % p
        assert orc.op("sub", a, b) == (a - b) % p
        if a:
            assert orc.op("inv", a) == pow(a, -1, p)
    for k, w in case["roots_of_unity"].items():
        assert orc.root_of_unity(int(k)) == int(w, 16)


def test_ghc_vector_checker_accepts_the_fixtures_and_names_a_difference(tmp_path):
    """tools/ghc_vectors/check.py is what turns a dump of the real Haskell library (tools/ghc_vectors/Main.hs, for a
    machine with GHC) into a verdict on the derived fixtures.  Fed a dump synthesised from the fixtures themselves, in
    the aeson shape the library would emit, it must accept; with one coefficient or one root of unity changed it must
    name the convention."""
    import json, os, subprocess, sys
    from oracle import ref_qap as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "qap_cases.json")))[:3]

    def qs(polys, dims):
        out = {"qapSetConstant": [int(x, 16) for x in polys.get("0", [])], "qapSetInput": {}, "qapSetIntermediate": {}, "qapSetOutput": {}}
        base = 1
        for key, size in (("qapSetInput", dims[0]), ("qapSetIntermediate", dims[1]), ("qapSetOutput", dims[2])):
            for k in range(size):
                if str(base + k) in polys:
                    out[key][str(k)] = [int(x, 16) for x in polys[str(base + k)]]
            base += size
        return out

    cases = []
    for g in gold:
        c = {"name": g["name"],
             "qap": {"qapTarget": [int(x, 16) for x in g["target"]], "qapInputsLeft": qs(g["polys"]["A"], g["dims"]),
                     "qapInputsRight": qs(g["polys"]["B"], g["dims"]), "qapOutputs": qs(g["polys"]["C"], g["dims"])},
             "assignments": [{"valid": a["valid"], "assignment": None,
                              "h": None if a.get("h") is None else [int(x, 16) for x in a["h"]],
                              "h_zk": None if a.get("h_zk") is None else [int(x, 16) for x in a["h_zk"]]} for a in g["assignments"]]}
        if g["name"] == "example_hs":
            c["circuit"] = json.load(open(os.path.join(root, "tests", "golden", "aeson_example_circuit.json")))
            c["assignments"][0]["assignment"] = json.load(open(os.path.join(root, "tests", "golden", "aeson_example_assignment.json")))
        cases.append(c)
    dump = {"roots_of_unity": [R.BN254.root_of_unity(k) for k in range(29)], "cases": cases}
    check = os.path.join(root, "tools", "ghc_vectors", "check.py")

    def run(d):
        path = tmp_path / "dump.json"
        path.write_text(json.dumps(d))
        return subprocess.run([sys.executable, check, str(path)], capture_output=True, text=True)

    ok = run(dump)
    assert ok.returncode == 0 and "all derived fixtures match" in ok.stdout, ok.stdout + ok.stderr
    bad = json.loads(json.dumps(dump))
    bad["roots_of_unity"][28] += 1
    bad["cases"][0]["qap"]["qapInputsLeft"]["qapSetInput"]["0"][1] += 1
    out = run(bad)
    assert out.returncode == 1 and "getRootOfUnity table differs" in out.stdout and "FFT.interpolate point order" in out.stdout
