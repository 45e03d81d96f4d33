#!/usr/bin/env python3
"""Compare a dump of the REAL Haskell library (tools/ghc_vectors/Main.hs, run inside a checkout of
sdiehl/arithmetic-circuits v0.2.0 on a machine with GHC) with this repository's derived fixtures:

    python tools/ghc_vectors/check.py ghc_vectors.json

  * pairing's getRootOfUnity table            vs  oracle/ref_qap.py BN254.root_of_unity
  * createPolynomialsFFT coefficients, target vs  tests/golden/qap_cases.json "polys", "target"
  * verificationWitness[Zk] quotients         vs  "h", "h_zk"
  * verifyAssignment Bools                    vs  "valid"
  * aeson encodings of the Example.hs program / assignment vs tests/golden/aeson_example_*.json
Any mismatch names the convention that differs (point order, padding, root table, JSON shape): that is the one
piece of parity this build could not pin without GHC.  Exit 0 = the derived fixtures ARE the
library's outputs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_qap as R  # noqa: E402


def strip(poly):
    poly = list(poly)
    while poly and poly[-1] == 0:
        poly.pop()
    return poly


def flat_polys(qs, dims):
    """aeson QapSet of coefficient arrays -> {flat wire index: coefficients} in qapSetToMap order (src/QAP.hs:605-620)."""
    out = {0: strip(qs["qapSetConstant"])}
    base = 1
    for key, size in (("qapSetInput", dims[0]), ("qapSetIntermediate", dims[1]), ("qapSetOutput", dims[2])):
        for k, v in qs[key].items():
            out[base + int(k)] = strip(v)
        base += size
    return out


def main():
    dump = json.load(open(sys.argv[1]))
    gold = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "qap_cases.json")))}
    bad = []
    want_roots = [R.BN254.root_of_unity(k) for k in range(29)]
    if [int(x) for x in dump["roots_of_unity"]] != want_roots:
        bad.append("getRootOfUnity table differs from 5^((r-1)/2^28) chain (SURVEY.md Appendix A.5): use acx_ctx_set_root / acx_mgpu_set_root")
    for case in dump["cases"]:
        g = gold[case["name"]]
        dims = g["dims"]
        qap = case["qap"]
        if strip(qap["qapTarget"]) != strip(int(x, 16) for x in g["target"]):
            bad.append(f"{case['name']}: qapTarget differs (FFT.fftTargetPoly / padding rule)")
        for mname, key in (("A", "qapInputsLeft"), ("B", "qapInputsRight"), ("C", "qapOutputs")):
            got = flat_polys(qap[key], dims)
            for idx, coeffs in g["polys"][mname].items():
                want = strip(int(x, 16) for x in coeffs)
                if got.get(int(idx), []) != want:
                    bad.append(f"{case['name']}: polynomial of flat wire {idx} in {key} differs (FFT.interpolate point order / padding)")
        for a_dump, a_gold in zip(case["assignments"], g["assignments"]):
            if bool(a_dump["valid"]) != bool(a_gold["valid"]):
                bad.append(f"{case['name']}: verifyAssignment Bool differs")
            for k in ("h", "h_zk"):
                if a_gold.get(k) is None and a_gold.get("delta") is None and k == "h_zk":
                    continue
                want = None if a_gold.get(k) is None else strip(int(x, 16) for x in a_gold[k])
                got = None if a_dump[k] is None else strip(a_dump[k])
                if got != want:
                    bad.append(f"{case['name']}: {k} differs")
        if case["name"] == "example_hs":
            gdir = os.path.join(ROOT, "tests", "golden")
            if case["circuit"] != json.load(open(os.path.join(gdir, "aeson_example_circuit.json"))):
                bad.append("aeson encoding of ArithCircuit differs from tests/golden/aeson_example_circuit.json (json_io.py)")
            if case["assignments"][0]["assignment"] != json.load(open(os.path.join(gdir, "aeson_example_assignment.json"))):
                bad.append("aeson encoding of QapSet differs from tests/golden/aeson_example_assignment.json (json_io.py)")
    for b in bad:
        print("MISMATCH:", b)
    print("checked", len(dump["cases"]), "cases:", "all derived fixtures match the Haskell library" if not bad else f"{len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
