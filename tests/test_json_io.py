"""aeson-shaped JSON round trips (host logic, CPU).  UNPINNED against the real library (no GHC)."""
import importlib
import json
import random

from oracle import ref_qap as R
from tests import helpers as H


def test_circuit_and_qapset_round_trip(acx):
    jio = importlib.import_module("arithmetic-circuits_amd.json_io")
    p = R.BN254.p
    rnd = random.Random(4)
    gates = H.arb_arith_circuit(rnd, p, 3, 9, dist=(5, 2, 2), split_bits=4)
    program = H.to_acx_circuit(acx, gates)
    text = jio.dumps(program)
    back = jio.circuit_from_json(json.loads(text))
    assert back.gates == program.gates
    j = json.loads(text)
    assert isinstance(j, list) and all("tag" in g for g in j)
    mul = next(g for g in j if g["tag"] == "Mul")
    assert set(mul) == {"tag", "mulLeft", "mulRight", "mulOutput"} and mul["mulOutput"]["tag"] in ("IntermediateWire", "OutputWire")
    a = acx.generateAssignment(program, {0: 5, 1: p - 1, 2: 123456789 ** 7 % p})
    ja = json.loads(jio.dumps(a))
    assert ja["qapSetConstant"] == 1 and all(isinstance(k, str) for k in ja["qapSetInput"])
    assert jio.qapset_from_json(ja) == a
    # field elements are bare (arbitrary-precision) JSON integers
    assert ja["qapSetInput"]["1"] == p - 1
