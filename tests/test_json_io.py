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


def test_hand_derived_aeson_fixtures(acx):
    """tests/golden/aeson_*.json were written BY HAND from aeson's documented generic encoding (defaultOptions:
    TaggedObject "tag"/"contents", record fields inlined beside the tag, single-constructor newtypes unwrapped,
    `Map Int v` as an object with decimal string keys; the orphan instances of src/QAP.hs:82-90 make `Prime n` a bare
    integer) -- independently of json_io's own output, for the reference's Example.hs program, its assignment, and a
    circuit that uses every constructor.  json_io must read them into the right objects and write them back
    identically.  Still unpinned against a real GHC build (none is available here): see json_io's header."""
    import os
    jio = importlib.import_module("arithmetic-circuits_amd.json_io")
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    load = lambda name: json.load(open(os.path.join(gdir, name)))
    p = R.BN254.p
    # Example.hs:10-20 -- shared fresh counter: inputs 0..2, intermediates 3, 4, no output wire
    example = acx.ArithCircuit([
        acx.Mul(acx.Var(acx.InputWire(0)), acx.Var(acx.InputWire(1)), acx.IntermediateWire(3)),
        acx.Mul(acx.Var(acx.IntermediateWire(3)), acx.Add(acx.Var(acx.InputWire(0)), acx.Var(acx.InputWire(2))), acx.IntermediateWire(4))])
    j = load("aeson_example_circuit.json")
    assert jio.circuit_from_json(j).gates == example.gates and json.loads(jio.dumps(example)) == j
    every = acx.ArithCircuit([
        acx.Mul(acx.ScalarMul(p - 1, acx.Var(acx.InputWire(0))), acx.Add(acx.ConstGate(10), acx.Var(acx.InputWire(1))), acx.IntermediateWire(0)),
        acx.Equal(acx.IntermediateWire(0), acx.IntermediateWire(1), acx.IntermediateWire(2)),
        acx.Split(acx.IntermediateWire(0), [acx.IntermediateWire(3), acx.IntermediateWire(4), acx.OutputWire(0)])])
    j = load("aeson_all_constructors.json")
    assert jio.circuit_from_json(j).gates == every.gates and json.loads(jio.dumps(every)) == j
    a = acx.generateAssignment(example, {0: 7, 1: 5, 2: 4})
    j = load("aeson_example_assignment.json")
    assert jio.qapset_from_json(j) == a and json.loads(jio.dumps(a)) == j
