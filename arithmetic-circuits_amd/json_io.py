"""JSON interchange in the shape of the reference's aeson instances.

The reference derives `ToJSON`/`FromJSON` generically (src/Circuit/Arithmetic.hs:36,59,150;
src/Circuit/Affine.hs:31; src/QAP.hs:71,79,99) with aeson's default options, plus two orphan
instances: `Prime n` as a bare JSON integer (`toJSON . fromP`, src/QAP.hs:87-90) and `VPoly` as
the coefficient array low to high (`toJSON . unPoly`, src/QAP.hs:82-85).  aeson's defaults
(`defaultOptions`: `sumEncoding = TaggedObject "tag" "contents"`, record fields inlined next to
the tag, newtypes unwrapped, `Map Int v` as an object with decimal string keys) give:

    Wire            {"tag":"InputWire","contents":3}
    AffineCircuit   {"tag":"Add","contents":[l,r]} | {"tag":"ScalarMul","contents":[f,e]}
                    | {"tag":"ConstGate","contents":f} | {"tag":"Var","contents":wire}
    Gate            {"tag":"Mul","mulLeft":..,"mulRight":..,"mulOutput":..}
                    | {"tag":"Equal","eqInput":..,"eqMagic":..,"eqOutput":..}
                    | {"tag":"Split","splitInput":..,"splitOutputs":[..]}
    ArithCircuit    [gate, ...]
    QapSet f        {"qapSetConstant":f,"qapSetInput":{"0":f,..},"qapSetIntermediate":{..},"qapSetOutput":{..}}
    QAP f           {"qapInputsLeft":QapSet [coeffs],"qapInputsRight":..,"qapOutputs":..,"qapTarget":[coeffs]}

No test of the reference pins these encodings and GHC is not available here, so this module is
UNPINNED against the real library: it follows aeson's documented defaults.  With no GHC on the
GPU box, JSON files are the practical way to exchange circuits and assignments with a Haskell
host."""
from __future__ import annotations

import json
from typing import Any, Dict, List

from .circuit import (Add, ArithCircuit, ConstGate, Equal, Mul, ScalarMul, Split, Var, Wire)
from .qap import QAP, QapSet

_WIRE_TAGS = ["InputWire", "IntermediateWire", "OutputWire"]


def wire_to_json(w: Wire) -> Dict[str, Any]:
    return {"tag": _WIRE_TAGS[w.kind], "contents": int(w.index)}


def wire_from_json(j: Dict[str, Any]) -> Wire:
    return Wire(_WIRE_TAGS.index(j["tag"]), int(j["contents"]))


def affine_to_json(c) -> Dict[str, Any]:
    if isinstance(c, Var):
        return {"tag": "Var", "contents": wire_to_json(c.wire)}
    if isinstance(c, ConstGate):
        return {"tag": "ConstGate", "contents": int(c.value)}
    if isinstance(c, ScalarMul):
        return {"tag": "ScalarMul", "contents": [int(c.scalar), affine_to_json(c.expr)]}
    if isinstance(c, Add):
        return {"tag": "Add", "contents": [affine_to_json(c.left), affine_to_json(c.right)]}
    raise TypeError(f"not an AffineCircuit: {c!r}")


def affine_from_json(j: Dict[str, Any]):
    tag, c = j["tag"], j.get("contents")
    if tag == "Var":
        return Var(wire_from_json(c))
    if tag == "ConstGate":
        return ConstGate(int(c))
    if tag == "ScalarMul":
        return ScalarMul(int(c[0]), affine_from_json(c[1]))
    if tag == "Add":
        return Add(affine_from_json(c[0]), affine_from_json(c[1]))
    raise ValueError(f"unknown AffineCircuit tag {tag!r}")


def gate_to_json(g) -> Dict[str, Any]:
    if isinstance(g, Mul):
        return {"tag": "Mul", "mulLeft": affine_to_json(g.mulLeft), "mulRight": affine_to_json(g.mulRight),
                "mulOutput": wire_to_json(g.mulOutput)}
    if isinstance(g, Equal):
        return {"tag": "Equal", "eqInput": wire_to_json(g.eqInput), "eqMagic": wire_to_json(g.eqMagic),
                "eqOutput": wire_to_json(g.eqOutput)}
    if isinstance(g, Split):
        return {"tag": "Split", "splitInput": wire_to_json(g.splitInput),
                "splitOutputs": [wire_to_json(w) for w in g.splitOutputs]}
    raise TypeError(f"not a Gate: {g!r}")


def gate_from_json(j: Dict[str, Any]):
    tag = j["tag"]
    if tag == "Mul":
        return Mul(affine_from_json(j["mulLeft"]), affine_from_json(j["mulRight"]), wire_from_json(j["mulOutput"]))
    if tag == "Equal":
        return Equal(wire_from_json(j["eqInput"]), wire_from_json(j["eqMagic"]), wire_from_json(j["eqOutput"]))
    if tag == "Split":
        return Split(wire_from_json(j["splitInput"]), [wire_from_json(w) for w in j["splitOutputs"]])
    raise ValueError(f"unknown Gate tag {tag!r}")


def circuit_to_json(c: ArithCircuit) -> List[Any]:
    return [gate_to_json(g) for g in c.gates]


def circuit_from_json(j: List[Any]) -> ArithCircuit:
    return ArithCircuit([gate_from_json(g) for g in j])


def _map_to_json(m: Dict[int, Any], f=int) -> Dict[str, Any]:
    return {str(k): f(v) for k, v in sorted(m.items())}


def qapset_to_json(qs: QapSet, f=int) -> Dict[str, Any]:
    return {"qapSetConstant": f(qs.qapSetConstant), "qapSetInput": _map_to_json(qs.qapSetInput, f),
            "qapSetIntermediate": _map_to_json(qs.qapSetIntermediate, f), "qapSetOutput": _map_to_json(qs.qapSetOutput, f)}


def qapset_from_json(j: Dict[str, Any], f=int) -> QapSet:
    conv = lambda d: {int(k): f(v) for k, v in d.items()}
    return QapSet(f(j["qapSetConstant"]), conv(j["qapSetInput"]), conv(j["qapSetIntermediate"]), conv(j["qapSetOutput"]))


def qap_to_json(qap: QAP) -> Dict[str, Any]:
    """Materialises EVERY per-wire polynomial (3*m*N coefficients): test-size QAPs only, exactly
    like the reference's own `QAP` value."""
    gen = qap.gen

    def side(getter) -> Dict[str, Any]:
        polys = QapSet(getter(flat=0))
        base = 1
        for part, size in ((polys.qapSetInput, gen.n_inputs), (polys.qapSetIntermediate, gen.n_intermediates),
                           (polys.qapSetOutput, gen.n_outputs)):
            for k in range(size):
                part[k] = getter(flat=base + k)
            base += size
        return qapset_to_json(polys, f=lambda poly: [int(c) for c in poly])

    return {"qapInputsLeft": side(qap.qapInputsLeft), "qapInputsRight": side(qap.qapInputsRight),
            "qapOutputs": side(qap.qapOutputs), "qapTarget": [int(c) for c in qap.qapTarget]}


def dumps(obj) -> str:
    if isinstance(obj, ArithCircuit):
        return json.dumps(circuit_to_json(obj))
    if isinstance(obj, QapSet):
        return json.dumps(qapset_to_json(obj))
    if isinstance(obj, QAP):
        return json.dumps(qap_to_json(obj))
    raise TypeError(type(obj))
