"""Loader for tests/golden/*.json (data written by tests/golden/gen_golden.py)."""
import json
import os

from oracle import ref_qap as R

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(HERE, name)))


def unhex(xs):
    return [int(x, 16) for x in xs]


def affine_from_json(j):
    if "var" in j:
        return R.Var(R.Wire(*j["var"]))
    if "const" in j:
        return R.ConstGate(int(j["const"], 16))
    if "smul" in j:
        return R.ScalarMul(int(j["smul"][0], 16), affine_from_json(j["smul"][1]))
    return R.Add(affine_from_json(j["add"][0]), affine_from_json(j["add"][1]))


def gates_from_json(js):
    out = []
    for g in js:
        if "mul" in g:
            out.append(R.Mul(affine_from_json(g["mul"][0]), affine_from_json(g["mul"][1]), R.Wire(*g["mul"][2])))
        elif "equal" in g:
            out.append(R.Equal(*[R.Wire(*w) for w in g["equal"]]))
        else:
            out.append(R.Split(R.Wire(*g["split"][0]), [R.Wire(*w) for w in g["split"][1]]))
    return out


def qapset_from_json(j):
    conv = lambda d: {int(k): int(v, 16) for k, v in d.items()}
    return R.QapSet(int(j["constant"], 16), conv(j["inputs"]), conv(j["intermediates"]), conv(j["outputs"]))
