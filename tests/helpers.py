"""Test glue shared by the CPU and GPU suites: generators restating the reference's QuickCheck
generators (test/Test/Circuit/Arithmetic.hs:50-126, test/Test/Circuit/Affine.hs:12-30) on a
seeded `random.Random`, and conversions between the three representations under test:
  oracle.ref_qap (literal Python restatement)  <->  CSR arrays  <->  acx (product) objects."""
from __future__ import annotations

import random
from typing import Dict, List, Sequence, Tuple

import numpy as np

from oracle import ref_qap as R


# ---------------------------------------------------------------------------- generators
def arb_affine_with_mids(rnd: random.Random, p: int, num_inps: int, mids: Sequence[int], size: int):
    """`arbAffineCircuitWithMids` (test/Test/Circuit/Arithmetic.hs:50-64)."""
    if size <= 0:
        choices = ["const"]
        if num_inps > 0:
            choices.append("inp")
        if mids:
            choices.append("mid")
        c = rnd.choice(choices)
        if c == "const":
            return R.ConstGate(rnd.randrange(p))
        if c == "inp":
            return R.Var(R.InputWire(rnd.randrange(num_inps)))
        return R.Var(R.IntermediateWire(rnd.choice(list(mids))))
    if rnd.random() < 0.5:
        return R.ScalarMul(rnd.randrange(p), arb_affine_with_mids(rnd, p, num_inps, mids, size - 1))
    return R.Add(arb_affine_with_mids(rnd, p, num_inps, mids, size - 1),
                 arb_affine_with_mids(rnd, p, num_inps, mids, size - 1))


def arb_arith_circuit(rnd: random.Random, p: int, num_inps: int, size: int, dist=(50, 10, 1),
                      split_bits: int = 256) -> List[tuple]:
    """`arbArithCircuit` (test/Test/Circuit/Arithmetic.hs:69-126): incremental builder; a gate
    may reference inputs and any earlier intermediate output; Split always `split_bits` wide
    (256 in the reference)."""
    gates: List[tuple] = []
    for _ in range(size):
        mids = [w.index for g in gates for w in R.output_wires(g) if w.kind == 1]
        options = [("mul", dist[0])]
        if mids:
            options += [("equal", dist[1]), ("split", dist[2])]
        pick = rnd.choices([o[0] for o in options], weights=[o[1] for o in options])[0]
        out_wire = max(mids) + 1 if mids else 0
        if pick == "mul":
            lhs = arb_affine_with_mids(rnd, p, num_inps, mids, 1)
            rhs = arb_affine_with_mids(rnd, p, num_inps, mids, 1)
            gates.append(R.Mul(lhs, rhs, R.IntermediateWire(out_wire)))
        elif pick == "equal":
            inp = rnd.choice(mids)
            gates.append(R.Equal(R.IntermediateWire(inp), R.IntermediateWire(out_wire),
                                 R.IntermediateWire(out_wire + 1)))
        else:
            inp = rnd.choice(mids)
            outs = [R.IntermediateWire(out_wire + j) for j in range(split_bits)]
            gates.append(R.Split(R.IntermediateWire(inp), outs))
    return gates


def degenerate_root_lists(rnd: random.Random, gates, mode: str, pool: int = 0) -> List[List[int]]:
    """Per-gate root lists (src/QAP.hs:530-539 takes `[[k]]`) that the reference accepts without being what `generateRoots`
    makes: `dup` -- roots drawn from a pool smaller than the row count, so rows of different gates (and of one Equal / Split
    gate) share roots; `surplus` -- extra lists behind the last gate, some repeating earlier roots; `missing` -- fewer lists
    than gates; `mixed` -- repeated roots with either of the two.  Every list keeps the length its gate demands (the reference panics otherwise)."""
    lists = R.fresh_roots(gates, 1)
    total = sum(len(rs) for rs in lists)
    if mode in ("dup", "mixed"):
        pool = pool or max(2, total // 2)
        lists = [[1 + rnd.randrange(pool) for _ in rs] for rs in lists]
    if mode == "mixed":                         # with lists missing, lists appended behind would pair with gates
        mode = rnd.choice(["missing", "surplus"])
    if mode == "missing" and len(lists) > 1:
        lists = lists[: rnd.randrange(1, len(lists))]
    if mode == "surplus":
        for _ in range(rnd.randrange(1, 4)):
            lists.append([rnd.randrange(1, total + 20) for _ in range(rnd.randrange(0, 4))])
    return lists


def arb_input_vector(rnd: random.Random, p: int, num_vars: int) -> Dict[int, int]:
    return {i: rnd.randrange(p) for i in range(num_vars)}


def arb_affine(rnd: random.Random, p: int, num_vars: int, size: int):
    """`arbAffineCircuit` (test/Test/Circuit/Affine.hs:12-30) over wires InputWire."""
    if size <= 0:
        if rnd.random() < 0.5:
            return R.ConstGate(rnd.randrange(p))
        return R.Var(R.InputWire(rnd.randrange(num_vars)))
    c = rnd.randrange(2)
    if c == 0:
        return R.ScalarMul(rnd.randrange(p), arb_affine(rnd, p, num_vars, size - 1))
    return R.Add(arb_affine(rnd, p, num_vars, size - 1), arb_affine(rnd, p, num_vars, size - 1))


# ---------------------------------------------------------------------------- conversions
def to_acx_affine(acx, circ):
    tag = circ[0]
    if tag == "var":
        w = circ[1]
        return acx.Var(acx.Wire(w.kind, w.index))
    if tag == "const":
        return acx.ConstGate(circ[1])
    if tag == "smul":
        return acx.ScalarMul(circ[1], to_acx_affine(acx, circ[2]))
    return acx.Add(to_acx_affine(acx, circ[1]), to_acx_affine(acx, circ[2]))


def to_acx_circuit(acx, gates):
    """oracle-form gate tuples -> product ArithCircuit (pure data translation)."""
    W = lambda w: acx.Wire(w.kind, w.index)
    out = []
    for g in gates:
        if g[0] == "mul":
            out.append(acx.Mul(to_acx_affine(acx, g[1]), to_acx_affine(acx, g[2]), W(g[3])))
        elif g[0] == "equal":
            out.append(acx.Equal(W(g[1]), W(g[2]), W(g[3])))
        else:
            out.append(acx.Split(W(g[1]), [W(o) for o in g[2]]))
    return acx.ArithCircuit(out)


def circuit_dims(gates) -> Tuple[int, int, int]:
    """max index + 1 per wire kind over every wire the circuit mentions."""
    d = [0, 0, 0]

    def see(w):
        d[w.kind] = max(d[w.kind], w.index + 1)

    for g in gates:
        if g[0] == "mul":
            for w in R.fetch_vars(g[1]) + R.fetch_vars(g[2]) + [g[3]]:
                see(w)
        elif g[0] == "equal":
            for w in g[1:4]:
                see(w)
        else:
            see(g[1])
            for w in g[2]:
                see(w)
    return tuple(d)


def flat_index(dims, w) -> int:
    return (1, 1 + dims[0], 1 + dims[0] + dims[1])[w.kind] + w.index


def gen_qap_to_csr(gen: R.GenQAP, dims, p: int):
    """Literal GenQAP (per-wire Map root -> value, densified) -> CSR triple in ascending-root
    row order with zero entries dropped.  Returns (n, m, [A, B, C]) with each matrix as
    (rowptr uint32, col uint32, val (nnz,4) uint64)."""
    from oracle.c_oracle import ints_to_limbs
    roots = sorted(gen.target)
    row_of = {r: i for i, r in enumerate(roots)}
    n, m = len(roots), 1 + sum(dims)
    mats = []
    for qs in (gen.left, gen.right, gen.out):
        rows: List[Dict[int, int]] = [dict() for _ in range(n)]
        for r, v in qs.constant.items():
            if v % p:
                rows[row_of[r]][0] = v % p
        for kind, part in enumerate((qs.inputs, qs.intermediates, qs.outputs)):
            for idx, mp in part.items():
                col = flat_index(dims, R.Wire(kind, idx))
                for r, v in mp.items():
                    if v % p:
                        rows[row_of[r]][col] = v % p
        rowptr, col, val = [0], [], []
        for row in rows:
            for c in sorted(row):
                col.append(c)
                val.append(row[c])
            rowptr.append(len(col))
        mats.append((np.array(rowptr, dtype=np.uint32), np.array(col, dtype=np.uint32),
                     ints_to_limbs(val) if val else np.zeros((0, 4), dtype=np.uint64)))
    return n, m, mats


def qapset_to_flat(qs: R.QapSet, dims, p: int) -> List[int]:
    m = 1 + sum(dims)
    w = [0] * m
    w[0] = qs.constant % p
    for kind, part in enumerate((qs.inputs, qs.intermediates, qs.outputs)):
        for idx, v in part.items():
            if idx < dims[kind]:
                w[flat_index(dims, R.Wire(kind, idx))] = v % p
    return w


def csr_equal(a, b) -> bool:
    return all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))


def free_port() -> int:
    """A TCP port that is free right now on 127.0.0.1 (torch.distributed rendezvous of the multi-process tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
