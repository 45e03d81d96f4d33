"""oracle/ref_qap.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Literal big-integer restatement (pure Python, arbitrary-precision ints) of the
hot path of the Haskell reference sdiehl/arithmetic-circuits v0.2.0.  It follows
the reference's *own algorithm* -- per-wire dense polynomials, scalar*poly sums,
schoolbook polynomial product, polynomial long division -- so that it is the
semantic ground truth for small sizes.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product never does.

Reference functions restated (all paths are into /root/reference):
  * Circuit.Affine   src/Circuit/Affine.hs:73-125
  * Circuit.Arithmetic src/Circuit/Arithmetic.hs:106-145,158-244
  * QAP              src/QAP.hs:104-110,163-181,226-239,276-352,366-620
  * Fresh            src/Fresh.hs:13-20
  * Circuit.Expr / Circuit.Lang (builder only, needed for Example.hs KAT)
                     src/Circuit/Expr.hs:186-305, src/Circuit/Lang.hs:26-78

Third-party arithmetic that is NOT in /root/reference (restated from the
published algorithms):
  * galois-field-1.0.2 / mod-0.1.1.0  `Prime p`: canonical residues mod p.
  * poly-0.4.0.0 `VPoly`: dense coefficient vectors low->high, no trailing zero;
    `quotRem` = schoolbook long division.
  * galois-fft-0.1.0 `FFT.interpolate`/`FFT.fftTargetPoly`: recursive radix-2
    FFT, natural order, zero-pad to the next power of two, inverse = FFT with
    inverse roots then divide by n; target = x^N - 1.
  * pairing-1.0.0 `getRootOfUnity k` = omega_28^(2^(28-k)), omega_28 = 5^((r-1)/2^28).

PARITY STATUS: the Haskell toolchain is absent (no ghc/cabal/stack) so the
reference cannot be run here.  This restatement is pinned against every
known-answer test the reference's own test-suite holds for the path (Bool
results of verifyAssignment; Equal/Split evaluation KATs) -- see
tests/test_oracle_kat.py.  The reference's tests never pin a polynomial
coefficient, an FFT output or a root of unity, so for those values
"parity unpinned": they are fixed only by mathematical uniqueness under the
conventions above.
"""
from __future__ import annotations

from dataclasses import dataclass, field as dc_field
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

# ----------------------------------------------------------------------------
# Fields (galois-field `Prime p`; pairing `Fr`, `getRootOfUnity`)
# ----------------------------------------------------------------------------

BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BLS12_381_R = 52435875175126190479447740508185965837690552500527637822603658699938581184513


@dataclass(frozen=True)
class Field:
    name: str
    p: int
    two_adicity: int
    generator: int  # multiplicative non-residue used to derive omega_max

    @property
    def omega_max(self) -> int:
        return pow(self.generator, (self.p - 1) >> self.two_adicity, self.p)

    def root_of_unity(self, k: int) -> int:
        """pairing-1.0.0 `getRootOfUnity k`: primitive 2^k-th root (table for 0<=k<=28,
        panic otherwise).  Call sites: bench/Circuit.hs:33, Example.hs:26,
        test/Test/QAP.hs:101, test/Test/Circuit/Arithmetic.hs:208."""
        if not 0 <= k <= self.two_adicity:
            raise ValueError("getRootOfUnity: exponent out of range")
        return pow(self.omega_max, 1 << (self.two_adicity - k), self.p)


BN254 = Field("bn254", BN254_R, 28, 5)
BLS12_381 = Field("bls12_381", BLS12_381_R, 32, 7)

# ----------------------------------------------------------------------------
# poly-0.4.0.0 VPoly restated: list of ints low->high, no trailing zeros
# ----------------------------------------------------------------------------

Poly = List[int]


def to_poly(coeffs: Sequence[int], p: int) -> Poly:
    """`toPoly`: reduce + strip trailing zeros (src/QAP.hs:84-85 uses it for JSON)."""
    out = [c % p for c in coeffs]
    while out and out[-1] == 0:
        out.pop()
    return out


def poly_add(a: Poly, b: Poly, p: int) -> Poly:
    n = max(len(a), len(b))
    return to_poly([(a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0) for i in range(n)], p)


def poly_sub(a: Poly, b: Poly, p: int) -> Poly:
    n = max(len(a), len(b))
    return to_poly([(a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0) for i in range(n)], p)


def poly_mul(a: Poly, b: Poly, p: int) -> Poly:
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                out[i + j] += ai * bj
    return to_poly(out, p)


def poly_scale(c: int, a: Poly, p: int) -> Poly:
    """`monomial 0 c * a` (src/QAP.hs:314)."""
    return to_poly([c * x for x in a], p)


def poly_quot_rem(a: Poly, b: Poly, p: int) -> Tuple[Poly, Poly]:
    """semirings `Euclidean` quotRem for field coefficients: schoolbook long
    division (src/QAP.hs:327)."""
    if not b:
        raise ZeroDivisionError("polynomial division by zero")
    rem = list(a)
    q = [0] * max(0, len(a) - len(b) + 1)
    inv_lead = pow(b[-1], -1, p)
    for k in range(len(a) - len(b), -1, -1):
        c = rem[k + len(b) - 1] * inv_lead % p
        q[k] = c
        if c:
            for j, bj in enumerate(b):
                rem[k + j] = (rem[k + j] - c * bj) % p
    return to_poly(q, p), to_poly(rem[: len(b) - 1], p)


def poly_eval(a: Poly, x: int, p: int) -> int:
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % p
    return acc


def poly_deriv(a: Poly, p: int) -> Poly:
    return to_poly([i * a[i] for i in range(1, len(a))], p)


# ----------------------------------------------------------------------------
# galois-fft-0.1.0 restated
# ----------------------------------------------------------------------------

def _log2_ceil(n: int) -> int:
    k = 0
    while (1 << k) < n:
        k += 1
    return k


def _fft_rec(omega_of: Callable[[int], int], xs: List[int], p: int) -> List[int]:
    """Recursive radix-2 FFT on a list: y_k = sum_i xs[i] * omega_n^(i k)."""
    n = len(xs)
    if n == 1:
        return xs[:]
    ev = _fft_rec(omega_of, xs[0::2], p)
    od = _fft_rec(omega_of, xs[1::2], p)
    w = omega_of(_log2_ceil(n))
    out = [0] * n
    wk = 1
    for k in range(n // 2):
        t = wk * od[k] % p
        out[k] = (ev[k] + t) % p
        out[k + n // 2] = (ev[k] - t) % p
        wk = wk * w % p
    return out


def fft(omega_of: Callable[[int], int], xs: Sequence[int], p: int) -> List[int]:
    n = 1 << _log2_ceil(max(1, len(xs)))
    return _fft_rec(omega_of, list(xs) + [0] * (n - len(xs)), p)


def inverse_dft(omega_of: Callable[[int], int], ys: Sequence[int], p: int) -> List[int]:
    n = 1 << _log2_ceil(max(1, len(ys)))
    inv_omega = lambda k: pow(omega_of(k), -1, p)
    out = _fft_rec(inv_omega, list(ys) + [0] * (n - len(ys)), p)
    n_inv = pow(n, -1, p)
    return [v * n_inv % p for v in out]


def fft_interpolate(omega_of: Callable[[int], int], ys: Sequence[int], p: int) -> Poly:
    """`FFT.interpolate primRoots pts` (call site src/QAP.hs:521-523): the unique
    polynomial of degree < N with P(omega_N^i) = ys[i], zero at the padded points."""
    return to_poly(inverse_dft(omega_of, ys, p), p)


def fft_target_poly(num_roots: int, p: int) -> Poly:
    """`FFT.fftTargetPoly primRoots numRoots` = x^N - 1 (call site src/QAP.hs:524)."""
    n = 1 << _log2_ceil(max(1, num_roots))
    return to_poly([p - 1] + [0] * (n - 1) + [1], p)


# ----------------------------------------------------------------------------
# Circuit.Affine (src/Circuit/Affine.hs)
# ----------------------------------------------------------------------------

@dataclass(frozen=True, order=True)
class Wire:
    """src/Circuit/Arithmetic.hs:32-36.  Derived `Ord`: constructor order
    InputWire < IntermediateWire < OutputWire, then index."""
    kind: int  # 0 InputWire, 1 IntermediateWire, 2 OutputWire
    index: int


def InputWire(i: int) -> Wire:
    return Wire(0, i)


def IntermediateWire(i: int) -> Wire:
    return Wire(1, i)


def OutputWire(i: int) -> Wire:
    return Wire(2, i)


# AffineCircuit (src/Circuit/Affine.hs:26-31) as tagged tuples:
#   ("add", l, r) | ("smul", scalar, e) | ("const", f) | ("var", wire)
Affine = tuple


def Add(l: Affine, r: Affine) -> Affine:
    return ("add", l, r)


def ScalarMul(s: int, e: Affine) -> Affine:
    return ("smul", s, e)


def ConstGate(f: int) -> Affine:
    return ("const", f)


def Var(w) -> Affine:
    return ("var", w)


def eval_affine_circuit(lookup, vars_, circ: Affine, p: int) -> int:
    """src/Circuit/Affine.hs:73-86 -- failed lookups are 0."""
    tag = circ[0]
    if tag == "const":
        return circ[1] % p
    if tag == "var":
        v = lookup(circ[1], vars_)
        return 0 if v is None else v % p
    if tag == "add":
        return (eval_affine_circuit(lookup, vars_, circ[1], p) + eval_affine_circuit(lookup, vars_, circ[2], p)) % p
    if tag == "smul":
        return eval_affine_circuit(lookup, vars_, circ[2], p) * circ[1] % p
    raise ValueError(tag)


def affine_circuit_to_affine_map(circ: Affine, p: int) -> Tuple[int, Dict]:
    """src/Circuit/Affine.hs:90-105.  Add merges duplicate wires with (+); ScalarMul
    scales the constant and every coefficient; explicit zero coefficients are kept."""
    tag = circ[0]
    if tag == "var":
        return 0, {circ[1]: 1}
    if tag == "add":
        cl, vl = affine_circuit_to_affine_map(circ[1], p)
        cr, vr = affine_circuit_to_affine_map(circ[2], p)
        out = dict(vl)
        for k, v in vr.items():
            out[k] = (out[k] + v) % p if k in out else v
        return (cl + cr) % p, out
    if tag == "smul":
        ce, ve = affine_circuit_to_affine_map(circ[2], p)
        s = circ[1] % p
        return s * ce % p, {k: s * v % p for k, v in ve.items()}
    if tag == "const":
        return circ[1] % p, {}
    raise ValueError(tag)


def dot_product(inp: Dict, comp: Dict, p: int) -> int:
    """src/Circuit/Affine.hs:121-125."""
    return sum(c * inp.get(ix, 0) for ix, c in comp.items()) % p


def eval_affine_map(amap: Tuple[int, Dict], inp: Dict, p: int) -> int:
    """src/Circuit/Affine.hs:111-119."""
    return (amap[0] + dot_product(inp, amap[1], p)) % p


def fetch_vars(circ: Affine) -> List[Wire]:
    """src/Circuit/Arithmetic.hs:187-191."""
    tag = circ[0]
    if tag == "var":
        return [circ[1]]
    if tag == "const":
        return []
    if tag == "smul":
        return fetch_vars(circ[2])
    return fetch_vars(circ[1]) + fetch_vars(circ[2])


# ----------------------------------------------------------------------------
# Circuit.Arithmetic (src/Circuit/Arithmetic.hs)
# ----------------------------------------------------------------------------

# Gate (src/Circuit/Arithmetic.hs:44-59) as tagged tuples:
#   ("mul", left, right, out) | ("equal", i, m, out) | ("split", inp, [outs])
Gate = tuple


def Mul(l: Affine, r: Affine, out: Wire) -> Gate:
    return ("mul", l, r, out)


def Equal(i: Wire, m: Wire, out: Wire) -> Gate:
    return ("equal", i, m, out)


def Split(inp: Wire, outs: Sequence[Wire]) -> Gate:
    return ("split", inp, list(outs))


def output_wires(g: Gate) -> List[Wire]:
    """src/Circuit/Arithmetic.hs:67-71."""
    if g[0] == "split":
        return list(g[2])
    return [g[3]]


class ReferencePanic(Exception):
    """Stands for the reference's `panic` calls."""


def eval_gate(lookup, update, vars_, gate: Gate, p: int):
    """src/Circuit/Arithmetic.hs:106-145."""
    if gate[0] == "mul":
        lval = eval_affine_circuit(lookup, vars_, gate[1], p)
        rval = eval_affine_circuit(lookup, vars_, gate[2], p)
        return update(gate[3], lval * rval % p, vars_)
    if gate[0] == "equal":
        inp = lookup(gate[1], vars_)
        if inp is None:
            raise ReferencePanic("evalGate: the impossible happened")
        res = 0 if inp == 0 else 1
        mid = 0 if inp == 0 else pow(inp, -1, p)
        return update(gate[3], res, update(gate[2], mid, vars_))
    if gate[0] == "split":
        inp = lookup(gate[1], vars_)
        if inp is None:
            raise ReferencePanic("evalGate: the impossible happened")
        for ix, out in enumerate(gate[2]):
            vars_ = update(out, (inp >> ix) & 1, vars_)
        return vars_
    raise ValueError(gate[0])


def eval_arith_circuit(lookup, update, gates: Sequence[Gate], vars_, p: int):
    """src/Circuit/Arithmetic.hs:221-235: left fold of evalGate."""
    for g in gates:
        vars_ = eval_gate(lookup, update, vars_, g, p)
    return vars_


def valid_arith_circuit(gates: Sequence[Gate]) -> bool:
    """src/Circuit/Arithmetic.hs:158-185."""
    res = True
    defined: List[Wire] = []
    for g in gates:
        if g[0] == "mul":
            fetched = fetch_vars(g[1]) + fetch_vars(g[2])
        else:
            fetched = [g[1]]
        outs = output_wires(g)
        ok_out = all(w.kind != 0 for w in outs)
        ok_in = all(w.kind == 0 or (w.kind == 1 and w in defined) for w in fetched)
        res = res and ok_out and ok_in
        defined = outs + defined
    return res


def generate_roots(take_root: Callable[[], int], gates: Sequence[Gate]) -> List[List[int]]:
    """src/Circuit/Arithmetic.hs:194-216: Mul -> 1 root, Equal -> 2, Split -> 1+#outs,
    handed out consecutively gate by gate."""
    out = []
    for g in gates:
        if g[0] == "mul":
            out.append([take_root()])
        elif g[0] == "equal":
            out.append([take_root(), take_root()])
        else:
            out.append([take_root() for _ in range(1 + len(g[2]))])
    return out


def fresh_roots(gates: Sequence[Gate], offset: int = 0) -> List[List[int]]:
    """`evalFresh $ generateRoots (fromIntegral . (+offset) <$> fresh)` (src/Fresh.hs:13-20;
    offset 0: bench/Circuit.hs:31-35; offset 1: Example.hs:23, test/.../Arithmetic.hs:195,207)."""
    counter = [0]

    def take():
        v = counter[0]
        counter[0] += 1
        return v + offset

    return generate_roots(take, gates)


def unsplit(wires: Sequence[Wire]) -> Affine:
    """src/Circuit/Arithmetic.hs:238-244."""
    rest: Affine = ConstGate(0)
    for ix, w in enumerate(wires):
        rest = Add(rest, ScalarMul(2 ** ix, Var(w)))
    return rest


# ----------------------------------------------------------------------------
# QAP (src/QAP.hs)
# ----------------------------------------------------------------------------

@dataclass
class QapSet:
    """src/QAP.hs:66-71."""
    constant: object
    inputs: Dict[int, object] = dc_field(default_factory=dict)
    intermediates: Dict[int, object] = dc_field(default_factory=dict)
    outputs: Dict[int, object] = dc_field(default_factory=dict)

    def copy(self) -> "QapSet":
        return QapSet(self.constant, dict(self.inputs), dict(self.intermediates), dict(self.outputs))

    def fmap(self, f) -> "QapSet":
        return QapSet(f(self.constant), {k: f(v) for k, v in self.inputs.items()},
                      {k: f(v) for k, v in self.intermediates.items()},
                      {k: f(v) for k, v in self.outputs.items()})

    def values(self) -> List[object]:
        """Foldable order: constant, inputs, intermediates, outputs (ascending keys)."""
        return ([self.constant] + [self.inputs[k] for k in sorted(self.inputs)]
                + [self.intermediates[k] for k in sorted(self.intermediates)]
                + [self.outputs[k] for k in sorted(self.outputs)])


def _part(qs: QapSet, kind: int) -> Dict[int, object]:
    return (qs.inputs, qs.intermediates, qs.outputs)[kind]


def lookup_at_wire(w: Wire, qs: QapSet):
    """src/QAP.hs:331-337."""
    return _part(qs, w.kind).get(w.index)


def update_at_wire(w: Wire, a, qs: QapSet) -> QapSet:
    """src/QAP.hs:341-347 (functional update)."""
    out = qs.copy()
    _part(out, w.kind)[w.index] = a
    return out


def update_at_wires(wire_vals, qs: QapSet) -> QapSet:
    """src/QAP.hs:350-352: left fold, later entries overwrite earlier ones."""
    for w, v in wire_vals:
        qs = update_at_wire(w, v, qs)
    return qs


def constant_qap_set(g) -> QapSet:
    """src/QAP.hs:113-119."""
    return QapSet(g)


def initial_qap_set(inputs: Dict[int, int]) -> QapSet:
    """src/QAP.hs:591-595: constant wire = 1."""
    return QapSet(1, dict(inputs))


def generate_assignment(gates: Sequence[Gate], inputs: Dict[int, int], p: int) -> QapSet:
    """src/QAP.hs:597-603."""
    return eval_arith_circuit(lookup_at_wire, update_at_wire, gates, initial_qap_set(inputs), p)


def generate_assignment_gate(gate: Gate, inputs: Dict[int, int], p: int) -> QapSet:
    """src/QAP.hs:579-589."""
    return eval_gate(lookup_at_wire, update_at_wire, initial_qap_set(inputs), gate, p)


def qap_set_to_map(qs: QapSet) -> Dict[int, object]:
    """src/QAP.hs:605-620: flat numbering 0 = constant, then inputs, intermediates, outputs,
    each block sized (max key + 1)."""
    def max_key(m):
        return max(m) + 1 if m else 0
    n_in = max_key(qs.inputs)
    n_mid = max_key(qs.intermediates)
    out = {0: qs.constant}
    out.update({1 + k: v for k, v in qs.inputs.items()})
    out.update({1 + n_in + k: v for k, v in qs.intermediates.items()})
    out.update({1 + n_in + n_mid + k: v for k, v in qs.outputs.items()})
    return out


@dataclass
class GenQAPRow:
    """One `GenQAP ((,) k) k` = one constraint row at one root (src/QAP.hs:94-99,366-474).
    Each QapSet holds (root, value) pairs."""
    left: QapSet
    right: QapSet
    out: QapSet
    target: Tuple[int, int]


def gate_to_gen_qap(roots: Sequence[int], gate: Gate, p: int) -> List[GenQAPRow]:
    """src/QAP.hs:366-474."""
    if gate[0] == "mul":
        if len(roots) != 1:
            raise ReferencePanic("gateToGenQAP: wrong number of roots supplied")
        root = roots[0]
        lc, lv = affine_circuit_to_affine_map(gate[1], p)
        rc, rv = affine_circuit_to_affine_map(gate[2], p)
        left = constant_qap_set((root, lc))
        for w, v in lv.items():
            left = update_at_wire(w, (root, v), left)
        right = constant_qap_set((root, rc))
        for w, v in rv.items():
            right = update_at_wire(w, (root, v), right)
        out = update_at_wire(gate[3], (root, 1), constant_qap_set((root, 0)))
        return [GenQAPRow(left, right, out, (root, 0))]
    if gate[0] == "equal":
        if len(roots) != 2:
            raise ReferencePanic("gateToGenQAP: wrong number of roots supplied")
        r0, r1 = roots
        i, m, o = gate[1], gate[2], gate[3]
        q0 = GenQAPRow(
            update_at_wires([(i, (r0, 1)), (m, (r0, 0)), (o, (r0, 0))], constant_qap_set((r0, 0))),
            update_at_wires([(i, (r0, 0)), (m, (r0, 1)), (o, (r0, 0))], constant_qap_set((r0, 0))),
            update_at_wires([(i, (r0, 0)), (m, (r0, 0)), (o, (r0, 1))], constant_qap_set((r0, 0))),
            (r0, 0))
        q1 = GenQAPRow(
            update_at_wires([(i, (r1, 0)), (m, (r1, 0)), (o, (r1, p - 1))], constant_qap_set((r1, 1))),
            update_at_wires([(i, (r1, 1)), (m, (r1, 0)), (o, (r1, 0))], constant_qap_set((r1, 0))),
            update_at_wires([(i, (r1, 0)), (m, (r1, 0)), (o, (r1, 0))], constant_qap_set((r1, 0))),
            (r1, 0))
        return [q0, q1]
    if gate[0] == "split":
        if not roots:
            raise ReferencePanic("gateToGenQAP: wrong number of roots supplied")
        root, rest = roots[0], list(roots[1:])
        inp, outs = gate[1], gate[2]
        if len(rest) != len(outs):
            raise ReferencePanic("gateToGenQAP: wrong number of roots supplied")
        q0 = GenQAPRow(
            update_at_wires([(inp, (root, 0))] + [(o, (root, pow(2, ix, p))) for ix, o in enumerate(outs)],
                            constant_qap_set((root, 0))),
            update_at_wires([(inp, (root, 0))], constant_qap_set((root, 1))),
            update_at_wires([(inp, (root, 1))], constant_qap_set((root, 0))),
            (root, 0))
        rows = [q0]
        for r, o in zip(rest, outs):
            rows.append(GenQAPRow(
                update_at_wire(o, (r, 1), constant_qap_set((r, 0))),
                update_at_wire(o, (r, p - 1), constant_qap_set((r, 1))),
                update_at_wire(o, (r, 0), constant_qap_set((r, 0))),
                (r, 0)))
        return rows
    raise ReferencePanic("gateToGenQAP: wrong number of roots supplied")


@dataclass
class GenQAP:
    """`GenQAP (Map k) k` (src/QAP.hs:94-99): per-wire Map root -> value."""
    left: QapSet
    right: QapSet
    out: QapSet
    target: Dict[int, int]


def _sequence_qap_set(sets: List[QapSet]) -> QapSet:
    """src/QAP.hs:104-110 followed by `Map.fromList` (src/QAP.hs:236-238): collect the
    (root, value) pairs per wire; a later pair with the same root overwrites."""
    def collect(getter):
        acc: Dict[int, Dict[int, int]] = {}
        for s in sets:
            for k, (root, val) in getter(s).items():
                acc.setdefault(k, {})[root] = val
        return acc
    const: Dict[int, int] = {}
    for s in sets:
        const[s.constant[0]] = s.constant[1]
    return QapSet(const, collect(lambda s: s.inputs), collect(lambda s: s.intermediates),
                  collect(lambda s: s.outputs))


def create_map_gen_qap(rows: List[GenQAPRow]) -> GenQAP:
    """src/QAP.hs:233-239."""
    return GenQAP(_sequence_qap_set([r.left for r in rows]), _sequence_qap_set([r.right for r in rows]),
                  _sequence_qap_set([r.out for r in rows]), {r.target[0]: r.target[1] for r in rows})


def add_missing_zeroes(all_roots: Sequence[int], g: GenQAP) -> GenQAP:
    """src/QAP.hs:566-576: left-biased union with {root: 0 for every root} -- densifies."""
    def fill(m):
        out = {r: 0 for r in all_roots}
        out.update(m)
        return out
    return GenQAP(g.left.fmap(fill), g.right.fmap(fill), g.out.fmap(fill), fill(g.target))


def arith_circuit_to_gen_qap(roots_per_gate: Sequence[Sequence[int]], gates: Sequence[Gate], p: int) -> GenQAP:
    """src/QAP.hs:530-539.  (`zipWith` truncates to the shorter list.)"""
    rows: List[GenQAPRow] = []
    for roots, gate in zip(roots_per_gate, gates):
        rows.extend(gate_to_gen_qap(roots, gate, p))
    all_roots = [r for rs in roots_per_gate for r in rs]
    return add_missing_zeroes(all_roots, create_map_gen_qap(rows))


@dataclass
class QAP:
    """src/QAP.hs:74-79."""
    left: QapSet
    right: QapSet
    out: QapSet
    target: Poly


def _elems(m: Dict[int, int]) -> List[int]:
    """`Map.elems`: values in ascending key (= root) order (src/QAP.hs:521-523)."""
    return [m[k] for k in sorted(m)]


def create_polynomials_fft(omega_of: Callable[[int], int], g: GenQAP, p: int) -> QAP:
    """src/QAP.hs:512-525."""
    interp = lambda m: fft_interpolate(omega_of, _elems(m), p)
    return QAP(g.left.fmap(interp), g.right.fmap(interp), g.out.fmap(interp),
               fft_target_poly(len(g.target), p))


def lagrange_interpolate(xys: List[Tuple[int, int]], p: int) -> Poly:
    """src/QAP.hs:494-508."""
    xs = [x for x, _ in xys]
    ys = [y for _, y in xys]
    roots: Poly = [1]
    for xi in xs:
        roots = poly_mul(roots, to_poly([-xi, 1], p), p)
    droots = poly_deriv(roots, p)
    acc: Poly = []
    for x, y in zip(xs, ys):
        phi = poly_eval(droots, x, p)
        f = y * pow(phi, -1, p) % p
        quot, _ = poly_quot_rem(roots, to_poly([-x, 1], p), p)
        acc = poly_add(acc, poly_scale(f, quot, p), p)
    return acc


def create_polynomials(g: GenQAP, p: int) -> QAP:
    """src/QAP.hs:486-508: naive Lagrange on the actual root values; target = prod (x - r)."""
    interp = lambda m: lagrange_interpolate(sorted(m.items()), p)
    target: Poly = [1]
    for root in sorted(g.target):
        target = poly_mul(target, to_poly([-root, 1], p), p)
    return QAP(g.left.fmap(interp), g.right.fmap(interp), g.out.fmap(interp), target)


def arith_circuit_to_qap(roots, gates, p: int) -> QAP:
    """src/QAP.hs:542-549."""
    return create_polynomials(arith_circuit_to_gen_qap(roots, gates, p), p)


def arith_circuit_to_qap_fft(omega_of, roots, gates, p: int) -> QAP:
    """src/QAP.hs:552-561."""
    return create_polynomials_fft(omega_of, arith_circuit_to_gen_qap(roots, gates, p), p)


def gate_to_qap(omega_of, roots: Sequence[int], gate: Gate, p: int) -> QAP:
    """src/QAP.hs:355-363."""
    g = add_missing_zeroes(list(roots), create_map_gen_qap(gate_to_gen_qap(roots, gate, p)))
    return create_polynomials_fft(omega_of, g, p)


def combine_with_defaults(f, default_a, default_b, a: QapSet, b: QapSet) -> QapSet:
    """src/QAP.hs:163-181: merge per part; a key missing on one side uses that side's default."""
    def comb(ma, mb):
        out = {}
        for k in set(ma) | set(mb):
            out[k] = f(ma.get(k, default_a), mb.get(k, default_b))
        return out
    return QapSet(f(a.constant, b.constant), comb(a.inputs, b.inputs),
                  comb(a.intermediates, b.intermediates), comb(a.outputs, b.outputs))


def verification_witness_zk(d1: int, d2: int, d3: int, qap: QAP, assignment: QapSet, p: int) -> Optional[Poly]:
    """src/QAP.hs:300-327: Just quotient iff target | (L*R - O)."""
    def scaled_sum(polys: QapSet) -> Poly:
        scaled = combine_with_defaults(lambda a, b: poly_scale(b % p, a, p), [], 0, polys, assignment)
        acc: Poly = []
        for v in scaled.values():
            acc = poly_add(acc, v, p)
        return acc
    left = poly_add(poly_scale(d1 % p, qap.target, p), scaled_sum(qap.left), p)
    right = poly_add(poly_scale(d2 % p, qap.target, p), scaled_sum(qap.right), p)
    outp = poly_add(poly_scale(d3 % p, qap.target, p), scaled_sum(qap.out), p)
    io = poly_sub(poly_mul(left, right, p), outp, p)
    quotient, remainder = poly_quot_rem(io, qap.target, p)
    return quotient if not remainder else None


def verification_witness(qap: QAP, assignment: QapSet, p: int) -> Optional[Poly]:
    """src/QAP.hs:292-298."""
    return verification_witness_zk(0, 0, 0, qap, assignment, p)


def verify_assignment(qap: QAP, assignment: QapSet, p: int) -> bool:
    """src/QAP.hs:276-282."""
    return verification_witness(qap, assignment, p) is not None


# ----------------------------------------------------------------------------
# Circuit.Expr / Circuit.Lang builder -- only what Example.hs needs, so the README
# known-answer program can be reproduced from source form (SURVEY 8c KAT 3).
# ----------------------------------------------------------------------------

class CircuitBuilder:
    """`ExprM` state monad (src/Circuit/Expr.hs:186-217): ONE shared counter for every
    wire kind; gates are emitted in order."""

    def __init__(self):
        self.gates: List[Gate] = []
        self.counter = 0

    def _fresh(self) -> int:
        v = self.counter
        self.counter += 1
        return v

    def input(self) -> Wire:  # Lang.hs:77-78
        return InputWire(self._fresh())

    def imm(self) -> Wire:
        return IntermediateWire(self._fresh())

    def fresh_output(self) -> Wire:
        return OutputWire(self._fresh())

    # expressions: ("var", wire) | ("const", n) | ("add"|"sub"|"mul", e1, e2)
    def compile(self, expr):
        """src/Circuit/Expr.hs:247-305 restricted to EVar/EConst/BAdd/BSub/BMul.
        Returns ("wire", w) for Left or ("affine", circ) for Right."""
        tag = expr[0]
        if tag == "const":
            return ("affine", ConstGate(expr[1]))
        if tag == "var":
            return ("wire", expr[1])
        l = self._add_var(self.compile(expr[1]))
        r = self._add_var(self.compile(expr[2]))
        if tag == "add":
            return ("affine", Add(l, r))
        if tag == "sub":
            return ("affine", Add(l, ScalarMul(-1, r)))
        if tag == "mul":
            o = self.imm()
            self.gates.append(Mul(l, r, o))
            return ("wire", o)
        raise ValueError(tag)

    @staticmethod
    def _add_var(x) -> Affine:
        return Var(x[1]) if x[0] == "wire" else x[1]

    def ret(self, expr) -> Wire:
        """src/Circuit/Lang.hs:67-75: no new gate when the expression already is a wire."""
        out = self.compile(expr)
        if out[0] == "wire":
            return out[1]
        w = self.fresh_output()
        self.gates.append(Mul(ConstGate(1), out[1], w))
        return w
