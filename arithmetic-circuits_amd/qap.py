"""Drop-in mirror of the reference's `QAP` module (export list src/QAP.hs:11-39) for the hot
path: same function names, argument order and results; the bodies marshal across the C ABI into
the HIP engine.  Differences forced by the boundary are stated per function.

  QapSet                       src/QAP.hs:66-71
  GenQAP / QAP                 src/QAP.hs:74-99   (device-resident handles here)
  arithCircuitToGenQAP         src/QAP.hs:530-539
  createPolynomialsFFT         src/QAP.hs:512-525
  arithCircuitToQAPFFT         src/QAP.hs:552-561
  gateToQAP                    src/QAP.hs:355-363
  verifyAssignment             src/QAP.hs:276-282
  verificationWitness[Zk]      src/QAP.hs:292-327
  generateAssignment[Gate]     src/QAP.hs:579-603
  qapSetToMap / initialQapSet  src/QAP.hs:591-620
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .circuit import ArithCircuit, Equal, Gate, Mul, Split, Wire
from .engine import Circuit, Context, Naive, R1CS, fr_to_ints, ints_to_fr


@dataclass
class QapSet:
    """`data QapSet f` -- constant, inputs, intermediates, outputs (src/QAP.hs:66-71)."""
    qapSetConstant: int
    qapSetInput: Dict[int, int] = field(default_factory=dict)
    qapSetIntermediate: Dict[int, int] = field(default_factory=dict)
    qapSetOutput: Dict[int, int] = field(default_factory=dict)


def initialQapSet(inputs: Dict[int, int]) -> QapSet:
    return QapSet(1, dict(inputs))


def lookupAtWire(wire: Wire, qs: QapSet) -> Optional[int]:
    return (qs.qapSetInput, qs.qapSetIntermediate, qs.qapSetOutput)[wire.kind].get(wire.index)


def qapSetToMap(qs: QapSet) -> Dict[int, int]:
    def max_key(m):
        return max(m) + 1 if m else 0
    n_in, n_mid = max_key(qs.qapSetInput), max_key(qs.qapSetIntermediate)
    out = {0: qs.qapSetConstant}
    out.update({1 + k: v for k, v in qs.qapSetInput.items()})
    out.update({1 + n_in + k: v for k, v in qs.qapSetIntermediate.items()})
    out.update({1 + n_in + n_mid + k: v for k, v in qs.qapSetOutput.items()})
    return out


class GenQAP:
    """`GenQAP (Map k) k`: the evaluation-form QAP.  Here: a device-resident sparse constraint
    system (never densified) plus the circuit's wire numbering."""

    def __init__(self, ctx: Context, r1cs: R1CS, n_inputs: int, n_intermediates: int, n_outputs: int):
        self.ctx, self.r1cs = ctx, r1cs
        self.n_inputs, self.n_intermediates, self.n_outputs = n_inputs, n_intermediates, n_outputs

    def flat_index(self, wire: Wire) -> int:
        base = (1, 1 + self.n_inputs, 1 + self.n_inputs + self.n_intermediates)[wire.kind]
        return base + wire.index

    def witness_vector(self, assignment: QapSet) -> np.ndarray:
        """Flat w[m]; wires the QAP does not know contribute nothing, wires the assignment lacks
        are 0 (`combineWithDefaults`, src/QAP.hs:163-181,314)."""
        p = self.ctx.p
        vals = [0] * self.r1cs.m
        vals[0] = assignment.qapSetConstant % p
        for part, base, size in ((assignment.qapSetInput, 1, self.n_inputs),
                                 (assignment.qapSetIntermediate, 1 + self.n_inputs, self.n_intermediates),
                                 (assignment.qapSetOutput, 1 + self.n_inputs + self.n_intermediates, self.n_outputs)):
            for k, v in part.items():
                if 0 <= k < size:
                    vals[base + k] = v % p
        return ints_to_fr(vals)


class QAP:
    """`data QAP f` for the FFT path: qapTarget = x^N - 1; the per-wire polynomials are
    materialised on demand (3*m*N coefficients do not fit any memory at bench sizes)."""

    def __init__(self, gen: GenQAP):
        self.gen = gen

    @property
    def qapTarget(self) -> List[int]:
        N = 1 << self.gen.r1cs.log_n
        return [self.gen.ctx.p - 1] + [0] * (N - 1) + [1]

    def _polys(self, matrix: int, wire: Wire = None, flat: int = None) -> List[int]:
        k = flat if flat is not None else self.gen.flat_index(wire)
        coeffs, lens = self.gen.r1cs.qap_columns(matrix, k, 1)
        return fr_to_ints(coeffs[0, : int(lens[0])])

    def qapInputsLeft(self, wire: Wire = None, flat: int = None) -> List[int]:
        return self._polys(0, wire, flat)

    def qapInputsRight(self, wire: Wire = None, flat: int = None) -> List[int]:
        return self._polys(1, wire, flat)

    def qapOutputs(self, wire: Wire = None, flat: int = None) -> List[int]:
        return self._polys(2, wire, flat)


class NaiveQAP(QAP):
    """`data QAP f` produced by `createPolynomials` / `arithCircuitToQAP` (src/QAP.hs:486-508,
    542-549): interpolation on the actual root values, qapTarget = prod (x - r)."""

    def __init__(self, gen: GenQAP, sorted_roots: Sequence[int]):
        super().__init__(gen)
        self.naive = Naive(gen.r1cs, sorted_roots)

    @property
    def qapTarget(self) -> List[int]:
        return fr_to_ints(self.naive.target())

    def _polys(self, matrix: int, wire: Wire = None, flat: int = None) -> List[int]:
        k = flat if flat is not None else self.gen.flat_index(wire)
        coeffs, lens = self.naive.columns(matrix, k, 1)
        return fr_to_ints(coeffs[0, : int(lens[0])])


def arithCircuitToGenQAP(ctx: Context, roots: Optional[Sequence[Sequence[int]]], circuit: ArithCircuit) -> GenQAP:
    """src/QAP.hs:530-539.  `roots` = one list per gate (None = fresh numbering).  The lists cross the C ABI as they are
    (acx_circuit_to_r1cs_lists with ACX_ROOTS_REFERENCE_SEMANTICS): a list of the wrong length for its gate is the reference's
    panic (src/QAP.hs:445,474) = AcxError ROOT_COUNT; repeated roots, surplus and missing lists give what the reference gives
    (`Map.fromList` merging, `addMissingZeroes`, `zipWith` truncation)."""
    c = circuit.marshal(ctx.field)
    if roots is None:
        r = c.to_r1cs(ctx)
    else:
        r = c.to_r1cs_lists(ctx, [[x % ctx.p for x in rs] for rs in roots])
    return GenQAP(ctx, r, c.n_inputs, c.n_intermediates, c.n_outputs)


def createPolynomialsFFT(gen: GenQAP) -> QAP:
    """src/QAP.hs:512-525.  The `primRoots` argument lives in the Context."""
    return QAP(gen)


def arithCircuitToQAPFFT(ctx: Context, roots, circuit: ArithCircuit) -> QAP:
    return createPolynomialsFFT(arithCircuitToGenQAP(ctx, roots, circuit))


def createPolynomials(gen: GenQAP, roots: Sequence[Sequence[int]]) -> NaiveQAP:
    """src/QAP.hs:486-508.  The GenQAP here does not carry the root values (rows are stored in
    ascending-root order), so they are passed again."""
    p = gen.ctx.p
    return NaiveQAP(gen, sorted({r % p for rs in roots for r in rs}))      # the Map's keys: distinct, ascending


def arithCircuitToQAP(ctx: Context, roots: Sequence[Sequence[int]], circuit: ArithCircuit) -> NaiveQAP:
    """src/QAP.hs:542-549."""
    return createPolynomials(arithCircuitToGenQAP(ctx, roots, circuit), roots)     # `concat rootsPerGate`: surplus lists included


def gateToQAP(ctx: Context, roots: Sequence[int], gate: Gate) -> QAP:
    return arithCircuitToQAPFFT(ctx, [list(roots)], ArithCircuit([gate]))


def verifyAssignment(qap: QAP, assignment: QapSet) -> bool:
    ok, _, _ = qap.gen.r1cs.verify(qap.gen.witness_vector(assignment))
    return ok


def verifyAssignments(qap: QAP, assignments: Sequence[QapSet]) -> List[bool]:
    """`map (verifyAssignment qap) assignments` -- the shape of test/Test/Circuit/Arithmetic.hs:209 (one QAP, many
    assignments) -- in one call across the C ABI (acx_r1cs_verify_many: one PCIe copy and one batched launch)."""
    if not assignments:
        return []
    w = np.stack([qap.gen.witness_vector(a) for a in assignments])
    ok, _, _ = qap.gen.r1cs.verify_many(w)
    return [bool(x) for x in ok]


def verificationWitnessZk(delta1: int, delta2: int, delta3: int, qap: QAP, assignment: QapSet) -> Optional[List[int]]:
    p = qap.gen.ctx.p
    d = [delta1 % p, delta2 % p, delta3 % p]
    if isinstance(qap, NaiveQAP):
        h, ok = qap.naive.h(qap.gen.witness_vector(assignment), d)
    else:
        h, ok = qap.gen.r1cs.qap_h(qap.gen.witness_vector(assignment), d)
    return fr_to_ints(h) if ok else None


def verificationWitness(qap: QAP, assignment: QapSet) -> Optional[List[int]]:
    return verificationWitnessZk(0, 0, 0, qap, assignment)


def _assignment_from_flat(c: Circuit, w: np.ndarray, assigned: np.ndarray) -> QapSet:
    vals = fr_to_ints(w)
    qs = QapSet(vals[0])
    base = 1
    for part, size in ((qs.qapSetInput, c.n_inputs), (qs.qapSetIntermediate, c.n_intermediates),
                       (qs.qapSetOutput, c.n_outputs)):
        for k in range(size):
            if assigned[base + k]:
                part[k] = vals[base + k]
        base += size
    return qs


def generateAssignment(circuit: ArithCircuit, inputs: Dict[int, int], field: str = "bn254") -> QapSet:
    """src/QAP.hs:597-603 (host-sequential, like the reference; needs no GPU)."""
    from .engine import FIELDS
    p = FIELDS[field][1]
    c = circuit.marshal(field)
    n = max(list(inputs) + [-1]) + 1
    vals = [inputs.get(i, 0) % p for i in range(n)]
    present = np.array([1 if i in inputs else 0 for i in range(n)], dtype=np.uint8)
    w, assigned = c.eval(ints_to_fr(vals) if n else np.zeros((0, 4), dtype=np.uint64), present)
    qs = _assignment_from_flat(c, w, assigned)
    for i, v in inputs.items():          # inputs the circuit never mentions stay in the QapSet
        qs.qapSetInput.setdefault(i, v % p)
    return qs


def generateAssignmentGate(gate: Gate, inputs: Dict[int, int], field: str = "bn254") -> QapSet:
    return generateAssignment(ArithCircuit([gate]), inputs, field)
