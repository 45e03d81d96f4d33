"""Host-side mirror of the reference's circuit types, same names and argument meaning, so that
callers (and the parity tests) read like the reference's own code:

  Wire / InputWire / IntermediateWire / OutputWire   src/Circuit/Arithmetic.hs:32-36
  AffineCircuit: Add / ScalarMul / ConstGate / Var    src/Circuit/Affine.hs:26-31
  Gate: Mul / Equal / Split                           src/Circuit/Arithmetic.hs:44-59
  ArithCircuit                                        src/Circuit/Arithmetic.hs:149-150
  generateRoots, unsplit, fresh                       src/Circuit/Arithmetic.hs:194-244, src/Fresh.hs

These are plain data carriers: all arithmetic happens behind the C ABI (libacx.so)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, NamedTuple, Optional, Sequence, Union

import numpy as np

from . import _lib
from .engine import Circuit, FIELDS, ints_to_fr


class Wire(NamedTuple):
    kind: int   # 0 InputWire, 1 IntermediateWire, 2 OutputWire  (derived Ord = this order)
    index: int


def InputWire(i: int) -> Wire:
    return Wire(0, i)


def IntermediateWire(i: int) -> Wire:
    return Wire(1, i)


def OutputWire(i: int) -> Wire:
    return Wire(2, i)


@dataclass(frozen=True)
class Add:
    left: "AffineCircuit"
    right: "AffineCircuit"


@dataclass(frozen=True)
class ScalarMul:
    scalar: int
    expr: "AffineCircuit"


@dataclass(frozen=True)
class ConstGate:
    value: int


@dataclass(frozen=True)
class Var:
    wire: Wire


AffineCircuit = Union[Add, ScalarMul, ConstGate, Var]


@dataclass(frozen=True)
class Mul:
    mulLeft: AffineCircuit
    mulRight: AffineCircuit
    mulOutput: Wire


@dataclass(frozen=True)
class Equal:
    eqInput: Wire
    eqMagic: Wire
    eqOutput: Wire


@dataclass(frozen=True)
class Split:
    splitInput: Wire
    splitOutputs: Sequence[Wire]


Gate = Union[Mul, Equal, Split]


def unsplit(wires: Sequence[Wire]) -> AffineCircuit:
    """src/Circuit/Arithmetic.hs:238-244."""
    rest: AffineCircuit = ConstGate(0)
    for ix, w in enumerate(wires):
        rest = Add(rest, ScalarMul(2 ** ix, Var(w)))
    return rest


class ArithCircuit:
    """`newtype ArithCircuit f = ArithCircuit [Gate Wire f]`."""

    def __init__(self, gates: Sequence[Gate]):
        self.gates = list(gates)

    # -- marshalling into the flat acx_gate_list of include/acx.h ------------------------------
    def marshal(self, field: str = "bn254") -> Circuit:
        p = FIELDS[field][1]
        kind, tok_ofs, tok_op, tok_arg = [], [0], [], []
        scalars: List[int] = []
        aff_wires: List[Wire] = []
        wire_ofs, wires = [0], []

        def emit(node: AffineCircuit):
            # iterative pre-order walk (unsplit chains are hundreds of nodes deep)
            stack = [node]
            while stack:
                nd = stack.pop()
                if isinstance(nd, Var):
                    tok_op.append(3)
                    tok_arg.append(len(aff_wires))
                    aff_wires.append(nd.wire)
                elif isinstance(nd, ConstGate):
                    tok_op.append(2)
                    tok_arg.append(len(scalars))
                    scalars.append(nd.value % p)      # `fromInteger` reduces mod p
                elif isinstance(nd, ScalarMul):
                    tok_op.append(1)
                    tok_arg.append(len(scalars))
                    scalars.append(nd.scalar % p)
                    stack.append(nd.expr)
                elif isinstance(nd, Add):
                    tok_op.append(0)
                    tok_arg.append(0)
                    stack.append(nd.right)
                    stack.append(nd.left)
                else:
                    raise TypeError(f"not an AffineCircuit node: {nd!r}")

        for g in self.gates:
            if isinstance(g, Mul):
                kind.append(0)
                emit(g.mulLeft)
                tok_ofs.append(len(tok_op))
                emit(g.mulRight)
                tok_ofs.append(len(tok_op))
                wires.append(g.mulOutput)
            elif isinstance(g, Equal):
                kind.append(1)
                tok_ofs += [len(tok_op)] * 2
                wires += [g.eqInput, g.eqMagic, g.eqOutput]
            elif isinstance(g, Split):
                kind.append(2)
                tok_ofs += [len(tok_op)] * 2
                wires += [g.splitInput] + list(g.splitOutputs)
            else:
                raise TypeError(f"not a Gate: {g!r}")
            wire_ofs.append(len(wires))

        def wire_array(ws):
            arr = np.zeros((max(len(ws), 1), 2), dtype=np.uint32)
            for i, w in enumerate(ws):
                arr[i, 0], arr[i, 1] = w.kind, w.index
            return arr

        a_kind = np.array(kind, dtype=np.uint8) if kind else np.zeros(1, dtype=np.uint8)
        a_tok_ofs = np.array(tok_ofs, dtype=np.uint64)
        a_tok_op = np.array(tok_op, dtype=np.uint8) if tok_op else np.zeros(1, dtype=np.uint8)
        a_tok_arg = np.array(tok_arg, dtype=np.uint32) if tok_arg else np.zeros(1, dtype=np.uint32)
        a_scalars = ints_to_fr(scalars) if scalars else np.zeros((1, 4), dtype=np.uint64)
        a_aff = wire_array(aff_wires)
        a_wire_ofs = np.array(wire_ofs, dtype=np.uint64)
        a_wires = wire_array(wires)
        gl = _lib.GateList(len(self.gates), a_kind.ctypes.data, a_tok_ofs.ctypes.data, a_tok_op.ctypes.data,
                           a_tok_arg.ctypes.data, a_scalars.ctypes.data, len(scalars), a_aff.ctypes.data,
                           len(aff_wires), a_wire_ofs.ctypes.data, a_wires.ctypes.data)
        keep = (a_kind, a_tok_ofs, a_tok_op, a_tok_arg, a_scalars, a_aff, a_wire_ofs, a_wires)
        return Circuit(field, gl, keep)


def generateRoots(takeRoot: Callable[[], int], circuit: ArithCircuit) -> List[List[int]]:
    """src/Circuit/Arithmetic.hs:194-216: Mul -> 1 root, Equal -> 2, Split -> 1 + #outputs."""
    out = []
    for g in circuit.gates:
        if isinstance(g, Mul):
            out.append([takeRoot()])
        elif isinstance(g, Equal):
            out.append([takeRoot(), takeRoot()])
        else:
            out.append([takeRoot() for _ in range(1 + len(g.splitOutputs))])
    return out


def freshRoots(circuit: ArithCircuit, offset: int = 0) -> List[List[int]]:
    """`evalFresh $ generateRoots (fromIntegral . (+offset) <$> fresh) circuit` (src/Fresh.hs:13-20)."""
    counter = [0]

    def take() -> int:
        v = counter[0]
        counter[0] += 1
        return v + offset

    return generateRoots(take, circuit)
