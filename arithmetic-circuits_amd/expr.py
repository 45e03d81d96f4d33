"""Host-side mirror of the reference's expression language and circuit builder:

  Circuit.Expr  src/Circuit/Expr.hs:31-63 (Expr GADT), :141-180 (evalExpr), :186-217 (builder monad),
                :247-305 (compile)
  Circuit.Lang  src/Circuit/Lang.hs:26-78 (c add sub mul and_ or_ xor_ not_ eq deref e cond ret input)

Pure Python data + a builder object standing in for the `ExprM` state monad; no arithmetic on the
hot path happens here -- it produces the `ArithCircuit` that `arithCircuitToQAPFFT` consumes, so that
`Example.hs` can be written from its source form.  The builder reproduces the reference's quirks:
ONE shared fresh-name counter for input, intermediate and output wires (Expr.hs:201-217) and `ret`
returning an existing wire without emitting a gate when the expression already is one
(Lang.hs:67-75).  NB: in the reference `evalExpr`'s `truncRotate` moves bit ix to (ix + r) mod n
(Expr.hs:121-138) while `compile` builds `unsplit (rotateList r outputs)` which moves bit ix to
(ix - r) mod n (Expr.hs:228-229,265-269); the two agree only for r = 0 (mod n).  Its test generator
never produces URot (test/Test/Circuit/Expr.hs:32-46), so the mismatch is invisible there; both
behaviours are mirrored as they are."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple, Union

from .circuit import (Add, AffineCircuit, ArithCircuit, ConstGate, Equal, Gate, InputWire, IntermediateWire, Mul,
                      OutputWire, ScalarMul, Split, Var, Wire, unsplit)


# ---- Expr (src/Circuit/Expr.hs:31-63) ---------------------------------------------------------------
@dataclass(frozen=True)
class EConst:
    value: int


@dataclass(frozen=True)
class EConstBool:
    value: bool


@dataclass(frozen=True)
class EVar:
    var: object


@dataclass(frozen=True)
class EVarBool:
    var: object


@dataclass(frozen=True)
class EUnOp:
    op: tuple          # ("UNeg",) | ("UNot",) | ("URot", truncBits, rotBits)
    e1: "Expr"


@dataclass(frozen=True)
class EBinOp:
    op: str            # BAdd BSub BMul BAnd BOr BXor
    e1: "Expr"
    e2: "Expr"


@dataclass(frozen=True)
class EIf:
    cond: "Expr"
    true: "Expr"
    false: "Expr"


@dataclass(frozen=True)
class EEq:
    lhs: "Expr"
    rhs: "Expr"


Expr = Union[EConst, EConstBool, EVar, EVarBool, EUnOp, EBinOp, EIf, EEq]


# ---- Circuit.Lang sugar (src/Circuit/Lang.hs:26-78) -----------------------------------------------
def c(f: int) -> Expr:
    return EConst(f)


def add(a: Expr, b: Expr) -> Expr:
    return EBinOp("BAdd", a, b)


def sub(a: Expr, b: Expr) -> Expr:
    return EBinOp("BSub", a, b)


def mul(a: Expr, b: Expr) -> Expr:
    return EBinOp("BMul", a, b)


def and_(a: Expr, b: Expr) -> Expr:
    return EBinOp("BAnd", a, b)


def or_(a: Expr, b: Expr) -> Expr:
    return EBinOp("BOr", a, b)


def xor_(a: Expr, b: Expr) -> Expr:
    return EBinOp("BXor", a, b)


def not_(a: Expr) -> Expr:
    return EUnOp(("UNot",), a)


def eq(a: Expr, b: Expr) -> Expr:
    return EEq(a, b)


def deref(w: Wire) -> Expr:
    return EVar(w)


def cond(b: Expr, t: Expr, f: Expr) -> Expr:
    return EIf(b, t, f)


def rotate_list(steps: int, xs: list) -> list:
    """src/Circuit/Expr.hs:228-229."""
    n = len(xs)
    return [xs[(i + steps) % n] for i in range(n)] if n else []


# ---- evalExpr (src/Circuit/Expr.hs:141-180) -------------------------------------------------------
def evalExpr(lookup: Callable[[object, object], Optional[int]], expr: Expr, vars_, p: int):
    def ev(e):
        if isinstance(e, EConst):
            return e.value % p
        if isinstance(e, EConstBool):
            return bool(e.value)
        if isinstance(e, EVar):
            v = lookup(e.var, vars_)
            if v is None:
                raise KeyError("evalExpr: incorrect var lookup")      # Expr.hs:156 panic
            return v % p
        if isinstance(e, EVarBool):
            v = lookup(e.var, vars_)
            if v is None:
                raise KeyError("evalExpr: incorrect var lookup")      # Expr.hs:159 panic
            return (v % p) == 1
        if isinstance(e, EUnOp):
            x = ev(e.e1)
            if e.op[0] == "UNeg":
                return (-x) % p
            if e.op[0] == "UNot":
                return not x
            trunc, rot = e.op[1], e.op[2]                              # URot: truncRotate, Expr.hs:121-138
            return sum(1 << ((ix + rot) % trunc) for ix in range(trunc) if (x >> ix) & 1) % p
        if isinstance(e, EBinOp):
            a, b = ev(e.e1), ev(e.e2)
            return {"BAdd": lambda: (a + b) % p, "BSub": lambda: (a - b) % p, "BMul": lambda: a * b % p,
                    "BAnd": lambda: bool(a) and bool(b), "BOr": lambda: bool(a) or bool(b),
                    "BXor": lambda: bool(a) != bool(b)}[e.op]()
        if isinstance(e, EIf):
            return ev(e.true) if ev(e.cond) else ev(e.false)
        if isinstance(e, EEq):
            return ev(e.lhs) == ev(e.rhs)
        raise TypeError(e)
    return ev(expr)


# ---- builder = ExprM state monad (src/Circuit/Expr.hs:186-245) -----------------------------------------
class CircuitBuilder:
    def __init__(self):
        self.gates: List[Gate] = []
        self.counter = 0

    def fresh(self) -> int:
        v = self.counter
        self.counter += 1
        return v

    def imm(self) -> Wire:
        return IntermediateWire(self.fresh())

    def freshInput(self) -> Wire:
        return InputWire(self.fresh())

    def freshOutput(self) -> Wire:
        return OutputWire(self.fresh())

    input = freshInput                      # Lang.hs:77-78

    def emit(self, g: Gate) -> None:
        self.gates.append(g)

    @staticmethod
    def _add_var(x) -> AffineCircuit:       # addVar, Expr.hs:232-234
        return Var(x[1]) if x[0] == "L" else x[1]

    def _add_wire(self, x) -> Wire:         # addWire, Expr.hs:237-242
        if x[0] == "L":
            return x[1]
        out = self.imm()
        self.emit(Mul(ConstGate(1), x[1], out))
        return out

    def _mul_to_imm(self, l: AffineCircuit, r: AffineCircuit) -> Wire:
        o = self.imm()
        self.emit(Mul(l, r, o))
        return o

    def compile(self, expr: Expr):
        """src/Circuit/Expr.hs:247-305.  Returns ("L", wire) or ("R", affine circuit)."""
        if isinstance(expr, EConst):
            return ("R", ConstGate(expr.value))
        if isinstance(expr, EConstBool):
            return ("R", ConstGate(1 if expr.value else 0))
        if isinstance(expr, (EVar, EVarBool)):
            return ("L", expr.var)
        if isinstance(expr, EUnOp):
            e1 = self.compile(expr.e1)
            if expr.op[0] == "UNeg":
                return ("R", ScalarMul(-1, self._add_var(e1)))
            if expr.op[0] == "UNot":
                return ("R", Add(ConstGate(1), ScalarMul(-1, self._add_var(e1))))
            trunc, rot = expr.op[1], expr.op[2]
            inp = self._add_wire(e1)
            outputs = [self.imm() for _ in range(trunc)]
            self.emit(Split(inp, outputs))
            return ("R", unsplit(rotate_list(rot, outputs)))
        if isinstance(expr, EBinOp):
            a = self._add_var(self.compile(expr.e1))
            b = self._add_var(self.compile(expr.e2))
            if expr.op == "BAdd":
                return ("R", Add(a, b))
            if expr.op == "BSub":
                return ("R", Add(a, ScalarMul(-1, b)))
            if expr.op in ("BMul", "BAnd"):
                return ("L", self._mul_to_imm(a, b))
            tmp = self.imm()
            self.emit(Mul(a, b, tmp))
            k = -1 if expr.op == "BOr" else -2          # OR: a+b-ab ; XOR: a+b-2ab
            return ("R", Add(Add(a, b), ScalarMul(k, Var(tmp))))
        if isinstance(expr, EIf):
            cnd = self._add_var(self.compile(expr.cond))
            t = self._add_var(self.compile(expr.true))
            f = self._add_var(self.compile(expr.false))
            t1, t2 = self.imm(), self.imm()
            self.emit(Mul(cnd, t, t1))
            self.emit(Mul(Add(ConstGate(1), ScalarMul(-1, cnd)), f, t2))
            return ("R", Add(Var(t1), Var(t2)))
        if isinstance(expr, EEq):
            diff = self.compile(EBinOp("BSub", expr.lhs, expr.rhs))
            eq_in = self._add_wire(diff)
            eq_free, eq_out = self.imm(), self.imm()
            self.emit(Equal(eq_in, eq_free, eq_out))
            return ("R", Add(ConstGate(1), ScalarMul(-1, Var(eq_out))))
        raise TypeError(expr)

    def _compile_with_wire(self, fresh_wire: Callable[[], Wire], expr: Expr) -> Wire:   # Lang.hs:67-75
        out = self.compile(expr)
        if out[0] == "L":
            return out[1]
        w = fresh_wire()
        self.emit(Mul(ConstGate(1), out[1], w))
        return w

    def e(self, expr: Expr) -> Wire:
        return self._compile_with_wire(self.imm, expr)

    def ret(self, expr: Expr) -> Wire:
        return self._compile_with_wire(self.freshOutput, expr)

    def exprToArithCircuit(self, expr: Expr, output: Wire) -> None:
        """src/Circuit/Expr.hs:308-322 (variables of `expr` are input indices)."""
        out = self.compile(mapVarsExpr(InputWire, expr))
        self.emit(Mul(ConstGate(1), self._add_var(out), output))


def mapVarsExpr(f: Callable, expr: Expr) -> Expr:
    if isinstance(expr, EVar):
        return EVar(f(expr.var))
    if isinstance(expr, EVarBool):
        return EVarBool(f(expr.var))
    if isinstance(expr, (EConst, EConstBool)):
        return expr
    if isinstance(expr, EBinOp):
        return EBinOp(expr.op, mapVarsExpr(f, expr.e1), mapVarsExpr(f, expr.e2))
    if isinstance(expr, EUnOp):
        return EUnOp(expr.op, mapVarsExpr(f, expr.e1))
    if isinstance(expr, EIf):
        return EIf(mapVarsExpr(f, expr.cond), mapVarsExpr(f, expr.true), mapVarsExpr(f, expr.false))
    return EEq(mapVarsExpr(f, expr.lhs), mapVarsExpr(f, expr.rhs))


def execCircuitBuilder(program: Callable[[CircuitBuilder], object]) -> ArithCircuit:
    """`execCircuitBuilder :: ExprM f a -> ArithCircuit f` (src/Circuit/Expr.hs:188-191): `program`
    is a function of the builder standing in for the do-block."""
    b = CircuitBuilder()
    program(b)
    return ArithCircuit(b.gates)
