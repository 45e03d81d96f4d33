"""Synthetic random circuits for parity tests at scale and for bench.py.

`mulgraph(n, n_in, k, window, seed)`: n `Mul` gates; each side is sum_t s_t * Var(x_t) (+ a
ConstGate with probability 1/3, mirroring test/Test/Circuit/Arithmetic.hs:59-64); x_t is an input
w.p. 1/2, else one of the last `window` intermediate outputs; gate g writes IntermediateWire g,
the last gate OutputWire 0.  All randomness is counter-based (SplitMix64 on (seed, stream,
index)) so any party regenerates identical bytes.  The marshalled gate list is built directly as
flat numpy arrays (no per-gate Python objects), then handed to the C ABI like any other circuit."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .engine import Circuit, FIELDS

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        return splitmix64(idx.astype(np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base)


def random_fr(count: int, seed: int, stream: int, field: str = "bn254") -> np.ndarray:
    """`count` uniform elements of [0,p) as (count,4) uint64, by rejection sampling."""
    p = FIELDS[field][1]
    bits = p.bit_length()
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    pl = [np.uint64((p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) for i in range(4)]
    out = np.zeros((count, 4), dtype=np.uint64)
    pending = np.arange(count, dtype=np.uint64)
    for attempt in range(24):
        if pending.size == 0:
            break
        cand = np.empty((pending.size, 4), dtype=np.uint64)
        for j in range(4):
            cand[:, j] = _stream(seed, stream * 4 + j, pending * np.uint64(32) + np.uint64(attempt))
        cand[:, 3] &= top_mask
        lt = np.zeros(pending.size, dtype=bool)
        eq = np.ones(pending.size, dtype=bool)
        for j in (3, 2, 1, 0):
            lt |= eq & (cand[:, j] < pl[j])
            eq &= cand[:, j] == pl[j]
        out[pending[lt]] = cand[lt]
        pending = pending[~lt]
    if pending.size:  # astronomically unlikely; stay deterministic
        out[pending, 0] = pending
    return out


def small_fr(count: int, seed: int, stream: int, field: str = "bn254", bits: int = 16) -> np.ndarray:
    """`count` coefficients of a compiled program's shape (src/Circuit/Expr.hs:256-305): +-c with 1 <= c <= 2^bits,
    negative ones as p - c; (count,4) uint64 canonical."""
    p = FIELDS[field][1]
    idx = np.arange(count, dtype=np.uint64)
    r = _stream(seed, stream * 4, idx)
    mag = (r % np.uint64(1 << bits)) + np.uint64(1)
    neg = ((r >> np.uint64(40)) & np.uint64(1)) == 1
    out = np.zeros((count, 4), dtype=np.uint64)
    out[:, 0] = mag
    pl = [(p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    # p - c for c < 2^64: borrow only from limb 0 (p's low limb exceeds 2^bits for both fields)
    assert pl[0] > (1 << bits)
    with np.errstate(over="ignore"):
        out[neg, 0] = np.uint64(pl[0]) - mag[neg]
    for j in (1, 2, 3):
        out[neg, j] = np.uint64(pl[j])
    return out


@dataclass
class SynthCircuit:
    circuit: Circuit           # marshalled (host) circuit
    inputs: np.ndarray         # (n_in, 4) canonical
    n: int
    n_in: int
    k: int

    def rows(self):
        return self.circuit.rows()

    def witness(self) -> np.ndarray:
        w, _ = self.circuit.eval(self.inputs)
        return w


def mulgraph(n: int, n_in: int = 1024, k: int = 2, window: int = 4096, seed: int = 0xAC355,
             field: str = "bn254", coeff: str = "random") -> SynthCircuit:
    if n < 1 or n_in < 1 or k < 1:
        raise ValueError("n, n_in, k must be positive")
    sides = 2 * n
    side_idx = np.arange(sides, dtype=np.uint64)
    gate_of_side = (side_idx // np.uint64(2)).astype(np.int64)
    has_const = (_stream(seed, 1, side_idx) % np.uint64(3)) == 0
    # term wires
    t_idx = np.arange(sides * k, dtype=np.uint64)
    gate_of_term = np.repeat(gate_of_side, k)
    use_mid = ((_stream(seed, 2, t_idx) & np.uint64(1)) == 1) & (gate_of_term > 0)
    r = _stream(seed, 3, t_idx)
    span = np.minimum(gate_of_term, window).astype(np.uint64)
    span_safe = np.maximum(span, np.uint64(1))
    mid_index = gate_of_term.astype(np.uint64) - np.uint64(1) - (r % span_safe)
    inp_index = r % np.uint64(n_in)
    aff_wires = np.zeros((sides * k, 2), dtype=np.uint32)
    aff_wires[:, 0] = use_mid.astype(np.uint32)          # 1 = IntermediateWire, 0 = InputWire
    aff_wires[:, 1] = np.where(use_mid, mid_index, inp_index).astype(np.uint32)
    # scalars: k coefficients per side, then one constant per side that has one
    n_const = int(has_const.sum())
    if coeff == "random":          # uniform in [0, p)
        coeffs, consts = random_fr(sides * k, seed, 10, field), random_fr(n_const, seed, 11, field)
    elif coeff == "small":         # the shape of a compiled program: +-c, c <= 2^16
        coeffs, consts = small_fr(sides * k, seed, 10, field), small_fr(n_const, seed, 11, field)
    else:
        raise ValueError("coeff must be 'random' or 'small'")
    scalars = np.concatenate([coeffs, consts], axis=0)
    const_slot = np.cumsum(has_const) - 1 + sides * k       # scalar index of a side's constant
    # tokens per side: (k-1) ADDs [+1 ADD if const], k x (SMUL, VAR) [, CONST]
    #   pre-order of  Add(t1, Add(t2, ... Add(t_k, const)))   /   Add(t1, ... Add(t_{k-1}, t_k))
    tok_per_side = (3 * k - 1) + np.where(has_const, 2, 0)
    side_ofs = np.concatenate([[0], np.cumsum(tok_per_side)]).astype(np.int64)
    n_tok = int(side_ofs[-1])
    tok_op = np.zeros(n_tok, dtype=np.uint8)
    tok_arg = np.zeros(n_tok, dtype=np.uint32)
    base = side_ofs[:-1]
    n_adds = (k - 1) + has_const.astype(np.int64)
    for t in range(k):
        # position of term t: ADDs interleave: ADD t1 ADD t2 ... ; term t preceded by min(t+1, n_adds) ADDs
        adds_before = np.minimum(t + 1, n_adds)
        pos = base + adds_before + 2 * t
        # mark the ADD that precedes this term (if any new one)
        new_add = adds_before > np.minimum(t, n_adds)
        tok_op[(pos - 1)[new_add]] = 0
        tok_op[pos] = 1
        tok_arg[pos] = (np.arange(sides) * k + t).astype(np.uint32)
        tok_op[pos + 1] = 3
        tok_arg[pos + 1] = (np.arange(sides) * k + t).astype(np.uint32)
    cpos = (base + n_adds + 2 * k)[has_const]
    tok_op[cpos] = 2
    tok_arg[cpos] = const_slot[has_const].astype(np.uint32)
    tok_ofs = side_ofs.astype(np.uint64)
    kind = np.zeros(n, dtype=np.uint8)
    wires = np.zeros((n, 2), dtype=np.uint32)
    wires[:, 0] = 1
    wires[:, 1] = np.arange(n, dtype=np.uint32)
    wires[n - 1] = (2, 0)                                  # last gate -> OutputWire 0
    wire_ofs = np.arange(n + 1, dtype=np.uint64)
    gl = _lib.GateList(n, kind.ctypes.data, tok_ofs.ctypes.data, tok_op.ctypes.data, tok_arg.ctypes.data,
                       scalars.ctypes.data, scalars.shape[0], aff_wires.ctypes.data, aff_wires.shape[0],
                       wire_ofs.ctypes.data, wires.ctypes.data)
    circ = Circuit(field, gl, (kind, tok_ofs, tok_op, tok_arg, scalars, aff_wires, wire_ofs, wires))
    inputs = random_fr(n_in, seed, 12, field)
    return SynthCircuit(circ, inputs, n, n_in, k)


def gatemix(n_gates: int, n_in: int = 64, seed: int = 0x6A7E, field: str = "bn254", weights=(50, 10, 1), split_bits: int = 256,
            window: int = 4096) -> SynthCircuit:
    """`n_gates` gates in the reference's own generator mix (test/Test/Circuit/Arithmetic.hs:69-136: Mul : Equal : Split =
    50 : 10 : 1, Split always 256 bits wide): a Mul gate multiplies two affine sides s1 * Var(x1) + s2 * Var(x2), x an input or
    one of the last `window` intermediate wires; an Equal or Split gate reads an earlier intermediate wire.  Every gate writes
    fresh IntermediateWires (Mul 1, Equal 2: magic and output, Split `split_bits`), so the list is in single-assignment form and
    Split rows (257 entries) take the long-row path.  Counter-based randomness like mulgraph; flat arrays, no per-gate objects."""
    g = np.arange(n_gates, dtype=np.uint64)
    pick = _stream(seed, 1, g) % np.uint64(sum(weights))
    kind = np.where(pick < weights[0], 0, np.where(pick < weights[0] + weights[1], 1, 2)).astype(np.uint8)
    kind[0] = 0                                              # the first gate has no intermediate wire to read
    new = np.where(kind == 0, 1, np.where(kind == 1, 2, split_bits)).astype(np.int64)
    first = np.concatenate([[0], np.cumsum(new)])[:-1]       # first intermediate wire a gate writes = wires before it
    # an Equal gate's magic wire is written but is not one of its `outputWires` (src/Circuit/Arithmetic.hs:158-185): a read
    # that lands on one moves to the gate's output wire next to it, so that validArithCircuit holds
    magic = np.zeros(int(first[-1] + new[-1]) + 1, dtype=np.uint32)
    magic[first[kind == 1]] = 1
    mul = np.nonzero(kind == 0)[0]
    nm = mul.shape[0]
    # Mul gates: two sides x two terms
    t = np.arange(4 * nm, dtype=np.uint64)
    gate_of_t = np.repeat(mul, 4)
    before = first[gate_of_t].astype(np.uint64)
    use_mid = ((_stream(seed, 2, t) & np.uint64(1)) == 1) & (before > 0)
    r = _stream(seed, 3, t)
    span = np.maximum(np.minimum(before, np.uint64(window)), np.uint64(1))
    aff = np.zeros((4 * nm, 2), dtype=np.uint32)
    aff[:, 0] = use_mid.astype(np.uint32)
    mid = (before - np.uint64(1) - (r % span)).astype(np.int64)
    mid = np.where(use_mid, mid, 0)
    aff[:, 1] = np.where(use_mid, mid + magic[mid], r % np.uint64(n_in)).astype(np.uint32)
    scalars = random_fr(4 * nm, seed, 10, field)
    # tokens per Mul side: ADD, SMUL, VAR, SMUL, VAR
    tok_per_gate = np.where(kind == 0, 10, 0).astype(np.int64)
    side = np.zeros(2 * n_gates + 1, dtype=np.int64)
    side[1::2] = np.where(kind == 0, 5, 0)
    side[2::2] = np.where(kind == 0, 5, 0)
    tok_ofs = np.concatenate([[0], np.cumsum(side[1:])]).astype(np.uint64)
    n_tok = int(tok_ofs[-1])
    tok_op = np.tile(np.array([0, 1, 3, 1, 3], dtype=np.uint8), 2 * nm)
    tok_arg = np.zeros(n_tok, dtype=np.uint32)
    term = np.arange(4 * nm, dtype=np.uint32)                # side-major: term 2 * side + {0, 1}
    base = (np.arange(2 * nm, dtype=np.int64) * 5)
    for j in (0, 1):
        tok_arg[base + 1 + 2 * j] = term[j::2]
        tok_arg[base + 2 + 2 * j] = term[j::2]
    assert tok_op.shape[0] == n_tok
    # wires per gate: Mul {out}; Equal {input, magic, out}; Split {input, bits...}
    nw = np.where(kind == 0, 1, np.where(kind == 1, 3, 1 + split_bits)).astype(np.int64)
    wire_ofs = np.concatenate([[0], np.cumsum(nw)]).astype(np.uint64)
    wires = np.zeros((int(wire_ofs[-1]), 2), dtype=np.uint32)
    wires[:, 0] = 1
    pos = wire_ofs[:-1].astype(np.int64)
    rin = _stream(seed, 4, g)
    src = (first.astype(np.uint64) - np.uint64(1) - (rin % np.maximum(np.minimum(first.astype(np.uint64), np.uint64(window)), np.uint64(1)))).astype(np.uint32)
    src = src + magic[np.where(first > 0, src, 0)]
    wires[pos[kind == 0], 1] = first[kind == 0]
    eq = kind == 1
    wires[pos[eq], 1] = src[eq]
    wires[pos[eq] + 1, 1] = first[eq]
    wires[pos[eq] + 2, 1] = first[eq] + 1
    sp = np.nonzero(kind == 2)[0]
    if sp.shape[0]:
        wires[pos[sp], 1] = src[sp]
        bits = (pos[sp] + 1).reshape(-1, 1) + np.arange(split_bits).reshape(1, -1)
        wires[bits.reshape(-1), 1] = (first[sp].reshape(-1, 1) + np.arange(split_bits).reshape(1, -1)).reshape(-1)
    gl = _lib.GateList(n_gates, kind.ctypes.data, tok_ofs.ctypes.data, tok_op.ctypes.data, tok_arg.ctypes.data,
                       scalars.ctypes.data, scalars.shape[0], aff.ctypes.data, aff.shape[0], wire_ofs.ctypes.data, wires.ctypes.data)
    circ = Circuit(field, gl, (kind, tok_ofs, tok_op, tok_arg, scalars, aff, wire_ofs, wires))
    return SynthCircuit(circ, random_fr(n_in, seed, 12, field), n_gates, n_in, 2)


class BlockSystem:
    """`blocks` block-diagonal copies of one mulgraph system that share only the constant wire: the shape of
    BASELINE.json configs[3] (2^24 constraints = 256 x 2^16) without ever materialising it on one host.  Any
    subset of the global rows -- e.g. one rank's block-cyclic share -- is produced directly (`rows_of`), so every
    process marshals only what it owns."""

    def __init__(self, base: SynthCircuit, blocks: int):
        self.base, self.blocks = base, blocks
        self.mats = base.rows()
        self.n0 = base.circuit.n_rows
        self.m0 = base.circuit.m
        self.n = self.n0 * blocks
        self.m = 1 + blocks * (self.m0 - 1)
        self._w0: Optional[np.ndarray] = None

    def witness(self) -> np.ndarray:
        if self._w0 is None:
            self._w0 = self.base.witness()
        return np.concatenate([self._w0[:1]] + [self._w0[1:]] * self.blocks)

    def wire(self, block: int, k: int) -> int:
        """global index of wire k (>= 1) of one block"""
        return 1 + block * (self.m0 - 1) + (k - 1)

    def full_rows(self) -> Tuple[Tuple[np.ndarray, np.ndarray, np.ndarray], ...]:
        """CSR triples (A, B, C) of the whole system (the base system's arrays tiled `blocks` times, wires renumbered per
        block): what a single-process host hands to acx_mgpu_r1cs_load."""
        out = []
        for rowptr, col, val in self.mats:
            rp = np.asarray(rowptr, dtype=np.int64)
            nnz0 = int(rp[-1])
            full_rp = np.concatenate([[0]] + [rp[1:] + b * nnz0 for b in range(self.blocks)])
            c = col.astype(np.int64)
            shift = (np.arange(self.blocks, dtype=np.int64) * (self.m0 - 1)).reshape(-1, 1)
            full_c = np.where(c.reshape(1, -1) == 0, 0, c.reshape(1, -1) + shift).reshape(-1)
            out.append((full_rp.astype(np.uint32), full_c.astype(np.uint32), np.tile(val, (self.blocks, 1))))
        return tuple(out)

    def rows_of(self, rows: np.ndarray) -> Tuple[Tuple[np.ndarray, np.ndarray, np.ndarray], ...]:
        """CSR triples (A, B, C) of the given global rows, in the given order; indices >= n are empty rows
        (the zero padding of the evaluation domain)."""
        rows = np.asarray(rows, dtype=np.int64)
        live = rows < self.n
        blk = np.where(live, rows // self.n0, 0)
        loc = np.where(live, rows % self.n0, 0)
        out = []
        for rowptr, col, val in self.mats:
            rp = np.asarray(rowptr, dtype=np.int64)
            lens = np.where(live, rp[loc + 1] - rp[loc], 0)
            new_rp = np.concatenate([[0], np.cumsum(lens)])
            total = int(new_rp[-1])
            # entry e of output row j comes from base entry rp[loc[j]] + (e - new_rp[j])
            owner = np.repeat(np.arange(rows.shape[0], dtype=np.int64), lens)
            src = rp[loc][owner] + (np.arange(total, dtype=np.int64) - new_rp[:-1][owner])
            c = col[src].astype(np.int64)
            c = np.where(c == 0, 0, c + blk[owner] * (self.m0 - 1))
            out.append((new_rp.astype(np.uint32), c.astype(np.uint32), np.ascontiguousarray(val[src])))
        return tuple(out)
