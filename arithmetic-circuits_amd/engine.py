"""Low-level object wrappers over the C ABI (include/acx.h): Context, Circuit, R1CS.

Bulk data crosses as numpy arrays: field elements are uint64 arrays of shape (..., 4)
(32-byte little-endian canonical integers)."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import AcxError, check

FIELDS = {
    "bn254": (_lib.FIELD_BN254_FR, 21888242871839275222246405745257275088548364400416034343698204186575808495617),
    "bls12_381": (_lib.FIELD_BLS12_381_FR, 52435875175126190479447740508185965837690552500527637822603658699938581184513),
}
_MASK64 = (1 << 64) - 1


def ints_to_fr(xs: Sequence[int]) -> np.ndarray:
    """Python ints (already reduced to [0,p)) -> (len, 4) uint64."""
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def fr_to_ints(a: np.ndarray) -> List[int]:
    raw = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None else a.ctypes.data


def _fr_array(a, count: Optional[int] = None) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    if count is not None and arr.shape[0] != count:
        raise ValueError(f"expected {count} field elements, got {arr.shape[0]}")
    return arr


class Context:
    """acx_ctx: one field on one GPU.  Replaces the `GaloisField k` dictionary and the
    `getRootOfUnity` argument of the reference (src/QAP.hs:513-514)."""

    def __init__(self, field: str = "bn254", device: int = 0):
        self.lib = _lib.load()
        self.field = field
        code, self.p = FIELDS[field]
        h = C.c_void_p()
        check(self.lib.acx_ctx_create(code, device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self.lib.acx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def root_of_unity(self, k: int) -> int:
        out = np.zeros(4, dtype=np.uint64)
        check(self.lib.acx_ctx_root_of_unity(self._h, k, _ptr(out)))
        return fr_to_ints(out)[0]

    def set_root(self, two_adicity: int, omega: int) -> None:
        w = ints_to_fr([omega])
        check(self.lib.acx_ctx_set_root(self._h, two_adicity, _ptr(w)))

    def sync(self) -> None:
        check(self.lib.acx_ctx_sync(self._h))

    @property
    def stream(self) -> int:
        return int(self.lib.acx_ctx_stream(self._h) or 0)

    def ntt(self, data: np.ndarray, log_n: int, inverse: bool = False, shift: Optional[int] = None) -> np.ndarray:
        n = 1 << log_n
        arr = _fr_array(data)
        if arr.shape[0] % n:
            raise ValueError("data length is not a multiple of 2^log_n")
        out = np.empty_like(arr)
        sh = ints_to_fr([shift]) if shift is not None else None
        check(self.lib.acx_ntt(self._h, log_n, arr.shape[0] // n, int(inverse), _ptr(sh), _ptr(arr), _ptr(out)))
        return out.reshape(np.asarray(data).shape)

    # device-pointer API (pointers are integers, e.g. torch.Tensor.data_ptr())
    def dev_from_canonical(self, count: int, d_in: int, d_out: int, d_err: int = 0) -> None:
        check(self.lib.acx_dev_from_canonical(self._h, count, d_in, d_out, d_err or None))

    def dev_to_canonical(self, count: int, d_in: int, d_out: int) -> None:
        check(self.lib.acx_dev_to_canonical(self._h, count, d_in, d_out))

    def ntt_dev(self, d_data: int, log_n: int, batch: int = 1, inverse: bool = False, shift: Optional[int] = None) -> None:
        sh = ints_to_fr([shift]) if shift is not None else None
        check(self.lib.acx_ntt_dev(self._h, log_n, batch, int(inverse), _ptr(sh), d_data))


    def qap_pointwise_dev(self, d_a: int, d_b: int, d_c: Optional[int], d_out: int, count: int, log_n: int, shift: int) -> None:
        """out = (a*b - c) / (shift^N - 1); d_c = None: out = a*b / (shift^N - 1)."""
        sh = ints_to_fr([shift])
        check(self.lib.acx_qap_pointwise_dev(self._h, log_n, count, _ptr(sh), d_a, d_b, d_c, d_out))

    def qap_sub_o_dev(self, d_h: int, d_o: int, count: int, log_n: int, shift: int) -> None:
        """h -= o / (shift^N - 1): O(x) enters the quotient in coefficient form (include/acx.h)."""
        sh = ints_to_fr([shift])
        check(self.lib.acx_qap_sub_o_dev(self._h, log_n, count, _ptr(sh), d_h, d_o))

    def ntt_dist_step_dev(self, d_in: int, d_out: int, log_n: int, log_r: int, world: int, rank: int, inverse: bool, step: int,
                          shift: Optional[int] = None, rows_t: bool = False, d_mul: int = 0, d_add: int = 0) -> None:
        """One local step of the distributed four-step NTT (include/acx.h: COLS / ROWS / XCHG layouts); rows_t: the input of
        an inverse step 0 is the transposed ROWS block (rows in ascending order), ACX_DIST_ROWS_T.  d_mul: the step transforms
        d_in[i] * d_mul[i]; d_add: d_out[k] = X[k] + d_add[k] (acx_ntt_dist_step_fused_dev: the h(x) pipeline's fused forms)."""
        sh = ints_to_fr([shift]) if shift is not None else None
        check(self.lib.acx_ntt_dist_step_fused_dev(self._h, log_n, log_r, world, rank, int(inverse), step, 1 if rows_t else 0, _ptr(sh),
                                                   d_in, d_mul or None, d_add or None, d_out))


class R1CS:
    """acx_r1cs: a device-resident GenQAP in row (constraint) form."""

    def __init__(self, ctx: Context, handle: C.c_void_p):
        self.ctx = ctx
        self._h = handle
        n, m, log_n = C.c_uint64(), C.c_uint64(), C.c_uint32()
        nnz = (C.c_uint64 * 3)()
        check(ctx.lib.acx_r1cs_dims(handle, C.byref(n), C.byref(m), C.byref(log_n), C.byref(nnz)))
        self.n, self.m, self.log_n, self.nnz = n.value, m.value, log_n.value, tuple(nnz)

    @classmethod
    def load(cls, ctx: Context, n: int, m: int, A, B, Cm) -> "R1CS":
        keep, structs = [], []
        for rowptr, col, val in (A, B, Cm):
            rp = np.ascontiguousarray(rowptr, dtype=np.uint32)
            cl = np.ascontiguousarray(col, dtype=np.uint32)
            vl = _fr_array(val)
            if rp.shape[0] != n + 1 or cl.shape[0] != vl.shape[0] or (n and int(rp[-1]) != cl.shape[0]):
                raise ValueError("inconsistent CSR arrays")
            keep.append((rp, cl, vl))
            structs.append(_lib.Csr(_ptr(rp), _ptr(cl), _ptr(vl)))
        h = C.c_void_p()
        check(ctx.lib.acx_r1cs_load(ctx._h, n, m, C.byref(structs[0]), C.byref(structs[1]), C.byref(structs[2]), C.byref(h)))
        return cls(ctx, h)

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx.lib.acx_r1cs_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def format(self) -> Tuple[int, bool, int]:
        """(small-coefficient mask over A, B, C; unit C; rows on the CSR path): how the residual kernel stores the system."""
        small, unit, n_long = C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(self.ctx.lib.acx_r1cs_format(self._h, C.byref(small), C.byref(unit), C.byref(n_long)))
        return small.value, bool(unit.value), n_long.value

    def export(self, matrix: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        rowptr = np.zeros(self.n + 1, dtype=np.uint32)
        col = np.zeros(self.nnz[matrix], dtype=np.uint32)
        val = np.zeros((self.nnz[matrix], 4), dtype=np.uint64)
        check(self.ctx.lib.acx_r1cs_export(self._h, matrix, _ptr(rowptr), _ptr(col), _ptr(val)))
        return rowptr, col, val

    def verify(self, witness: np.ndarray) -> Tuple[bool, int, int]:
        w = _fr_array(witness, self.m)
        ok, nbad, first = C.c_int(), C.c_uint64(), C.c_uint64()
        check(self.ctx.lib.acx_r1cs_verify(self._h, _ptr(w), C.byref(ok), C.byref(nbad), C.byref(first)))
        return bool(ok.value), nbad.value, first.value

    def verify_many(self, witnesses: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """`all (verifyAssignment qap) assignments` in one call: witnesses (count, m, 4) canonical ->
        (ok[count] bool, n_bad[count], first_bad[count])."""
        w = np.ascontiguousarray(witnesses, dtype=np.uint64)
        if w.ndim != 3 or w.shape[1:] != (self.m, 4):
            raise ValueError(f"witnesses must have shape (count, {self.m}, 4)")
        count = w.shape[0]
        ok = np.zeros(count, dtype=np.uint8)
        nbad = np.zeros(count, dtype=np.uint64)
        first = np.zeros(count, dtype=np.uint64)
        check(self.ctx.lib.acx_r1cs_verify_many(self._h, count, w.ctypes.data if count else None, ok.ctypes.data,
                                                nbad.ctypes.data, first.ctypes.data))
        return ok.astype(bool), nbad, first

    def eval_witness(self, inputs: np.ndarray, present: Optional[np.ndarray] = None,
                     download: bool = True) -> Tuple[Optional[np.ndarray], np.ndarray]:
        """generateAssignment on the GPU (level-parallel); the witness stays device resident."""
        inp = _fr_array(inputs) if len(inputs) else np.zeros((0, 4), dtype=np.uint64)
        pres = np.ascontiguousarray(present, dtype=np.uint8) if present is not None else None
        w = np.zeros((self.m, 4), dtype=np.uint64) if download else None
        assigned = np.zeros(self.m, dtype=np.uint8)
        check(self.ctx.lib.acx_r1cs_eval(self._h, _ptr(inp), _ptr(pres), inp.shape[0], _ptr(w), _ptr(assigned)))
        return w, assigned

    def verify_resident(self) -> Tuple[bool, int, int]:
        ok, nbad, first = C.c_int(), C.c_uint64(), C.c_uint64()
        check(self.ctx.lib.acx_r1cs_verify_resident(self._h, C.byref(ok), C.byref(nbad), C.byref(first)))
        return bool(ok.value), nbad.value, first.value

    def residuals(self, witness: np.ndarray) -> np.ndarray:
        w = _fr_array(witness, self.m)
        out = np.zeros((self.n, 4), dtype=np.uint64)
        check(self.ctx.lib.acx_r1cs_residuals(self._h, _ptr(w), _ptr(out)))
        return out

    def qap_h(self, witness: np.ndarray, delta: Optional[Sequence[int]] = None) -> Tuple[Optional[np.ndarray], bool]:
        w = _fr_array(witness, self.m)
        N = 1 << self.log_n
        out = np.zeros((N + 1, 4), dtype=np.uint64)
        dl = ints_to_fr(list(delta)) if delta is not None else None
        hlen, ok = C.c_uint64(), C.c_int()
        check(self.ctx.lib.acx_qap_h(self._h, _ptr(w), _ptr(dl), _ptr(out), C.byref(hlen), C.byref(ok)))
        return (out[: hlen.value] if ok.value else None), bool(ok.value)

    def qap_h_dev(self, d_witness: int, d_h: int, d_result: int, delta: Optional[Sequence[int]] = None) -> None:
        """verificationWitnessZk on device pointers (asynchronous): d_h gets N+1 dev elements."""
        dl = ints_to_fr(list(delta)) if delta is not None else None
        check(self.ctx.lib.acx_qap_h_dev(self._h, d_witness, _ptr(dl), d_h, d_result))

    def qap_columns(self, matrix: int, wire_begin: int, wire_count: int) -> Tuple[np.ndarray, np.ndarray]:
        N = 1 << self.log_n
        out = np.zeros((wire_count, N, 4), dtype=np.uint64)
        lens = np.zeros(wire_count, dtype=np.uint64)
        check(self.ctx.lib.acx_qap_columns(self._h, matrix, wire_begin, wire_count, _ptr(out), _ptr(lens)))
        return out, lens

    def qap_columns_dev(self, matrix: int, wire_begin: int, wire_count: int, d_out: int, d_len: int = 0) -> None:
        check(self.ctx.lib.acx_qap_columns_dev(self._h, matrix, wire_begin, wire_count, d_out, d_len or None))

    def dots_h_dev(self, d_witness: int, d_result: int, d_dots: int, h_log_n: int, shift: Optional[int] = None, row_offset: int = 0) -> None:
        """verify_dev storing the dot products for h(x) over 2^h_log_n points on the coset shift * <omega> (None: the field's
        generator): <A_i,w> / z, <B_i,w>, -<C_i,w> / z with z = shift^N - 1 (acx_r1cs_dots_h_dev)."""
        sh = ints_to_fr([shift]) if shift is not None else None
        check(self.ctx.lib.acx_r1cs_dots_h_dev(self._h, d_witness, row_offset, d_result, d_dots, h_log_n, _ptr(sh)))

    def verify_dev(self, d_witness: int, d_result: int, row_offset: int = 0, d_residuals: int = 0, d_dots: int = 0) -> None:
        check(self.ctx.lib.acx_r1cs_verify_dev(self._h, d_witness, row_offset, d_result, d_residuals or None, d_dots or None))


class Naive:
    """acx_naive: createPolynomials on arbitrary distinct roots (Lagrange), n <= 4096."""

    def __init__(self, r1cs: R1CS, roots: Sequence[int]):
        self.r1cs = r1cs
        self.lib = r1cs.ctx.lib
        rr = ints_to_fr(list(roots))
        h = C.c_void_p()
        check(self.lib.acx_naive_create(r1cs._h, _ptr(rr), rr.shape[0], C.byref(h)))
        self._h = h
        self.n = r1cs.n

    def close(self):
        if getattr(self, "_h", None) and getattr(self.r1cs, "_h", None):
            self.lib.acx_naive_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def target(self) -> np.ndarray:
        out = np.zeros((self.n + 1, 4), dtype=np.uint64)
        check(self.lib.acx_naive_target(self._h, _ptr(out)))
        return out

    def columns(self, matrix: int, wire_begin: int, wire_count: int) -> Tuple[np.ndarray, np.ndarray]:
        out = np.zeros((wire_count, self.n, 4), dtype=np.uint64)
        lens = np.zeros(wire_count, dtype=np.uint64)
        check(self.lib.acx_naive_columns(self._h, matrix, wire_begin, wire_count, _ptr(out), _ptr(lens)))
        return out, lens

    def h(self, witness: np.ndarray, delta: Optional[Sequence[int]] = None) -> Tuple[Optional[np.ndarray], bool]:
        w = _fr_array(witness, self.r1cs.m)
        out = np.zeros((self.n + 1, 4), dtype=np.uint64)
        dl = ints_to_fr(list(delta)) if delta is not None else None
        hlen, ok = C.c_uint64(), C.c_int()
        check(self.lib.acx_naive_h(self._h, _ptr(w), _ptr(dl), _ptr(out), C.byref(hlen), C.byref(ok)))
        return (out[: hlen.value] if ok.value else None), bool(ok.value)


class Batch:
    """acx_batch: many (R1CS, device witness) pairs verified by one launch."""

    def __init__(self, ctx: Context, systems: Sequence[R1CS], d_witnesses: Sequence[int], d_results: int,
                 per_system: bool = False):
        self.ctx = ctx
        self._systems = list(systems)          # keep alive
        n = len(self._systems)
        hs = (C.c_void_p * n)(*[s._h for s in self._systems])
        ws = (C.c_void_p * n)(*[int(p) for p in d_witnesses])
        h = C.c_void_p()
        check(ctx.lib.acx_batch_create(ctx._h, n, hs, ws, d_results, 2 if per_system else 0, C.byref(h)))
        self._h = h
        self.rows = sum(s.n for s in self._systems)

    def verify_dev(self) -> None:
        check(self.ctx.lib.acx_batch_verify_dev(self._h))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx.lib.acx_batch_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Circuit:
    """acx_circuit: a marshalled `ArithCircuit` (pure host object: needs no GPU)."""

    def __init__(self, field: str, gate_list: "_lib.GateList", keep):
        self.lib = _lib.load()
        self.field = field
        self._keep = keep
        self._gate_list = gate_list            # the marshalled form itself (its arrays are in _keep): re-creatable, e.g. to time acx_circuit_create
        h = C.c_void_p()
        check(self.lib.acx_circuit_create(FIELDS[field][0], C.byref(gate_list), C.byref(h)))
        self._h = h
        vals = [C.c_uint64() for _ in range(5)]
        check(self.lib.acx_circuit_dims(h, *[C.byref(v) for v in vals]))
        self.n_rows, self.m, self.n_inputs, self.n_intermediates, self.n_outputs = [v.value for v in vals]
        self.n_gates = gate_list.n_gates

    def close(self):
        if getattr(self, "_h", None):
            self.lib.acx_circuit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check_root_counts(self, counts: Sequence[int]) -> None:
        """One root list per gate, each of the gate's row count (src/QAP.hs:444-445,474): AcxError ROOT_COUNT otherwise."""
        arr = np.ascontiguousarray(list(counts) or [0], dtype=np.uint32)
        check(self.lib.acx_circuit_check_root_counts(self._h, _ptr(arr), len(counts)))

    def rows_per_gate(self) -> np.ndarray:
        out = np.zeros(max(self.n_gates, 1), dtype=np.uint32)
        check(self.lib.acx_circuit_rows_per_gate(self._h, _ptr(out)))
        return out[: self.n_gates]

    def valid(self) -> bool:
        v = C.c_int()
        check(self.lib.acx_circuit_valid(self._h, C.byref(v)))
        return bool(v.value)

    def eval(self, inputs: np.ndarray, present: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        inp = _fr_array(inputs) if len(inputs) else np.zeros((0, 4), dtype=np.uint64)
        pres = np.ascontiguousarray(present, dtype=np.uint8) if present is not None else None
        w = np.zeros((self.m, 4), dtype=np.uint64)
        assigned = np.zeros(self.m, dtype=np.uint8)
        check(self.lib.acx_circuit_eval(self._h, _ptr(inp), _ptr(pres), inp.shape[0], _ptr(w), _ptr(assigned)))
        return w, assigned

    def rows(self, roots: Optional[np.ndarray] = None):
        """Host CSR triple [(rowptr, col, val)] x 3 of `arithCircuitToGenQAP` (pure host)."""
        nnz = (C.c_uint64 * 3)()
        check(self.lib.acx_circuit_nnz(self._h, C.byref(nnz)))
        r = _fr_array(roots) if roots is not None else None
        out = []
        for k in range(3):
            rowptr = np.zeros(self.n_rows + 1, dtype=np.uint32)
            col = np.zeros(nnz[k], dtype=np.uint32)
            val = np.zeros((nnz[k], 4), dtype=np.uint64)
            check(self.lib.acx_circuit_rows(self._h, _ptr(r), 0 if r is None else r.shape[0], k,
                                            _ptr(rowptr), _ptr(col), _ptr(val)))
            out.append((rowptr, col, val))
        return out

    @staticmethod
    def _lists(root_lists: Sequence[Sequence[int]]):
        counts = np.ascontiguousarray([len(rs) for rs in root_lists] or [0], dtype=np.uint32)
        flat = [int(r) for rs in root_lists for r in rs]
        roots = ints_to_fr(flat) if flat else np.zeros((1, 4), dtype=np.uint64)
        return roots, counts, len(root_lists)

    def rows_lists(self, root_lists: Sequence[Sequence[int]], reference_semantics: bool = True):
        """`arithCircuitToGenQAP rootsPerGate circuit` on the host with the roots as ONE LIST PER GATE (canonical ints):
        ([(rowptr, col, val)] x 3, sorted distinct roots).  reference_semantics: duplicated roots merge rows the way
        `Map.fromList` does, surplus lists append zero rows, missing lists drop gates (src/QAP.hs:233-239,530-539,566-576)."""
        roots, counts, n_lists = self._lists(root_lists)
        flags = 1 if reference_semantics else 0
        out, sorted_roots = [], None
        for k in range(3):
            n, nnz = C.c_uint64(), C.c_uint64()
            check(self.lib.acx_circuit_rows_lists(self._h, _ptr(roots), _ptr(counts), n_lists, flags, k, C.byref(n), C.byref(nnz),
                                                  None, None, None, None))
            rowptr = np.zeros(n.value + 1, dtype=np.uint32)
            col = np.zeros(nnz.value, dtype=np.uint32)
            val = np.zeros((nnz.value, 4), dtype=np.uint64)
            sr = np.zeros((n.value, 4), dtype=np.uint64)
            check(self.lib.acx_circuit_rows_lists(self._h, _ptr(roots), _ptr(counts), n_lists, flags, k, None, None,
                                                  _ptr(rowptr), _ptr(col), _ptr(val), _ptr(sr)))
            out.append((rowptr, col, val))
            sorted_roots = fr_to_ints(sr)
        return out, sorted_roots

    def to_r1cs_lists(self, ctx: Context, root_lists: Sequence[Sequence[int]], reference_semantics: bool = True) -> R1CS:
        """acx_circuit_to_r1cs_lists: the device system of `arithCircuitToGenQAP rootsPerGate circuit`, roots as per-gate lists."""
        if ctx.field != self.field:
            raise ValueError("context and circuit are over different fields")
        roots, counts, n_lists = self._lists(root_lists)
        h = C.c_void_p()
        check(self.lib.acx_circuit_to_r1cs_lists(ctx._h, self._h, _ptr(roots), _ptr(counts), n_lists, 1 if reference_semantics else 0, C.byref(h)))
        return R1CS(ctx, h)

    def to_r1cs(self, ctx: Context, roots: Optional[np.ndarray] = None) -> R1CS:
        if ctx.field != self.field:
            raise ValueError("context and circuit are over different fields")
        h = C.c_void_p()
        if roots is None:
            check(self.lib.acx_circuit_to_r1cs(ctx._h, self._h, None, 0, C.byref(h)))
        else:
            r = _fr_array(roots)
            check(self.lib.acx_circuit_to_r1cs(ctx._h, self._h, _ptr(r), r.shape[0], C.byref(h)))
        return R1CS(ctx, h)


class MultiGpu:
    """acx_mgpu: ONE process, several GPUs behind the C ABI (include/acx.h): the library shards the constraint rows,
    replicates the witness and issues the RCCL collectives itself, so `verifyAssignment` / `verificationWitness`
    (src/QAP.hs:276-327) keep the reference's one-call shape.  `devices` may repeat an ordinal (several shards on one
    GPU, exchanged by device copies: the n_devices = 2 / 4 / 8 paths on a one-GPU machine)."""

    TRANSPORTS = {0: "rccl", 1: "peer-copy"}

    def __init__(self, field: str = "bn254", devices: Sequence[int] = (0,)):
        self.lib = _lib.load()
        self.field = field
        code, self.p = FIELDS[field]
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        check(self.lib.acx_mgpu_create(code, ids, len(devices), C.byref(h)))
        self._h = h
        self.devices = list(devices)
        n, tr, thr = C.c_uint32(), C.c_int(), C.c_uint32()
        check(self.lib.acx_mgpu_info(h, C.byref(n), C.byref(tr), C.byref(thr)))
        self.n_devices, self.transport, self.shard_threshold = n.value, self.TRANSPORTS[tr.value], thr.value

    def close(self):
        if getattr(self, "_h", None):
            self.lib.acx_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_shard_threshold(self, log_n: int) -> None:
        check(self.lib.acx_mgpu_set_shard_threshold(self._h, log_n))
        self.shard_threshold = max(10, log_n)

    def set_root(self, two_adicity: int, omega: int) -> None:
        w = ints_to_fr([omega])
        check(self.lib.acx_mgpu_set_root(self._h, two_adicity, _ptr(w)))

    def sync(self) -> None:
        check(self.lib.acx_mgpu_sync(self._h))

    def stream(self, shard: int = 0) -> int:
        c = self.lib.acx_mgpu_ctx(self._h, shard)
        return int(self.lib.acx_ctx_stream(c) or 0) if c else 0

    def ntt(self, data: np.ndarray, log_n: int, inverse: bool = False, shift: Optional[int] = None) -> np.ndarray:
        arr = _fr_array(data, 1 << log_n)
        out = np.empty_like(arr)
        sh = ints_to_fr([shift]) if shift is not None else None
        check(self.lib.acx_mgpu_ntt(self._h, log_n, int(inverse), _ptr(sh), _ptr(arr), _ptr(out)))
        return out

    def load(self, n: int, m: int, A, B, Cm, verify_only: bool = False) -> "MgR1CS":
        keep, structs = [], []
        for rowptr, col, val in (A, B, Cm):
            rp = np.ascontiguousarray(rowptr, dtype=np.uint32)
            cl = np.ascontiguousarray(col, dtype=np.uint32)
            vl = _fr_array(val)
            if rp.shape[0] != n + 1 or cl.shape[0] != vl.shape[0] or (n and int(rp[-1]) != cl.shape[0]):
                raise ValueError("inconsistent CSR arrays")
            keep.append((rp, cl, vl))
            structs.append(_lib.Csr(_ptr(rp), _ptr(cl), _ptr(vl)))
        h = C.c_void_p()
        check(self.lib.acx_mgpu_r1cs_load(self._h, n, m, C.byref(structs[0]), C.byref(structs[1]), C.byref(structs[2]),
                                          1 if verify_only else 0, C.byref(h)))
        return MgR1CS(self, h)

    def from_circuit(self, circuit: "Circuit", roots: Optional[np.ndarray] = None, verify_only: bool = False) -> "MgR1CS":
        if circuit.field != self.field:
            raise ValueError("context and circuit are over different fields")
        h = C.c_void_p()
        r = _fr_array(roots) if roots is not None else None
        check(self.lib.acx_mgpu_circuit_to_r1cs(self._h, circuit._h, _ptr(r), 0 if r is None else r.shape[0],
                                                1 if verify_only else 0, C.byref(h)))
        return MgR1CS(self, h)


class MgR1CS:
    """acx_mgpu_r1cs: a constraint system sharded over the devices of a MultiGpu (or held whole on its first device
    when it is below the shard threshold).  Same methods and results as R1CS."""

    def __init__(self, mg: MultiGpu, handle: C.c_void_p):
        self.mg = mg
        self._h = handle
        n, m, log_n, sh = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        check(mg.lib.acx_mgpu_r1cs_dims(handle, C.byref(n), C.byref(m), C.byref(log_n), C.byref(sh)))
        self.n, self.m, self.log_n, self.n_shards = n.value, m.value, log_n.value, sh.value

    def close(self):
        if getattr(self, "_h", None) and getattr(self.mg, "_h", None):
            self.mg.lib.acx_mgpu_r1cs_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def verify(self, witness: np.ndarray, want_first: bool = True) -> Tuple[bool, int, int]:
        w = _fr_array(witness, self.m)
        ok, nbad, first = C.c_int(), C.c_uint64(), C.c_uint64(2**64 - 1)
        check(self.mg.lib.acx_mgpu_r1cs_verify(self._h, _ptr(w), C.byref(ok), C.byref(nbad), C.byref(first) if want_first else None))
        return bool(ok.value), nbad.value, first.value

    def qap_columns(self, matrix: int, wire_begin: int, wire_count: int) -> Tuple[np.ndarray, np.ndarray]:
        """createPolynomialsFFT for a wire range of one matrix, the wires shared out over the devices (no exchange)."""
        N = 1 << self.log_n
        out = np.zeros((wire_count, N, 4), dtype=np.uint64)
        lens = np.zeros(wire_count, dtype=np.uint64)
        check(self.mg.lib.acx_mgpu_qap_columns(self._h, matrix, wire_begin, wire_count, _ptr(out), _ptr(lens)))
        return out, lens

    def verify_many(self, witnesses: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """`all (verifyAssignment qap) assignments` in one call: witnesses (count, m, 4) canonical -> (ok[count] bool, n_bad[count])."""
        w = np.ascontiguousarray(witnesses, dtype=np.uint64)
        if w.ndim != 3 or w.shape[1:] != (self.m, 4):
            raise ValueError(f"witnesses must have shape (count, {self.m}, 4)")
        count = w.shape[0]
        ok = np.zeros(count, dtype=np.uint8)
        nbad = np.zeros(count, dtype=np.uint64)
        check(self.mg.lib.acx_mgpu_r1cs_verify_many(self._h, count, w.ctypes.data if count else None, ok.ctypes.data, nbad.ctypes.data))
        return ok.astype(bool), nbad

    def qap_h(self, witness: np.ndarray, delta: Optional[Sequence[int]] = None) -> Tuple[Optional[np.ndarray], bool]:
        w = _fr_array(witness, self.m)
        out = np.zeros(((1 << self.log_n) + 1, 4), dtype=np.uint64)
        dl = ints_to_fr(list(delta)) if delta is not None else None
        hlen, ok = C.c_uint64(), C.c_int()
        check(self.mg.lib.acx_mgpu_qap_h(self._h, _ptr(w), _ptr(dl), _ptr(out), C.byref(hlen), C.byref(ok)))
        return (out[: hlen.value] if ok.value else None), bool(ok.value)

    # resident-witness form (what bench.py times)
    def upload_witness(self, witness: np.ndarray) -> None:
        w = _fr_array(witness, self.m)
        check(self.mg.lib.acx_mgpu_witness_upload(self._h, _ptr(w)))

    def verify_resident(self, want_first: bool = False) -> Tuple[bool, int, int]:
        ok, nbad, first = C.c_int(), C.c_uint64(), C.c_uint64(2**64 - 1)
        check(self.mg.lib.acx_mgpu_r1cs_verify_resident(self._h, C.byref(ok), C.byref(nbad), C.byref(first) if want_first else None))
        return bool(ok.value), nbad.value, first.value

    def verify_enqueue(self, slot: int) -> None:
        """asynchronous: one check of the resident witness accumulated into result slot `slot` (< 16) on every device"""
        check(self.mg.lib.acx_mgpu_r1cs_verify_enqueue(self._h, slot))

    def verdicts(self, slot0: int, count: int) -> np.ndarray:
        """ONE collective: violated-row counts of slots [slot0, slot0 + count); waits and clears them"""
        out = np.zeros(count, dtype=np.uint64)
        check(self.mg.lib.acx_mgpu_r1cs_verdicts(self._h, slot0, count, _ptr(out)))
        return out

    def qap_h_resident(self, delta: Optional[Sequence[int]] = None) -> bool:
        dl = ints_to_fr(list(delta)) if delta is not None else None
        ok = C.c_int()
        check(self.mg.lib.acx_mgpu_qap_h_resident(self._h, _ptr(dl), C.byref(ok)))
        return bool(ok.value)

    def qap_h_fetch(self) -> np.ndarray:
        out = np.zeros(((1 << self.log_n) + 1, 4), dtype=np.uint64)
        hlen = C.c_uint64()
        check(self.mg.lib.acx_mgpu_qap_h_fetch(self._h, _ptr(out), C.byref(hlen)))
        return out[: hlen.value]
