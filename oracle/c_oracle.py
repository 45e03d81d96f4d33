"""oracle/c_oracle.py -- ctypes wrapper over oracle/_build/libacx_oracle.so (TEST INFRASTRUCTURE).

See oracle/acx_oracle.c for what is restated and the reference file:line citations.
Elements are numpy uint64 arrays of shape (..., 4): 32-byte little-endian canonical integers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libacx_oracle.so")

_FIELDS = {
    # name: (modulus, generator, two-adicity)   -- SURVEY.md Appendix A.5
    "bn254": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 5, 28),
    "bls12_381": (52435875175126190479447740508185965837690552500527637822603658699938581184513, 7, 32),
}


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "acx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def int_to_limbs(x: int) -> np.ndarray:
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def ints_to_limbs(xs: Sequence[int]) -> np.ndarray:
    out = np.empty((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        for j in range(4):
            out[i, j] = (x >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a: np.ndarray) -> list:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


def _p(a: Optional[np.ndarray], ty):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ty))


class COracle:
    def __init__(self, field: str = "bn254"):
        self.lib = C.CDLL(build())
        self.name = field
        self.p, gen, s = _FIELDS[field]
        self.two_adicity = s
        self.generator = gen
        self.lib.orc_field_sizeof.restype = C.c_size_t
        self._F = C.create_string_buffer(self.lib.orc_field_sizeof())
        pl = int_to_limbs(self.p)
        self.lib.orc_field_init(self._F, _p(pl, C.c_uint64), C.c_uint64(gen), C.c_int(s))

    # -- field -----------------------------------------------------------
    def op(self, op: str, a: int, b: Optional[int] = None) -> int:
        code = {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op]
        al = int_to_limbs(a)
        bl = int_to_limbs(b) if b is not None else None
        out = np.zeros(4, dtype=np.uint64)
        rc = self.lib.orc_field_op(self._F, code, _p(al, C.c_uint64), _p(bl, C.c_uint64), _p(out, C.c_uint64))
        assert rc == 0
        return limbs_to_ints(out)[0]

    def root_of_unity(self, k: int) -> int:
        out = np.zeros(4, dtype=np.uint64)
        rc = self.lib.orc_root_of_unity(self._F, k, _p(out, C.c_uint64))
        if rc:
            raise ValueError("root of unity exponent out of range")
        return limbs_to_ints(out)[0]

    # -- R1CS ------------------------------------------------------------
    @staticmethod
    def _csr_args(M):
        rowptr, col, val = M
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4)
        return (rowptr, col, val), [_p(rowptr, C.c_uint32), _p(col, C.c_uint32), _p(val, C.c_uint64)]

    def r1cs_residuals(self, n: int, m: int, A, B, Cm, witness: np.ndarray, want_residuals: bool = True,
                       nthreads: int = 1, repeat: int = 1) -> Tuple[Optional[np.ndarray], int, int]:
        keep, args = [], []
        for M in (A, B, Cm):
            k, a = self._csr_args(M)
            keep.append(k)
            args += a
        w = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
        assert w.shape[0] == m
        res = np.zeros((n, 4), dtype=np.uint64) if want_residuals else None
        nbad, first = C.c_uint64(0), C.c_uint64(0)
        rc = self.lib.orc_r1cs_residuals(self._F, C.c_uint64(n), C.c_uint64(m), *args, _p(w, C.c_uint64),
                                         _p(res, C.c_uint64), C.byref(nbad), C.byref(first), C.c_int(nthreads), C.c_int(repeat))
        assert rc == 0
        return res, nbad.value, first.value

    # -- NTT -------------------------------------------------------------
    def ntt(self, data: np.ndarray, log_n: int, inverse: bool = False, shift: Optional[int] = None,
            nthreads: int = 1) -> np.ndarray:
        d = np.array(data, dtype=np.uint64).reshape(-1, 1 << log_n, 4)
        sh = int_to_limbs(shift) if shift is not None else None
        rc = self.lib.orc_ntt(self._F, C.c_int(log_n), C.c_uint64(d.shape[0]), C.c_int(int(inverse)),
                              _p(sh, C.c_uint64), _p(d, C.c_uint64), C.c_int(nthreads))
        if rc:
            raise ValueError("orc_ntt failed")
        return d.reshape(np.asarray(data).shape)

    def qap_columns(self, n: int, log_n: int, M, wire_begin: int, wire_count: int, nthreads: int = 1) -> np.ndarray:
        keep, args = self._csr_args(M)
        out = np.zeros((wire_count, 1 << log_n, 4), dtype=np.uint64)
        rc = self.lib.orc_qap_columns(self._F, C.c_uint64(n), C.c_int(log_n), *args, C.c_uint64(wire_begin),
                                      C.c_uint64(wire_count), _p(out, C.c_uint64), C.c_int(nthreads))
        if rc:
            raise ValueError("orc_qap_columns failed")
        return out

    def ref_verify(self, m: int, log_n: int, cols: np.ndarray, witness: np.ndarray) -> Tuple[np.ndarray, bool]:
        """The reference's own polynomial-domain verificationWitness (dense per-wire polynomials, dense product, long
        division; orc_ref_verify): cols = (3, m, N, 4) canonical coefficients -> (quotient N coefficients, remainder == 0)."""
        N = 1 << log_n
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        assert cols.shape == (3, m, N, 4)
        w = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((N, 4), dtype=np.uint64)
        ok = C.c_int(0)
        rc = self.lib.orc_ref_verify(self._F, C.c_uint64(m), C.c_int(log_n), _p(cols, C.c_uint64), _p(w, C.c_uint64),
                                     _p(out, C.c_uint64), C.byref(ok))
        if rc:
            raise ValueError("orc_ref_verify failed")
        return out, bool(ok.value)

    def qap_h(self, n: int, m: int, log_n: int, A, B, Cm, witness: np.ndarray, delta: Optional[Sequence[int]] = None,
              nthreads: int = 1) -> Tuple[np.ndarray, bool]:
        keep, args = [], []
        for M in (A, B, Cm):
            k, a = self._csr_args(M)
            keep.append(k)
            args += a
        w = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
        dl = ints_to_limbs(list(delta)) if delta is not None else None
        out = np.zeros(((1 << log_n) + 1, 4), dtype=np.uint64)
        ok = C.c_int(0)
        rc = self.lib.orc_qap_h(self._F, C.c_uint64(n), C.c_uint64(m), C.c_int(log_n), *args, _p(w, C.c_uint64),
                                _p(dl, C.c_uint64), _p(out, C.c_uint64), C.byref(ok), C.c_int(nthreads))
        if rc:
            raise ValueError("orc_qap_h failed")
        return out, bool(ok.value)
