"""ctypes binding of libacx.so (include/acx.h).  There is no fallback: if the shared library
is missing the import of any compute entry point raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACX_LIB", os.path.join(HERE, "libacx.so"))   # ACX_LIB: development override

ACX_OK = 0
STATUS = {
    "INVALID_ARG": -1, "NONCANONICAL": -2, "NO_DEVICE": -3, "HIP": -4, "ROOT_COUNT": -5,
    "UNDEFINED_WIRE": -6, "DUPLICATE_ROOT": -7, "TOO_LARGE": -8, "OOM": -9, "BAD_CIRCUIT": -10,
    "UNSUPPORTED": -11,
}
FIELD_BN254_FR = 0
FIELD_BLS12_381_FR = 1


class AcxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"acx error {status}: {message}")
        self.status = status


class Wire(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32)]


class GateList(C.Structure):
    _fields_ = [
        ("n_gates", C.c_uint64), ("kind", C.c_void_p), ("tok_ofs", C.c_void_p), ("tok_op", C.c_void_p),
        ("tok_arg", C.c_void_p), ("scalars", C.c_void_p), ("n_scalars", C.c_uint64),
        ("aff_wires", C.c_void_p), ("n_aff_wires", C.c_uint64), ("wire_ofs", C.c_void_p), ("wires", C.c_void_p),
    ]


class Csr(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p)]


# every symbol include/acx.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_U64 = C.c_uint64
_U32 = C.c_uint32
_I = C.c_int
SYMBOLS = {
    "acx_strerror": (C.c_char_p, [_I]),
    "acx_last_error": (C.c_char_p, []),
    "acx_version": (_U32, []),
    "acx_ctx_create": (_I, [_I, _I, C.POINTER(_P)]),
    "acx_ctx_destroy": (None, [_P]),
    "acx_ctx_set_root": (_I, [_P, _U32, _P]),
    "acx_ctx_root_of_unity": (_I, [_P, _U32, _P]),
    "acx_ctx_sync": (_I, [_P]),
    "acx_ctx_stream": (_P, [_P]),
    "acx_circuit_create": (_I, [_I, C.POINTER(GateList), C.POINTER(_P)]),
    "acx_circuit_destroy": (None, [_P]),
    "acx_circuit_dims": (_I, [_P] + [C.POINTER(_U64)] * 5),
    "acx_circuit_rows_per_gate": (_I, [_P, _P]),
    "acx_circuit_valid": (_I, [_P, C.POINTER(_I)]),
    "acx_circuit_eval": (_I, [_P, _P, _P, _U64, _P, _P]),
    "acx_circuit_to_r1cs": (_I, [_P, _P, _P, _U64, C.POINTER(_P)]),
    "acx_circuit_check_root_counts": (_I, [_P, _P, _U64]),
    "acx_circuit_to_r1cs_lists": (_I, [_P, _P, _P, _P, _U64, _U32, C.POINTER(_P)]),
    "acx_circuit_rows_lists": (_I, [_P, _P, _P, _U64, _U32, _I, C.POINTER(_U64), C.POINTER(_U64), _P, _P, _P, _P]),
    "acx_circuit_nnz": (_I, [_P, C.POINTER(_U64 * 3)]),
    "acx_circuit_rows": (_I, [_P, _P, _U64, _I, _P, _P, _P]),
    "acx_r1cs_load": (_I, [_P, _U64, _U64, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr), C.POINTER(_P)]),
    "acx_r1cs_destroy": (None, [_P]),
    "acx_r1cs_dims": (_I, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U32), C.POINTER(_U64 * 3)]),
    "acx_r1cs_format": (_I, [_P, C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U64)]),
    "acx_r1cs_verify_many": (_I, [_P, _U64, _P, _P, _P, _P]),
    "acx_r1cs_export": (_I, [_P, _I, _P, _P, _P]),
    "acx_r1cs_verify": (_I, [_P, _P, C.POINTER(_I), C.POINTER(_U64), C.POINTER(_U64)]),
    "acx_r1cs_eval": (_I, [_P, _P, _P, _U64, _P, _P]),
    "acx_r1cs_verify_resident": (_I, [_P, C.POINTER(_I), C.POINTER(_U64), C.POINTER(_U64)]),
    "acx_r1cs_residuals": (_I, [_P, _P, _P]),
    "acx_qap_h": (_I, [_P, _P, _P, _P, C.POINTER(_U64), C.POINTER(_I)]),
    "acx_qap_columns": (_I, [_P, _I, _U64, _U64, _P, _P]),
    "acx_naive_create": (_I, [_P, _P, _U64, C.POINTER(_P)]),
    "acx_naive_destroy": (None, [_P]),
    "acx_naive_target": (_I, [_P, _P]),
    "acx_naive_columns": (_I, [_P, _I, _U64, _U64, _P, _P]),
    "acx_naive_h": (_I, [_P, _P, _P, _P, C.POINTER(_U64), C.POINTER(_I)]),
    "acx_ntt": (_I, [_P, _U32, _U64, _I, _P, _P, _P]),
    "acx_dev_from_canonical": (_I, [_P, _U64, _P, _P, _P]),
    "acx_dev_to_canonical": (_I, [_P, _U64, _P, _P]),
    "acx_r1cs_verify_dev": (_I, [_P, _P, _U64, _P, _P, _P]),
    "acx_ntt_dev": (_I, [_P, _U32, _U64, _I, _P, _P]),
    "acx_qap_h_dev": (_I, [_P, _P, _P, _P, _P]),
    "acx_qap_columns_dev": (_I, [_P, _I, _U64, _U64, _P, _P]),
    "acx_qap_pointwise_dev": (_I, [_P, _U32, _U64, _P, _P, _P, _P, _P]),
    "acx_qap_sub_o_dev": (_I, [_P, _U32, _U64, _P, _P, _P]),
    "acx_ntt_dist_step_dev": (_I, [_P, _U32, _U32, _U32, _U32, _I, _I, _P, _P, _P]),
    "acx_ntt_dist_step_ex_dev": (_I, [_P, _U32, _U32, _U32, _U32, _I, _I, _U32, _P, _P, _P]),
    "acx_ntt_dist_step_fused_dev": (_I, [_P, _U32, _U32, _U32, _U32, _I, _I, _U32, _P, _P, _P, _P, _P]),
    "acx_r1cs_dots_h_dev": (_I, [_P, _P, _U64, _P, _P, _U32, _P]),
    "acx_batch_create": (_I, [_P, _U64, _P, _P, _P, _U64, C.POINTER(_P)]),
    "acx_batch_verify_dev": (_I, [_P]),
    "acx_batch_destroy": (None, [_P]),
    "acx_mgpu_create": (_I, [_I, C.POINTER(_I), _U32, C.POINTER(_P)]),
    "acx_mgpu_destroy": (None, [_P]),
    "acx_mgpu_info": (_I, [_P, C.POINTER(_U32), C.POINTER(_I), C.POINTER(_U32)]),
    "acx_mgpu_ctx": (_P, [_P, _U32]),
    "acx_mgpu_set_shard_threshold": (_I, [_P, _U32]),
    "acx_mgpu_set_root": (_I, [_P, _U32, _P]),
    "acx_mgpu_sync": (_I, [_P]),
    "acx_mgpu_r1cs_load": (_I, [_P, _U64, _U64, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr), _U32, C.POINTER(_P)]),
    "acx_mgpu_circuit_to_r1cs": (_I, [_P, _P, _P, _U64, _U32, C.POINTER(_P)]),
    "acx_mgpu_r1cs_verify_enqueue": (_I, [_P, _U32]),
    "acx_mgpu_r1cs_verdicts": (_I, [_P, _U32, _U32, _P]),
    "acx_mgpu_r1cs_destroy": (None, [_P]),
    "acx_mgpu_r1cs_dims": (_I, [_P, C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U32), C.POINTER(_U32)]),
    "acx_mgpu_r1cs_verify": (_I, [_P, _P, C.POINTER(_I), C.POINTER(_U64), C.POINTER(_U64)]),
    "acx_mgpu_r1cs_verify_many": (_I, [_P, _U64, _P, _P, _P]),
    "acx_mgpu_qap_h": (_I, [_P, _P, _P, _P, C.POINTER(_U64), C.POINTER(_I)]),
    "acx_mgpu_ntt": (_I, [_P, _U32, _I, _P, _P, _P]),
    "acx_mgpu_qap_columns": (_I, [_P, _I, _U64, _U64, _P, _P]),
    "acx_mgpu_witness_upload": (_I, [_P, _P]),
    "acx_mgpu_r1cs_verify_resident": (_I, [_P, C.POINTER(_I), C.POINTER(_U64), C.POINTER(_U64)]),
    "acx_mgpu_qap_h_resident": (_I, [_P, _P, C.POINTER(_I)]),
    "acx_mgpu_qap_h_fetch": (_I, [_P, _P, C.POINTER(_U64)]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libacx.so and declare prototypes.  Loading needs no GPU; creating a context does.
    torch is imported first so that libacx binds to the HIP runtime torch already mapped: two HIP
    runtimes in one process cannot both own the device ("No HIP GPUs are available")."""
    global _lib
    if _lib is None:
        try:
            import torch  # noqa: F401
        except ImportError:   # a host without torch uses the system ROCm runtime
            pass
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        # ACX_LIB_HOST_ONLY=1 (with ACX_LIB): the library is the sanitizer build of the pure-host entry points
        # (csrc/host_only.cpp, tests/test_host_sanitized.py) -- only the symbols it has are bound; everything else of the ABI
        # is absent there and using it fails loudly (AttributeError), as it must
        host_only = os.environ.get("ACX_LIB_HOST_ONLY") == "1" and "ACX_LIB" in os.environ
        for name, (res, args) in SYMBOLS.items():
            if host_only and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int) -> None:
    if status != ACX_OK:
        lib = load()
        detail = lib.acx_last_error().decode() or lib.acx_strerror(status).decode()
        raise AcxError(status, detail)
