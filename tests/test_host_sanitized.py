"""the host marshalling code under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only.

`csrc/host_only.cpp` is the pure-host part of the C ABI (acx_circuit_*: the same source text libacx.so compiles, no HIP) built
by g++ with -fsanitize=address,undefined into tests/_build/libacx_host_asan.so.  Two runs in child processes with the
sanitizer runtime preloaded (every heap buffer, numpy's included, then has red zones):
  * tests/test_host_logic.py's host-only tests against that library (ACX_LIB + ACX_LIB_HOST_ONLY=1);
  * tests/host_fuzz_worker.py: 10 000 byte-level mutations of marshalled gate lists (truncated offsets, ADD towers, tok_arg
    beyond the tables, wild wire kinds / indices, NULL arrays with counts).
A sanitizer report aborts the child: the test fails with its stderr."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
LIB = os.path.join(BUILD, "libacx_host_asan.so")
SRC = os.path.join(ROOT, "arithmetic-circuits_amd", "csrc")


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.fixture(scope="module")
def sanitized_env():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan next to this gcc")
    os.makedirs(BUILD, exist_ok=True)
    deps = [os.path.join(SRC, f) for f in ("host_only.cpp", "abi_common.inc.h", "circuit_abi.inc.h", "circuit_host.h", "host_field.h", "field_consts.h")]
    deps.append(os.path.join(ROOT, "include", "acx.h"))
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-pthread", deps[0], "-o", LIB])
    # leak checking stays off: the interpreter itself never frees everything; allocator_may_return_null turns an absurd
    # allocation (a gate list naming wire 2^31) into std::bad_alloc -> ACX_ERR_OOM instead of an abort of the allocator
    return dict(os.environ, LD_PRELOAD=asan, ACX_LIB=LIB, ACX_LIB_HOST_ONLY="1",
                ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_host_logic_suite_under_sanitizers(sanitized_env):
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_logic.py"), "-x", "-q", "-p", "no:cacheprovider",
           "-k", "not abi_exports and not no_gpu and not c_host and not wire_ranges"]
    out = subprocess.run(cmd, cwd=ROOT, env=sanitized_env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_gate_list_fuzz_under_sanitizers(sanitized_env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_fuzz_worker.py"), "10000", "20260929"], cwd=ROOT,
                         env=sanitized_env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-4000:])
    tag, word, accepted, rejected = out.stdout.split()[-4:]
    assert (tag, word) == ("fuzz", "ok") and int(accepted) + int(rejected) == 10000
    assert int(accepted) > 500 and int(rejected) > 2000       # both sides of the validator are really exercised


def test_the_sanitizers_are_live(sanitized_env, tmp_path):
    """A deliberately broken caller (token array one element shorter than its offsets claim) must be REPORTED: proof that
    the library is instrumented and that numpy's buffers carry red zones in these child processes."""
    prog = (
        "import ctypes as C, importlib, sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "L = importlib.import_module('arithmetic-circuits_amd._lib'); lib = L.load()\n"
        "kind = np.zeros(1, dtype=np.uint8); tok_ofs = np.array([0, 1, 2], dtype=np.uint64)\n"
        "tok_op = np.full(1, 2, dtype=np.uint8)      # two tokens claimed, one allocated\n"
        "tok_arg = np.zeros(2, dtype=np.uint32); sc = np.zeros((1, 4), dtype=np.uint64); aff = np.zeros((1, 2), dtype=np.uint32)\n"
        "wofs = np.array([0, 1], dtype=np.uint64); wires = np.array([[1, 0]], dtype=np.uint32)\n"
        "gl = L.GateList(1, kind.ctypes.data, tok_ofs.ctypes.data, tok_op.ctypes.data, tok_arg.ctypes.data, sc.ctypes.data, 1,\n"
        "                aff.ctypes.data, 1, wofs.ctypes.data, wires.ctypes.data)\n"
        "h = C.c_void_p(); lib.acx_circuit_create(0, C.byref(gl), C.byref(h)); print('not reported')\n")
    out = subprocess.run([sys.executable, "-c", prog], cwd=ROOT, env=sanitized_env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "AddressSanitizer" in out.stderr and "heap-buffer-overflow" in out.stderr, (out.stdout, out.stderr[-2000:])
