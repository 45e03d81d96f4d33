"""Build recipe for libacx.so (hipcc, gfx950 only; cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libacx.so")
SOURCES = ["engine.hip"]
HEADERS = ["fr.hip.h", "kernels.hip.h", "ntt_r4.hip.h", "field_consts.h", "host_field.h", "circuit_host.h", "mgpu.inc.h",
           os.path.join("..", "..", "include", "acx.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> arithmetic-circuits_amd/libacx.so (in-tree)."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-function", "-pthread", "-ldl"] + os.environ.get("ACX_EXTRA_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
