"""Build recipe for libacx.so (hipcc, gfx950 only; cross-compiles without a GPU).  Two translation units, compiled in
parallel: engine.hip (host code, the C ABI, every kernel but the NTT pass kernels) and ntt_r4.hip (the k_ntt_r4 instances,
with the register-minimising instruction scheduler: see the header of that file)."""
from __future__ import annotations

import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libacx.so")
# unit -> LLVM options of its DEVICE code generation only (the x86 pass of a HIP compilation must not see AMDGPU scheduler
# names: `-mllvm` reaches both passes and -Xarch_device takes no options with arguments, so such a unit is compiled the way
# the driver does it internally, in three steps: device code object, offload bundle, host object with the bundle embedded)
UNITS = {"engine.hip": [], "ntt_r4.hip": []}
if os.environ.get("ACX_NTT_MISCHED"):          # development A/B: another instruction scheduler for the pass kernels
    UNITS["ntt_r4.hip"] = ["-misched=" + os.environ["ACX_NTT_MISCHED"]]
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
HEADERS = ["fr.hip.h", "mem.hip.h", "ntt_pass.hip.h", "kernels.hip.h", "ntt_r4.hip.h", "field_consts.h", "host_field.h", "circuit_host.h",
           "mgpu.inc.h", os.path.join("..", "..", "include", "acx.h")]
# --offload-compress: the code objects travel zstd-compressed inside the library (1.9 MB -> under 1 MB); the HIP runtime
# of this ROCm decompresses them at load (checked on the MI355X box: the whole -m gpu suite runs from the compressed library)
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "--offload-compress"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in list(UNITS) + HEADERS)


def build_to(out: str, extra_flags=(), verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = list(extra_flags) + os.environ.get("ACX_EXTRA_FLAGS", "").split()
    with tempfile.TemporaryDirectory() as tmp:
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)

        def compile_unit(item):
            src, dev_llvm = item
            path, obj = os.path.join(CSRC, src), os.path.join(tmp, src + ".o")
            if not dev_llvm:
                run([hipcc] + COMMON + extra + ["-c", path, "-o", obj])
                return obj
            co, fb = os.path.join(tmp, src + ".co"), os.path.join(tmp, src + ".hipfb")
            mllvm = [x for opt in dev_llvm for x in ("-mllvm", opt)]
            run([hipcc] + COMMON + extra + mllvm + ["--cuda-device-only", "--no-gpu-bundle-output", "-c", path, "-o", co])
            run([BUNDLER, "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
                 "-input=/dev/null", "-input=" + co, "-output=" + fb])
            run([hipcc] + COMMON + extra + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", path, "-o", obj])
            return obj
        with ThreadPoolExecutor(len(UNITS)) as pool:
            objs = list(pool.map(compile_unit, UNITS.items()))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-ldl", "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> arithmetic-circuits_amd/libacx.so (in-tree)."""
    if not force and not needs_build():
        return LIB
    return build_to(LIB, verbose=verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
