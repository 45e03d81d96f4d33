#!/usr/bin/env python3
"""Another build of libacx.so for same-box A/B measurements: python tools/build_variant.py NAME -DFLAG=1 ...
-> arithmetic-circuits_amd/variants/libacx_NAME.so (git-ignored, travels with gpurun); select it with ACX_LIB=<path>
(tools/k2_ab.py takes such paths as variants)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "arithmetic-circuits_amd", "variants")
os.makedirs(out_dir, exist_ok=True)
out = os.path.join(out_dir, f"libacx_{name}.so")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-function", "-Wno-pass-failed", "-pthread", "-ldl"] + flags + \
      [os.path.join(ROOT, "arithmetic-circuits_amd", "csrc", "engine.hip"), "-o", out]
subprocess.check_call(cmd)
print(out)
