#!/usr/bin/env python3
"""Another build of libacx.so for same-box A/B measurements: python tools/build_variant.py NAME -DFLAG=1 ...
-> arithmetic-circuits_amd/variants/libacx_NAME.so (git-ignored, travels with gpurun); select it with ACX_LIB=<path>
(tools/k2_ab.py takes such paths as variants)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
builder = importlib.import_module("arithmetic-circuits_amd.build")
name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "arithmetic-circuits_amd", "variants")
os.makedirs(out_dir, exist_ok=True)
print(builder.build_to(os.path.join(out_dir, f"libacx_{name}.so"), flags))
