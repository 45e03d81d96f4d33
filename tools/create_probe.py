import importlib, sys, time
import os; sys.path.insert(0, os.getcwd())
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
eng = importlib.import_module("arithmetic-circuits_amd.engine")
import numpy as np
orig = eng.Circuit.__init__
def timed(self, *a, **k):
    t0 = time.perf_counter(); orig(self, *a, **k); print(f"  acx_circuit_create: {1e3*(time.perf_counter()-t0):.1f} ms")
eng.Circuit.__init__ = timed
for ln in (16, 20):
    t0 = time.perf_counter(); s = synth.mulgraph(1 << ln); t1 = time.perf_counter()
    print(f"n=2^{ln}: mulgraph total {1e3*(t1-t0):.1f} ms")
    t0 = time.perf_counter(); w = s.witness(); print(f"  host eval (acx_circuit_eval): {1e3*(time.perf_counter()-t0):.1f} ms")
    t0 = time.perf_counter(); rows = s.rows(); print(f"  acx_circuit_rows export: {1e3*(time.perf_counter()-t0):.1f} ms")
