import importlib, os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo") else os.getcwd())
sys.path.insert(0, os.getcwd())
acx = importlib.import_module("arithmetic-circuits_amd")
synth = importlib.import_module("arithmetic-circuits_amd.synth")
ctx = acx.Context("bn254", 0)
for ln in (10, 16, 20):
    t0 = time.perf_counter(); s = synth.mulgraph(1 << ln); t1 = time.perf_counter()
    r = s.circuit.to_r1cs(ctx); ctx.sync(); t2 = time.perf_counter()
    r2 = s.circuit.to_r1cs(ctx); ctx.sync(); t3 = time.perf_counter()
    print(f"n=2^{ln}: mulgraph+marshal {1e3*(t1-t0):8.1f} ms | acx_circuit_to_r1cs first {1e3*(t2-t1):8.1f} ms, again {1e3*(t3-t2):8.1f} ms = {(1<<ln)/(t3-t2):.3e} constraints/s")
