import importlib, sys, time, torch, numpy as np
sys.path.insert(0, ".")
acx = importlib.import_module("arithmetic-circuits_amd"); synth = importlib.import_module("arithmetic-circuits_amd.synth")
ctx = acx.Context("bn254", 0)
stream = torch.cuda.ExternalStream(ctx.stream)
systems, wit = [], []
for c in range(32):
    s = synth.mulgraph(1 << 16, seed=0xAC355 + c)
    systems.append(s.circuit.to_r1cs(ctx))
    t = torch.from_numpy(s.witness().view(np.int64).copy()).cuda(); torch.cuda.synchronize()
    ctx.dev_from_canonical(t.shape[0], t.data_ptr(), t.data_ptr()); wit.append(t)
res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
b = acx.Batch(ctx, systems, [w.data_ptr() for w in wit], res.data_ptr())
ctx.sync(); torch.cuda.synchronize()
time.sleep(2.0)
with torch.cuda.stream(stream):
    t0 = time.perf_counter()
    for chunk in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(50): b.verify_dev()
        e1.record(stream); e1.synchronize()
        print(f"chunk {chunk:2d} t={time.perf_counter()-t0:6.3f}s  {e0.elapsed_time(e1)*1e3/50:7.2f} us/step")
