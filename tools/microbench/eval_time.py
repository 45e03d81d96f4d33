import importlib,sys,time,numpy as np
sys.path.insert(0,".")
acx=importlib.import_module("arithmetic-circuits_amd"); synth=importlib.import_module("arithmetic-circuits_amd.synth")
ctx=acx.Context()
for ln in (16,20):
    s=synth.mulgraph(1<<ln); r=s.circuit.to_r1cs(ctx)
    t=time.time(); wh=s.witness(); th=time.time()-t
    for _ in range(3): r.eval_witness(s.inputs, download=False)
    ts=[]
    for _ in range(5):
        t=time.time(); r.eval_witness(s.inputs, download=False); ts.append(time.time()-t)
    w,_=r.eval_witness(s.inputs)
    print("2^%d gates: host %.3f s, gpu %.4f s, equal=%s, verify=%s"%(ln,th,min(ts),np.array_equal(w,wh), r.verify_resident()[0]))
