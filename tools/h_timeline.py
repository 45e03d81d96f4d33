#!/usr/bin/env python3
"""Kernel timeline of ONE device-resident h(x) computation (acx_qap_h_dev at 2^20 constraints): runs the pipeline under
rocprofv3 --kernel-trace (csv) and prints, for the last repetition, every kernel with its start offset, duration and the
idle gap before it.   python tools/h_timeline.py [--logn 20] [--coeff random]   (on an MI355X)"""
import argparse, csv, glob, importlib, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    acx = importlib.import_module("arithmetic-circuits_amd")
    synth = importlib.import_module("arithmetic-circuits_amd.synth")
    ctx = acx.Context("bn254", 0)
    n = 1 << a.logn
    s = synth.mulgraph(n, seed=0xAC3, coeff=a.coeff)
    r = s.circuit.to_r1cs(ctx)
    w = s.witness()
    dw = torch.from_numpy(w.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.dev_from_canonical(w.shape[0], dw.data_ptr(), dw.data_ptr())
    dh = torch.zeros((n + 1, 4), dtype=torch.int64, device="cuda")
    res = torch.tensor([0, -1], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(a.reps):
        r.qap_h_dev(dw.data_ptr(), dh.data_ptr(), res.data_ptr())
    ctx.sync()
    assert int(res[0]) == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--coeff", default="random")
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--out", default="gpurun_out/h_trace")
    a = ap.parse_args()
    if a.child:
        return child(a)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", a.out, "-o", "h", "--",
                           sys.executable, os.path.abspath(__file__), "--child", "--logn", str(a.logn), "--coeff", a.coeff,
                           "--reps", str(a.reps)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for f in glob.glob(os.path.join(a.out, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    rows.sort()
    # one repetition = from one k_r1cs_sell to the next
    starts = [i for i, r in enumerate(rows) if "k_r1cs_sell" in r[2]]
    lo, hi = starts[-2], starts[-1]
    t0, prev_end = rows[lo][0], rows[lo][0]
    busy = 0
    print(f"h(x) at 2^{a.logn}, one repetition ({hi - lo} kernels), period {(rows[hi][0] - t0) / 1e3:.1f} us:")
    for s, e, name in rows[lo:hi]:
        short = name.split("(")[0].replace("void acx::", "").replace("acx::", "")[:60]
        print(f"  +{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  run {(e - s) / 1e3:7.1f}  {short}")
        busy += e - s
        prev_end = e
    print(f"  busy {busy / 1e3:.1f} us, idle {(rows[hi][0] - t0 - busy) / 1e3:.1f} us")
    for f in glob.glob(os.path.join(a.out, "**", "*.csv"), recursive=True):
        os.remove(f)


if __name__ == "__main__":
    main()
