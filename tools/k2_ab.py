#!/usr/bin/env python3
"""A/B of K2 variants on the bench workload in ONE process run each, alternating: python tools/k2_ab.py 0 1 2  (values of
ACX_K2_MULTI, or paths of other builds of libacx.so -> ACX_LIB).  Prints kernel us per launch of bench.py --no-cpu --no-ntt for every variant, three rounds."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1:] or ["0"]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        env = dict(os.environ)
        if v.endswith(".so"):
            env["ACX_LIB"] = os.path.abspath(v)         # another build of the library (A/B of two source states on one box)
        else:
            env["ACX_K2_MULTI"] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-ntt", "--sustain", "0.3"], env=env, capture_output=True, text=True)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        res[v].append((d["roofline"]["kernel_us"], d["sustained"]["us_per_step_median"], d["config"]["single_system_launch_us"]))
for v in variants:
    print(v, "  ".join("%.2f/%.2f/%.2f" % x for x in res[v]))
