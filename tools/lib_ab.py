#!/usr/bin/env python3
"""Same-box A/B of builds of libacx.so (tools/build_variant.py) on bench.py's secondary objects: alternating processes,
`rounds` rounds, both fields on request; prints the 2^20 NTT, the batch-of-64 NTT per transform, h(x) at 2^20 and the
headline launch, in us.   python tools/lib_ab.py [--rounds 2] [--fields bn254,bls12_381] NAME=path/to/libacx_NAME.so ..."""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--fields", default="bn254", help="comma separated: bn254,bls12_381")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
rows = {}
for rnd in range(a.rounds):
    for field in a.fields.split(","):
        for v in a.variants:
            name, _, path = v.partition("=")
            env = dict(os.environ, ACX_LIB=os.path.abspath(path))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only", "ntt,qap_h", "--steps", "20", "--sustain", "0", "--field", field], env=env, capture_output=True, text=True)
            try:
                line = json.loads(out.stdout.strip().splitlines()[-1])
                rows.setdefault((field, name), []).append((line["ntt"]["us"], line["ntt"]["batch"]["us_per_transform"], line["qap_h"]["us"],
                                                           line["roofline"]["kernel_us"], line["ntt"]["parity_vs_oracle"] and line["qap_h"]["parity_vs_oracle"]))
            except Exception as e:
                rows.setdefault((field, name), []).append((None, None, None, None, f"failed: {e}: {out.stderr[-300:]}"))
print(f"{'field':10s} {'build':10s}  ntt 2^20 | batch/64 | h(x) 2^20 | K2 launch   (us, one column per round)   parity")
for (field, name), r in rows.items():
    cols = [" / ".join("%7.1f" % x[i] if x[i] is not None else "   fail" for x in r) for i in range(4)]
    print(f"{field:10s} {name:10s}  {cols[0]} | {cols[1]} | {cols[2]} | {cols[3]}   {all(x[4] is True for x in r)}")
