#!/usr/bin/env python3
"""acx_r1cs_eval on a circuit in the reference's gate mix: launches per evaluation and their duration (run under tools/prof.py)."""
import importlib, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acx = importlib.import_module("arithmetic-circuits_amd")
from tests import helpers as H
size = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
ctx = acx.Context("bn254", 0)
rnd = random.Random(5)
gates = H.arb_arith_circuit(rnd, ctx.p, 6, size, dist=(50, 10, 1), split_bits=256)
circ = H.to_acx_circuit(acx, gates).marshal("bn254")
r = circ.to_r1cs(ctx)
inp = acx.ints_to_fr([rnd.randrange(ctx.p) for _ in range(6)])
for _ in range(20):
    r.eval_witness(inp, download=False)
ctx.sync()
