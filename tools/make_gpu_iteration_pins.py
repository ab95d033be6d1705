#!/usr/bin/env python3
"""Iteration counts of the GPU path on the golden instances -> tests/golden/gpu_iteration_counts.json.

Every sum on the device is taken in a fixed order, so a solve is reproducible bit for bit and its iteration count
is a regression pin: a change means a summation order (or a decision) changed.  Run on the GPU box; the tests
compare exactly (tests/test_gpu_parity.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from highs_amd import solver, lp as L
GOLD = os.path.join(ROOT, "tests", "golden")
names = sorted(set(json.load(open(os.path.join(GOLD, "reference_pdlp.json")))) | set(json.load(open(os.path.join(GOLD, "reference_pdlp_more.json")))))
out = {}
for name in names:
    o = solver.solveLpCupdlp(L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz")))
    out[name] = {"pdlp_iteration_count": int(o.pdlp_iteration_count), "term_code": int(o.result.term_code)}
    print(name, out[name], flush=True)
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(GOLD, "gpu_iteration_counts.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
