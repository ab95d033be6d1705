#!/usr/bin/env python3
"""tools/r5_hard.py [names...] — the LPs of check/instances that stress a first-order method, on the GPU (round 5):
default tolerance, time limit HARD_TIME_LIMIT seconds (default 240) each; prints status, iterations, objective against the
reference simplex optimum of tests/golden/reference_hard.json, wall time and microseconds per iteration (JSON lines)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from highs_amd import solver, lp as L
gold = json.load(open(os.path.join(R, "tests/golden/reference_hard.json")))
limit = float(os.environ.get("HARD_TIME_LIMIT", "240"))
solver.solveLpCupdlp(L.HighsLp.from_npz(os.path.join(R, "tests/golden/instances/afiro.npz")))
for name in sys.argv[1:] or ["perold", "gas11", "greenbea", "primal1"]:
    lp = L.HighsLp.from_npz(os.path.join(R, "tests/golden/instances/%s.npz" % name))
    t = time.time()
    o = solver.solveLpCupdlp(lp, time_limit=limit)
    dt = time.time() - t
    g = gold.get(name, {}).get("simplex", {})
    obj = lp.objective_value(o.solution.col_value)
    ref = g.get("objective_value")
    print(json.dumps({"name": name, "model_status": o.model_status, "term_code": int(o.result.term_code), "iterations": o.pdlp_iteration_count,
                      "objective": obj, "reference_simplex": ref, "reference_status": g.get("model_status"),
                      "rel_err": (abs(obj - ref) / max(1.0, abs(ref))) if ref is not None else None,
                      "wall_s": round(dt, 2), "us_per_iter": round(o.result.solve_seconds / max(o.pdlp_iteration_count, 1) * 1e6, 2)}), flush=True)
