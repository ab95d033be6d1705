#!/usr/bin/env python3
"""debug: sharded (2 ranks on one device) vs single GPU after k iterations, both mesh layouts"""
import os, sys, pathlib, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_mesh as T
from highs_amd import solver
for k in (1, 2, 3, 5, 9, 10, 11, 40, 120):
    sp_ = solver.SyntheticProblem(20000, 20000, 160000, 3)
    S = solver.DeviceSolver(problem_struct=sp_.struct)
    S.iterate(k); x1 = S.get("x", S.n); s1 = S.get("steps", 8); S.close()
    for lay in ("colblock", "partial"):
        with tempfile.TemporaryDirectory() as d:
            res = T._run_ranks(2, "iterate:synth:%d" % k, pathlib.Path(d), extra_env={"PDLP_MI355X_MESH_LAYOUT": lay})
        err = np.linalg.norm(res[0]["x"] - x1) / (1e-300 + np.linalg.norm(x1))
        print(k, lay, "err", err, "trials", int(res[0]["trials"]), "restarts", int(res[0]["restarts"]), "steps", res[0]["steps"][:3], s1[:3], flush=True)
