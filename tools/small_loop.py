#!/usr/bin/env python3
"""Small-LP loop time with and without the check iterations (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highs_amd import solver, lp as L
for name in sys.argv[1:] or ["25fv47", "80bau3b"]:
    lp = L.HighsLp.from_npz("tests/golden/instances/%s.npz" % name)
    for ci in (0, 100000):
        S = solver.DeviceSolver(lp=lp, **({"check_interval": ci} if ci else {}))
        S.iterate(400)
        st = S.iterate(8000)
        print(name, "check_interval", ci or 40, "->", round(1e3 * st.gpu_ms / st.iters, 2), "us/iter", "trials/iter", round(st.trials / st.iters, 3), "checks", st.checks, flush=True)
        S.close()
