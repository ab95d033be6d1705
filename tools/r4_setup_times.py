#!/usr/bin/env python3
"""Wall time of a whole solve (create + solve + read back + destroy) of small LPs, and the set-up share (development)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highs_amd import solver, lp as L
solver.solveLpCupdlp(L.HighsLp.from_npz("tests/golden/instances/afiro.npz"))  # context, module load
for name in sys.argv[1:] or ["afiro", "adlittle", "scagr7", "25fv47"]:
    lp = L.HighsLp.from_npz("tests/golden/instances/%s.npz" % name)
    for rep in range(2):
        t = time.time(); o = solver.solveLpCupdlp(lp); dt = time.time() - t
        r = o.result
        print(name, "iters", o.pdlp_iteration_count, "wall %.1f ms" % (dt * 1e3), "setup %.1f ms" % (r.setup_seconds * 1e3), "loop %.1f ms" % (r.solve_seconds * 1e3), flush=True)
