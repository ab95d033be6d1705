import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from highs_amd import solver, lp as L
for name in ["25fv47", "80bau3b"]:
    lp = L.HighsLp.from_npz("tests/golden/instances/%s.npz" % name)
    solver.solveLpCupdlp(L.HighsLp.from_npz("tests/golden/instances/afiro.npz"))
    t = time.time(); o = solver.solveLpCupdlp(lp); dt = time.time() - t
    print(name, "pdlp", o.pdlp_iteration_count, "iters", round(dt, 2), "s wall,", round(o.result.solve_seconds, 2), "s loop ->", round(o.result.solve_seconds / max(o.pdlp_iteration_count, 1) * 1e6, 1), "us/iter")
