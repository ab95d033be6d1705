"""One rank of a multi-process sharded solve (launched by test_gpu_mesh.py, one process per rank; on the
1-GPU test box every rank uses the same device — the xGMI mesh exchange only needs HIP IPC).

usage: mesh_worker.py RANK WORLD IDHEX CASE OUTFILE
  CASE = solve:<instance>[:features_off[:iteration_limit]]  -> full solve through create_sharded / run
         iterate:<instance|synth>:<k>     -> k fixed iterations, dumps x and the step sizes
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from highs_amd import solver  # noqa: E402
from highs_amd import lp as L  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    rank, world, idhex, case, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    uid = (C.c_ubyte * 128).from_buffer_copy(bytes.fromhex(idhex))
    kind, name, *rest = case.split(":")
    sp_ = None
    if name in ("synth", "synthbig"):
        dims = (20000, 20000, 160000, 3) if name == "synth" else (1000000, 1000000, 8000000, 1)
        sp_ = solver.SyntheticProblem(*dims)
        kw = dict(problem_struct=sp_.struct)
        nc, nr = dims[1], dims[0]
    else:
        lp = L.special_lps()[name] if name in L.special_lps() else L.HighsLp.from_npz(os.path.join(GOLD, "instances", name + ".npz"))
        kw = dict(lp=lp)
        nc, nr = lp.num_col, lp.num_row
    if kind == "die":
        # rank 1 disappears right after the collective creation; rank 0 must get an error, not a hung GPU
        S = solver.DeviceSolver(rank=rank, world=world, unique_id=uid, **kw)
        if rank == 1:
            os._exit(0)
        import time
        t0 = time.time()
        try:
            S.iterate(200)
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        np.savez(out, msg=msg, seconds=time.time() - t0)
        S.close()
        return
    if kind == "hsolve":  # the HiPDLP path, sharded
        S = solver.DeviceSolver(rank=rank, world=world, unique_id=uid, solver="hipdlp", **kw)
        ex = S.stage("exchange")[0]
        R = S.run(nc, nr)
        np.savez(out, exchange=ex, col_value=R.col_value, col_dual=R.col_dual, row_value=R.row_value,
                 row_dual=R.row_dual, num_iter=R.num_iter, num_restarts=R.num_restarts, term=R.term_code,
                 primal_obj=R.primal_obj, dual_obj=R.dual_obj, primal_feas=R.primal_feas, dual_feas=R.dual_feas,
                 rel_gap=R.rel_gap, norm_rhs=R.norm_rhs, norm_cost=R.norm_cost)
        S.close()
        return
    if kind == "solve":
        foff = int(rest[0]) if rest else 0
        if len(rest) > 1:  # a fixed amount of work on an LP that takes millions of iterations: tolerance out of reach
            kw.update(pdlp_iteration_limit=int(rest[1]), kkt_tolerance=1e-12)
        S = solver.DeviceSolver(rank=rank, world=world, unique_id=uid, time_limit=1000.0, pdlp_features_off=foff, **kw)
        ex = S.stage("exchange")[0]
        R = S.run(nc, nr)
        np.savez(out, exchange=ex, col_value=R.col_value, col_dual=R.col_dual, row_value=R.row_value,
                 row_dual=R.row_dual, num_iter=R.num_iter, num_trials=R.num_trials, term=R.term_code,
                 primal_obj=R.primal_obj, dual_obj=R.dual_obj, primal_feas=R.primal_feas, dual_feas=R.dual_feas,
                 rel_gap=R.rel_gap, norm_rhs=R.norm_rhs)
    else:
        k = int(rest[0])
        S = solver.DeviceSolver(rank=rank, world=world, unique_id=uid, **kw)
        ex = S.stage("exchange")[0]
        st = S.iterate(k)
        np.savez(out, exchange=ex, x=S.get("x", S.n), steps=S.get("steps", 8), iters=st.iters, trials=st.trials,
                 restarts=st.restarts)
    S.close()


if __name__ == "__main__":
    main()
