#!/usr/bin/env python3
"""Converged-solution goldens for BASELINE.json configs 2 and 4 (run in the build container only).

Runs the REAL cuPDLP-C core compiled from the reference sources (oracle/_ref/libpdlp_ref.so,
`make -C oracle ref`) on the seed-1 synthetic LPs of SURVEY §8d — 100k x 100k / 1M nnz and
1M x 1M / 8M nnz — at the reference's DEFAULT tolerance (kkt 1e-7), single thread, and stores what the
north_star parity criterion compares: objective, cuPDLP primal / dual objective, residual norms,
HiGHS-style KKT measures and the iteration count.

    python tests/golden/make_golden_synth.py [a] [b] [c_small] [c] [d_small] [d] [e_small] [e] [f_small] [f]   (default: a b; "b" takes ~1-2 h of one core)

Output: tests/golden/reference_synth.json (one record per config; existing records of configs that are
not re-run are kept).  The LP itself is regenerated on the GPU box by the library's seeded generator
(pdlp_mi355x_gen_synthetic), so only these scalars are committed.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oraclelib as O  # noqa: E402
from highs_amd import lp as L  # noqa: E402
from highs_amd import solver  # noqa: E402

CONFIGS = {"a": (100_000, 100_000, 1_000_000), "b": (1_000_000, 1_000_000, 8_000_000),
           # BASELINE config 3 stand-in (pds-100 is not in the reference tree): tests/lpgen.py::structured_lp,
           # block-angular network LP with dense linking rows; "c" = the bench size (5.3M nnz), "c_small" = 1/16
           "c_small": dict(commodities=16, nodes=1024, arcs=8192, link_rows=64, link_nnz=2048, extra_rows=128),
           "c": dict(),
           # second structured family (round 4): tests/lpgen.py::dense_column_lp — staircase LP with DENSE COLUMNS and
           # power-law row lengths; "d" = the bench size (~4.4M nnz), "d_small" ~ 1/8
           "d_small": dict(family="dense_column", periods=64, rows_per=1024, cols_per=896, dense_cols=48, dense_nnz=4000,
                           tail_rows=512, tail_max=2500),
           "d": dict(family="dense_column"),
           # HELD-OUT families (round 6; the slab partition's constants were not tuned on them): tests/lpgen.py::tall_lp
           # (m >> n, dense coupling rows) and powerlaw_band_lp (power-law lengths in both orientations, hub columns)
           "e_small": dict(family="tall", n=20_000, m=150_000, window=1024, dense_rows=24, dense_nnz=3000),
           "e": dict(family="tall"),
           "f_small": dict(family="plband", n=90_000, m=80_000, band=2048, hubs=400),
           "f": dict(family="plband")}
OUT = os.path.join(HERE, "reference_synth.json")


def record(key, tol):
    if isinstance(CONFIGS[key], dict):
        from lpgen import dense_column_lp, powerlaw_band_lp, structured_lp, tall_lp
        kw = dict(CONFIGS[key])
        fam = kw.pop("family", None)
        lp = {"dense_column": dense_column_lp, "tall": tall_lp, "plband": powerlaw_band_lp, None: structured_lp}[fam](1, **kw)
        m, n, nnz = lp.num_row, lp.num_col, lp.num_nz
    else:
        m, n, nnz = CONFIGS[key]
        sp = solver.SyntheticProblem(m, n, nnz, 1)
        lp = sp.to_lp()
    t0 = time.time()
    r = O.ref_solve(lp, kkt_tolerance=tol, pdlp_iteration_limit=2_000_000)
    wall = time.time() - t0
    return {"m": m, "n": n, "nnz_requested": nnz, "nnz": int(lp.num_nz), "seed": 1, "kkt_tolerance": tol,
            "term_code": r.term_code, "term_iterate": r.term_iterate, "num_iter": r.num_iter, "num_trials": r.num_trials,
            "num_restarts": r.num_restarts, "primal_obj": r.primal_obj, "dual_obj": r.dual_obj,
            "primal_feas": r.primal_feas, "dual_feas": r.dual_feas, "rel_gap": r.rel_gap,
            "norm_rhs": r.norm_rhs, "norm_cost": r.norm_cost,
            "objective_function_value": lp.objective_value(r.col_value),
            "kkt": L.kkt_measures(lp, r.col_value, r.col_dual, r.row_value, r.row_dual),
            "reference_wall_seconds": wall, "reference_solve_seconds": r.solve_seconds,
            "generator": "oracle/_ref/libpdlp_ref.so = /root/reference/highs/pdlp/cupdlp/*.c (oracle/Makefile), 1 thread"}


def main():
    keys = [a for a in sys.argv[1:] if a in CONFIGS] or ["a", "b"]
    tols = [1e-7] + ([1e-4] if "--also-1e-4" in sys.argv else [])
    for key in keys:
        for tol in tols:
            rec = record(key, tol)
            recs = json.load(open(OUT)) if os.path.exists(OUT) else {}
            recs["%s_tol%g" % (key, tol)] = rec
            json.dump(recs, open(OUT, "w"), indent=1, sort_keys=True)
            print(key, tol, rec["num_iter"], rec["objective_function_value"], "%.0f s" % rec["reference_wall_seconds"], flush=True)


if __name__ == "__main__":
    main()
