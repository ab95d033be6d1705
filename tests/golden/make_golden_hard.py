#!/usr/bin/env python3
"""The LPs of check/instances that stress a first-order method (build container only) — VERDICT round 4, item 4.

greenbea (2 392 x 5 405, the second-largest LP of the directory: the reference's CPU pdlp runs 1.59 M iterations in
120 s without converging), perold, gas11 (unbounded) and primal1 (a QP with a diagonal Hessian).
Goldens, from the REFERENCE's own code:
  "simplex": model status and optimal objective of the reference CLI with --solver=simplex (for primal1, a QP, the CLI runs
             the reference's QP solver) — a 1e-6 target that needs no CPU-pdlp convergence;
  "cupdlp":  the real cuPDLP-C core compiled from the reference sources (oracle/_ref) at the default tolerance with a
             time budget of REF_TIME_LIMIT seconds per instance: where it converges, iteration count and objectives; where
             it does not, the record says so (term_code) with the iterations it got through.
The reference CLI here is integration/_build/highs_reference_cli: the UNMODIFIED reference (app/RunHighs.cpp on
libhighs_reference.so.1, `make -C integration reference` — nothing of this repository in it).

    python tests/golden/make_golden_hard.py   -> tests/golden/reference_hard.json, tests/golden/instances/*.npz
"""
import json
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
from highs_amd import solver  # noqa: E402

CLI = os.environ.get("HIGHS_REF_BIN", os.path.join(ROOT, "integration", "_build", "highs_reference_cli"))
NAMES = ["perold", "gas11", "primal1", "greenbea"]
REF_TIME_LIMIT = float(os.environ.get("REF_TIME_LIMIT", "3000"))


def simplex_record(mps):
    out = subprocess.run([CLI, "--solver=simplex", mps], capture_output=True, text=True).stdout
    g = lambda pat: (re.search(pat, out) or [None, None])[1]
    return {"model_status": g(r"Model status\s*:\s*(.+)"), "objective_value": float(g(r"Objective value\s*:\s*(\S+)") or "nan"),
            "is_qp": bool(re.search(r"^QP ", out, re.M))}


def main():
    out_path = os.path.join(HERE, "reference_hard.json")
    recs = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for name in NAMES:
        mps = f"{MG.REF}/check/instances/{name}.mps"
        lp, _ = solver.read_mps(mps)
        lp.to_npz(os.path.join(HERE, "instances", name + ".npz"))
        rec = {"rows": lp.num_row, "cols": lp.num_col, "nnz": int(lp.num_nz), "simplex": simplex_record(mps)}
        if not rec["simplex"]["is_qp"]:  # (the cuPDLP-C core is an LP code)
            t0 = time.time()
            rec["cupdlp"] = MG.cupdlp_record(lp, time_limit=REF_TIME_LIMIT)
            rec["cupdlp"]["seconds"] = round(time.time() - t0, 1)
            rec["cupdlp"]["time_limit"] = REF_TIME_LIMIT
        recs[name] = rec
        print(name, rec["simplex"], (rec.get("cupdlp") or {}).get("num_iter"), (rec.get("cupdlp") or {}).get("term_code"), flush=True)
        json.dump(recs, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
