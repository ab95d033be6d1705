#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference tree (run in the build container only).

Inputs  : /root/reference/check/instances/<name>.mps (13 ctest PDLP instances,
          check/CMakeLists.txt:321-335) read with highs_amd.lp.read_mps.
Outputs : tests/golden/instances/<name>.npz   the LP as HighsLp arrays (CSC)
          tests/golden/reference_pdlp.json    per instance:
             - "highs": what the reference BINARY prints for
               `highs --solver=pdlp --presolve=off <mps>` (model status, PDLP
               iterations, objective, P-D objective error), if a reference build
               exists at $HIGHS_REF_BIN (default integration/_build/highs_reference_cli:
               `make -C integration reference`);
             - "cupdlp": full-precision outputs of the real cuPDLP-C core
               compiled from the reference sources (oracle/_ref), default
               tolerances 1e-7: iterations, trials, term code, cuPDLP primal /
               dual objective, residual norms, HiGHS-style objective c'x + offset.
          tests/golden/special_lps.json  same "cupdlp" record for the in-code
          LPs of check/TestPdlp.cpp at kkt_tolerance 1e-4 (+ the pinned counts).
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oraclelib as O  # noqa: E402
from highs_amd import lp as L  # noqa: E402

REF = "/root/reference"
# the UNMODIFIED reference binary, built from /root/reference by this repository's own recipe: make -C integration reference
HIGHS = os.environ.get("HIGHS_REF_BIN", os.path.join(ROOT, "integration", "_build", "highs_reference_cli"))
NAMES = ["25fv47", "adlittle", "afiro", "avgas", "blending", "chip", "e226", "scrs8", "sctest", "shell", "stair",
         "standata", "standgub"]
# largest LP bundled with the reference (BASELINE.json config 3 stand-in: pds-100 is not in the tree);
# not part of the reference's ctest list, so no ctest prefix
EXTRA = ["80bau3b"]
# CPU objective prefixes the reference's ctest greps for, check/CMakeLists.txt:321-335
CTEST_PREFIX = {"25fv47": "5.5018469", "adlittle": "2.254949", "afiro": "-4.64753150", "avgas": "-7.7499999",
                "blending": "-3.1999999", "chip": "-8.9999999", "e226": "-1.16389294", "scrs8": "9.04297094",
                "sctest": "5.75000000", "shell": "1.20882534", "stair": "-2.51266942", "standata": "1.25769944",
                "standgub": "1.25769944"}


def cupdlp_record(lp, **kw):
    r = O.ref_solve(lp, **kw)
    return {"term_code": r.term_code, "term_iterate": r.term_iterate, "num_iter": r.num_iter,
            "num_trials": r.num_trials, "primal_obj": r.primal_obj, "dual_obj": r.dual_obj,
            "primal_feas": r.primal_feas, "dual_feas": r.dual_feas, "rel_gap": r.rel_gap,
            "norm_rhs": r.norm_rhs, "norm_cost": r.norm_cost,
            "objective_function_value": lp.objective_value(r.col_value),
            "kkt": L.kkt_measures(lp, r.col_value, r.col_dual, r.row_value, r.row_dual)}


def highs_record(mps):
    if not os.path.exists(HIGHS):
        return None
    out = subprocess.run([HIGHS, "--solver=pdlp", "--presolve=off", mps], capture_output=True, text=True).stdout
    g = lambda pat: (re.search(pat, out) or [None, None])[1]
    return {"model_status": g(r"Model status\s*:\s*(.+)"), "pdlp_iterations": int(g(r"PDLP\s+iterations:\s*(\d+)") or -1),
            "objective_value": float(g(r"Objective value\s*:\s*(\S+)") or "nan"),
            "pd_objective_error": float(g(r"P-D objective error\s*:\s*(\S+)") or "nan")}


def main():
    os.makedirs(os.path.join(HERE, "instances"), exist_ok=True)
    if not O.ref_available():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    recs = {}
    for name in NAMES + EXTRA:
        mps = f"{REF}/check/instances/{name}.mps"
        lp = L.read_mps(mps)
        lp.to_npz(os.path.join(HERE, "instances", name + ".npz"))
        recs[name] = {"rows": lp.num_row, "cols": lp.num_col, "nnz": lp.num_nz, "ctest_cpu_prefix": CTEST_PREFIX.get(name),
                      "highs": highs_record(mps), "cupdlp": cupdlp_record(lp)}
        print(name, recs[name]["highs"], recs[name]["cupdlp"]["num_iter"])
    json.dump(recs, open(os.path.join(HERE, "reference_pdlp.json"), "w"), indent=1, sort_keys=True)
    sp = {}
    for name, lp in L.special_lps().items():
        sp[name] = cupdlp_record(lp, kkt_tolerance=1e-4)
    sp["distillation"]["pinned_iterations"] = 160  # check/TestPdlp.cpp:29,44
    sp["distillation_limit80"] = cupdlp_record(L.special_lps()["distillation"], kkt_tolerance=1e-4, pdlp_iteration_limit=80)
    sp["distillation_limit80"]["pinned_iterations"] = 79  # check/TestPdlp.cpp:61
    json.dump(sp, open(os.path.join(HERE, "special_lps.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
