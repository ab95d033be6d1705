#!/usr/bin/env python3
"""QP goldens (run in the build container only).  The reference has NO PDLP on QPs (HighsOptions.cpp:1178-1181
gates solver="pdlp" to LPs), so parity of the QP prox path is pinned on the OPTIMAL OBJECTIVES of the
reference's own QP solver: random convex QPs with a diagonal Hessian (tests/lpgen.py::random_diag_qp) are written
as .mps with a QUADOBJ section and solved by the reference binary ($HIGHS_REF_BIN, default
/tmp/ref_build/bin/highs; its default QP solver is the active-set `qpasm`).  Output: tests/golden/qp/qp<seed>.npz
(the model incl. the Hessian) and tests/golden/reference_qp.json (objective, model status)."""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from highs_amd import lp as L  # noqa: E402
from lpgen import random_diag_qp  # noqa: E402

HIGHS = os.environ.get("HIGHS_REF_BIN", "/tmp/ref_build/bin/highs")


def main():
    os.makedirs(os.path.join(HERE, "qp"), exist_ok=True)
    recs = {}
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(16):
            lp = random_diag_qp(seed)
            mps = os.path.join(tmp, "qp%d.mps" % seed)
            L.write_mps(lp, mps)
            out = subprocess.run([HIGHS, mps], capture_output=True, text=True, cwd=tmp).stdout
            g = lambda pat: (re.search(pat, out) or [None, None])[1]
            status = (g(r"Model status\s*:\s*(.+)") or "").strip()
            obj = g(r"Objective value\s*:\s*(\S+)")
            if status != "Optimal" or obj is None:
                print("seed", seed, "skipped:", status)
                continue
            lp.to_npz(os.path.join(HERE, "qp", "qp%d.npz" % seed))
            recs["qp%d" % seed] = {"objective_value": float(obj), "model_status": status, "rows": lp.num_row,
                                   "cols": lp.num_col, "sense": lp.sense,
                                   "solver": "reference binary, default QP solver (qpasm)"}
            print(seed, status, obj)
    json.dump(recs, open(os.path.join(HERE, "reference_qp.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
