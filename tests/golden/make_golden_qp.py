#!/usr/bin/env python3
"""QP goldens (run in the build container only).  The reference has NO PDLP on QPs (HighsOptions.cpp:1178-1181
gates solver="pdlp" to LPs), so parity of the QP prox path is pinned on the OPTIMAL OBJECTIVES of the
reference's own QP solver: random convex QPs with a diagonal Hessian (tests/lpgen.py::random_diag_qp) are written
as .mps with a QUADOBJ section and solved by the reference binary ($HIGHS_REF_BIN, default
integration/_build/highs_reference_cli = `make -C integration reference`; its default QP solver is the active-set `qpasm`).  Output: tests/golden/qp/qp<seed>.npz
(the model incl. the Hessian) and tests/golden/reference_qp.json (objective, model status).

Round 3: Hessians with OFF-DIAGONAL entries — random sparse PSD Q = G'G + diag(d) on the same LPs
(tests/lpgen.py::random_sparse_qp: 16 small ones and three larger ones, up to 1200 columns) and the reference's own
QP instances (check/instances/qjh.mps, qjh_quadobj.mps, qjh_qmatrix.mps, qptestnw.lp — the .lp file goes through
the reference's own reader: the binary writes it out as .mps first) -> tests/golden/qp/sq<seed>.npz, <instance>.npz
and tests/golden/reference_qp_sparse.json."""
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
from lpgen import random_diag_qp, random_sparse_qp  # noqa: E402

# the UNMODIFIED reference binary, built from /root/reference by this repository's own recipe: make -C integration reference
HIGHS = os.environ.get("HIGHS_REF_BIN", os.path.join(ROOT, "integration", "_build", "highs_reference_cli"))


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
    sparse()


def run_ref(mps, cwd):
    out = subprocess.run([HIGHS, mps], capture_output=True, text=True, cwd=cwd).stdout
    g = lambda pat: (re.search(pat, out) or [None, None])[1]
    return (g(r"Model status\s*:\s*(.+)") or "").strip(), g(r"Objective value\s*:\s*(\S+)")


def sparse():
    from highs_amd import solver
    inst = os.path.join(os.environ.get("HIGHS_REFERENCE", "/root/reference"), "check", "instances")
    recs = {}
    with tempfile.TemporaryDirectory() as tmp:
        cases = [("sq%d" % s, random_sparse_qp(s)) for s in range(16)]
        cases += [("sq100", random_sparse_qp(100, m=80, n=200)), ("sq101", random_sparse_qp(101, m=200, n=500)),
                  ("sq102", random_sparse_qp(102, m=400, n=1200, density=0.004))]
        for name, lp in cases:
            mps = os.path.join(tmp, name + ".mps")
            L.write_mps(lp, mps)
            status, obj = run_ref(mps, tmp)
            if status != "Optimal" or obj is None:
                print(name, "skipped:", status)
                continue
            lp.to_npz(os.path.join(HERE, "qp", name + ".npz"))
            nnz_off = int((lp.hessian[1] != __import__("numpy").repeat(__import__("numpy").arange(lp.num_col), __import__("numpy").diff(lp.hessian[0]))).sum())
            recs[name] = {"objective_value": float(obj), "model_status": status, "rows": lp.num_row, "cols": lp.num_col,
                          "hessian_off_diagonal_entries": nnz_off, "sense": lp.sense,
                          "solver": "reference binary, default QP solver (qpasm)"}
            print(name, status, obj)
        for f in ("qjh.mps", "qjh_quadobj.mps", "qjh_qmatrix.mps", "qptestnw.lp"):
            src = os.path.join(inst, f)
            name = f.replace(".", "_")
            mps = os.path.join(tmp, name + ".mps")
            if f.endswith(".lp"):  # through the reference's own .lp reader
                subprocess.run([HIGHS, src, "--write_model_file", mps, "--solver", "simplex", "--time_limit", "0"], capture_output=True,
                               text=True, cwd=tmp)
            else:
                mps = src
            status, obj = run_ref(src, tmp)
            if status != "Optimal" or obj is None or not os.path.exists(mps):
                print(name, "skipped:", status)
                continue
            lp, _info = solver.read_mps(mps)  # the product's reader (QUADOBJ / QMATRIX -> lower triangle)
            lp.to_npz(os.path.join(HERE, "qp", name + ".npz"))
            recs[name] = {"objective_value": float(obj), "model_status": status, "rows": lp.num_row, "cols": lp.num_col,
                          "sense": lp.sense, "solver": "reference binary, default QP solver (qpasm)", "file": "check/instances/" + f}
            print(name, status, obj)
    json.dump(recs, open(os.path.join(HERE, "reference_qp_sparse.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
