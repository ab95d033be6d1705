#!/usr/bin/env python3
"""More instances of the reference's check/instances for the converged-solution parity test (build container only).

The 13 ctest instances + 80bau3b live in reference_pdlp.json (make_golden.py).  This adds every other LP of
check/instances that the reference's CPU pdlp finishes within minutes at its default tolerance: five it solves to
optimality and nine it reports as primal infeasible or unbounded.  Same record layout as make_golden.py ("highs": what
the reference binary prints, "cupdlp": the real cuPDLP-C core compiled from the reference sources, oracle/_ref); the
LPs are read with the library's reader (pinned on the reference's, tests/test_mps_reader.py) and stored as .npz.

    python tests/golden/make_golden_more.py       -> tests/golden/reference_pdlp_more.json, tests/golden/instances/*.npz
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
from highs_amd import solver  # noqa: E402

OPTIMAL = ["egout-ac", "qap04", "israel", "etamacro", "standmps"]
INFEASIBLE_OR_UNBOUNDED = ["galenet", "woodinfe", "forest6", "gams10am", "ex72a", "box1", "bgetam", "cplex1", "refinery"]


def main():
    recs = {}
    for name in OPTIMAL + INFEASIBLE_OR_UNBOUNDED:
        mps = f"{MG.REF}/check/instances/{name}.mps"
        lp, _ = solver.read_mps(mps)
        lp.to_npz(os.path.join(HERE, "instances", name + ".npz"))
        recs[name] = {"rows": lp.num_row, "cols": lp.num_col, "nnz": int(lp.num_nz), "expect": "optimal" if name in OPTIMAL else "infeasible_or_unbounded",
                      "highs": MG.highs_record(mps), "cupdlp": MG.cupdlp_record(lp)}
        print(name, recs[name]["highs"], recs[name]["cupdlp"]["num_iter"], recs[name]["cupdlp"]["term_code"], flush=True)
    json.dump(recs, open(os.path.join(HERE, "reference_pdlp_more.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
