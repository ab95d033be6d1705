#!/bin/bash
# Runs the REFERENCE's own unmodified CLI and Catch2 unit tests against the drop-in libhighs
# (integration/_build, built by integration/build_dropin.sh) on a GPU box.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
export LD_LIBRARY_PATH=$R/integration/_build:$R/highs_amd/lib:${LD_LIBRARY_PATH:-}
mkdir -p $R/gpurun_out/mps
python - <<PY
import sys; sys.path.insert(0, "$R")
from highs_amd import lp as L
for name in ["25fv47", "afiro", "adlittle", "shell"]:
    L.write_mps(L.HighsLp.from_npz("$R/tests/golden/instances/%s.npz" % name), "$R/gpurun_out/mps/%s.mps" % name)
PY
for name in 25fv47 afiro adlittle shell; do
  echo "== highs --solver=pdlp --presolve=off $name.mps"
  timeout 120 $R/integration/_build/highs_ref_cli --solver=pdlp --presolve=off $R/gpurun_out/mps/$name.mps | grep -E "Model status|PDLP +iter|Objective value|P-D objective|run time"
done
echo "== highs --solver=pdlp (presolve on) 25fv47.mps"
timeout 120 $R/integration/_build/highs_ref_cli --solver=pdlp $R/gpurun_out/mps/25fv47.mps | grep -E "Model status|PDLP +iter|Objective value|P-D objective"
for name in afiro adlittle shell; do
  echo "== highs --solver=hipdlp --presolve=off $name.mps"
  timeout 120 $R/integration/_build/highs_ref_cli --solver=hipdlp --presolve=off $R/gpurun_out/mps/$name.mps | grep -E "Model status|PDLP +iter|Objective value|P-D objective|run time"
done
echo "== reference Catch2 unit tests (in-code LPs)"
for t in pdlp-distillation-lp pdlp-3d-lp pdlp-boxed-row-lp pdlp-infeasible-lp pdlp-unbounded-lp pdlp-restart-lp pdlp-restart-add-row; do
  timeout 120 $R/integration/_build/unit_tests_ref "$t" 2>&1 | tail -3 | tr '\n' ' '; echo " <- $t"
done
