#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03k
python tools/make_gpu_iteration_pins.py gpurun_out/r03k/gpu_iteration_counts.json 2>&1 | grep -v amdgpu.ids | tail -40
python tools/make_gpu_iteration_pins.py gpurun_out/r03k/gpu_iteration_counts_2.json > /dev/null 2>&1
cmp gpurun_out/r03k/gpu_iteration_counts.json gpurun_out/r03k/gpu_iteration_counts_2.json && echo "second run: identical"
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from highs_amd import solver, lp as L
lp = L.HighsLp.from_npz("tests/golden/instances/afiro.npz")
lp.col_cost = lp.col_cost.copy(); lp.col_cost[3] = float("nan")
o = solver.solveLpCupdlp(lp, pdlp_iteration_limit=100000, time_limit=60.0)
print("nan:", o.status, o.model_status, o.pdlp_iteration_count, solver.kError)
PY
