#!/bin/bash
# hierarchical barrier: 80bau3b (48 workgroups) both ways, 100k x 100k, parity
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
O=gpurun_out/r03h; mkdir -p $O
echo "80bau3b sweep barrier:"; PDLP_MI355X_HIER_BARRIER=0 python tools/small_loop.py 80bau3b 2>&1 | grep -v amdgpu.ids
echo "80bau3b hierarchical:"; PDLP_MI355X_HIER_BARRIER=1 python tools/small_loop.py 80bau3b 2>&1 | grep -v amdgpu.ids
echo "25fv47 all XCDs, sweep / hierarchical:"; PDLP_MI355X_XCD_LOCAL=0 PDLP_MI355X_HIER_BARRIER=0 python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids
PDLP_MI355X_XCD_LOCAL=0 PDLP_MI355X_HIER_BARRIER=1 python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids
timeout 300 python bench.py --config a > $O/a_persistent.json 2> $O/a_persistent.err
python - $O/a_persistent.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("trial_launches"))
PY
PDLP_MI355X_HIER_BARRIER=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
