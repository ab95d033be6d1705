#!/bin/bash
# small LPs after the window-of-nine row sums: phases, loop times, parity
cd "$(dirname "$0")/.."
O=gpurun_out/r03j; mkdir -p $O
PDLP_MI355X_SMALL_PROF=1 python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids
python tools/small_loop.py 25fv47 80bau3b 2>&1 | grep -v amdgpu.ids
python tools/solve_times.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
