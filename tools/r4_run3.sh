#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4_run3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py tests/test_gpu_qp.py tests/test_gpu_dropin.py -q -x -m gpu \
  -k "device_driven or bit_exact or barrier_launch or trial_loop_variants or qp_through or special_lps or hot_start or iteration_limit or time_limit" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
for cs in 0 1; do echo "== CHECK_SMALL=$cs"; PDLP_MI355X_CHECK_SMALL=$cs python tools/solve_times.py 2>&1 | grep -v amdgpu.ids; PDLP_MI355X_CHECK_SMALL=$cs python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids; done
