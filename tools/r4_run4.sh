#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4_run4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py -q -x -m gpu \
  -k "device_driven or instances_bit_exact or barrier_launch or trial_loop_variants or two_large or concurrent" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
python tools/solve_times.py 2>&1 | grep -v amdgpu.ids; python tools/small_loop.py 25fv47 80bau3b 2>&1 | grep -v amdgpu.ids
