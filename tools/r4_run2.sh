#!/bin/bash
# round 4: barrier fall-back / contention tests, QP through Highs::run, dense-column LP
cd "$(dirname "$0")/.."
O=gpurun_out/r4_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -x -m gpu \
  -k "barrier_launch or two_large_contexts or concurrent_solver or qp_through or unpatched or dense_column or slab_width or c_api_client or device_driven" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
