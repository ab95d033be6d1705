#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r03g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py tests/test_gpu_qp.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 $O/pytest.log
python tools/solve_times.py > $O/small_persistent.log 2>&1; PDLP_MI355X_PERSISTENT=0 python tools/solve_times.py > $O/small_launches.log 2>&1
tail -n 3 $O/small_*.log
