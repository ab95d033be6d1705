#!/bin/bash
# round-3 exploration D: fused 2-launch trial — parity, then timings fused / unfused on configs b and c
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r03d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py tests/test_gpu_qp.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 $O/pytest.log
for F in 1 0; do
  PDLP_MI355X_FUSED=$F timeout 300 python tools/spmv_sweep.py --iters 800 --variants "slab=1" > $O/sweep_b_fused$F.log 2>&1
  PDLP_MI355X_FUSED=$F timeout 300 python tools/spmv_sweep.py --structured --iters 400 --variants "slab=1,w=14" > $O/sweep_c_fused$F.log 2>&1
done
tail -n 3 $O/sweep*.log
