#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r03f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "staged or structured" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 $O/pytest.log
for ST in 1 0; do
  PDLP_MI355X_SLAB_STAGE=$ST timeout 300 python tools/spmv_sweep.py --structured --iters 400 --variants "slab=1" > $O/sweep_c_stage$ST.log 2>&1
  PDLP_MI355X_SLAB_STAGE=$ST python tools/kbench.py --structured --reps 30 --kernels spmv_ax_plain_nolong,spmv_ax_plain,spmv_aty_plain,spmv_ax,spmv_aty > $O/kbench_c_stage$ST.log 2>&1
done
tail -n 3 $O/sweep*.log $O/kbench*.log
