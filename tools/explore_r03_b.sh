#!/bin/bash
# round-3 exploration B: segment tasks for long majors — parity first, then config-c timings with one / two blocks per CU
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py tests/test_gpu_hipdlp.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 15 $O/pytest.log
for OCC in 1 0; do
  PDLP_MI355X_SLAB_OCC2=$OCC python tools/spmv_sweep.py --structured --iters 400 --variants "slab=1;slab=0,xcd=0" > $O/sweep_c_occ$OCC.log 2>&1
  PDLP_MI355X_SLAB_OCC2=$OCC python tools/kbench.py --structured --reps 30 --kernels spmv_ax_plain_nolong,spmv_ax_plain,spmv_aty_plain,decide_primal,spmv_ax,spmv_aty > $O/kbench_c_occ$OCC.log 2>&1
done
tail -n 5 $O/sweep*.log $O/kbench*.log
