#!/bin/bash
# round-3 exploration A: config-c layouts with the round-2 kernels (baseline for the long-major rework)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out/r03a
python tools/spmv_sweep.py --structured --iters 400 --variants "slab=1;slab=1,xcd=0;slab=1,xcd=1;slab=0,xcd=0;slab=0,xcd=1" > gpurun_out/r03a/sweep_c.log 2>&1
python tools/kbench.py --structured --reps 30 --kernels spmv_ax_plain_slab,spmv_ax_plain_side,spmv_ax_plain,spmv_aty_plain,decide_primal,spmv_ax,spmv_aty > gpurun_out/r03a/kbench_c.log 2>&1
tail -n 20 gpurun_out/r03a/*.log
