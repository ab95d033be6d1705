#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r03c; mkdir -p $O
python tools/spmv_sweep.py --structured --iters 200 --variants "slab=1,w=9;slab=1,w=10;slab=1,w=11;slab=1,w=12;slab=1,w=13;slab=1,w=14;slab=1,w=15;slab=1,w=17" > $O/sweep_c_w.log 2>&1
cat $O/sweep_c_w.log
