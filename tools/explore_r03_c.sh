#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r03c; mkdir -p $O
python tools/spmv_sweep.py --structured --iters 200 --variants "slab=1,w=14;slab=1,w=14;slab=1,w=17;slab=1,w=14,occ2=0;slab=1,w=14" > $O/rep.log 2>&1
cat $O/rep.log; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
