#!/bin/bash
# tools/r4_graph_ab.sh — A/B on one box: the 42-trial hipGraph vs eager launches queued ahead (PDLP_MI355X_GRAPH=0), configs b and c
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r4_graph_ab.log; : > $O
cd $R
for rep in 1 2; do
  for cfg in b c; do
    for g in -1 0; do
      echo "== config $cfg PDLP_MI355X_GRAPH=$g rep $rep" >> $O
      PDLP_MI355X_GRAPH=$g python bench.py --config $cfg --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step']*1e3, d.get('trial_launches'))" >> $O
    done
  done
done
cat $O
