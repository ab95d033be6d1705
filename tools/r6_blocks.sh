#!/bin/bash
# tools/r6_blocks.sh TAG [configs] — per-block phase tables of the two slab launches (raw + JSON) and the bench lines
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
TAG=${1:-r06_blocks}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; }
for cfg in ${*:-c d}; do
  python bench.py --config $cfg --cpu-iters 0 2>$OUT/bench_$cfg.err | tee $OUT/bench_$cfg.json | line $cfg
  PDLP_MI355X_SLAB_PROF=$OUT/prof_$cfg.bin python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab " | tee $OUT/prof_$cfg.log
  python tools/slab_blocks.py $OUT/prof_$cfg.bin
done
