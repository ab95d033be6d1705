#!/bin/bash
# tools/r6_xp.sh TAG — round 6 experiments on one box (EXPERIMENT switches: bits may differ from the oracle's)
export PDLP_MI355X_DEV=1
cd "$(dirname "$0")/.."
TAG=${1:-r06_xp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; }
run() { # name cfg env...
  local name=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --cpu-iters 0 2>/dev/null | line $name
  env "$@" PDLP_MI355X_SLAB_PROF=$OUT/prof_$name.bin python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch" | grep -E "fused" | grep -E "epilogue|barrier|kernel"
  python tools/slab_blocks.py $OUT/prof_$name.bin | grep task
}
run d_nocap d PDLP_MI355X_XP_COLS_CAP=0
run c c X=1
run b b X=1
run qp qp X=1
