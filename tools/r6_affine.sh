#!/bin/bash
# tools/r6_affine.sh TAG — round 6, step 1 on the GPU box: XCD-affine segment tasks + the 5/4 column cap of the transposed
# operand.  The GPU tests that see the slab layout / long majors first (log kept), then A/B bench lines of configs c, d, b
# (PDLP_MI355X_AFFINE_TASKS=0: tasks in (major, segment) order as in round 5), the per-block phase profile, and the PMC
# traffic of configs c and d (separate --pmc passes, tools/pmc_traffic.sh).
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
TAG=${1:-r06_affine}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -z "${SKIP_TESTS:-}" ]; then
  PYTEST_TIMEOUT=${PYTEST_TIMEOUT:-900} bash tools/gpu_pytest.sh $TAG/pytest_gpu tests -m gpu -q -x --timeout 300 \
    -k "${KEXPR:-bit_exact or long or dense or structured or spmv or fused or setup or trial_loop or hard_instances_first}"
fi
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; }
for cfg in ${CONFIGS:-c d b}; do
  python bench.py --config $cfg --cpu-iters 0 2>$OUT/bench_$cfg.err | tee $OUT/bench_$cfg.json | line $cfg
  if [ $cfg != b ]; then
    PDLP_MI355X_AFFINE_TASKS=0 python bench.py --config $cfg --cpu-iters 0 2>$OUT/bench_${cfg}_major_order.err | tee $OUT/bench_${cfg}_major_order.json | line ${cfg}_tasks_in_major_order
  fi
done
for cfg in ${PROF_CONFIGS:-c d}; do
  echo "== bench.py --config $cfg (PDLP_MI355X_SLAB_PROF=1)"
  PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch"
  echo "== the same, PDLP_MI355X_AFFINE_TASKS=0"
  PDLP_MI355X_AFFINE_TASKS=0 PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch"
done | tee $OUT/slab_phase_profile.log
if [ -z "${SKIP_PMC:-}" ]; then
  bash tools/pmc_traffic.sh $TAG ${PMC_CONFIGS:-c d} > $OUT/pmc_traffic.log 2>&1
  cp gpurun_out/profiles_$TAG/pmc_traffic.json $OUT/pmc_traffic.json
  python -c "import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k:{kk:round(vv/1e6,1) for kk,vv in v.items()} for k,v in d.items() if k in 'bcd'})"
fi
