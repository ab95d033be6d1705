#!/bin/bash
# tools/r5_balance.sh TAG — round 5: the work-balanced slab partition on the GPU box.  GPU tests first (log kept), then the
# bench lines of configs b / c / d / a / qp and the per-block phase profile of the slab launches.
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -n "${SKIP_TESTS:-}" ]; then
  echo "(GPU tests skipped)"
elif [ -n "${SUBSET:-}" ]; then  # the tests that see the slab layout, the loop variants and the device gate
  PYTEST_TIMEOUT=${PYTEST_TIMEOUT:-600} bash tools/gpu_pytest.sh $TAG/pytest_gpu tests -m gpu -q -x --timeout 200 \
    -k "bit_exact or long or dense or structured or spmv or fused or two_large or setup or trial_loop or hard_instances_first"
else
  PYTEST_TIMEOUT=${PYTEST_TIMEOUT:-1500} bash tools/gpu_pytest.sh $TAG/pytest_gpu tests -m gpu -q ${PYTEST_ARGS:-}
fi
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; }
for cfg in ${CONFIGS:-b c d a qp}; do
  python bench.py --config $cfg --cpu-iters 0 2>$OUT/bench_$cfg.err | tee $OUT/bench_$cfg.json | line $cfg
done
for cfg in ${PROF_CONFIGS:-b c d}; do
  echo "== bench.py --config $cfg (PDLP_MI355X_SLAB_PROF=1)"
  PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch"
done | tee $OUT/slab_phase_profile.log
if [ -n "${EXTRA_A:-}" ]; then
  echo "== config a forced onto the slab layout (fused 2-launch trial) vs the persistent loop"
  PDLP_MI355X_SLAB=1 python bench.py --config a --cpu-iters 0 2>/dev/null | line a_slab_fused
  PDLP_MI355X_SLAB=1 PDLP_MI355X_FUSED=0 python bench.py --config a --cpu-iters 0 2>/dev/null | line a_slab_3launch
fi
