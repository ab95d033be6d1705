#!/bin/bash
# tools/r6_heldout.sh TAG — round 6: the two HELD-OUT structured families (bench.py --config e / f) with the partition constants as
# they are: bench lines, per-block phase profile (raw tables + JSON), and their GPU tests.
export PDLP_MI355X_DEV=1
cd "$(dirname "$0")/.."
TAG=${1:-r06_heldout}; OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()}, 'launches', d['trial_launches'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; }
for cfg in ${CONFIGS:-e f}; do
  python bench.py --config $cfg ${BENCH_ARGS:---cpu-iters 0} 2>$OUT/bench_$cfg.err | tee $OUT/bench_$cfg.json | line $cfg
  PDLP_MI355X_SLAB_PROF=$OUT/prof_$cfg.bin python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab " | tee $OUT/prof_$cfg.log
  python tools/slab_blocks.py $OUT/prof_$cfg.bin
done
if [ -z "${SKIP_TESTS:-}" ]; then
  PYTEST_TIMEOUT=900 bash tools/gpu_pytest.sh $TAG/pytest tests -m gpu -q -x --timeout 600 -k "held_out"
fi
