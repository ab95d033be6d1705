#!/bin/bash
# tools/r6_xp.sh [TAG] — round-6 A/B of the fused trial's barrier with task workgroups (configs d, f): streaming blocks
# vouch for "their" task workgroups' arrival (PDLP_MI355X_PAIRED_TASKS=1, default) vs every word swept by everybody (0)
export PDLP_MI355X_DEV=1
cd "$(dirname "$0")/.."
TAG=${1:-r06_xp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()}, 'setup', round(d['setup_seconds'],3))"; }
run() { # name cfg env...
  local name=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --cpu-iters 0 2>$OUT/$name.err | line $name
  env "$@" PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch" | grep -E "fused" | grep -E "epilogue|barrier|kernel"
}
for cfg in d f; do
  for rep in 1 2; do
    run ${cfg}_paired_$rep $cfg PDLP_MI355X_PAIRED_TASKS=1
    run ${cfg}_swept_$rep $cfg PDLP_MI355X_PAIRED_TASKS=0
  done
done
bash tools/gpu_pytest.sh $TAG/pytest tests -m gpu -q -x --timeout 600 -k "fused or bit_exact or two_large or fault or barrier or held_out"
