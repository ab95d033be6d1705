#!/bin/bash
# tools/r6_xp.sh TAG — round 6 experiments on one box (A/B of development switches)
export PDLP_MI355X_DEV=1
cd "$(dirname "$0")/.."
TAG=${1:-r06_xp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()}, 'setup', round(d['setup_seconds'],3))"; }
run() { # name cfg env...
  local name=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --cpu-iters 0 2>$OUT/$name.err | line $name
  grep "slab operand" $OUT/$name.err
}
for cfg in b d e c f; do
  run ${cfg}_tuned $cfg PDLP_MI355X_SLAB_PROF=1
  run ${cfg}_rule $cfg PDLP_MI355X_SLAB_TUNE=0
done
python bench.py --solver hipdlp --cpu-iters 0 2>/dev/null | line hipdlp_tuned
bash tools/gpu_pytest.sh $TAG/pytest tests -m gpu -q -x --timeout 600 -k "bit_exact or setup or held_out"
