#!/bin/bash
# tools/r6_xp.sh TAG "CONFIGS" NAME[=ENV[,ENV...]] ... — same-box A/B harness of round 6 (run through gpurun): every variant on
# every bench configuration, one bench line each (it/s, us per iteration, the two launches of a trial, set-up) plus the
# per-block phase profile of the fused launch, then the bit-exact / fused / barrier / held-out tests on the shipped build.
#   a variant is a name and the environment it runs in, e.g.
#     bash tools/r6_xp.sh r06_cc "b c" default nt=PDLP_MI355X_CONST_CACHED=0 cached=PDLP_MI355X_CONST_CACHED=1
#     bash tools/r6_xp.sh r06_k3 "d f" k3 k2=PDLP_MI355X_LIB=$PWD/highs_amd/lib/alt/lib_k2.so   (tools/build_alt.sh k2 -D...)
#   REPS=n repeats every variant n times (boxes differ by +-1.5 %, runs on one box by +-0.5 %).
# The logs under profiles/r06_*_ab.log were made with this script as it stood for each experiment (git history).
export PDLP_MI355X_DEV=1
cd "$(dirname "$0")/.."
TAG=${1:-r06_xp}; CFGS=${2:-"b c d"}; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()}, 'setup', round(d['setup_seconds'],3))"; }
run() { # name cfg env...
  local name=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --cpu-iters 0 2>$OUT/$name.err | line $name
  env "$@" PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch" | grep -E "fused" | grep -E "arrival|barrier|kernel"
}
for cfg in $CFGS; do
  for rep in $(seq 1 ${REPS:-1}); do
    for v in "$@"; do
      name=${v%%=*}; envs=PDLP_X=0
      [ "$v" != "$name" ] && envs=$(echo "${v#*=}" | tr ',' ' ')
      run ${cfg}_${name}_$rep $cfg $envs
    done
  done
done
bash tools/gpu_pytest.sh $TAG/pytest tests -m gpu -q -x --timeout 600 -k "fused or bit_exact or two_large or fault or barrier or held_out"
