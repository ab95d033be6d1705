#!/bin/bash
# tools/r6_xp.sh [TAG] — round-6 A/B of the fused trial's barrier: arrival words 8 bytes apart (the build) vs 128 B / 256 B /
# 4 KB apart (highs_amd/lib/alt/lib_st{16,32,512}.so): do the sweeps of 256 blocks queue on one memory channel?
export PDLP_MI355X_DEV=1
cd "$(dirname "$0")/.."
TAG=${1:-r06_xp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()}, 'setup', round(d['setup_seconds'],3))"; }
run() { # name cfg env...
  local name=$1 cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --cpu-iters 0 2>$OUT/$name.err | line $name
  env "$@" PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch" | grep -E "fused" | grep -E "barrier|kernel"
}
for cfg in qp d e c f; do
  run ${cfg}_nt $cfg PDLP_X=0
  run ${cfg}_ld $cfg PDLP_MI355X_LIB=$PWD/highs_amd/lib/alt/lib_nont1.so
  run ${cfg}_ldst $cfg PDLP_MI355X_LIB=$PWD/highs_amd/lib/alt/lib_nont2.so
done
bash tools/gpu_pytest.sh $TAG/pytest tests -m gpu -q -x --timeout 600 -k "fused or bit_exact or two_large or fault or barrier or held_out"
