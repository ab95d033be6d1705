#!/bin/bash
# tools/r6_flake.sh — how often the 8-rank folded afiro solve times out, with and without a ninth process that holds
# a used HIP context on the device (what the pytest parent is).  Development.
export PDLP_MI355X_DEV=1 HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r06_flake}; mkdir -p $OUT; rm -f $OUT/idle.ready
W=${WORLD:-8}; CASE=${CASE:-solve:afiro}; N=${N:-8}
run() { # name env...
  local name=$1; shift
  local ok=0 bad=0
  for i in $(seq 1 $N); do
    uid=$(python -c "import os;print(os.urandom(128).hex())")
    pids=()
    for r in $(seq 0 $((W-1))); do
      env "$@" timeout 60 python tests/mesh_worker.py $r $W $uid $CASE $OUT/${name}_r$r.npz > $OUT/${name}_${i}_r$r.log 2>&1 &
      pids+=($!)
    done
    fail=0
    for p in "${pids[@]}"; do wait $p || fail=1; done
    if [ $fail = 0 ]; then ok=$((ok+1)); rm -f $OUT/${name}_${i}_r*.log; else bad=$((bad+1)); fi
  done
  echo "$name ok=$ok bad=$bad"
}
run default PDLP_X=0
cat > $OUT/idle.py <<PYEOF
import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from highs_amd import solver, lp as L
lp = L.HighsLp.from_npz('tests/golden/instances/afiro.npz')
solver.solveLpCupdlp(lp)
open('$OUT/idle.ready', 'w').write('1')
time.sleep(10000)
PYEOF
python $OUT/idle.py &
IDLE=$!
while [ ! -f $OUT/idle.ready ]; do sleep 0.2; done
run with_ninth ${NINTH_ENV:-PDLP_X=0}
kill $IDLE
run default_again PDLP_X=0
