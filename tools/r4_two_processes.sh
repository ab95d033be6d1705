#!/bin/bash
# two PROCESSES with grid-barrier launches on one device at the same time (the in-process gate does not reach across
# processes: the roll call / the all-or-none barrier and the fall-back have to carry this case)
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_two_processes; mkdir -p $O
for cfg in a b; do
  echo "== two processes, config $cfg"
  PDLP_MI355X_BARRIER_TIMEOUT_MS=${TMO:-1000} python tools/kbench.py --config $cfg --iters 3000 --reps 1 --kernels spmv_ax > $O/${cfg}_1.log 2>&1 &
  P1=$!
  PDLP_MI355X_BARRIER_TIMEOUT_MS=${TMO:-1000} python tools/kbench.py --config $cfg --iters 3000 --reps 1 --kernels spmv_ax > $O/${cfg}_2.log 2>&1 &
  P2=$!
  wait $P1; R1=$?; wait $P2; R2=$?
  echo "exit codes $R1 $R2"
  grep -h -E "Note:|iterate_ms_per_iter|Error|error" $O/${cfg}_1.log $O/${cfg}_2.log | cut -c1-220
done
echo "== alone, config a / b"
python tools/kbench.py --config a --iters 3000 --reps 1 --kernels spmv_ax 2>/dev/null | cut -c1-200
python tools/kbench.py --config b --iters 3000 --reps 1 --kernels spmv_ax 2>/dev/null | cut -c1-200
