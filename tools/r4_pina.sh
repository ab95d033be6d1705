#!/bin/bash
# round 4: the persistent loop without its P phase (PDLP_MI355X_PRIMAL_IN_A) — parity subset, then A/B on the same box
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_pina; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bitexact.py tests/test_gpu_parity.py -q -x -m gpu \
  -k "bit_exact or trial_loop_variants or device_driven or hot_start or special_lps or dense_column or barrier_launch or concurrently" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
for v in 0 1; do
  echo "== PRIMAL_IN_A=$v"
  PDLP_MI355X_PRIMAL_IN_A=$v timeout 300 python bench.py --config a --cpu-iters 0 2>/dev/null > $O/bench_a_$v.json
  python -c "import json,sys; d=json.loads(open('$O/bench_a_$v.json').read().strip().splitlines()[-1]); print('100k', round(d['value']), round(d['ms_per_step']*1e3,2), d['trial_launches'], d['checks'])"
  PDLP_MI355X_PRIMAL_IN_A=$v PDLP_MI355X_SMALL_PROF=1 python tools/small_loop.py 25fv47 80bau3b standmps 2>&1 | grep -v amdgpu.ids
  PDLP_MI355X_PRIMAL_IN_A=$v python tools/solve_times.py 2>&1 | grep -v amdgpu.ids
done
