#!/bin/bash
# round 4, device-driven checks: parity subset + A/B of the host-driven and the device-driven loop on the same box
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_devcheck; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bitexact.py tests/test_gpu_parity.py -q -x -m gpu \
  -k "bit_exact or device_driven or iteration_limit or time_limit or nan_in or hot_start or infeasible_and or special_lps or trial_loop_variants" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
for dc in 0 1; do
  echo "== DEVICE_CHECK=$dc"
  PDLP_MI355X_DEVICE_CHECK=$dc timeout 300 python bench.py --cpu-iters 0 2>/dev/null > $O/bench_b_dc$dc.json
  python -c "import json,sys; d=json.loads(open('$O/bench_b_dc$dc.json').read().strip().splitlines()[-1]); print('1M', round(d['value']), round(d['ms_per_step']*1e3,2), d['trial_launches'], d['checks'], {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"
  PDLP_MI355X_DEVICE_CHECK=$dc timeout 300 python bench.py --config a --cpu-iters 0 2>/dev/null > $O/bench_a_dc$dc.json
  python -c "import json,sys; d=json.loads(open('$O/bench_a_dc$dc.json').read().strip().splitlines()[-1]); print('100k', round(d['value']), round(d['ms_per_step']*1e3,2), d['trial_launches'], d['checks'])"
  PDLP_MI355X_DEVICE_CHECK=$dc timeout 300 python bench.py --config c --cpu-iters 0 2>/dev/null > $O/bench_c_dc$dc.json
  python -c "import json,sys; d=json.loads(open('$O/bench_c_dc$dc.json').read().strip().splitlines()[-1]); print('struct', round(d['value']), round(d['ms_per_step']*1e3,2), d['trial_launches'], d['checks'])"
  PDLP_MI355X_DEVICE_CHECK=$dc python tools/solve_times.py 2>&1 | grep -v amdgpu.ids
done
