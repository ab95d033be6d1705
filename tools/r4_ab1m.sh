#!/bin/bash
# quick check of bench lines (development)
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py -q -x -m gpu -k "bit_exact or trial_loop_variants or fused or device_driven or synthetic or check_interval" 2>&1 | tail -2
for cfg in b a; do for i in 1 2; do python bench.py --config $cfg --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; done; done
PDLP_MI355X_SMALL_PROF=1 python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids
python tools/small_loop.py 80bau3b afiro 2>&1 | grep -v amdgpu.ids
