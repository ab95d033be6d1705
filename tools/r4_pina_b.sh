#!/bin/bash
# quick timing of the small / mid-size persistent loop (development)
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
for i in 1 2; do python bench.py --config a --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('100k', d['value'], d['ms_per_step']*1e3)"; done
PDLP_MI355X_SMALL_PROF=1 python bench.py --config a --cpu-iters 0 2>&1 | grep "small-LP phases"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py -q -m gpu -k "trial_loop_variants or synthetic or structured or two_large" 2>&1 | tail -2
