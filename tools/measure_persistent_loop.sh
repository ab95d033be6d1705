#!/bin/bash
# persistent trial loop: phases and loop times on 25fv47 / 80bau3b, 100k x 100k; the 1M headline; bit-identity of the loop variants
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
O=gpurun_out/persistent; mkdir -p $O
PDLP_MI355X_SMALL_PROF=1 python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids
python tools/small_loop.py 25fv47 80bau3b 2>&1 | grep -v amdgpu.ids
PDLP_MI355X_SMALL_PROF=1 timeout 300 python bench.py --config a 2>&1 >/dev/null | grep phases
timeout 300 python bench.py --config a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('100k', round(d['value']), round(d['ms_per_step']*1e3,2), d['trial_launches'])"
timeout 300 python bench.py --cpu-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1M', round(d['value']), round(d['ms_per_step']*1e3,2), d['trial_launches'], {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_qp.py tests/test_gpu_bitexact.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
