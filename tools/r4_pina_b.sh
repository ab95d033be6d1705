#!/bin/bash
# quick timing of the small-LP loop (development)
cd "$(dirname "$0")/.."
PDLP_MI355X_SMALL_PROF=1 python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids
PDLP_MI355X_SMALL_PROF=1 python tools/small_loop.py 80bau3b 2>&1 | grep -v amdgpu.ids
python tools/small_loop.py standmps afiro 2>&1 | grep -v amdgpu.ids
python bench.py --config a --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('100k', d['value'], d['ms_per_step']*1e3)"
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "trial_loop_variants" 2>&1 | tail -2
python bench.py --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('1M', d['value'], d['ms_per_step']*1e3, {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"
python bench.py --config c --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c', d['value'], d['ms_per_step']*1e3)"
