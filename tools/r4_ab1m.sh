#!/bin/bash
# quick check of the bench lines and the per-block phase profile (development)
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py tests/test_gpu_hipdlp.py -q -x -m gpu -k "bit_exact or trial_loop_variants or fused or synthetic or structured or dense_column or device_driven or slab or long" 2>&1 | tail -3
for cfg in c d b; do for i in 1 2; do python bench.py --config $cfg --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; done; done
for c in c d; do PDLP_MI355X_SLAB_PROF=1 python bench.py --config $c --cpu-iters 0 2>&1 | grep -E "slab launch"; done
