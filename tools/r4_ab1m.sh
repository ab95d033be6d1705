#!/bin/bash
# quick check of bench lines (development)
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_qp.py tests/test_gpu_parity.py -q -x -m gpu -k "qp or fused or trial_loop_variants" 2>&1 | tail -2
for cfg in qp b; do for i in 1 2; do python bench.py --config $cfg --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,1) for k,v in d['roofline']['per_kernel'].items()})"; done; done
