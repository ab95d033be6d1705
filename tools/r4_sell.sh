#!/bin/bash
# round 4: lane-per-major (sliced ELL) layout vs the slab stream — parity, then A/B per config on the same box
cd "$(dirname "$0")/.."
O=gpurun_out/r4_sell; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "lane_per_major or slab_width or layout" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log; grep -E "^(FAILED|ERROR)|Error" $O/pytest.log | head
for cfg in c d b qp; do
  for v in 0 1 -1; do
    PDLP_MI355X_SELL=$v timeout 300 python bench.py --config $cfg --cpu-iters 0 2>$O/err_${cfg}_$v.log > $O/bench_${cfg}_$v.json
    python - <<PY
import json
try:
    d=json.loads(open('$O/bench_${cfg}_$v.json').read().strip().splitlines()[-1])
    print('$cfg SELL=$v', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(x['ms']*1e3,1) for k,x in d['roofline']['per_kernel'].items()}, 'setup', round(d['setup_seconds'],3))
except Exception as e:
    print('$cfg SELL=$v failed', e, open('$O/err_${cfg}_$v.log').read()[-400:])
PY
  done
done
