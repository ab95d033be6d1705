#!/bin/bash
# persistent loop at 100k x 100k with cached gathers behind an agent-scope acquire at each barrier
cd "$(dirname "$0")/.."
O=gpurun_out/r03j; mkdir -p $O
PDLP_MI355X_SMALL_PROF=1 timeout 300 python bench.py --config a 2>&1 >/dev/null | grep phases
timeout 300 python bench.py --config a > $O/a_1.json 2> $O/a_1.err
python - $O/a_1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"]*1e3,2), d.get("trial_launches"))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "100k or bit_exact or synthetic" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
