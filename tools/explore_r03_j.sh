#!/bin/bash
# shared majorSum in the stream kernel and the persistent loop: parity, 100k x 100k both ways, small LPs with launches
cd "$(dirname "$0")/.."
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hipdlp.py -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for pm in 1 0; do
PDLP_MI355X_PERSISTENT=$pm timeout 300 python bench.py --config a > $O/a_$pm.json 2> $O/a_$pm.err
python - $O/a_$pm.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"]*1e3,2), d.get("trial_launches"), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), {k:round(v["ms"]*1e3,1) for k,v in d["roofline"]["per_kernel"].items()})
PY
done
PDLP_MI355X_PERSISTENT=0 python tools/small_loop.py 25fv47 2>&1 | grep -v amdgpu.ids
