#!/bin/bash
# structured LP (config c): stream layout for both operands against the slab layout
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
cd "$(dirname "$0")/.."
O=gpurun_out/r03i; mkdir -p $O
for mode in auto 0; do
  if [ $mode = auto ]; then unset PDLP_MI355X_SLAB; else export PDLP_MI355X_SLAB=$mode; fi
  timeout 300 python bench.py --config c > $O/c_$mode.json 2> $O/c_$mode.err
  python - $O/c_$mode.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"]*1e3,1), d.get("trial_launches"), {k:round(v["ms"]*1e3,1) for k,v in d["roofline"]["per_kernel"].items()}, d["roofline"]["isolated_relaunch_ms"])
PY
done
