#!/bin/bash
# where a check iteration of a small LP goes: kernel timeline of 25fv47 in the persistent loop
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03m; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ITERS=8000 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python /root/repo/tools/trace_small.py run 25fv47 > $O/run.log 2>&1
python /root/repo/tools/trace_small.py summarise $O/t_kernel_trace.csv
