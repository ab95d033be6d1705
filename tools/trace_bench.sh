#!/bin/bash
# rocprofv3 kernel trace of the bench command (N=1): per-kernel average durations and gaps in the steady loop.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mkdir -p "$1" && cd "$1" && pwd); shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o t --output-format csv -- python "$R/bench.py" --gpus 1 --steps ${STEPS:-400} --warmup 40 --cpu-iters 0 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python "$R/tools/trace_small.py" summarise "$OUT/t_kernel_trace.csv"
