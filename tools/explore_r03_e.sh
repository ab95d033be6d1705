#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; O=gpurun_out/r03e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bitexact.py tests/test_gpu_qp.py tests/test_gpu_hipdlp.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 6 $O/pytest.log
python tools/solve_times.py > $O/small_fused.log 2>&1; PDLP_MI355X_FUSED=0 python tools/solve_times.py > $O/small_unfused.log 2>&1
tail -n 4 $O/small_*.log
python bench.py --cpu-iters 0 > $O/bench_b.json 2> $O/bench.err; python bench.py --config a --cpu-iters 0 > $O/bench_a.json 2>> $O/bench.err; python bench.py --config c --cpu-iters 0 > $O/bench_c.json 2>> $O/bench.err
PDLP_MI355X_FUSED=0 python bench.py --config a --cpu-iters 0 > $O/bench_a_unfused.json 2>> $O/bench.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1].split("/")[-1], round(d["value"]), "it/s", round(d["ms_per_step"]*1e3,1), "us/iter, launches", d.get("trial_launches"), "ax", round(r["other_kernels_ms"]["spmv_ax_dual"]*1e3,1), "aty", round(r["other_kernels_ms"]["spmv_aty_interact"]*1e3,1), "frac", round(r["frac"],3), r["kernel"])
PY
done
