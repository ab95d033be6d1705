#!/bin/bash
# tools/make_profiles.sh ROUND — regenerates every file under profiles/ with ONE command on a GPU box
# (run through gpurun; everything is written to gpurun_out/profiles_ROUND/, then copied to profiles/ by hand or
# by `tools/make_profiles.sh ROUND collect` in the build container).
#   bench lines     : python bench.py for the headline config (default flags and the driver's --steps 20 --warmup 5),
#                     configs a / c / qp and the HiPDLP path
#   kernel stats    : rocprofv3 --kernel-trace --stats of the headline bench command (N=1)
#   PMC traffic     : FETCH_SIZE / WRITE_SIZE of the fused SpMV kernels, one counter per pass (never combined with
#                     trace domains), corrected as MI355X_MICROARCH.md prescribes: bytes = (2*FETCH + WRITE) * 1024
#   ingest / e2e    : tools/mps_bench.py on this box's host cores, tools/e2e_cli.sh (reference CLI on the drop-in, MPS -> solution)
#   test logs       : pytest -m gpu (includes the drop-in tests: reference CLI, Catch2 cases, C API client)
export PDLP_MI355X_DEV=1  # the switches below are development switches (highs_amd/csrc/pdlp_env.hpp)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
RND=${1:-r06}
if [ "${2:-}" = "collect" ]; then
  S=$R/gpurun_out/profiles_$RND
  cp $S/r*.json $S/r*.jsonl $S/r*.csv $S/r*.log $R/profiles/ 2>/dev/null
  cp $S/pmc_traffic.json $R/profiles/pmc_traffic.json
  ls -la $R/profiles
  exit 0
fi
OUT=$R/gpurun_out/profiles_$RND
mkdir -p $OUT
cd $R
# PMC traffic of the dominant kernels of every configuration first (separate --pmc passes): the bench lines below then
# carry THIS tree's counters in roofline.traffic (bench.py reads profiles/pmc_traffic.json)
bash tools/pmc_traffic.sh $RND b a c d e f qp qpn > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json
python bench.py > $OUT/${RND}_bench_1M.json 2> $OUT/bench_1M.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${RND}_bench_1M_driver_flags.json 2>> $OUT/bench_1M.err
python bench.py --config a > $OUT/${RND}_bench_100k.json 2>> $OUT/bench_1M.err
python bench.py --config c > $OUT/${RND}_bench_structured.json 2>> $OUT/bench_1M.err
python bench.py --config d > $OUT/${RND}_bench_staircase_dense_columns.json 2>> $OUT/bench_1M.err
python bench.py --config e > $OUT/${RND}_bench_heldout_tall.json 2>> $OUT/bench_1M.err
python bench.py --config f > $OUT/${RND}_bench_heldout_powerlaw_band.json 2>> $OUT/bench_1M.err
python bench.py --config qp > $OUT/${RND}_bench_qp.json 2>> $OUT/bench_1M.err
python bench.py --config qpn > $OUT/${RND}_bench_qp_sparse_hessian.json 2>> $OUT/bench_1M.err
python bench.py --solver hipdlp > $OUT/${RND}_bench_hipdlp_1M.json 2>> $OUT/bench_1M.err
for cfg in b c d e f qp; do  # per-block phase profile of the two slab launches of a trial (development buffer, 100 MHz wall clock)
  echo "== bench.py --config $cfg (PDLP_MI355X_SLAB_PROF=1)"
  PDLP_MI355X_SLAB_PROF=1 python bench.py --config $cfg --cpu-iters 0 2>&1 >/dev/null | grep "slab launch"
done > $OUT/${RND}_slab_phase_profile.log
python tools/solve_times.py > $OUT/${RND}_small_lp_times.log 2>&1
python tools/small_loop.py 25fv47 80bau3b >> $OUT/${RND}_small_lp_times.log 2>&1
( cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $R/bench.py --steps 2000 --warmup 200 --cpu-iters 0 > /dev/null 2>&1
  cp $OUT/trace/t_kernel_stats.csv $OUT/${RND}_bench_1M_kernel_stats.csv
  cp $OUT/trace/t_kernel_trace.csv $OUT/kernel_trace_1M.csv 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $OUT/trace_h -o t --output-format csv -- python $R/bench.py --solver hipdlp --cpu-iters 0 > /dev/null 2>&1
  cp $OUT/trace_h/t_kernel_stats.csv $OUT/${RND}_bench_hipdlp_1M_kernel_stats.csv
)
python - <<PY
import csv, collections, glob, json, re
# per-kernel averages over WORKING launches only: launches queued after the device halted return at once (a few hundred ns)
# and must not dilute the averages rocprofv3 --stats prints
try:
    rows = list(csv.DictReader(open("$OUT/kernel_trace_1M.csv")))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, v in per.items():
        if "pdlp" not in k or len(v) < 20:
            continue
        v.sort()
        med = v[len(v) // 2]
        w = [x for x in v if x >= 0.2 * med]
        short = re.sub(r"pdlp::\(anonymous namespace\)::", "", k)[:80]
        out[short] = {"launches": len(v), "working_launches": len(w), "avg_ns_all": sum(v) / len(v), "avg_ns_working": sum(w) / len(w)}
    json.dump(out, open("$OUT/${RND}_bench_1M_kernel_stats_working_launches.json", "w"), indent=1)
except Exception as e:
    print("kernel trace post-processing skipped:", e)
PY
# host-side ingest on this box and the end-to-end run through the reference CLI (both need integration/_build)
if [ -e $R/integration/_build/libhighs.so.1 ]; then
  python tools/mps_bench.py --config c --threads 1,8,16,32,64,0 --reps 2 > $OUT/${RND}_mps_ingest_gpu_box_host.json 2> $OUT/mps_bench.err
  bash tools/e2e_cli.sh b > $OUT/${RND}_e2e_reference_cli_1M.log 2>&1
  # Highs::run()-level time to solution, reference CLI (CPU) vs drop-in CLI (GPU), same .mps, same options
  timeout 3000 python tools/time_to_solution.py a b > $OUT/${RND}_time_to_solution.json 2> $OUT/time_to_solution.err
fi
bash tools/bringup_multi_gpu.sh 4 $OUT/bringup > $OUT/${RND}_bringup_multi_gpu.log 2>&1
cp $OUT/bringup/bringup.jsonl $OUT/${RND}_bringup_multi_gpu.jsonl 2>/dev/null
PYTEST_TIMEOUT=2400 bash tools/gpu_pytest.sh profiles_$RND/${RND}_pytest_gpu tests -m gpu -q
echo done
