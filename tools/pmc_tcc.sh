#!/bin/bash
# tools/pmc_tcc.sh OUTDIR "ENV=val ENV=val" ... -- L2 hit/miss of the plain A x SpMV per variant (development tool)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mkdir -p "$1" && cd "$1" && pwd); shift
cd /tmp; export TMPDIR=/tmp
i=0
for V in "$@"; do
  i=$((i+1))
  env $V timeout 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d "$OUT/v$i" -o p --output-format csv -- python "$R/tools/kbench.py" --reps 3 --kernels ${KERNELS:-spmv_ax_plain} > "$OUT/v$i.log" 2>&1
  python - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/v$i/p_counter_collection.csv")):
    if "k_spmv" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for r in csv.DictReader(open("$OUT/v$i/p_kernel_trace.csv")):
    if "k_spmv" in r["Kernel_Name"]:
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$V", {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())}, "us", round(sum(dur) / max(len(dur), 1), 1))
PY
done
