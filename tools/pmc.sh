#!/bin/bash
# tools/pmc.sh OUTDIR "CTR CTR ..." ["CTR ..."] ... -- one rocprofv3 --pmc pass per group (never combined
# with other trace domains), each under its own timeout, on tools/kbench.py kernels.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
KERNELS=${KERNELS:-spmv_ax_plain,spmv_ax}
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
i=0
for G in "$@"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $G --kernel-trace -d "$OUT/g$i" -o p --output-format csv -- python "$R/tools/kbench.py" --reps 3 --iters ${ITERS:-0} ${KBENCH_ARGS:-} --kernels $KERNELS > "$OUT/g$i.log" 2>&1
  echo "group $i ($G) rc=$?"
done
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        if "k_spmv" in nm or "k_primal" in nm:
            key = nm.split("(")[0].replace("void ", "").replace("pdlp::(anonymous namespace)::", "")
            agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {}
for (k, c), v in sorted(agg.items()):
    out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
