#!/bin/bash
# HBM traffic of the HiPDLP step kernels (the SpMV kernels with the Halpern epilogues) on the 1M x 1M LP:
# one rocprofv3 --pmc pass per counter, each under its own timeout (never combined with trace domains).
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/pmc_hipdlp
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace -d "$OUT/$C" -o p --output-format csv -- \
      python "$R/tools/kbench.py" --solver hipdlp --reps 3 --kernels spmv_ax,spmv_aty > "$OUT/$C.log" 2>&1
  echo "$C rc=$?"
done
python - <<PY
import csv, collections, glob, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        if "k_spmv_slab" in nm:
            key = "k_spmv_slab" + nm.split("k_spmv_slab")[1].split("(")[0]
            agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {}
for (k, c), v in sorted(agg.items()):
    out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
