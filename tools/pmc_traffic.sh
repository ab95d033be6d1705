#!/bin/bash
# tools/pmc_traffic.sh ROUND [configs...] — HBM-side bytes per launch of the dominant kernels of EVERY bench configuration
# (default: b a c d qp qpn on the pdlp path, b on the hipdlp path): FETCH_SIZE / WRITE_SIZE / TCC hits and misses with
# rocprofv3 --pmc, one counter group per pass, never combined with trace domains other than --kernel-trace.
# Result: gpurun_out/profiles_ROUND/pmc_traffic.json (copy to profiles/pmc_traffic.json: bench.py's roofline.traffic).
#   bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE counts coalesced streams at half their
#   size, MI355X_MICROARCH.md); TCC_MISS * 128 B is the cross-check.  Launches queued behind a device halt return at once:
#   only launches with at least a fifth of the median counter value enter the means.  The persistent trial loop (config a)
#   is ONE launch per check period: its bytes are divided by the trials it ran (the per-trial traffic of the loop).
set -u
R=$(cd "$(dirname "$0")/.." && pwd); RND=${1:-r05}; shift
CONFIGS=${*:-b a c d qp qpn}
OUT=$R/gpurun_out/profiles_$RND; mkdir -p $OUT
( cd /tmp; export TMPDIR=/tmp
  for CFG in $CONFIGS hipdlp; do
    if [ $CFG = hipdlp ]; then ARGS="--config b --solver hipdlp"; else ARGS="--config $CFG"; fi
    for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      D=$OUT/pmc_${CFG}_$(echo $C | tr ' ' '_')
      rm -rf $D
      timeout 200 rocprofv3 --pmc $C --kernel-trace -d $D -o p --output-format csv -- python $R/tools/kbench.py $ARGS --iters 160 --reps 3 --kernels spmv_ax,spmv_aty > $D.log 2>&1
    done
  done )
python - <<PY
import csv, collections, glob, json, os, re
OUT = "$OUT"
names = {1: "spmv_ax_dual", 2: "spmv_aty_interact", 4: "spmv_aty_halpern_primal", 5: "spmv_ax_halpern_dual",
         6: "spmv_aty_interact_decide_primal", 7: "spmv_qx_interact"}
res, raw = {}, {}
for d in sorted(glob.glob(OUT + "/pmc_*_FETCH_SIZE")):
    cfg = re.match(r"pmc_(.+)_FETCH_SIZE", os.path.basename(d)).group(1)
    agg = collections.defaultdict(list)
    for f in glob.glob(OUT + "/pmc_%s_*/p_counter_collection.csv" % cfg):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"k_spmv(?:_slab)?<\(?(?:pdlp::\(anonymous namespace\)::Epilogue\)?)?(\d)", k)
            if m and int(m.group(1)) in names:
                key = names[int(m.group(1))]
            elif "k_trials_small" in k:
                key = "trials_persistent"
            elif "k_decide_primal" in k:
                key = "decide_primal"
            else:
                continue
            agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
    trials = None
    for lg in glob.glob(OUT + "/pmc_%s_FETCH_SIZE.log" % cfg):
        for ln in open(lg):
            if ln.startswith("{"):
                try:
                    trials = json.loads(ln).get("trials")
                except Exception:
                    pass
    rw = {}
    for (key, c), v in sorted(agg.items()):
        v.sort()
        med = v[len(v) // 2]
        w = [x for x in v if x >= 0.2 * med] or v
        rw.setdefault(key, {})[c] = {"mean_working": sum(w) / len(w), "launches": len(v), "working": len(w), "sum": sum(v)}
    tr = {}
    for key, d2 in rw.items():
        if "FETCH_SIZE" in d2 and "WRITE_SIZE" in d2:
            if key == "trials_persistent" and trials:  # one launch per check period: per trial of the loop
                tr["trials_persistent_per_trial"] = (2 * d2["FETCH_SIZE"]["sum"] + d2["WRITE_SIZE"]["sum"]) * 1024 / trials
            else:
                tr[key] = (2 * d2["FETCH_SIZE"]["mean_working"] + d2["WRITE_SIZE"]["mean_working"]) * 1024
    if cfg == "hipdlp":
        res.setdefault("b", {}).update({k: v for k, v in tr.items() if "halpern" in k})
        raw["hipdlp_b"] = rw
    else:
        res.setdefault(cfg, {}).update(tr)
        raw[cfg] = rw
res["raw_per_launch"] = raw
res["note"] = ("HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 --pmc (one counter group per pass, "
               "tools/pmc_traffic.sh: the kernels inside 160 PDHG iterations of each bench configuration plus three isolated launches); on "
               "gfx950 FETCH_SIZE counts coalesced streams at half their size (MI355X_MICROARCH.md), so it is doubled; TCC_MISS * 128 B is "
               "the cross-check; means over working launches (launches queued behind a device halt return at once)")
json.dump(res, open(OUT + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("raw_per_launch", "note")}, indent=1))
PY
