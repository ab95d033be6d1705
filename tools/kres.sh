#!/bin/bash
# tools/kres.sh FILE.hip [filter] — register / scratch / occupancy table of the kernels in one translation unit
cd "$(dirname "$0")/../highs_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -x hip -c $1 -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re,sys,subprocess
cur=None;rows=[]
for ln in sys.stdin:
    m=re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/\w+\])?(?: \[waves/SIMD\])?: (\S+)",ln)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name":
        cur={"name":subprocess.run(["c++filt",v],capture_output=True,text=True).stdout.strip()[:90]};rows.append(cur)
    elif cur is not None: cur[k]=v
f=sys.argv[1] if len(sys.argv)>1 else ""
for r in rows:
    if f in r["name"]: print("%-92s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s lds %s"%(r["name"],r.get("VGPRs"),r.get("AGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize"),r.get("Occupancy"),r.get("LDS Size")))
' "${2:-}"
