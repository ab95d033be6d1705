#!/usr/bin/env python3
"""Kernel timeline of a small LP (development tool): run under rocprofv3 --kernel-trace, then summarise
durations and inter-kernel gaps:  rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- python tools/trace_small.py run
                                   python tools/trace_small.py summarise OUT/t_kernel_trace.csv"""
import csv
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    from highs_amd import lp as L
    from highs_amd import solver
    name = sys.argv[2] if len(sys.argv) > 2 else "25fv47"
    lp = L.HighsLp.from_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "instances", name + ".npz"))
    o = solver.solveLpCupdlp(lp, pdlp_iteration_limit=int(os.environ.get("ITERS", "4000")))
    print(o.pdlp_iteration_count, o.result.solve_seconds)
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]  # steady part
    import collections
    dur = collections.defaultdict(list)
    gap = collections.defaultdict(list)
    for a, b in zip(rows, rows[1:]):
        mm = re.search(r"(k_\w+(<[^>]*>)?)", a["Kernel_Name"])
        nm = mm.group(1) if mm else a["Kernel_Name"][:40]
        dur[nm].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        gap[nm].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    tot = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print("span %.1f us, %d kernels" % (tot / 1e3, len(rows)))
    for nm in sorted(dur, key=lambda k: -sum(dur[k]) - sum(gap[k])):
        d, g = dur[nm], gap[nm]
        print("%-50s n=%5d dur avg %6.2f us  gap-after avg %6.2f us  share %4.1f%%" % (
            nm[:50], len(d), sum(d) / len(d) / 1e3, sum(g) / len(g) / 1e3, 100.0 * (sum(d) + sum(g)) / tot))

