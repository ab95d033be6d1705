#!/usr/bin/env python3
"""tools/time_to_solution.py [a] [b] — Highs::run()-level time to solution, MPS file to solution, CPU reference vs drop-in
(VERDICT round 5, item 4): the reference's UNMODIFIED CLI (app/RunHighs.cpp) runs twice on the same .mps file with the same
options (solver=pdlp, presolve=off, kkt_tolerance) —

    integration/_build/highs_reference_cli   libhighs_reference.so.1: every TU the reference's, CPU cuPDLP-C, its MPS reader
    integration/_build/highs_ref_cli         libhighs.so.1: the PDLP wrapper TUs and the MPS reader front end replaced by
                                             this repository's (pdlp on the MI355X, multi-threaded reader)

— and reports process wall clock, the CLI's own "HiGHS run time", the MPS read time (wall minus run time), iterations,
status and objective of both.  Config a (100k x 100k) and b (1M x 1M) at kkt_tolerance 1e-4; b at the default 1e-7 on the
GPU only, with the CPU time EXTRAPOLATED from its measured iterations/s at 1e-4 and the GPU's iteration count (stated in
the record: the CPU run itself would take an hour).  One JSON line per run; run on the GPU box (both binaries travel
with the snapshot)."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from highs_amd import lp as L  # noqa: E402
from highs_amd import solver  # noqa: E402

BUILD = os.path.join(ROOT, "integration", "_build")
DIMS = {"a": (100_000, 100_000, 1_000_000), "b": (1_000_000, 1_000_000, 8_000_000)}


def mps_of(cfg):
    path = "/tmp/mps_bench_%s.mps" % cfg
    if not os.path.exists(path):
        L.write_mps(solver.SyntheticProblem(*DIMS[cfg], 1).to_lp(), path)
    return path


def run_cli(binary, mps, tol, time_limit):
    opt = "/tmp/tts_options_%g.txt" % tol
    open(opt, "w").write("kkt_tolerance = %g\ntime_limit = %g\n" % (tol, time_limit))
    env = dict(os.environ, LD_LIBRARY_PATH=BUILD + ":" + os.path.join(ROOT, "highs_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    t0 = time.time()
    out = subprocess.run([os.path.join(BUILD, binary), "--solver=pdlp", "--presolve=off", "--options_file=" + opt, mps],
                         capture_output=True, text=True, env=env, cwd="/tmp", timeout=time_limit + 600).stdout
    wall = time.time() - t0
    g = lambda pat, conv=str: (lambda m_: conv(m_.group(1)) if m_ else None)(re.search(pat, out))
    return {"wall_s": round(wall, 3), "highs_run_time_s": g(r"HiGHS run time\s*:\s*(\S+)", float),
            "model_status": g(r"Model status\s*:\s*(.+)"), "pdlp_iterations": g(r"PDLP\s+iterations:\s*(\d+)", int),
            "objective": g(r"Objective value\s*:\s*(\S+)", float)}


def main():
    for cfg in [a for a in sys.argv[1:] if a in DIMS] or ["a", "b"]:
        mps = mps_of(cfg)
        base = {"config": cfg, "m": DIMS[cfg][0], "n": DIMS[cfg][1], "nnz_target": DIMS[cfg][2], "mps_mb": round(os.path.getsize(mps) / 1e6, 1),
                "host_cpus": os.cpu_count()}
        gpu = run_cli("highs_ref_cli", mps, 1e-4, 3600)
        gpu["read_and_setup_outside_run_s"] = round(gpu["wall_s"] - (gpu["highs_run_time_s"] or 0.0), 3)
        print(json.dumps(dict(base, kkt_tolerance=1e-4, side="drop-in (MI355X)", **gpu)), flush=True)
        cpu = None
        if os.path.exists(os.path.join(BUILD, "highs_reference_cli")):
            cpu = run_cli("highs_reference_cli", mps, 1e-4, 3600)
            cpu["read_and_setup_outside_run_s"] = round(cpu["wall_s"] - (cpu["highs_run_time_s"] or 0.0), 3)
            print(json.dumps(dict(base, kkt_tolerance=1e-4, side="reference (CPU pdlp, 1 thread)", **cpu,
                                  speedup_wall=round(cpu["wall_s"] / gpu["wall_s"], 1))), flush=True)
        if cfg == "b":
            g7 = run_cli("highs_ref_cli", mps, 1e-7, 3600)
            rec = dict(base, kkt_tolerance=1e-7, side="drop-in (MI355X)", **g7)
            if cpu and cpu["pdlp_iterations"] and cpu["highs_run_time_s"]:
                its = cpu["pdlp_iterations"] / max(cpu["highs_run_time_s"], 1e-9)  # (includes the CPU's set-up: a lower bound of its rate)
                rec["cpu_extrapolated_s"] = round(g7["pdlp_iterations"] / its + cpu["read_and_setup_outside_run_s"], 0)
                rec["cpu_extrapolation"] = ("GPU iteration count at 1e-7 / the reference's measured iterations per second of run time at 1e-4 "
                                            "(%.1f it/s, set-up included) + its read time: NOT a measurement of the CPU at 1e-7" % its)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
