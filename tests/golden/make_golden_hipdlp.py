#!/usr/bin/env python3
"""Golden outputs of the reference's HiPDLP path (solver="hipdlp"), produced by RUNNING THE REFERENCE
BINARY in the build container (it cannot travel to the GPU box):

    $HIGHS_REF_BIN --solver=hipdlp --presolve=off [--options_file=<kkt_tolerance>] --solution_file=... <mps>

Outputs: tests/golden/reference_hipdlp.json — per instance and tolerance: model status, PDLP iteration
count, the full-precision objective the solution file prints, and the solution file's primal column
values / row duals (%.15g).  The oracle (oracle/hipdlp_oracle.c) is pinned on these in
tests/test_hipdlp_oracle.py.  Instances whose CPU run exceeds the time budget are recorded as skipped."""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
# the UNMODIFIED reference binary, built from /root/reference by this repository's own recipe: make -C integration reference
HIGHS = os.environ.get("HIGHS_REF_BIN", os.path.join(ROOT, "integration", "_build", "highs_reference_cli"))
NAMES = ["afiro", "adlittle", "avgas", "blending", "chip", "sctest", "shell", "standata", "standgub", "scrs8",
         "stair", "e226", "25fv47"]
BUDGET = float(os.environ.get("HIPDLP_GOLDEN_BUDGET", "400"))


def parse_solution(path):
    txt = open(path).read().split("\n")
    out = {}
    i = 0
    section = None
    while i < len(txt):
        line = txt[i]
        if line.startswith("# Primal solution values"):
            section = "primal"
        elif line.startswith("# Dual solution values"):
            section = "dual"
        elif line.startswith("# Basis"):
            section = None
        elif line.startswith("Objective ") and section == "primal":
            out["objective"] = line.split()[1]
        elif line.startswith("# Columns") and section:
            k = int(line.split()[2])
            out[section + "_col"] = [float(t.split()[1]) for t in txt[i + 1:i + 1 + k]]
            i += k
        elif line.startswith("# Rows") and section:
            k = int(line.split()[2])
            out[section + "_row"] = [float(t.split()[1]) for t in txt[i + 1:i + 1 + k]]
            i += k
        i += 1
    return out


def run(name, kkt):
    mps = f"{REF}/check/instances/{name}.mps"
    with tempfile.TemporaryDirectory() as td:
        sol = os.path.join(td, "s.sol")
        cmd = [HIGHS, "--solver=hipdlp", "--presolve=off", f"--solution_file={sol}"]
        if kkt is not None:
            opt = os.path.join(td, "o.txt")
            open(opt, "w").write(f"kkt_tolerance = {kkt}\n")
            cmd.append(f"--options_file={opt}")
        try:
            out = subprocess.run(cmd + [mps], capture_output=True, text=True, timeout=BUDGET, cwd=td).stdout
        except subprocess.TimeoutExpired:
            return {"skipped": f"reference CPU run exceeded {BUDGET:.0f} s"}
        g = lambda pat: (re.search(pat, out) or [None, None])[1]
        rec = {"model_status": g(r"Model status\s*:\s*(.+)"), "pdlp_iterations": int(g(r"PDLP\s+iterations:\s*(\d+)") or -1)}
        if os.path.exists(sol):
            s = parse_solution(sol)
            rec.update(objective=s.get("objective"), col_value=s.get("primal_col"), row_value=s.get("primal_row"),
                       col_dual=s.get("dual_col"), row_dual=s.get("dual_row"))
        return rec


def random_lps():
    """Small random LPs with every row / column kind (tests/lpgen.py), written as MPS for the reference binary:
    tests/golden/reference_hipdlp_random.json = {seed: {model_status, pdlp_iterations, objective}} at
    kkt_tolerance 1e-6 (minimisation seeds), and at an iteration limit of 400 for maximisation seeds (the
    reference does not apply the objective sense on this path)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import lpgen
    from highs_amd import lp as L
    out = {}
    for seed in list(range(0, 40, 2)) + [1, 3]:
        lp = lpgen.drop_free_rows(lpgen.random_lp(seed))
        with tempfile.TemporaryDirectory() as td:
            mps, sol, opt = (os.path.join(td, f) for f in ("a.mps", "s.sol", "o.txt"))
            L.write_mps(lp, mps)
            limit = 20000 if seed % 2 == 0 else 400
            open(opt, "w").write(f"kkt_tolerance = 1e-6\npdlp_iteration_limit = {limit}\n")
            txt = subprocess.run([HIGHS, "--solver=hipdlp", "--presolve=off", f"--options_file={opt}",
                                  f"--solution_file={sol}", mps], capture_output=True, text=True, timeout=BUDGET, cwd=td).stdout
            g = lambda pat: (re.search(pat, txt) or [None, None])[1]
            rec = {"model_status": g(r"Model status\s*:\s*(.+)"), "pdlp_iterations": int(g(r"PDLP\s+iterations:\s*(\d+)") or -1),
                   "iteration_limit": limit}
            if os.path.exists(sol):
                rec["objective"] = parse_solution(sol).get("objective")
                rec["col_value"] = parse_solution(sol).get("primal_col")
        out[str(seed)] = rec
        print("random", seed, rec["model_status"], rec["pdlp_iterations"], rec.get("objective"), flush=True)
    json.dump(out, open(os.path.join(HERE, "reference_hipdlp_random.json"), "w"), indent=0, sort_keys=True)


def main():
    if "--random-only" in sys.argv:
        return random_lps()
    recs = {}
    for name in NAMES:
        recs[name] = {"default": run(name, None)}
        if name in ("afiro", "adlittle", "shell", "sctest"):
            recs[name]["kkt1e-4"] = run(name, 1e-4)  # the tolerance check/TestPdlpHi.cpp uses
        print(name, {k: (v.get("pdlp_iterations"), v.get("objective"), v.get("skipped")) for k, v in recs[name].items()},
              flush=True)
        json.dump(recs, open(os.path.join(HERE, "reference_hipdlp.json"), "w"), indent=0, sort_keys=True)
    random_lps()


if __name__ == "__main__":
    main()
