#!/usr/bin/env python3
"""Differential fuzzing of the MPS reader against the reference's (build container only).

Random small MPS texts — every section, duplicate / undefined / keyword-like names, zero and repeated coefficients,
MARKER blocks, RHS / BOUNDS lines with and without set names, RANGES of both signs, every bound type, columns first met
in BOUNDS, OBJSENSE in its three spellings, comments, blank lines, tabs, CRLF, odd number formats — are read by
pdlp_mi355x_read_mps (random piece count) and by Highs_readModel of integration/_build/libhighs_ref_reader.so, and the
two models are compared exactly as tests/test_mps_reader.py compares them.  A disagreement is written to
tests/golden/mps_cases/fuzz_<seed>.mps so that it becomes a regression case (after `make_golden_mps.py`).

    python tools/mps_fuzz.py [--cases 2000] [--seed 1]
"""
import argparse
import ctypes as C
import os
import random
import json
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_mps as MG  # noqa: E402
import test_mps_reader as TR  # noqa: E402
from highs_amd import solver  # noqa: E402

NUMS = ["1", "-1", "2.5", "-0.75", "0", "0.0", "1e3", "1.5D2", "2d-1", "+3", ".5", "5.", "-0", "1e-12", "7E+0", "1e19", "-1e19",
        "123456.789", "3.0000000001"]


def ws(r):
    return r.choice([" ", "  ", "\t", "   ", " \t "])


def gen(r):
    nrow, ncol = r.randint(1, 7), r.randint(1, 6)
    keywordish = ["RHS", "BOUNDS", "RANGES", "ROWS", "COLUMNS", "NAME", "MAX", "ENDATA", "OBJSENSE"]
    rows = [r.choice(["r%d" % i, "R%d" % i, "row_%d" % i, "c%d" % i]) for i in range(nrow)]
    if r.random() < 0.15:
        rows[r.randrange(nrow)] = r.choice(keywordish[:5])
    if nrow > 1 and r.random() < 0.15:
        rows[-1] = rows[0]  # duplicate row name
    cols = ["x%d" % j for j in range(ncol)]
    if r.random() < 0.15:
        cols[r.randrange(ncol)] = r.choice(keywordish[:5])
    name = r.choice(["m", "MODEL", "fz"])
    out = []
    eol = r.choice(["\n", "\n", "\r\n"])
    if r.random() < 0.3:
        out.append("* a comment")
    out.append("NAME" + ws(r) + name + (ws(r) + "extra" if r.random() < 0.2 else ""))
    style = r.random()
    if style < 0.2:
        out += ["OBJSENSE", ws(r) + r.choice(["MAX", "MIN", "MAXIMIZE", "max"])]
    elif style < 0.35:
        out.append("OBJSENSE" + ws(r) + r.choice(["MAX", "MIN", "max"]))
    out.append("ROWS")
    obj = r.choice(["obj", "COST", "z"])
    row_lines = [(" N", obj)] if r.random() < 0.92 else []
    types = {}
    for rn in rows:
        t = r.choice("LGE")
        types.setdefault(rn, t)
        row_lines.append((" " + t, rn))
    if r.random() < 0.3:
        row_lines.insert(r.randrange(len(row_lines) + 1), (" N", "free%d" % r.randint(0, 2)))
    r.shuffle(row_lines) if r.random() < 0.3 else None
    for t, rn in row_lines:
        out.append(t + ws(r) + rn)
        if r.random() < 0.08:
            out.append("")
    out.append("COLUMNS")
    allrows = rows + [obj, "nosuchrow", "free0"]
    integral = False
    order = list(cols)
    if r.random() < 0.15:
        order.append(order[0])  # the column reappears later: a new column
    for cn in order:
        if r.random() < 0.2:
            integral = not integral
            out.append(" MK" + ws(r) + "'MARKER'" + ws(r) + ("'INTORG'" if integral else "'INTEND'"))
        for _ in range(r.randint(1, 5)):
            ln = ws(r) + cn + ws(r) + r.choice(allrows) + ws(r) + r.choice(NUMS)
            if r.random() < 0.4:
                ln += ws(r) + r.choice(allrows) + ws(r) + r.choice(NUMS)
            out.append(ln)
        if r.random() < 0.1:
            out.append("* c")
    if integral:
        out.append(" MK 'MARKER' 'INTEND'")
    if r.random() < 0.9:
        out.append("RHS")
        for _ in range(r.randint(0, nrow + 1)):
            setname = r.choice(["rhs", "RHS", "", name, "B"])
            ln = ws(r) + (setname + ws(r) if setname else "") + r.choice(rows + [obj, "nosuchrow"]) + ws(r) + r.choice(NUMS)
            if r.random() < 0.3:
                ln += ws(r) + r.choice(rows + [obj]) + ws(r) + r.choice(NUMS)
            out.append(ln)
    if r.random() < 0.5:
        out.append("RANGES")
        for _ in range(r.randint(0, nrow)):
            ln = ws(r) + "rng" + ws(r) + r.choice(rows + [obj, "nosuchrow", "free0"]) + ws(r) + r.choice(NUMS)
            if r.random() < 0.3:
                ln += ws(r) + r.choice(rows) + ws(r) + r.choice(NUMS)
            out.append(ln)
    if r.random() < 0.8:
        out.append("BOUNDS")
        for _ in range(r.randint(0, ncol + 2)):
            bt = r.choice(["UP", "LO", "FX", "MI", "PL", "BV", "LI", "UI", "FR", "SC", "SI"])
            cn = r.choice(cols + (["newcol"] if r.random() < 0.1 else []))
            setname = r.choice(["bnd", "BND", "", "BOUND"])
            ln = " " + bt + ws(r) + (setname + ws(r) if setname else "") + cn
            if bt not in ("MI", "PL", "BV", "FR") or r.random() < 0.2:
                ln += ws(r) + r.choice(NUMS)
            out.append(ln)
    if r.random() < 0.15:
        out.append(r.choice(["QUADOBJ", "QMATRIX"]))
        for cn in cols[: r.randint(1, ncol)]:
            out.append(ws(r) + cn + ws(r) + cn + ws(r) + r.choice(["1", "2.5", "0", "4"]))
    out.append("ENDATA")
    if r.random() < 0.2:
        out.append("trailing junk")
    text = eol.join(out)
    if r.random() < 0.8:
        text += eol
    return text


REF_WORKER = r"""
import ctypes as C, json, sys
sys.path.insert(0, sys.argv[1])
import make_golden_mps as MG
H = C.CDLL(MG.LIBHIGHS)
H.Highs_create.restype = C.c_void_p
H.Highs_readModel.argtypes = [C.c_void_p, C.c_char_p]
H.Highs_setBoolOptionValue.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
H.Highs_destroy.argtypes = [C.c_void_p]
for f in ("Highs_getNumCol", "Highs_getNumRow", "Highs_getNumNz", "Highs_getHessianNumNz"):
    getattr(H, f).argtypes = [C.c_void_p]
H.Highs_getColName.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
H.Highs_getRowName.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
H.Highs_getModel.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 18
out = {}
for path in sys.argv[2:]:
    out[path] = MG.read_with_reference(H, path)
    print(json.dumps({path: out[path]}), flush=True)
"""


def reference_records(paths, timeout=30):
    """What the reference builds for each file, computed in a child process: the reference itself can hang on a
    fuzzed file (seed 1000043 does), such files are skipped."""
    recs = {}
    todo = list(paths)
    while todo:
        try:
            p = subprocess.run([sys.executable, "-c", REF_WORKER, os.path.join(ROOT, "tests", "golden")] + todo,
                               capture_output=True, text=True, timeout=timeout)
            lines = p.stdout.splitlines()
        except subprocess.TimeoutExpired as e:
            lines = (e.stdout or b"").decode().splitlines() if isinstance(e.stdout, bytes) else (e.stdout or "").splitlines()
        for ln in lines:
            recs.update(json.loads(ln))
        done = [q for q in todo if q in recs]
        rest = [q for q in todo if q not in recs]
        if rest and len(done) < len(todo):
            print("reference did not finish", rest[0], "- skipped", flush=True)
            recs[rest[0]] = None
            rest = rest[1:]
        todo = rest
    return recs


PASS_WORKER = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import make_golden_mps as MG
from highs_amd import solver
try:
    lp, info = solver.read_mps(sys.argv[3], 2)
except Exception as e:
    print("native-refuses"); sys.exit(0)
H = C.CDLL(MG.LIBHIGHS)
H.Highs_create.restype = C.c_void_p
H.Highs_setBoolOptionValue.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
H.Highs_passMip.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double] + [C.c_void_p] * 9
h = H.Highs_create()
H.Highs_setBoolOptionValue(h, b"output_flag", 0)
integ = (info["integrality"] if info["integrality"] is not None else np.zeros(lp.num_col, np.uint8)).astype(np.int32)
p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
arrs = [np.ascontiguousarray(a) for a in (lp.col_cost, lp.col_lower, lp.col_upper, lp.row_lower, lp.row_upper,
                                          lp.a_start.astype(np.int32), lp.a_index.astype(np.int32), lp.a_value, integ)]
st = H.Highs_passMip(h, lp.num_col, lp.num_row, int(lp.num_nz), 1, int(lp.sense), float(lp.offset), *[p(a) for a in arrs])
print("pass-model-status", st)
"""


def rejected_by_pass_model(path):
    """Highs_readModel refused the file.  Either the reference's parser did — then this reader must refuse it too — or
    Highs::passModel refused the model the parser built: then it must also refuse the model THIS reader builds."""
    p = subprocess.run([sys.executable, "-c", PASS_WORKER, os.path.join(ROOT, "tests", "golden"), ROOT, path],
                       capture_output=True, text=True, timeout=30)
    out = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "crash: " + p.stderr[-300:]
    assert out in ("native-refuses", "pass-model-status -1"), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad = 0
    tmp = tempfile.mkdtemp()
    batch = 50
    for k0 in range(0, args.cases, batch):
        files = {}
        for k in range(k0, min(args.cases, k0 + batch)):
            seed = args.seed * 1000003 + k
            path = os.path.join(tmp, "f%d.mps" % seed)
            text = gen(random.Random(seed))
            if seed % 10 == 3:  # a gzip stream now and then: both readers inflate it (zstr in the reference)
                import gzip
                open(path, "wb").write(gzip.compress(text.encode()))
            else:
                open(path, "w", newline="").write(text)
            files[path] = seed
        recs = reference_records(list(files))
        for path, seed in files.items():
            ref = recs.get(path)
            if ref is None:
                continue
            key = "fuzz/%d" % seed
            TR.REF[key] = ref
            try:
                if ref["status"] == -1:
                    rejected_by_pass_model(path)
                else:
                    TR.check_against_reference(path, key, random.Random(seed).choice([1, 2, 3, 5, 9, 17, 64]))
            except BaseException as e:  # noqa: BLE001 - any disagreement is a finding
                bad += 1
                keep = os.path.join(ROOT, "tests", "golden", "mps_cases", "fuzz_%d.mps" % seed)
                open(keep, "wb").write(open(path, "rb").read())
                print("MISMATCH seed", seed, type(e).__name__, str(e)[:200], "->", keep, flush=True)
            os.remove(path)
        if bad >= 10:
            break
    print("cases", min(args.cases, k0 + batch), "mismatches", bad)


if __name__ == "__main__":
    main()
