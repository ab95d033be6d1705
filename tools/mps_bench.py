#!/usr/bin/env python3
"""MPS ingest at scale: wall time of the library's multi-threaded reader against the reference's
reader on the same file (host-only measurement; no GPU involved).

    python tools/mps_bench.py [--config b|c|a] [--dir /tmp] [--threads 1,2,4,8,0]

Writes the synthetic LP of the bench config as a free-format MPS file (once; kept in --dir), reads it with
pdlp_mi355x_read_mps for every thread count, and — when integration/_build/libhighs_ref_reader.so exists (build container) —
with the reference itself (Highs_readModel: io/FilereaderMps.cpp -> HMpsFF.cpp, plus Highs::passModel), checks that
both built the same model, and prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from highs_amd import lp as L  # noqa: E402
from highs_amd import solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c", choices=["a", "b", "c"])
ap.add_argument("--dir", default="/tmp")
ap.add_argument("--threads", default="1,2,4,8,0")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

path = os.path.join(args.dir, "mps_bench_%s.mps" % args.config)
if not os.path.exists(path):
    if args.config == "c":
        from lpgen import structured_lp
        lp = structured_lp(1)
    else:
        m, n, nnz = {"a": (100_000, 100_000, 1_000_000), "b": (1_000_000, 1_000_000, 8_000_000)}[args.config]
        lp = solver.SyntheticProblem(m, n, nnz, 1).to_lp()
    t0 = time.time()
    L.write_mps(lp, path)
    print("wrote %s (%.0f MB) in %.0f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0), file=sys.stderr)

out = {"file_mb": round(os.path.getsize(path) / 1e6, 1), "host_cpus": os.cpu_count(), "native": {}}
ref_model = None
for t in [int(x) for x in args.threads.split(",")]:
    best = None
    for _ in range(args.reps):
        t0 = time.time()
        lp, info = solver.read_mps(path, t)
        wall = time.time() - t0
        best = min(best or 1e9, info["seconds"])
    out["native"]["threads=%d" % t] = {"parse_s": round(best, 3), "used": info["threads"], "MB_per_s": round(out["file_mb"] / best)}
    out.update(num_col=lp.num_col, num_row=lp.num_row, num_nz=int(lp.num_nz))
    if ref_model is None:
        ref_model = lp
    else:
        for k in ("a_start", "a_index", "a_value", "col_cost", "col_lower", "col_upper", "row_lower", "row_upper"):
            assert np.array_equal(getattr(lp, k), getattr(ref_model, k)), k

libhighs = os.path.join(ROOT, "integration", "_build", "libhighs_ref_reader.so")
if os.path.exists(libhighs):
    H = C.CDLL(libhighs)
    H.Highs_create.restype = C.c_void_p
    H.Highs_readModel.argtypes = [C.c_void_p, C.c_char_p]
    H.Highs_setBoolOptionValue.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    H.Highs_destroy.argtypes = [C.c_void_p]
    for f in ("Highs_getNumCol", "Highs_getNumRow", "Highs_getNumNz"):
        getattr(H, f).argtypes = [C.c_void_p]
    best = None
    for _ in range(max(1, args.reps - 1)):
        h = H.Highs_create()
        H.Highs_setBoolOptionValue(h, b"output_flag", 0)
        t0 = time.time()
        st = H.Highs_readModel(h, os.fsencode(path))
        best = min(best or 1e9, time.time() - t0)
        dims = (H.Highs_getNumCol(h), H.Highs_getNumRow(h), H.Highs_getNumNz(h))
        H.Highs_destroy(h)
    assert st in (0, 1) and dims == (ref_model.num_col, ref_model.num_row, int(ref_model.num_nz)), (st, dims)
    out["reference_readModel_s"] = round(best, 3)
    out["speedup_vs_reference"] = {k: round(best / v["parse_s"], 1) for k, v in out["native"].items()}
print(json.dumps(out))
