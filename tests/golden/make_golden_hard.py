#!/usr/bin/env python3
"""The LPs of check/instances that stress a first-order method (build container only) — VERDICT round 4, item 4.

greenbea (2 392 x 5 405, the second-largest LP of the directory: the reference's CPU pdlp runs 1.59 M iterations in
120 s without converging), perold, gas11 (unbounded) and primal1 (a QP with a diagonal Hessian).
Goldens, from the REFERENCE's own code:
  "simplex": model status and optimal objective of the reference CLI with --solver=simplex (for primal1, a QP, the CLI runs
             the reference's QP solver) — a 1e-6 target that needs no CPU-pdlp convergence;
  "cupdlp":  the real cuPDLP-C core compiled from the reference sources (oracle/_ref) at the default tolerance with a
             time budget of REF_TIME_LIMIT seconds per instance: where it converges, iteration count and objectives; where
             it does not, the record says so (term_code) with the iterations it got through.
The reference CLI here is integration/_build/highs_reference_cli: the UNMODIFIED reference (app/RunHighs.cpp on
libhighs_reference.so.1, `make -C integration reference` — nothing of this repository in it).

    python tests/golden/make_golden_hard.py   -> tests/golden/reference_hard.json, tests/golden/instances/*.npz
"""
import json
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
from highs_amd import solver  # noqa: E402

CLI = os.environ.get("HIGHS_REF_BIN", os.path.join(ROOT, "integration", "_build", "highs_reference_cli"))
NAMES = ["perold", "gas11", "primal1", "greenbea"]
REF_TIME_LIMIT = float(os.environ.get("REF_TIME_LIMIT", "3000"))


def simplex_record(mps):
    out = subprocess.run([CLI, "--solver=simplex", mps], capture_output=True, text=True).stdout
    g = lambda pat: (re.search(pat, out) or [None, None])[1]
    return {"model_status": g(r"Model status\s*:\s*(.+)"), "objective_value": float(g(r"Objective value\s*:\s*(\S+)") or "nan"),
            "is_qp": bool(re.search(r"^QP ", out, re.M))}


def reference_lp(mps):
    """The model as the REFERENCE's reader builds it (Highs_readModel + Highs_getModel of integration/_build/
    libhighs_ref_reader.so, the unmodified reference library): the .npz inputs of the hard-instance tests do not pass
    through the product's MPS reader (round 6)."""
    import ctypes as C
    import numpy as np
    import make_golden_mps as MM
    from highs_amd import lp as L
    H = C.CDLL(MM.LIBHIGHS)
    H.Highs_create.restype = C.c_void_p
    H.Highs_readModel.argtypes = [C.c_void_p, C.c_char_p]
    H.Highs_setBoolOptionValue.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    H.Highs_destroy.argtypes = [C.c_void_p]
    for f in ("Highs_getNumCol", "Highs_getNumRow", "Highs_getNumNz", "Highs_getHessianNumNz"):
        getattr(H, f).argtypes = [C.c_void_p]
    H.Highs_getModel.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 18
    h = H.Highs_create()
    H.Highs_setBoolOptionValue(h, b"output_flag", 0)
    assert H.Highs_readModel(h, os.fsencode(mps)) in (0, 1)
    n, m, nz, qnz = H.Highs_getNumCol(h), H.Highs_getNumRow(h), H.Highs_getNumNz(h), H.Highs_getHessianNumNz(h)
    i32 = lambda k: np.zeros(max(k, 1), np.int32)
    f64 = lambda k: np.zeros(max(k, 1), np.float64)
    cost, cl, cu, rl, ru = f64(n), f64(n), f64(n), f64(m), f64(m)
    a_start, a_index, a_value = i32(n + 1), i32(nz), f64(nz)
    q_start, q_index, q_value = i32(n + 1), i32(qnz), f64(qnz)
    integrality = i32(n)
    sense, offset = C.c_int32(), C.c_double()
    nc, nr, nn, qn = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    H.Highs_getModel(h, 1, 1, C.byref(nc), C.byref(nr), C.byref(nn), C.byref(qn), C.byref(sense), C.byref(offset), p(cost), p(cl), p(cu),
                     p(rl), p(ru), p(a_start), p(a_index), p(a_value), p(q_start), p(q_index), p(q_value), p(integrality))
    H.Highs_destroy(h)
    a_start[n] = nz  # (Highs_getModel copies num_col starts, not num_col + 1)
    lp = L.HighsLp(n, m, cost[:n], cl[:n], cu[:n], rl[:m], ru[:m], a_start[:n + 1], a_index[:nz], a_value[:nz], int(sense.value),
                   float(offset.value), os.path.splitext(os.path.basename(mps))[0]).normalise()
    if qnz:
        q_start[n] = qnz
        lp.hessian = (q_start[:n + 1].copy(), q_index[:qnz].copy(), q_value[:qnz].copy())
    return lp


def main():
    import numpy as np
    out_path = os.path.join(HERE, "reference_hard.json")
    recs = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for name in NAMES:
        mps = f"{MG.REF}/check/instances/{name}.mps"
        lp = reference_lp(mps)
        mine, _ = solver.read_mps(mps)  # (the product's reader agrees, array for array: a check, not the source of the fixture)
        if int(mine.num_nz) != int(lp.num_nz):
            # gas11.mps holds 12 entries of magnitude <= 1e-9: the parser keeps them (the reference's HMpsFF does too),
            # Highs::passModel drops them (lp_data/HighsLpUtils.cpp assessMatrix, small_matrix_value = 1e-9), so Highs_getModel
            # shows the cleaned matrix.  The fixture keeps the file's entries (what round 5 pinned); the check: without the
            # tiny ones it IS the reference's model.
            keep = np.abs(mine.a_value) > 1e-9
            cnt = np.add.reduceat(keep.astype(np.int64), mine.a_start[:-1]) if mine.num_col else np.zeros(0, np.int64)
            cnt[np.diff(mine.a_start) == 0] = 0
            cleaned = (np.r_[0, np.cumsum(cnt)].astype(np.int32), mine.a_index[keep], mine.a_value[keep])
            for a, k in zip(cleaned, ("a_start", "a_index", "a_value")):
                assert np.array_equal(a, getattr(lp, k)), (name, k)
            print(name, "entries <= 1e-9 in the file:", int(mine.num_nz) - int(lp.num_nz), "(fixture keeps them)")
            lp.a_start, lp.a_index, lp.a_value = mine.a_start, mine.a_index, mine.a_value
        for k in ("a_start", "a_index", "a_value", "col_cost", "col_lower", "col_upper", "row_lower", "row_upper"):
            assert np.array_equal(getattr(lp, k), getattr(mine, k)), (name, k)
        assert (lp.hessian is None) == (mine.hessian is None)
        if lp.hessian is not None:
            # (the same lower triangle, entry order inside a column as each reader leaves it: compared in canonical form; the
            # fixture keeps the order round 5 pinned)
            import make_golden_mps as MM
            assert MM.hessian_canonical(*lp.hessian) == MM.hessian_canonical(*mine.hessian), name
            lp.hessian = mine.hessian
        lp.model_name = mine.model_name
        lp.to_npz(os.path.join(HERE, "instances", name + ".npz"))
        if os.environ.get("ONLY_INSTANCES") == "1":
            continue
        rec = {"rows": lp.num_row, "cols": lp.num_col, "nnz": int(lp.num_nz), "simplex": simplex_record(mps)}
        if not rec["simplex"]["is_qp"]:  # (the cuPDLP-C core is an LP code)
            t0 = time.time()
            rec["cupdlp"] = MG.cupdlp_record(lp, time_limit=REF_TIME_LIMIT)
            rec["cupdlp"]["seconds"] = round(time.time() - t0, 1)
            rec["cupdlp"]["time_limit"] = REF_TIME_LIMIT
        recs[name] = rec
        print(name, rec["simplex"], (rec.get("cupdlp") or {}).get("num_iter"), (rec.get("cupdlp") or {}).get("term_code"), flush=True)
        json.dump(recs, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
