#!/usr/bin/env python3
"""Golden digests of what the REFERENCE's MPS reader builds (run in the build container only).

Reads every MPS file of the reference's check/instances and of tests/golden/mps_cases/ (hand-written edge cases of
this repository) with the reference itself — Highs_readModel of the C API in integration/_build/libhighs_reference.so.1, the
UNMODIFIED reference library (`make -C integration reference`: every TU compiled from /root/reference; file reading: io/FilereaderMps.cpp -> io/HMpsFF.cpp, fixed-format fallback io/HMPSIO.cpp) — and stores
dimensions, sense, offset and sha256 digests of every array of the incumbent model in
tests/golden/reference_mps.json.  tests/test_mps_reader.py compares the library's multi-threaded reader
(pdlp_mi355x_read_mps) with these records after applying the two normalisations Highs::passModel performs on
any model it is given (|a_ij| <= 1e-9 dropped, |bound| >= 1e20 -> infinite).

    python tests/golden/make_golden_mps.py
"""
import ctypes as C
import glob
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_INSTANCES = "/root/reference/check/instances"
LIBHIGHS = os.path.join(ROOT, "integration", "_build", "libhighs_reference.so.1")  # the unmodified reference: make -C integration reference
OUT = os.path.join(HERE, "reference_mps.json")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def record_of(arrays):
    """arrays: dict of numpy arrays / scalars describing a model -> json record (shared with the test)."""
    rec = {}
    for k, v in arrays.items():
        rec[k] = digest(v) if isinstance(v, np.ndarray) else v
    return rec


def hessian_canonical(start, index, value):
    """Digest of the lower triangle as sorted (col, row, value) triplets without explicit zeros (Highs::passModel
    completes the diagonal with zeros and puts the diagonal entry first; neither is the reader's business)."""
    col = np.repeat(np.arange(len(start) - 1), np.diff(start))
    keep = value != 0.0
    col, row, val = col[keep], np.asarray(index)[keep], np.asarray(value)[keep]
    order = np.lexsort((row, col))
    return digest(col[order].astype(np.int32)) + digest(row[order].astype(np.int32)) + digest(val[order].astype(np.float64))


def read_with_reference(H, path):
    h = H.Highs_create()
    H.Highs_setBoolOptionValue(h, b"output_flag", 0)
    status = H.Highs_readModel(h, os.fsencode(path))
    rec = {"status": int(status)}
    if status != -1:
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
        H.Highs_getModel(h, 1, 1, C.byref(nc), C.byref(nr), C.byref(nn), C.byref(qn), C.byref(sense), C.byref(offset),
                         p(cost), p(cl), p(cu), p(rl), p(ru), p(a_start), p(a_index), p(a_value),
                         p(q_start), p(q_index), p(q_value), p(integrality))
        buf = C.create_string_buffer(1024)
        cn, rn = [], []
        names_ok = True
        for j in range(n):
            if H.Highs_getColName(h, j, buf) != 0:
                names_ok = False
                break
            cn.append(buf.value.decode())
        for i in range(m):
            if not names_ok or H.Highs_getRowName(h, i, buf) != 0:
                names_ok = False
                break
            rn.append(buf.value.decode())
        # (Highs_getModel copies num_col starts, not num_col + 1)
        rec.update(record_of(dict(
            num_col=n, num_row=m, num_nz=nz, sense=int(sense.value), offset=float(offset.value),
            col_cost=cost[:n], col_lower=cl[:n], col_upper=cu[:n], row_lower=rl[:m], row_upper=ru[:m],
            a_start=a_start[:n], a_index=a_index[:nz], a_value=a_value[:nz],
            integrality=integrality[:n].astype(np.uint8), hessian_num_nz=qnz,
            q_start=q_start[:n] if qnz else np.zeros(0, np.int32), q_index=q_index[:qnz], q_value=q_value[:qnz])))
        rec["hessian_canonical"] = hessian_canonical(np.r_[q_start[:n], qnz], q_index[:qnz], q_value[:qnz]) if qnz else None
        rec["names"] = hashlib.sha256("\n".join(cn + ["--"] + rn).encode()).hexdigest()[:24] if names_ok else None
    H.Highs_destroy(h)
    return rec


def main():
    H = C.CDLL(LIBHIGHS)
    H.Highs_create.restype = C.c_void_p
    for f in ("Highs_readModel", "Highs_getNumCol", "Highs_getNumRow", "Highs_getNumNz", "Highs_getHessianNumNz",
              "Highs_getModel", "Highs_getColName", "Highs_getRowName", "Highs_setBoolOptionValue", "Highs_destroy"):
        getattr(H, f).argtypes = None
    H.Highs_readModel.argtypes = [C.c_void_p, C.c_char_p]
    H.Highs_setBoolOptionValue.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    H.Highs_destroy.argtypes = [C.c_void_p]
    for f in ("Highs_getNumCol", "Highs_getNumRow", "Highs_getNumNz", "Highs_getHessianNumNz"):
        getattr(H, f).argtypes = [C.c_void_p]
    H.Highs_getColName.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
    H.Highs_getRowName.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
    H.Highs_getModel.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 18
    recs = {}
    files = sorted(glob.glob(os.path.join(REF_INSTANCES, "*.mps"))) + sorted(glob.glob(os.path.join(HERE, "mps_cases", "*.mps")))
    for f in files:
        key = ("ref/" if f.startswith(REF_INSTANCES) else "case/") + os.path.basename(f)
        recs[key] = read_with_reference(H, f)
        print(key, recs[key].get("status"), recs[key].get("num_col"), recs[key].get("num_row"), recs[key].get("num_nz"), flush=True)
    json.dump(recs, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, len(recs), "records")


if __name__ == "__main__":
    sys.exit(main())
