"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when built, the
real cuPDLP-C core (oracle/_ref/libpdlp_ref.so).  TEST INFRASTRUCTURE: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

from highs_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class Trace(C.Structure):
    _fields_ = [("iter", C.c_int), ("trials", C.c_int)] + [
        (k, C.c_double) for k in ("beta", "primal_step", "dual_step", "primal_obj", "dual_obj", "primal_feas",
                                  "dual_feas", "primal_obj_avg", "dual_obj_avg", "primal_feas_avg", "dual_feas_avg")]


TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Trace))


class Formulated(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("n_eqs", C.c_int), ("nnz", C.c_long),
                ("csc_beg", abi.c_i32p), ("csc_idx", abi.c_i32p), ("csc_val", abi.c_f64p),
                ("csr_beg", abi.c_i32p), ("csr_idx", abi.c_i32p), ("csr_val", abi.c_f64p),
                ("cost", abi.c_f64p), ("rhs", abi.c_f64p), ("lower", abi.c_f64p), ("upper", abi.c_f64p),
                ("col_scale", abi.c_f64p), ("row_scale", abi.c_f64p),
                ("row_type", abi.c_i32p), ("row_new_idx", abi.c_i32p),
                ("norm_cost", C.c_double), ("norm_rhs", C.c_double), ("mat_norm_inf", C.c_double)]


_oracle = None
_ref = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "pdlp_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build_oracle()
        lib = C.CDLL(path)
        lib.pdlp_oracle_solve.argtypes = [C.POINTER(abi.PdlpProblem), C.POINTER(abi.PdlpParams), C.POINTER(abi.PdlpResult)]
        lib.pdlp_oracle_solve.restype = C.c_int
        lib.pdlp_oracle_solve_traced.argtypes = lib.pdlp_oracle_solve.argtypes + [TRACE_FN, C.c_void_p]
        lib.pdlp_oracle_solve_traced.restype = C.c_int
        lib.pdlp_oracle_formulate_scale.argtypes = [C.POINTER(abi.PdlpProblem), C.POINTER(abi.PdlpParams), C.POINTER(Formulated)]
        lib.pdlp_oracle_formulate_scale.restype = C.c_int
        lib.pdlp_oracle_free_formulated.argtypes = [C.POINTER(Formulated)]
        lib.pdlp_oracle_spmv_csr.argtypes = [C.c_int, abi.c_i32p, abi.c_i32p, abi.c_f64p, abi.c_f64p, abi.c_f64p]
        lib.pdlp_oracle_trial_step.argtypes = [C.POINTER(Formulated), C.c_double, C.c_double] + [abi.c_f64p] * 9
        _oracle = lib
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libpdlp_ref.so"))


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libpdlp_ref.so"))
        lib.pdlp_ref_solve.argtypes = [C.POINTER(abi.PdlpProblem), C.POINTER(abi.PdlpParams), C.POINTER(abi.PdlpResult)]
        lib.pdlp_ref_solve.restype = C.c_int
        lib.pdlp_ref_scale.argtypes = [C.c_int, C.c_int, abi.c_i32p, abi.c_i32p, abi.c_f64p] + [abi.c_f64p] * 6
        lib.pdlp_ref_scale.restype = C.c_int
        _ref = lib
    return _ref


def _solve_with(fn, lp, params, start=None, trace=None):
    P = abi.ProblemHandle(lp, start)
    R = abi.ResultHandle(lp.num_col, lp.num_row)
    if trace is not None:
        recs = trace

        def cb(_ctx, t):
            recs.append({k: getattr(t.contents, k) for k, _ in Trace._fields_})

        rc = oracle().pdlp_oracle_solve_traced(C.byref(P.struct), C.byref(params), C.byref(R.struct), TRACE_FN(cb), None)
    else:
        rc = fn(C.byref(P.struct), C.byref(params), C.byref(R.struct))
    if rc != 0:
        raise RuntimeError("oracle solve failed rc=%d" % rc)
    return R


def oracle_solve(lp, params=None, start=None, trace=None, **kw):
    params = params or abi.default_params(**kw)
    return _solve_with(oracle().pdlp_oracle_solve, lp, params, start, trace)


def ref_solve(lp, params=None, start=None, **kw):
    params = params or abi.default_params(**kw)
    return _solve_with(ref().pdlp_ref_solve, lp, params, start)


class FormulatedView:
    """numpy views of an oracle-formulated problem (copies; the C memory is freed)."""

    def __init__(self, lp, params=None, **kw):
        params = params or abi.default_params(**kw)
        P = abi.ProblemHandle(lp)
        F = Formulated()
        rc = oracle().pdlp_oracle_formulate_scale(C.byref(P.struct), C.byref(params), C.byref(F))
        if rc:
            raise RuntimeError("formulate failed")
        n, m, nnz = F.n, F.m, F.nnz
        self.n, self.m, self.n_eqs, self.nnz = n, m, F.n_eqs, nnz
        g = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(dt).copy()
        self.csc_beg = g(F.csc_beg, n + 1, np.int32); self.csc_idx = g(F.csc_idx, nnz, np.int32); self.csc_val = g(F.csc_val, nnz, np.float64)
        self.csr_beg = g(F.csr_beg, m + 1, np.int32); self.csr_idx = g(F.csr_idx, nnz, np.int32); self.csr_val = g(F.csr_val, nnz, np.float64)
        self.cost = g(F.cost, n, np.float64); self.rhs = g(F.rhs, m, np.float64)
        self.lower = g(F.lower, n, np.float64); self.upper = g(F.upper, n, np.float64)
        self.col_scale = g(F.col_scale, n, np.float64); self.row_scale = g(F.row_scale, m, np.float64)
        self.row_type = g(F.row_type, m, np.int32); self.row_new_idx = g(F.row_new_idx, m, np.int32)
        self.norm_cost, self.norm_rhs, self.mat_norm_inf = F.norm_cost, F.norm_rhs, F.mat_norm_inf
        oracle().pdlp_oracle_free_formulated(C.byref(F))
