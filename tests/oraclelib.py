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
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("pdlp_oracle.c", "hipdlp_oracle.c", "gpu_order.h", "det_math.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
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
        lib.pdlp_oracle_spmv_csr_device_order.argtypes = lib.pdlp_oracle_spmv_csr.argtypes + [C.c_int]
        lib.pdlp_oracle_det_exp_log.argtypes = [C.c_int, abi.c_f64p, abi.c_f64p, abi.c_f64p]
        lib.pdlp_oracle_slab_blocks.argtypes = [C.c_int, C.c_int, abi.c_i32p, abi.c_i32p, C.c_int, C.c_int, abi.c_i32p]
        lib.pdlp_oracle_slab_blocks.restype = C.c_int
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


# ---- HiPDLP path (oracle/hipdlp_oracle.c) -------------------------------------------------------
class HipdlpProbe(C.Structure):
    _fields_ = [("steps", C.c_int), ("tau", C.c_double), ("sigma", C.c_double),
                ("x_cur", abi.c_f64p), ("y_cur", abi.c_f64p), ("x_next", abi.c_f64p), ("y_next", abi.c_f64p),
                ("out_tau", C.c_double), ("out_sigma", C.c_double), ("out_fpe", C.c_double), ("out_lambda", C.c_double),
                ("n", C.c_int), ("m", C.c_int)]


class HipdlpPrepared(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("n_eqs", C.c_int), ("nnz", C.c_long),
                ("beg", abi.c_i32p), ("idx", abi.c_i32p), ("val", abi.c_f64p),
                ("cost", abi.c_f64p), ("lower", abi.c_f64p), ("upper", abi.c_f64p), ("row_lower", abi.c_f64p),
                ("row_upper", abi.c_f64p), ("col_scale", abi.c_f64p), ("row_scale", abi.c_f64p),
                ("norm_cost", C.c_double), ("norm_rhs", C.c_double)]


def _hipdlp_lib():
    lib = oracle()
    if not getattr(lib, "_hipdlp_ready", False):
        sig = [C.POINTER(abi.PdlpProblem), C.POINTER(abi.PdlpParams)]
        lib.hipdlp_oracle_solve.argtypes = sig + [C.POINTER(abi.PdlpResult)]
        lib.hipdlp_oracle_solve.restype = C.c_int
        lib.hipdlp_oracle_probe.argtypes = sig + [C.POINTER(HipdlpProbe)]
        lib.hipdlp_oracle_probe.restype = C.c_int
        lib.hipdlp_oracle_prepare.argtypes = sig + [C.POINTER(HipdlpPrepared)]
        lib.hipdlp_oracle_prepare.restype = C.c_int
        lib.hipdlp_oracle_free_prepared.argtypes = [C.POINTER(HipdlpPrepared)]
        lib._hipdlp_ready = True
    return lib


def hipdlp_solve_fn():
    return _hipdlp_lib().hipdlp_oracle_solve


def hipdlp_probe(lp, steps, tau=0.0, sigma=0.0, **kw):
    """State of the oracle after `steps` Halpern steps of the first block (last step major)."""
    params = abi.default_params(solver="hipdlp", **kw)
    P = abi.ProblemHandle(lp)
    prep = hipdlp_prepared(lp, **kw)
    n, m = prep["n"], prep["m"]
    out = {k: np.zeros(n if k[0] == "x" else m) for k in ("x_cur", "y_cur", "x_next", "y_next")}
    pr = HipdlpProbe()
    pr.steps, pr.tau, pr.sigma = steps, tau, sigma
    for k, v in out.items():
        setattr(pr, k, v.ctypes.data_as(abi.c_f64p))
    rc = _hipdlp_lib().hipdlp_oracle_probe(C.byref(P.struct), C.byref(params), C.byref(pr))
    if rc:
        raise RuntimeError("hipdlp probe failed")
    out.update(tau=pr.out_tau, sigma=pr.out_sigma, fpe=pr.out_fpe, lam=pr.out_lambda)
    return out


def hipdlp_prepared(lp, **kw):
    params = abi.default_params(solver="hipdlp", **kw)
    P = abi.ProblemHandle(lp)
    F = HipdlpPrepared()
    if _hipdlp_lib().hipdlp_oracle_prepare(C.byref(P.struct), C.byref(params), C.byref(F)):
        raise RuntimeError("hipdlp prepare failed")
    n, m, nnz = F.n, F.m, F.nnz
    g = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(dt).copy()
    out = dict(n=n, m=m, n_eqs=F.n_eqs, nnz=nnz, beg=g(F.beg, n + 1, np.int32), idx=g(F.idx, nnz, np.int32),
               val=g(F.val, nnz, np.float64), cost=g(F.cost, n, np.float64), lower=g(F.lower, n, np.float64),
               upper=g(F.upper, n, np.float64), row_lower=g(F.row_lower, m, np.float64),
               row_upper=g(F.row_upper, m, np.float64), col_scale=g(F.col_scale, n, np.float64),
               row_scale=g(F.row_scale, m, np.float64), norm_cost=F.norm_cost, norm_rhs=F.norm_rhs)
    _hipdlp_lib().hipdlp_oracle_free_prepared(C.byref(F))
    return out
