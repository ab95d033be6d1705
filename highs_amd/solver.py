"""Host-side mirror of the reference interface of the PDLP path.

`solveLpCupdlp(lp, options)` has the role of the reference's
`HighsStatus solveLpCupdlp(HighsLpSolverObject&)` (highs/pdlp/CupdlpWrapper.cpp:23-278):
options in (`getUserParamsFromOptions` :642-717), HighsSolution / model status
/ pdlp_iteration_count out (status map :225-251).  All the arithmetic happens
in the C-ABI library libpdlp_mi355x.so on the GPU; there is NO CPU fallback —
if the library or a HIP device is missing this raises.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

from . import abi
from .lp import HighsLp, kkt_measures

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpdlp_mi355x.so")
if os.environ.get("PDLP_MI355X_LIB"):  # development: an experimental build of the same library
    LIB_PATH = os.environ["PDLP_MI355X_LIB"]

# HighsModelStatus values (lp_data/HConst.h, highs_c_api.h:74-91)
kSolveError, kOptimal, kInfeasible, kUnboundedOrInfeasible, kUnbounded = 4, 7, 8, 9, 10
kTimeLimit, kIterationLimit, kUnknown = 13, 14, 15
MODEL_STATUS_NAME = {4: "Solve error", 7: "Optimal", 8: "Infeasible", 9: "Primal infeasible or unbounded",
                     10: "Unbounded", 13: "Time limit reached", 14: "Iteration limit reached", 15: "Unknown"}
# HighsStatus
kOk, kWarning, kError = 0, 1, -1

_lib = None

EXPORTS = [
    "pdlp_mi355x_default_params", "pdlp_mi355x_solve", "pdlp_mi355x_create", "pdlp_mi355x_run",
    "pdlp_mi355x_destroy", "pdlp_mi355x_dims", "pdlp_mi355x_reset", "pdlp_mi355x_iterate",
    "pdlp_mi355x_get_vector", "pdlp_mi355x_set_vector", "pdlp_mi355x_stage", "pdlp_mi355x_time_kernel",
    "pdlp_mi355x_comm_unique_id", "pdlp_mi355x_create_sharded", "pdlp_mi355x_gen_synthetic",
    "pdlp_mi355x_free_problem", "pdlp_mi355x_last_error", "pdlp_mi355x_abi_version",
    "pdlp_mi355x_host_prepare", "pdlp_mi355x_free_prepared", "pdlp_mi355x_row_partition", "pdlp_mi355x_sizeof",
    "pdlp_mi355x_host_slab_layout", "pdlp_mi355x_free_slab_layout", "pdlp_mi355x_det_exp_log",
    "pdlp_mi355x_host_task_plan", "pdlp_mi355x_free_task_plan",
    "pdlp_mi355x_read_mps", "pdlp_mi355x_read_mps_timed", "pdlp_mi355x_free_mps_model",
]


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", src])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the PDLP path has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        pP, pO, pR = C.POINTER(abi.PdlpProblem), C.POINTER(abi.PdlpParams), C.POINTER(abi.PdlpResult)
        H = C.c_void_p
        L.pdlp_mi355x_default_params.argtypes = [pO]
        L.pdlp_mi355x_solve.argtypes = [pP, pO, pR]
        L.pdlp_mi355x_create.argtypes = [pP, pO, C.POINTER(H)]
        L.pdlp_mi355x_create_sharded.argtypes = [pP, pO, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(H)]
        L.pdlp_mi355x_run.argtypes = [H, pR]
        L.pdlp_mi355x_destroy.argtypes = [H]
        L.pdlp_mi355x_destroy.restype = None
        L.pdlp_mi355x_dims.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.pdlp_mi355x_reset.argtypes = [H]
        L.pdlp_mi355x_iterate.argtypes = [H, C.c_int32, C.POINTER(abi.PdlpIterStats)]
        L.pdlp_mi355x_get_vector.argtypes = [H, C.c_char_p, abi.c_f64p, C.c_int64]
        L.pdlp_mi355x_set_vector.argtypes = [H, C.c_char_p, abi.c_f64p, C.c_int64]
        L.pdlp_mi355x_stage.argtypes = [H, C.c_char_p, abi.c_f64p, C.c_int32]
        L.pdlp_mi355x_time_kernel.argtypes = [H, C.c_char_p, C.c_int32, C.POINTER(C.c_double)]
        L.pdlp_mi355x_comm_unique_id.argtypes = [C.c_void_p]
        L.pdlp_mi355x_gen_synthetic.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_uint64, pP]
        L.pdlp_mi355x_free_problem.argtypes = [pP]
        L.pdlp_mi355x_free_problem.restype = None
        pPrep = C.POINTER(abi.PdlpPrepared)
        L.pdlp_mi355x_host_prepare.argtypes = [pP, pO, pPrep]
        L.pdlp_mi355x_free_prepared.argtypes = [pPrep]
        L.pdlp_mi355x_free_prepared.restype = None
        L.pdlp_mi355x_row_partition.argtypes = [pPrep, C.c_int32, abi.c_i32p]
        pSlab = C.POINTER(abi.PdlpSlabLayout)
        L.pdlp_mi355x_host_slab_layout.argtypes = [pPrep, C.c_int32, C.c_int32, pSlab]
        L.pdlp_mi355x_free_slab_layout.argtypes = [pSlab]
        L.pdlp_mi355x_free_slab_layout.restype = None
        pTask = C.POINTER(abi.PdlpTaskPlan)
        L.pdlp_mi355x_host_task_plan.argtypes = [pPrep, C.c_int32, C.c_int32, C.c_int32, pTask]
        L.pdlp_mi355x_free_task_plan.argtypes = [pTask]
        L.pdlp_mi355x_free_task_plan.restype = None
        pMps = C.POINTER(abi.PdlpMpsModel)
        L.pdlp_mi355x_read_mps.argtypes = [C.c_char_p, C.c_int32, pMps]
        L.pdlp_mi355x_read_mps_timed.argtypes = [C.c_char_p, C.c_int32, C.c_double, pMps]
        L.pdlp_mi355x_free_mps_model.argtypes = [pMps]
        L.pdlp_mi355x_free_mps_model.restype = None
        L.pdlp_mi355x_det_exp_log.argtypes = [C.c_int32, abi.c_f64p, abi.c_f64p, abi.c_f64p]
        L.pdlp_mi355x_det_exp_log.restype = None
        L.pdlp_mi355x_sizeof.argtypes = [C.c_int32]
        L.pdlp_mi355x_sizeof.restype = C.c_int64
        L.pdlp_mi355x_last_error.restype = C.c_char_p
        L.pdlp_mi355x_abi_version.restype = C.c_int
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().pdlp_mi355x_last_error().decode()}")


class MpsFixedFormat(RuntimeError):
    """The file has names with spaces: a fixed-column reader is needed (return code 3 of pdlp_mi355x_read_mps)."""


class MpsTimeout(RuntimeError):
    """options.time_limit passed while the file was read (return code 5: FilereaderRetcode::kTimeout)."""


def read_mps(path, threads=0, time_limit=0.0):
    """Highs::readModel for an MPS file through the library's multi-threaded reader (csrc/pdlp_mps.cpp; the
    reference: io/FilereaderMps.cpp:24-58 -> io/HMpsFF.cpp).  Returns (HighsLp, info); info carries what HighsLp
    has no field for: integrality, names, objective name, cost row location, warnings, threads, seconds.
    time_limit: HMpsFF::time_limit_ (seconds; <= 0: none)."""
    M = abi.PdlpMpsModel()
    L = lib()
    rc = L.pdlp_mi355x_read_mps_timed(os.fsencode(path), int(threads), float(time_limit), C.byref(M))
    if rc == 2:
        raise FileNotFoundError(L.pdlp_mi355x_last_error().decode())
    if rc == 3:
        raise MpsFixedFormat(L.pdlp_mi355x_last_error().decode())
    if rc == 5:
        raise MpsTimeout(L.pdlp_mi355x_last_error().decode())
    _check(rc, "pdlp_mi355x_read_mps")
    try:
        P = M.lp
        n, m, nz = P.num_col, P.num_row, P.num_nz
        arr = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(k,)).astype(dt).copy() if k else np.zeros(0, dt)
        lp = HighsLp(n, m, arr(P.col_cost, n, np.float64), arr(P.col_lower, n, np.float64), arr(P.col_upper, n, np.float64),
                     arr(P.row_lower, m, np.float64), arr(P.row_upper, m, np.float64),
                     np.ctypeslib.as_array(P.a_start, shape=(n + 1,)).astype(np.int32).copy(),
                     arr(P.a_index, nz, np.int32), arr(P.a_value, nz, np.float64), P.sense, P.offset,
                     (M.model_name or b"").decode())
        if P.q_dim > 0:
            qs = np.ctypeslib.as_array(P.q_start, shape=(P.q_dim + 1,)).astype(np.int32).copy()
            lp.hessian = (qs, arr(P.q_index, int(qs[-1]), np.int32), arr(P.q_value, int(qs[-1]), np.float64))

        def names(pool, start, k):
            if not start or k == 0:
                return None
            st = np.ctypeslib.as_array(start, shape=(k + 1,))
            raw = C.string_at(pool, int(st[k]))
            return [raw[int(st[i]):int(st[i + 1]) - 1].decode() for i in range(k)]

        info = dict(cost_row_location=M.cost_row_location, objective_name=(M.objective_name or b"").decode(),
                    integrality=arr(M.integrality, M.num_integrality, np.uint8) if M.num_integrality else None,
                    col_names=names(M.col_name_pool, M.col_name_start, n), row_names=names(M.row_name_pool, M.row_name_start, m),
                    num_warnings=M.num_warnings, warning_issued=bool(M.warning_issued), warnings=(M.warnings or b"").decode().splitlines(), threads=M.threads,
                    file_bytes=M.file_bytes, seconds=M.seconds)
        if M.hessian_dim > 0:
            hs = np.ctypeslib.as_array(M.hessian_start, shape=(M.hessian_dim + 1,)).astype(np.int32).copy()
            info["hessian_square"] = (hs, arr(M.hessian_index, int(hs[-1]), np.int32), arr(M.hessian_value, int(hs[-1]), np.float64))
        return lp, info
    finally:
        L.pdlp_mi355x_free_mps_model(C.byref(M))


@dataclass
class HighsSolution:
    col_value: np.ndarray
    col_dual: np.ndarray
    row_value: np.ndarray
    row_dual: np.ndarray
    value_valid: bool = False
    dual_valid: bool = False


@dataclass
class PdlpOutcome:
    status: int  # HighsStatus
    model_status: int  # HighsModelStatus
    solution: HighsSolution
    pdlp_iteration_count: int
    info: dict = field(default_factory=dict)
    result: object = None


def model_status_from_term(term_code, num_iter, iter_limit, rc=0):
    """Status map of CupdlpWrapper.cpp:225-251."""
    if rc != 0:
        return kSolveError
    if term_code == abi.TERM_OPTIMAL:
        return kOptimal
    if term_code == abi.TERM_INFEASIBLE:
        return kInfeasible
    if term_code == abi.TERM_UNBOUNDED:
        return kUnbounded
    if term_code == abi.TERM_INFEASIBLE_OR_UNBOUNDED:
        return kUnboundedOrInfeasible
    if term_code == abi.TERM_TIMELIMIT_OR_ITERLIMIT:
        return kIterationLimit if num_iter >= iter_limit - 1 else kTimeLimit
    return kUnknown


def solveLpCupdlp(lp: HighsLp, start=None, solve_fn=None, **options):
    """Solve `lp` with the MI355X PDLP path.  Keyword options use the HiGHS
    option names: kkt_tolerance, primal_feasibility_tolerance (primal_tol),
    pdlp_iteration_limit, pdlp_features_off, time_limit, log_level, ...
    `solve_fn` lets the tests run the same marshalling against the oracle."""
    params = options.pop("params", None) or abi.default_params(**options)
    P = abi.ProblemHandle(lp, start)
    R = abi.ResultHandle(lp.num_col, lp.num_row)
    fn = solve_fn or lib().pdlp_mi355x_solve
    rc = fn(C.byref(P.struct), C.byref(params), C.byref(R.struct))
    ms = model_status_from_term(R.term_code, R.num_iter, params.iter_limit, rc)
    sol = HighsSolution(R.col_value, R.col_dual, R.row_value, R.row_dual, bool(R.value_valid), bool(R.dual_valid))
    info = kkt_measures(lp, sol.col_value, sol.col_dual, sol.row_value, sol.row_dual) if rc == 0 else {}
    info["pdlp_iteration_count"] = int(R.num_iter)
    status = kError if rc != 0 else (kOk if ms == kOptimal or ms == kUnboundedOrInfeasible else kWarning)
    return PdlpOutcome(status, ms, sol, int(R.num_iter), info, R)


def run_model_file(path, solver="pdlp", threads=0, **options):
    """Highs::readModel + Highs::run with solver="pdlp" / "hipdlp" and presolve off, on this package's side of the C
    ABI: the library's MPS reader (read_mps), then the solve.  Returns (PdlpOutcome, HighsLp, read info)."""
    lp, info = read_mps(path, threads)
    if info["integrality"] is not None and info["integrality"].any():
        raise ValueError("the model has integer columns: solver=\"pdlp\" is for LPs (and diagonal QPs)")
    out = solveLpCupdlp(lp, **options) if solver == "pdlp" else solveLpHiPdlp(lp, **options)
    return out, lp, info


def solveLpHiPdlp(lp: HighsLp, solve_fn=None, **options):
    """Mirror of the reference's second PDLP entry point, solveLpHiPdlp (highs/pdlp/HiPdlpWrapper.cpp:26-141):
    restarted Halpern PDHG.  Same option names as HiGHS (kkt_tolerance / pdlp_optimality_tolerance ->
    gap_tol, pdlp_iteration_limit, time_limit, pdlp_features_off, pdlp_scaling_mode, pdlp_ruiz_iterations,
    pdlp_step_size_strategy); status map of HiPdlpWrapper.cpp:99-128."""
    options = dict(options)
    options["solver"] = "hipdlp"
    params = options.pop("params", None) or abi.default_params(**options)
    P = abi.ProblemHandle(lp)
    R = abi.ResultHandle(lp.num_col, lp.num_row)
    fn = solve_fn or lib().pdlp_mi355x_solve
    rc = fn(C.byref(P.struct), C.byref(params), C.byref(R.struct))
    if rc != 0:
        ms = kSolveError
    elif R.term_code == abi.TERM_OPTIMAL:
        ms = kOptimal
    elif R.term_code == abi.TERM_TIMELIMIT_OR_ITERLIMIT:
        ms = kTimeLimit if R.reserved_i == 1 else kIterationLimit
    else:
        ms = kUnknown
    sol = HighsSolution(R.col_value, R.col_dual, R.row_value, R.row_dual, bool(R.value_valid), bool(R.dual_valid))
    info = kkt_measures(lp, sol.col_value, sol.col_dual, sol.row_value, sol.row_dual) if rc == 0 else {}
    info["pdlp_iteration_count"] = int(R.num_iter)
    return PdlpOutcome(kError if rc != 0 else kOk, ms, sol, int(R.num_iter), info, R)


class DeviceSolver:
    """Long-lived solver context with the problem resident in HBM (pdlp_mi355x_create ... destroy)."""

    def __init__(self, lp=None, problem_struct=None, params=None, rank=0, world=1, unique_id=None, **options):
        self.params = params or abi.default_params(**options)
        self._keep = None
        if problem_struct is None:
            self._keep = abi.ProblemHandle(lp)
            problem_struct = self._keep.struct
        self.h = C.c_void_p()
        if world > 1:
            rc = lib().pdlp_mi355x_create_sharded(C.byref(problem_struct), C.byref(self.params), rank, world,
                                                  unique_id, C.byref(self.h))
        else:
            rc = lib().pdlp_mi355x_create(C.byref(problem_struct), C.byref(self.params), C.byref(self.h))
        _check(rc, "pdlp_mi355x_create")
        n, m, nnz, ne = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32()
        _check(lib().pdlp_mi355x_dims(self.h, C.byref(n), C.byref(m), C.byref(nnz), C.byref(ne)), "dims")
        self.n, self.m, self.nnz, self.n_eqs = n.value, m.value, nnz.value, ne.value

    def close(self):
        if self.h:
            lib().pdlp_mi355x_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, num_col, num_row):
        R = abi.ResultHandle(num_col, num_row)
        _check(lib().pdlp_mi355x_run(self.h, C.byref(R.struct)), "pdlp_mi355x_run")
        return R

    def reset(self):
        _check(lib().pdlp_mi355x_reset(self.h), "reset")

    def iterate(self, n_iters):
        st = abi.PdlpIterStats()
        _check(lib().pdlp_mi355x_iterate(self.h, n_iters, C.byref(st)), "iterate")
        return st

    def get(self, name, length):
        out = np.zeros(length)
        _check(lib().pdlp_mi355x_get_vector(self.h, name.encode(), out.ctypes.data_as(abi.c_f64p), length), "get " + name)
        return out

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        _check(lib().pdlp_mi355x_set_vector(self.h, name.encode(), arr.ctypes.data_as(abi.c_f64p), arr.size), "set " + name)

    def stage(self, name, n_out=16, init=None):
        out = np.zeros(n_out)
        if init is not None:  # a few stages read their argument from the scalar array
            out[:len(init)] = init
        _check(lib().pdlp_mi355x_stage(self.h, name.encode(), out.ctypes.data_as(abi.c_f64p), n_out), "stage " + name)
        return out

    def time_kernel(self, name, reps=20):
        ms = C.c_double()
        _check(lib().pdlp_mi355x_time_kernel(self.h, name.encode(), reps, C.byref(ms)), "time " + name)
        return ms.value


class SyntheticProblem:
    """The synthetic LP of SURVEY §8d, generated by the library (C++ std::mt19937_64)."""

    def __init__(self, m, n, nnz, seed=1):
        self.struct = abi.PdlpProblem()
        rc = lib().pdlp_mi355x_gen_synthetic(m, n, nnz, seed, C.byref(self.struct))
        if rc != 0:
            raise RuntimeError("gen_synthetic failed")

    def to_lp(self):
        P = self.struct
        n, m, nnz = P.num_col, P.num_row, P.num_nz
        g = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
        return HighsLp(n, m, g(P.col_cost, n), g(P.col_lower, n), g(P.col_upper, n), g(P.row_lower, m), g(P.row_upper, m),
                       g(P.a_start, n + 1), g(P.a_index, nnz), g(P.a_value, nnz), int(P.sense), float(P.offset),
                       "synthetic").normalise()

    def close(self):
        if self.struct.a_start:
            lib().pdlp_mi355x_free_problem(C.byref(self.struct))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Prepared:
    """Host-side standard form built by the PRODUCT library (pdlp_mi355x_host_prepare); numpy copies."""

    def __init__(self, lp=None, params=None, problem_struct=None, slab_long_limit=256, **options):
        params = params or abi.default_params(**options)
        keep = None
        if problem_struct is None:
            keep = abi.ProblemHandle(lp)
            problem_struct = keep.struct
        F = abi.PdlpPrepared()
        _check(lib().pdlp_mi355x_host_prepare(C.byref(problem_struct), C.byref(params), C.byref(F)), "host_prepare")
        n, m, nnz = F.n, F.m, F.nnz
        self.n, self.m, self.n_eqs, self.n_orig, self.nnz = n, m, F.n_eqs, F.n_orig, nnz
        g = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(dt).copy()
        self.csr_beg = g(F.csr_beg, m + 1, np.int32); self.csr_idx = g(F.csr_idx, nnz, np.int32); self.csr_val = g(F.csr_val, nnz, np.float64)
        self.csc_beg = g(F.csc_beg, n + 1, np.int32); self.csc_idx = g(F.csc_idx, nnz, np.int32); self.csc_val = g(F.csc_val, nnz, np.float64)
        self.cost = g(F.cost, n, np.float64); self.rhs = g(F.rhs, m, np.float64)
        self.lower = g(F.lower, n, np.float64); self.upper = g(F.upper, n, np.float64)
        self.col_scale = g(F.col_scale, n, np.float64); self.row_scale = g(F.row_scale, m, np.float64)
        self.row_kind = g(F.row_kind, m, np.int32); self.row_new_idx = g(F.row_new_idx, m, np.int32)
        self.norm_cost, self.norm_rhs, self.mat_norm_inf = F.norm_cost, F.norm_rhs, F.mat_norm_inf
        self.spmv_blocks_ax, self.spmv_blocks_aty = F.spmv_blocks_ax, F.spmv_blocks_aty
        self._slabs = {}
        for which in (0, 1):
            SL = abi.PdlpSlabLayout()
            _check(lib().pdlp_mi355x_host_slab_layout(C.byref(F), which, slab_long_limit, C.byref(SL)), "slab_layout")
            nb, R = SL.n_blocks, SL.rows_per_block
            self._slabs[which] = dict(
                rows_per_block=R, n_blocks=nb, minor_bits=SL.minor_bits, wave_beg=g(SL.wave_beg, 16 * nb + 1, np.int64),
                slab_width_log2=SL.slab_width_log2, wave_ptr=g(SL.wave_ptr, 16 * nb + 1, np.int64),
                ent=g(SL.ent, SL.nnz_short, np.uint32), val=g(SL.val, SL.nnz_short, np.float64),
                long_mask=g(SL.long_mask, (n if which else m) // 32 + 1, np.uint32), long_map=g(SL.long_map, SL.n_long, np.int32))
            lib().pdlp_mi355x_free_slab_layout(C.byref(SL))
        self._tasks = {}
        for which in (0, 1):
            for balance in (0, 1):
                TP = abi.PdlpTaskPlan()
                _check(lib().pdlp_mi355x_host_task_plan(C.byref(F), which, slab_long_limit, balance, C.byref(TP)), "task_plan")
                nl = TP.n_long
                lb = g(TP.long_beg, nl + 1, np.int64)
                self._tasks[which, balance] = dict(
                    n_tasks=TP.n_tasks, task_group=TP.task_group, n_seg_slots=TP.n_seg_slots, n_long=nl, n_blocks=TP.n_blocks,
                    tile_log2=TP.tile_log2, tasks=g(TP.tasks, 8 * TP.n_tasks, np.int64).reshape(-1, 8),
                    tile_owner=g(TP.tile_owner, TP.n_tiles, np.int64), long_beg=lb, long_idx=g(TP.long_idx, int(lb[-1]) if nl else 0, np.int64))
                lib().pdlp_mi355x_free_task_plan(C.byref(TP))
        self._parts = {}
        for w in (1, 2, 3, 4, 8):
            off = np.zeros(w + 1, dtype=np.int32)
            _check(lib().pdlp_mi355x_row_partition(C.byref(F), w, off.ctypes.data_as(abi.c_i32p)), "row_partition")
            self._parts[w] = off
        lib().pdlp_mi355x_free_prepared(C.byref(F))

    def row_partition(self, world):
        return self._parts[world]

    def task_plan(self, which, balance=1):
        """Segment tasks of the long majors of operand `which` as the slab launches run them (pdlp_task_plan_t)."""
        return self._tasks[which, balance]

    def slab_layout(self, which):
        """which = 0: A by rows, 1: A' by columns."""
        return self._slabs[which]
