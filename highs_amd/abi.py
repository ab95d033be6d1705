"""ctypes mirror of include/pdlp_mi355x.h (the C-ABI drop-in boundary).

Only plain C types cross the boundary: int32/int64/double and pointers to
caller-owned numpy buffers.  Field order and types must match the header
exactly; tests/test_host.py::test_struct_sizes_match_ctypes_mirror checks the struct sizes against the library.
"""
import ctypes as C

import numpy as np

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)

# termination codes, cuPDLP-C termination_code (cupdlp_defs.h:61-68)
TERM_OPTIMAL = 0
TERM_INFEASIBLE = 1
TERM_UNBOUNDED = 2
TERM_INFEASIBLE_OR_UNBOUNDED = 3
TERM_TIMELIMIT_OR_ITERLIMIT = 4
TERM_FEASIBLE = 5

# pdlp_features_off bits, HConst.h:417-422
FEATURE_SCALING_OFF = 1
FEATURE_RESTART_OFF = 2
FEATURE_ADAPTIVE_STEP_OFF = 4


class PdlpProblem(C.Structure):
    _fields_ = [
        ("num_col", C.c_int32),
        ("num_row", C.c_int32),
        ("num_nz", C.c_int64),
        ("a_start", c_i32p),
        ("a_index", c_i32p),
        ("a_value", c_f64p),
        ("col_cost", c_f64p),
        ("col_lower", c_f64p),
        ("col_upper", c_f64p),
        ("row_lower", c_f64p),
        ("row_upper", c_f64p),
        ("offset", C.c_double),
        ("sense", C.c_int32),
        ("start_col_value", c_f64p),
        ("start_row_value", c_f64p),
        ("start_row_dual", c_f64p),
        ("start_value_valid", C.c_int32),
        ("start_dual_valid", C.c_int32),
        ("q_dim", C.c_int32),          # HighsHessian (lower-triangular CSC); only diagonal Q is solved
        ("reserved_q", C.c_int32),
        ("q_start", c_i32p),
        ("q_index", c_i32p),
        ("q_value", c_f64p),
    ]


class PdlpParams(C.Structure):
    _fields_ = [
        ("primal_tol", C.c_double),
        ("dual_tol", C.c_double),
        ("gap_tol", C.c_double),
        ("time_limit", C.c_double),
        ("iter_limit", C.c_int32),
        ("features_off", C.c_int32),
        ("restart_method", C.c_int32),
        ("log_level", C.c_int32),
        ("device", C.c_int32),
        ("check_interval", C.c_int32),
        ("reserved", C.c_int32 * 2),
        ("algorithm", C.c_int32),           # 0 = cuPDLP-C path, 1 = HiPDLP (solver="hipdlp")
        ("scaling_mode", C.c_int32),        # pdlp_scaling_mode: 1 Ruiz | 2 L2 | 4 PC
        ("ruiz_iterations", C.c_int32),     # pdlp_ruiz_iterations
        ("step_size_strategy", C.c_int32),  # pdlp_step_size_strategy: 0 fixed, else PID
        ("num_devices", C.c_int32),         # pdlp_mi355x_solve: shard over this many devices of the process
        ("reserved2", C.c_int32),
        ("log_callback", C.c_void_p),       # void (*)(void* ctx, int level, const char* text); NULL = stdout
        ("log_ctx", C.c_void_p),
    ]


class PdlpResult(C.Structure):
    _fields_ = [
        ("col_value", c_f64p),
        ("col_dual", c_f64p),
        ("row_value", c_f64p),
        ("row_dual", c_f64p),
        ("value_valid", C.c_int32),
        ("dual_valid", C.c_int32),
        ("term_code", C.c_int32),
        ("term_iterate", C.c_int32),
        ("num_iter", C.c_int32),
        ("num_trials", C.c_int32),
        ("num_restarts", C.c_int32),
        ("reserved_i", C.c_int32),
        ("primal_obj", C.c_double),
        ("dual_obj", C.c_double),
        ("primal_feas", C.c_double),
        ("dual_feas", C.c_double),
        ("rel_gap", C.c_double),
        ("norm_rhs", C.c_double),
        ("norm_cost", C.c_double),
        ("setup_seconds", C.c_double),
        ("solve_seconds", C.c_double),
        ("reserved_d", C.c_double * 4),
    ]


class PdlpIterStats(C.Structure):
    _fields_ = [
        ("iters", C.c_int32),
        ("trials", C.c_int32),
        ("checks", C.c_int32),
        ("restarts", C.c_int32),
        ("gpu_ms", C.c_double),
        ("wall_ms", C.c_double),
        ("spmv_ax_ms", C.c_double),
        ("spmv_aty_ms", C.c_double),
        ("reserved", C.c_double * 4),
    ]


class PdlpPrepared(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("n_eqs", C.c_int32), ("n_orig", C.c_int32),
        ("nnz", C.c_int64),
        ("csr_beg", c_i32p), ("csr_idx", c_i32p), ("csr_val", c_f64p),
        ("csc_beg", c_i32p), ("csc_idx", c_i32p), ("csc_val", c_f64p),
        ("cost", c_f64p), ("rhs", c_f64p), ("lower", c_f64p), ("upper", c_f64p),
        ("col_scale", c_f64p), ("row_scale", c_f64p),
        ("row_kind", c_i32p), ("row_new_idx", c_i32p),
        ("norm_cost", C.c_double), ("norm_rhs", C.c_double), ("mat_norm_inf", C.c_double),
        ("spmv_blocks_ax", C.c_int32), ("spmv_blocks_aty", C.c_int32),
    ]


class PdlpSlabLayout(C.Structure):
    _fields_ = [
        ("rows_per_block", C.c_int32), ("rows_per_wave", C.c_int32), ("n_blocks", C.c_int32), ("minor_bits", C.c_int32),
        ("slab_width_log2", C.c_int32), ("n_long", C.c_int32),
        ("nnz_short", C.c_int64),
        ("wave_ptr", c_i32p), ("ent", C.POINTER(C.c_uint32)), ("val", c_f64p),
        ("long_mask", C.POINTER(C.c_uint32)), ("long_map", c_i32p), ("wave_beg", c_i32p),
    ]


class PdlpTaskPlan(C.Structure):
    """pdlp_task_plan_t (include/pdlp_mi355x.h): the segment tasks of a slab operand's long majors, for the CPU tests."""
    _fields_ = [
        ("n_tasks", C.c_int32), ("task_group", C.c_int32), ("n_seg_slots", C.c_int32), ("n_long", C.c_int32),
        ("n_blocks", C.c_int32), ("tile_log2", C.c_int32), ("n_tiles", C.c_int32), ("reserved", C.c_int32),
        ("tasks", c_i32p), ("tile_owner", C.POINTER(C.c_int8)), ("long_beg", c_i32p), ("long_idx", c_i32p),
    ]


class PdlpMpsModel(C.Structure):
    """pdlp_mps_model_t (include/pdlp_mi355x.h): what pdlp_mi355x_read_mps fills."""
    _fields_ = [
        ("lp", PdlpProblem),
        ("cost_row_location", C.c_int32), ("num_integrality", C.c_int32),
        ("integrality", C.POINTER(C.c_uint8)),
        ("model_name", C.c_char_p), ("objective_name", C.c_char_p),
        ("col_name_pool", C.POINTER(C.c_char)), ("col_name_start", C.POINTER(C.c_int64)),
        ("row_name_pool", C.POINTER(C.c_char)), ("row_name_start", C.POINTER(C.c_int64)),
        ("hessian_dim", C.c_int32), ("warning_issued", C.c_int32),
        ("hessian_start", c_i32p), ("hessian_index", c_i32p), ("hessian_value", c_f64p),
        ("num_warnings", C.c_int32), ("threads", C.c_int32),
        ("warnings", C.c_char_p),
        ("file_bytes", C.c_int64), ("seconds", C.c_double),
    ]


def default_params(**kw):
    """Defaults HiGHS passes for default options (CupdlpWrapper.cpp:642-717;
    kkt_tolerance default 1e-7, HConst.h:345)."""
    p = PdlpParams()
    p.primal_tol = 1e-7
    p.dual_tol = 1e-7
    p.gap_tol = 1e-7
    p.time_limit = float("inf")
    p.iter_limit = 2**31 - 1
    p.features_off = 0
    p.restart_method = 1
    p.log_level = 0
    p.device = 0
    p.check_interval = 0
    p.algorithm = 0
    p.scaling_mode = 5          # kPdlpScalingRuiz + kPdlpScalingPC, HighsOptions.h:1345-1349
    p.ruiz_iterations = 10      # HighsOptions.h:1353-1355
    p.step_size_strategy = 1    # kPdlpStepSizeStrategyAdaptive, HighsOptions.h:1374-1378 (HiPDLP: -> PID)
    for k, v in kw.items():
        if k == "kkt_tolerance":
            p.primal_tol = p.dual_tol = p.gap_tol = float(v)
        elif k == "pdlp_iteration_limit":
            p.iter_limit = int(min(v, 2**31 - 1))
        elif k == "pdlp_features_off":
            p.features_off = int(v)
        elif k == "solver":
            p.algorithm = {"pdlp": 0, "hipdlp": 1}[v]
        elif k in ("pdlp_scaling_mode", "pdlp_ruiz_iterations", "pdlp_step_size_strategy"):
            setattr(p, k[5:], int(v))
        elif k == "device_reduction_order":
            # ORACLE ONLY: sum the reductions in the HIP kernels' order (oracle/pdlp_oracle.c, GPU-ORDER)
            p.reserved[0] = 1 if v else 0
        elif k == "device_layout":
            # ORACLE ONLY: which SpMV work plan to restate: "auto" (the product's rule), "csr", "slab"
            p.reserved[1] = {"auto": 0, "csr": 1, "slab": 2}[v]
        else:
            setattr(p, k, v)
    return p


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, ty):
    return a.ctypes.data_as(ty)


class ProblemHandle:
    """Owns the numpy buffers a pdlp_problem_t points to."""

    def __init__(self, lp, start=None):
        self.lp = lp
        self.a_start = _i32(lp.a_start)
        self.a_index = _i32(lp.a_index)
        self.a_value = _f64(lp.a_value)
        self.col_cost = _f64(lp.col_cost)
        self.col_lower = _f64(lp.col_lower)
        self.col_upper = _f64(lp.col_upper)
        self.row_lower = _f64(lp.row_lower)
        self.row_upper = _f64(lp.row_upper)
        P = PdlpProblem()
        P.num_col = int(lp.num_col)
        P.num_row = int(lp.num_row)
        P.num_nz = int(self.a_start[lp.num_col])
        P.a_start = _ptr(self.a_start, c_i32p)
        P.a_index = _ptr(self.a_index, c_i32p)
        P.a_value = _ptr(self.a_value, c_f64p)
        P.col_cost = _ptr(self.col_cost, c_f64p)
        P.col_lower = _ptr(self.col_lower, c_f64p)
        P.col_upper = _ptr(self.col_upper, c_f64p)
        P.row_lower = _ptr(self.row_lower, c_f64p)
        P.row_upper = _ptr(self.row_upper, c_f64p)
        P.offset = float(lp.offset)
        P.sense = int(lp.sense)
        if start is not None:
            self.s_col = _f64(start["col_value"])
            self.s_rowv = _f64(start["row_value"])
            self.s_rowd = _f64(start["row_dual"])
            P.start_col_value = _ptr(self.s_col, c_f64p)
            P.start_row_value = _ptr(self.s_rowv, c_f64p)
            P.start_row_dual = _ptr(self.s_rowd, c_f64p)
            P.start_value_valid = 1
            P.start_dual_valid = 1
        hess = getattr(lp, "hessian", None)
        if hess is not None:  # (start, index, value) of a lower-triangular column-wise HighsHessian
            self.q_start, self.q_index, self.q_value = _i32(hess[0]), _i32(hess[1]), _f64(hess[2])
            P.q_dim = len(self.q_start) - 1
            P.q_start = _ptr(self.q_start, c_i32p)
            P.q_index = _ptr(self.q_index, c_i32p)
            P.q_value = _ptr(self.q_value, c_f64p)
        self.struct = P


class ResultHandle:
    def __init__(self, num_col, num_row):
        self.col_value = np.zeros(num_col)
        self.col_dual = np.zeros(num_col)
        self.row_value = np.zeros(max(num_row, 1))[:num_row]
        self.row_dual = np.zeros(max(num_row, 1))[:num_row]
        R = PdlpResult()
        R.col_value = _ptr(self.col_value, c_f64p)
        R.col_dual = _ptr(self.col_dual, c_f64p)
        R.row_value = _ptr(self.row_value, c_f64p)
        R.row_dual = _ptr(self.row_dual, c_f64p)
        self.struct = R

    def __getattr__(self, k):
        return getattr(self.__dict__["struct"], k)
