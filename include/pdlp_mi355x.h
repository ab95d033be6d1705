/*
 * pdlp_mi355x.h — C ABI of the MI355X-native PDLP hot path for HiGHS.
 *
 * This is the drop-in boundary for ONE path of ERGO-Code/HiGHS: the PDLP
 * first-order LP solver reached through Highs::run() with solver="pdlp"
 * (and, with pdlp_params_t.algorithm = 1, its sibling solver="hipdlp":
 * highs/pdlp/HiPdlpWrapper.cpp, replaced by integration/HiPdlpWrapperMi355x.cpp).
 * The entry points below are exactly what the reference wrapper
 * (highs/pdlp/CupdlpWrapper.cpp) would bind instead of its calls into the
 * vendored cuPDLP-C:
 *
 *   reference call (file:line)                               replaced by
 *   -------------------------------------------------------  --------------------------
 *   formulateLP_highs      CupdlpWrapper.cpp:104,280-448  \
 *   Init_Scaling           CupdlpWrapper.cpp:110           |
 *   PDHG_Scale_Data        CupdlpWrapper.cpp:153           |  pdlp_mi355x_create
 *   problem_alloc          CupdlpWrapper.cpp:161,517-585   |
 *   PDHG_Alloc             CupdlpWrapper.cpp:167          /
 *   LP_SolvePDHG           CupdlpWrapper.cpp:199,
 *                          cupdlp_solver.c:1437-1498          pdlp_mi355x_run
 *   PDHG_Destroy + frees   CupdlpWrapper.cpp:218,253-269      pdlp_mi355x_destroy
 *   (all of the above, one shot)                              pdlp_mi355x_solve
 *
 * Plain C types only: pointers, sizes, doubles, int32. No C++/torch types.
 * All input arrays are caller-owned and read-only; all output arrays are
 * caller-allocated (mirrors highs_solution.*.resize, CupdlpWrapper.cpp:190-193).
 * No function here ever calls exit()/abort() or throws across the boundary;
 * failures are reported through the return code (0 = RETCODE_OK,
 * cupdlp glbopts.h:250-256) and pdlp_mi355x_last_error().
 *
 * Arithmetic is fp64, indices are int32 (cupdlp_int / HighsInt default,
 * glbopts.h:258-263).
 */
#ifndef PDLP_MI355X_H_
#define PDLP_MI355X_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDLP_MI355X_ABI_VERSION 6

/* Termination codes: same numbering as cuPDLP-C's termination_code
 * (cupdlp_defs.h:61-68) so the status map of CupdlpWrapper.cpp:225-251
 * applies unchanged. */
enum {
  PDLP_TERM_OPTIMAL = 0,
  PDLP_TERM_INFEASIBLE = 1,
  PDLP_TERM_UNBOUNDED = 2,
  PDLP_TERM_INFEASIBLE_OR_UNBOUNDED = 3,
  PDLP_TERM_TIMELIMIT_OR_ITERLIMIT = 4,
  PDLP_TERM_FEASIBLE = 5
};

/* pdlp_features_off bitmask, HConst.h:417-422 */
enum {
  PDLP_FEATURE_SCALING_OFF = 1,
  PDLP_FEATURE_RESTART_OFF = 2,
  PDLP_FEATURE_ADAPTIVE_STEP_OFF = 4
};

/* The LP exactly as HiGHS holds it in HighsLp (lp_data/HighsLp.h), column-wise:
 *   min/max  sense * (col_cost' x) + offset
 *   s.t.     row_lower <= A x <= row_upper,  col_lower <= x <= col_upper
 * Infinite bounds are +-inf or any |value| >= 1e20 (CupdlpWrapper.cpp:316-317). */
typedef struct pdlp_problem {
  int32_t num_col;
  int32_t num_row;
  int64_t num_nz;
  const int32_t* a_start; /* [num_col+1] CSC column starts   (lp.a_matrix_.start_) */
  const int32_t* a_index; /* [num_nz]    row indices         (lp.a_matrix_.index_) */
  const double* a_value;  /* [num_nz]                         (lp.a_matrix_.value_) */
  const double* col_cost; /* [num_col] */
  const double* col_lower;
  const double* col_upper;
  const double* row_lower; /* [num_row] */
  const double* row_upper;
  double offset;
  int32_t sense; /* +1 minimise, -1 maximise (ObjSense) */
  /* Optional hot start (PDHG_PreSolve, cupdlp_solver.c:1217-1279); used only
   * when BOTH value_valid and dual_valid are non-zero. May be NULL. */
  const double* start_col_value; /* [num_col] */
  const double* start_row_value; /* [num_row] */
  const double* start_row_dual;  /* [num_row] */
  int32_t start_value_valid;
  int32_t start_dual_valid;
  /* Optional quadratic objective  + 1/2 x' Q x  (no reference counterpart on the PDLP
   * path: HiGHS gates solver="pdlp" to LPs, lp_data/HighsOptions.cpp:1178-1181).  Q as HiGHS holds it in
   * HighsHessian (model/HighsHessian.h:22-34): lower-triangular, column-wise, dimension q_dim <= num_col.
   * The diagonal of Q enters the primal step in closed form (proximal step), its off-diagonal part as an explicit
   * N x term (a third SpMV per trial; single GPU only).  Entries above the diagonal, and a diagonal of the wrong sign
   * for the objective sense, are errors.  q_dim = 0 / NULL arrays = LP.  Through Highs::run() the QP case needs the
   * reference's QP gate lifted (integration/qp_gate_patch.py). */
  int32_t q_dim;
  int32_t reserved_q;
  const int32_t* q_start; /* [q_dim+1] */
  const int32_t* q_index; /* [q_start[q_dim]] row indices (>= column index: lower triangle) */
  const double* q_value;
} pdlp_problem_t;

/* Options, one field per entry that getUserParamsFromOptions
 * (CupdlpWrapper.cpp:642-717) forwards to cuPDLP-C. */
typedef struct pdlp_params {
  double primal_tol;      /* D_PRIMAL_TOL  <- primal_feasibility_tolerance | kkt_tolerance */
  double dual_tol;        /* D_DUAL_TOL    <- dual_feasibility_tolerance   | kkt_tolerance */
  double gap_tol;         /* D_GAP_TOL     <- pdlp_optimality_tolerance    | kkt_tolerance */
  double time_limit;      /* D_TIME_LIM    <- time_limit (seconds; +inf = none) */
  int32_t iter_limit;     /* N_ITER_LIM    <- pdlp_iteration_limit (clamped to int32) */
  int32_t features_off;   /* pdlp_features_off bitmask (scaling/restart/adaptive) */
  int32_t restart_method; /* pdlp_cupdlpc_restart_method; 0 disables restart */
  int32_t log_level;      /* 0 silent, 1 summary, 2 verbose (CupdlpWrapper.cpp:839-848) */
  /* --- MI355X-specific knobs (no reference counterpart) --- */
  int32_t device;         /* HIP device ordinal of this process (default 0) */
  int32_t check_interval; /* 0 = reference schedule (CUPDLP_RELEASE_INTERVAL 40) */
  int32_t reserved[2];    /* test-infrastructure switches (ignored by the product) */
  /* --- second reference path: solver="hipdlp" (HiPdlpWrapper.cpp, hipdlp/pdhg.cc:1783-1874) --- */
  int32_t algorithm;          /* 0 = cuPDLP-C path (solver="pdlp"), 1 = HiPDLP restarted Halpern PDHG */
  int32_t scaling_mode;       /* pdlp_scaling_mode bitmask: 1 Ruiz, 2 L2, 4 Pock-Chambolle (default 5) */
  int32_t ruiz_iterations;    /* pdlp_ruiz_iterations (default 10) */
  int32_t step_size_strategy; /* pdlp_step_size_strategy: 0 fixed; anything else = PID primal weight
                                 (HiGHS default 1 -> PID, pdhg.cc:1856-1864) */
  /* --- multi-GPU behind the one-call boundary (no reference counterpart: Ax_multi_gpu / ATy_multi_gpu
   *     are exit(1) stubs, cupdlp_linalg.c:420-423,453-456) --- */
  int32_t num_devices;        /* pdlp_mi355x_solve only: 0 = take PDLP_MI355X_DEVICES from the environment
                                 (default 1); G > 1 = the constraint matrix is row-block sharded over the
                                 devices device, device+1, ... device+G-1 of THIS process (one host thread
                                 per device, direct xGMI exchange through peer access) */
  int32_t reserved2;
  /* --- log sink (HiGHS: highsLogUser).  NULL = stdout, as the reference's cuPDLP-C prints --- */
  void (*log_callback)(void* ctx, int level, const char* text); /* level 1 = summary, 2 = verbose */
  void* log_ctx;
} pdlp_params_t;

typedef struct pdlp_result {
  double* col_value; /* [num_col] caller-allocated, may be NULL */
  double* col_dual;  /* [num_col] */
  double* row_value; /* [num_row] */
  double* row_dual;  /* [num_row] */
  int32_t value_valid;
  int32_t dual_valid;
  int32_t term_code;   /* PDLP_TERM_* */
  int32_t term_iterate; /* 0 = last iterate, 1 = average iterate */
  int32_t num_iter;    /* outer PDHG iterations = highs_info.pdlp_iteration_count */
  int32_t num_trials;  /* trial steps incl. rejected ones (nStepSizeIter) */
  int32_t num_restarts;
  int32_t reserved_i;  /* HiPDLP path: 1 = stopped by the time limit (TerminationStatus::TIMEOUT) */
  /* cuPDLP's own view of the returned iterate (resobj, scaled-problem space
   * mapped back with row/col scale as in cupdlp_solver.c:12-204) */
  double primal_obj;
  double dual_obj;
  double primal_feas; /* ||r_p||_2 */
  double dual_feas;   /* ||r_d||_2 */
  double rel_gap;
  double norm_rhs;  /* ||b||_2 of the formulated, unscaled problem */
  double norm_cost; /* ||c||_2 */
  /* timings, seconds (steady clock; never time(NULL)) */
  double setup_seconds; /* formulate + scale + transpose + upload */
  double solve_seconds; /* PDHG loop */
  double reserved_d[4];
} pdlp_result_t;

typedef struct pdlp_mi355x_solver pdlp_mi355x_solver_t; /* opaque */

/* Fill *opt with the defaults HiGHS would pass for default options
 * (tolerances 1e-7, iteration limit INT32_MAX, time limit +inf, all features on). */
void pdlp_mi355x_default_params(pdlp_params_t* opt);

/* One-shot solve. Returns 0 on success (term_code says how it ended),
 * non-zero on failure (-> HighsStatus::kError / kSolveError). */
int pdlp_mi355x_solve(const pdlp_problem_t* P, const pdlp_params_t* opt,
                      pdlp_result_t* R);

/* Split form (what a long-lived Highs instance would hold). */
int pdlp_mi355x_create(const pdlp_problem_t* P, const pdlp_params_t* opt,
                       pdlp_mi355x_solver_t** out);
int pdlp_mi355x_run(pdlp_mi355x_solver_t* s, pdlp_result_t* R);
void pdlp_mi355x_destroy(pdlp_mi355x_solver_t* s);

/* ---- measurement / parity hooks (device-resident state) ----------------
 * These exist so that tests and bench.py can drive and observe the hot loop
 * with all inputs already resident in HBM. They are not needed by HiGHS. */

/* Formulated sizes: n = nCols (incl. slack columns), m = nRows, nnz, nEqs. */
int pdlp_mi355x_dims(const pdlp_mi355x_solver_t* s, int32_t* n_cols,
                     int32_t* n_rows, int64_t* nnz, int32_t* n_eqs);

/* (Re)initialise step sizes and iterates: PDHG_Init_Step_Sizes + PDHG_Init_Variables. */
int pdlp_mi355x_reset(pdlp_mi355x_solver_t* s);

/* Run exactly n_iters accepted PDHG iterations starting from the current
 * state, following the reference's check/restart schedule but never
 * terminating on optimality (fixed work for timing). Reports the number of
 * trial steps taken and the GPU time of the loop measured with HIP events on
 * the solver's stream. */
typedef struct pdlp_iter_stats {
  int32_t iters;
  int32_t trials;
  int32_t checks;
  int32_t restarts;
  double gpu_ms;      /* hipEvent elapsed over the whole loop on the solver stream */
  double wall_ms;     /* host steady clock over the same region */
  double spmv_ax_ms;  /* filled only when profiling is enabled, else 0 */
  double spmv_aty_ms;
  double reserved[4];
} pdlp_iter_stats_t;
int pdlp_mi355x_iterate(pdlp_mi355x_solver_t* s, int32_t n_iters,
                        pdlp_iter_stats_t* st);

/* Device vector access by name for kernel-level parity tests. Names:
 * "x","y","ax","aty" (current iterate), "x_next","y_next","ax_next","aty_next",
 * "x_avg","y_avg","ax_avg","aty_avg","x_sum","y_sum","cost","rhs","lower",
 * "upper","col_scale","row_scale","slack_pos","slack_neg".
 * len must equal the vector's length (n or m). */
int pdlp_mi355x_get_vector(pdlp_mi355x_solver_t* s, const char* name,
                           double* host, int64_t len);
int pdlp_mi355x_set_vector(pdlp_mi355x_solver_t* s, const char* name,
                           const double* host, int64_t len);

/* Run one named kernel stage on the current device state. Stages:
 *  "ax"        ax      = A x            (CSR SpMV,  cupdlp_linalg.c:460 Ax)
 *  "aty"       aty     = A' y           (CSC SpMV,  cupdlp_linalg.c:496 ATy)
 *  "trial"     one trial step with the current step sizes (cupdlp_step.c:241-257)
 *  "residuals" PDHG_Compute_Residuals on current+average (cupdlp_solver.c:473)
 *  "profile_on" / "profile_off"  bracket the two SpMV launches of every trial with HIP events
 *              (eager launches, no hipGraph): the next pdlp_mi355x_iterate then reports the
 *              IN-LOOP average launch durations in spmv_ax_ms / spmv_aty_ms (reserved[0] = launches)
 *  "exchange"  scalars_out[0] = 0 not sharded, 1 RCCL all-reduce, 2 direct xGMI mesh
 * HiPDLP solvers (algorithm = 1) instead know:
 *  "steps"     scalars_out[0] holds k on entry (1..40): run Halpern steps 1..k of a block, first and
 *              last one "major" (pdhg.cc:961-1018), from the current state and step sizes
 *  "block"     one whole block of 40 steps, then scalars_out[0] = fixed-point error (pdhg.cc:709-739)
 * scalars_out receives stage-specific scalars, n_scalars its capacity. */
int pdlp_mi355x_stage(pdlp_mi355x_solver_t* s, const char* stage,
                      double* scalars_out, int32_t n_scalars);

/* Time `reps` launches of one kernel with HIP events on the solver stream;
 * returns average milliseconds per launch in *avg_ms. Kernels: "spmv_ax",
 * "spmv_aty", "primal_step", "trial" (whole trial sequence), "copy" (n+m doubles
 * device copy, the measured HBM ceiling). */
int pdlp_mi355x_time_kernel(pdlp_mi355x_solver_t* s, const char* kernel,
                            int32_t reps, double* avg_ms);

/* Multi-GPU (row-block sharding): the process owning rank r of
 * world w passes the FULL problem; create() keeps only its row block.
 * id is the 128-byte communicator id obtained on rank 0 with
 * pdlp_mi355x_comm_unique_id and broadcast by the caller (torch.distributed /
 * MPI / anything).  All ranks must be processes of ONE node: the n-vector
 * exchange writes directly into the peers' HIP-IPC-mapped device memory over
 * xGMI; the id names their rendezvous and is also a valid
 * ncclUniqueId for the RCCL all-reduce fallback.  run/iterate/stage("residuals")
 * are collective afterwards.  Both algorithms shard (algorithm = 1 over the direct
 * exchange only). */
int pdlp_mi355x_comm_unique_id(void* id128);
int pdlp_mi355x_create_sharded(const pdlp_problem_t* P, const pdlp_params_t* opt,
                               int32_t rank, int32_t world, const void* id128,
                               pdlp_mi355x_solver_t** out);

/* Synthetic LP generator of SURVEY §8d / BASELINE.md §3 (std::mt19937_64(seed),
 * box 0<=x<=1, k = nnz/m draws per row, even rows equalities, odd rows <=).
 * One call: *P_out is filled with arrays malloc'ed by the library (release them with
 * pdlp_mi355x_free_problem).  Returns 1 on bad arguments (P_out == NULL, m or n <= 0, nnz_target < m) or
 * when memory runs out; never throws. */
int pdlp_mi355x_gen_synthetic(int32_t m, int32_t n, int64_t nnz_target,
                              uint64_t seed, pdlp_problem_t* P_out);
void pdlp_mi355x_free_problem(pdlp_problem_t* P);

/* ---- host-only introspection (no GPU needed) ---------------------------
 * The standard form the device iterates on: formulate (CupdlpWrapper.cpp:280-448)
 * + scaling (cupdlp_scaling.c) + both matrix orientations (cupdlp_utils.c:1222).
 * Used by the CPU test-suite to check the host logic against the oracle and
 * by the multi-GPU tests to check the row-block partition.  Arrays are
 * malloc'ed by the library; release with pdlp_mi355x_free_prepared. */
typedef struct pdlp_prepared {
  int32_t n, m, n_eqs, n_orig;
  int64_t nnz;
  int32_t *csr_beg, *csr_idx; /* rows, ascending column */
  double* csr_val;
  int32_t *csc_beg, *csc_idx; /* columns, ascending row */
  double* csc_val;
  double *cost, *rhs, *lower, *upper, *col_scale, *row_scale;
  int32_t *row_kind, *row_new_idx; /* per original row */
  double norm_cost, norm_rhs, mat_norm_inf;
  int32_t spmv_blocks_ax, spmv_blocks_aty; /* CSR-adaptive work blocks */
} pdlp_prepared_t;
int pdlp_mi355x_host_prepare(const pdlp_problem_t* P, const pdlp_params_t* opt,
                             pdlp_prepared_t* out);
void pdlp_mi355x_free_prepared(pdlp_prepared_t* out);
/* Row-block partition used by create_sharded: offsets[world+1]. */
int pdlp_mi355x_row_partition(const pdlp_prepared_t* prep, int32_t world,
                              int32_t* offsets);
/* The slab layout the GPU SpMV uses for large operands (see DESIGN.md "slab SpMV"), built on the
 * host for inspection by the CPU tests: which = 0 for A (rows), 1 for A' (columns).  Free with
 * pdlp_mi355x_free_slab_layout. */
typedef struct pdlp_slab_layout {
  int32_t rows_per_block; /* the most majors any block owns (size of the LDS accumulators) */
  int32_t rows_per_wave;  /* reserved (0): waves own variable runs of majors, see wave_beg */
  int32_t n_blocks, minor_bits, slab_width_log2, n_long;
  int64_t nnz_short;
  int32_t* wave_ptr;  /* [16*n_blocks+1] entry offsets */
  uint32_t* ent;      /* [nnz_short] (local_major << minor_bits | minor), local = major - first major of the owning wave */
  double* val;        /* [nnz_short] */
  uint32_t* long_mask;/* [n_major/32 + 1] bit r: major r is a long one */
  int32_t* long_map;  /* [n_long] */
  int32_t* wave_beg;  /* [16*n_blocks+1] first major of every wave: blocks and waves are cut by work, not by major
                         count (csrc/pdlp_host.hpp slabPartition holds the rule) */
} pdlp_slab_layout_t;
int pdlp_mi355x_host_slab_layout(const pdlp_prepared_t* prep, int32_t which,
                                 int32_t long_limit, pdlp_slab_layout_t* out);
void pdlp_mi355x_free_slab_layout(pdlp_slab_layout_t* out);
/* The segment tasks of that operand's long majors (more than long_limit entries) as the slab SpMV launches run them
 * (csrc/pdlp_host.hpp planSlabTasks): task t belongs to task workgroup t / task_group, which is workgroup
 * n_blocks + t / task_group of its launch and runs on XCD (n_blocks + t / task_group) % 8; a task is dealt to a workgroup
 * of the XCD whose streaming blocks gather from the stretch of the vector its entries lie in (tile_owner).  For the CPU
 * tests; balance = 1: at least one task workgroup per CU.  Free with pdlp_mi355x_free_task_plan. */
typedef struct pdlp_task_plan {
  int32_t n_tasks, task_group, n_seg_slots, n_long, n_blocks, tile_log2, n_tiles, reserved;
  int32_t* tasks;     /* [8*n_tasks] entry range [p_beg, p_end) in long_idx, long-major index c (-1: idle), first
                         segment-sum slot of the major, its segment count, its index in the result vector, contained,
                         segment number */
  int8_t* tile_owner; /* [n_tiles] XCD (contiguous block -> XCD map) that gathers most from minors [t << tile_log2, ...) */
  int32_t* long_beg;  /* [n_long+1] compact CSR of the long majors */
  int32_t* long_idx;  /* [long_beg[n_long]] */
} pdlp_task_plan_t;
int pdlp_mi355x_host_task_plan(const pdlp_prepared_t* prep, int32_t which, int32_t long_limit, int32_t balance,
                               pdlp_task_plan_t* out);
void pdlp_mi355x_free_task_plan(pdlp_task_plan_t* out);
/* exp(x[i]) and log(x[i]) as the solver computes them in the restart's primal-weight update (plain IEEE arithmetic,
 * csrc/pdlp_detmath.h: the same bits on host and device) — for the CPU tests, which compare them with long-double libm
 * and with the oracle's separately written restatement.  Reference arithmetic: cupdlp_step.c:165-170 (libm). */
void pdlp_mi355x_det_exp_log(int32_t n, const double* x, double* exp_out, double* log_out);
/* ---- MPS ingest (host-only, no GPU needed) --------------------------------------
 * Multi-threaded reader of free-format MPS files (fixed-format files without spaces in names are free
 * format too).  Replaces, for such files, the reference's single-threaded parser
 *   io/FilereaderMps.cpp:24-58 -> free_format_parser::HMpsFF::loadProblem (io/HMpsFF.cpp:21-133)
 * and builds the same model: first N row = objective (other N rows dropped), duplicate row / column names
 * are distinct rows / columns and only the first occurrence can be addressed, undefined rows and repeated
 * (column, row) pairs are ignored with a warning, zero coefficients dropped, entries of a column in file
 * order, RHS of the cost row = -offset, RANGES by sign, integer columns of a MARKER block are [0,1] until
 * a bound says otherwise, BOUNDS may introduce columns, QUADOBJ / QMATRIX -> Hessian, OBJSENSE either style.
 * integration/FilereaderMpsMi355x.cpp shows the binding inside Highs::readModel.
 * Return: 0 ok (out filled; release with pdlp_mi355x_free_mps_model), 1 malformed file / unsupported
 * section (pdlp_mi355x_last_error), 2 file cannot be opened, 3 names contain spaces: a fixed-COLUMN reader
 * is needed (FreeFormatParserReturnCode::kFixedFormat — the reference then falls back to io/HMPSIO.cpp),
 * 4 the file is a gzip stream and libz could not be loaded (gzip files are inflated by the reader itself otherwise,
 * as the reference does through zstr when built with zlib, HMpsFF.cpp:253-261). */
typedef struct pdlp_mps_model {
  pdlp_problem_t lp;           /* HighsLp fields; lp.q_* = LOWER TRIANGLE of the Hessian (what this library's
                                  QP path and HighsHessian::kTriangular expect), NULL / 0 for an LP */
  int32_t cost_row_location;   /* lp.cost_row_location_ (HMpsFF.cpp:639) */
  int32_t num_integrality;     /* 0: every column continuous (lp.integrality_ stays empty), else num_col */
  const uint8_t* integrality;  /* HighsVarType per column: 0 continuous, 1 integer, 2 semi-continuous, 3 semi-integer */
  const char* model_name;      /* NAME line */
  const char* objective_name;  /* name of the cost row ("Objective" if none) */
  /* names: one pool of NUL-terminated strings + [num+1] start offsets; NULL when the file repeats a name
   * (the reference clears its name arrays then, HMpsFF.cpp:63-80) */
  const char* col_name_pool;
  const int64_t* col_name_start;
  const char* row_name_pool;
  const int64_t* row_name_start;
  /* the Hessian exactly as the parser leaves it (square, column-wise, file order: fillHessian, HMpsFF.cpp:177-216) */
  int32_t hessian_dim;
  int32_t warning_issued;      /* HMpsFF::warning_issued_ at the end of the read: FilereaderRetcode::kWarning if set.
                                  (The reference ASSIGNS the flag at the end of COLUMNS / RHS / BOUNDS / RANGES, so
                                  earlier warnings may be forgotten; num_warnings below counts every class met.) */
  const int32_t* hessian_start;
  const int32_t* hessian_index;
  const double* hessian_value;
  int32_t num_warnings;        /* warning classes met */
  int32_t threads;             /* host threads used */
  const char* warnings;        /* one line per warning class */
  int64_t file_bytes;
  double seconds;              /* wall time of the call */
} pdlp_mps_model_t;
/* num_threads <= 0: one per hardware thread (at least 1 MB of file each, at most 64); > 0: exactly that many. */
int pdlp_mi355x_read_mps(const char* path, int32_t num_threads, pdlp_mps_model_t* out);
/* The same with HMpsFF::time_limit_ (io/FilereaderMps.cpp:30-31, io/HMpsFF.cpp:218-220): seconds from the start of
   the call, checked between the phases of the read; <= 0 or infinite: none.  Returns 5 when it has passed
   (FreeFormatParserReturnCode::kTimeout -> FilereaderRetcode::kTimeout). */
int pdlp_mi355x_read_mps_timed(const char* path, int32_t num_threads, double time_limit, pdlp_mps_model_t* out);
void pdlp_mi355x_free_mps_model(pdlp_mps_model_t* out);

/* sizeof() of the ABI structs: 0 problem, 1 params, 2 result, 3 iter_stats, 4 prepared, 5 slab_layout, 6 mps_model */
int64_t pdlp_mi355x_sizeof(int32_t which);

const char* pdlp_mi355x_last_error(void);
int pdlp_mi355x_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PDLP_MI355X_H_ */
