// CupdlpWrapperMi355x.cpp — the reference-side binding: a replacement translation
// unit for highs/pdlp/CupdlpWrapper.cpp (+ the vendored highs/pdlp/cupdlp/*.c)
// that keeps HiGHS' own entry point
//
//     HighsStatus solveLpCupdlp(HighsLpSolverObject& solver_object);      // CupdlpWrapper.h:92
//     HighsStatus solveLpCupdlp(const HighsOptions&, HighsTimer&, const HighsLp&, HighsBasis&,
//                               HighsSolution&, HighsModelStatus&, HighsInfo&, HighsCallback&);  // :93-98
//
// and forwards the work to the C-ABI library libpdlp_mi355x.so
// (include/pdlp_mi355x.h).  With this TU compiled into libhighs instead of the
// two reference files, `Highs::run()` with solver="pdlp", `bin/highs
// --solver=pdlp` and the C API (Highs_run / Highs_lpCall) drive the MI355X
// path unchanged: the call site stays HighsSolve.cpp:97-99.
//
// Build (in a tree that has the reference sources):
//   g++ -std=c++17 -O2 -fPIC -I$REF/highs -I$REF_BUILD -I<repo>/include -c CupdlpWrapperMi355x.cpp
// and link libhighs with -lpdlp_mi355x instead of CupdlpWrapper.cpp.o / cupdlp/*.c.o.
//
// Only HiGHS public headers are used; nothing here is derived from the body of
// the reference wrapper beyond the documented option/status mapping it must
// reproduce (CupdlpWrapper.cpp:225-251 status map, :642-717 option map).
#include <algorithm>
#include <cmath>
#include <limits>

#include "lp_data/HighsLpSolverObject.h"
#include "lp_data/HighsSolution.h"
#include "model/HighsHessian.h"
#include "pdlp_mi355x.h"

HighsStatus solveLpCupdlp(const HighsOptions& options, HighsTimer& timer, const HighsLp& lp,
                          HighsBasis& highs_basis, HighsSolution& highs_solution,
                          HighsModelStatus& model_status, HighsInfo& highs_info, HighsCallback& callback);
// QPs on the same path.  The reference gates solver="pdlp" to LPs (lp_data/HighsOptions.cpp:1178-1181
// solverValidForQp, lp_data/Highs.cpp:4139 callSolveQp); integration/qp_gate_patch.py lifts that gate in build-time
// COPIES of those two TUs (libhighs_qp.so.1), whose callSolveQp then calls this function when solver == "pdlp".
HighsStatus solveQpPdlpMi355x(const HighsOptions& options, HighsTimer& timer, const HighsLp& lp, const HighsHessian& hessian,
                              HighsBasis& highs_basis, HighsSolution& highs_solution, HighsModelStatus& model_status,
                              HighsInfo& highs_info, HighsCallback& callback);
static HighsStatus solveOnMi355x(const HighsOptions& options, const HighsLp& lp, const HighsHessian* hessian,
                                 HighsBasis& highs_basis, HighsSolution& highs_solution, HighsModelStatus& model_status,
                                 HighsInfo& highs_info);

HighsStatus solveLpCupdlp(HighsLpSolverObject& solver_object) {
  return solveLpCupdlp(solver_object.options_, solver_object.timer_, solver_object.lp_, solver_object.basis_,
                       solver_object.solution_, solver_object.model_status_, solver_object.highs_info_,
                       solver_object.callback_);
}

HighsStatus solveLpCupdlp(const HighsOptions& options, HighsTimer& timer, const HighsLp& lp,
                          HighsBasis& highs_basis, HighsSolution& highs_solution,
                          HighsModelStatus& model_status, HighsInfo& highs_info, HighsCallback& callback) {
  (void)timer;
  (void)callback;  // accepted but unused, as in the reference
  return solveOnMi355x(options, lp, nullptr, highs_basis, highs_solution, model_status, highs_info);
}

HighsStatus solveQpPdlpMi355x(const HighsOptions& options, HighsTimer& timer, const HighsLp& lp, const HighsHessian& hessian,
                              HighsBasis& highs_basis, HighsSolution& highs_solution, HighsModelStatus& model_status,
                              HighsInfo& highs_info, HighsCallback& callback) {
  (void)timer;
  (void)callback;
  return solveOnMi355x(options, lp, &hessian, highs_basis, highs_solution, model_status, highs_info);
}

static HighsStatus solveOnMi355x(const HighsOptions& options, const HighsLp& lp, const HighsHessian* hessian,
                                 HighsBasis& highs_basis, HighsSolution& highs_solution, HighsModelStatus& model_status,
                                 HighsInfo& highs_info) {
  resetModelStatusAndHighsInfo(model_status, highs_info);

  // HighsLp -> pdlp_problem_t: zero-copy views of HiGHS' own storage (column-wise matrix)
  pdlp_problem_t P{};
  P.num_col = lp.num_col_;
  P.num_row = lp.num_row_;
  P.num_nz = lp.a_matrix_.start_[lp.num_col_];
  P.a_start = lp.a_matrix_.start_.data();
  P.a_index = lp.a_matrix_.index_.data();
  P.a_value = lp.a_matrix_.value_.data();
  P.col_cost = lp.col_cost_.data();
  P.col_lower = lp.col_lower_.data();
  P.col_upper = lp.col_upper_.data();
  P.row_lower = lp.row_lower_.data();
  P.row_upper = lp.row_upper_.data();
  P.offset = lp.offset_;
  P.sense = lp.sense_ == ObjSense::kMaximize ? -1 : 1;
  if (hessian && hessian->dim_ > 0) {
    // HighsHessian as HiGHS holds it (model/HighsHessian.h:22-34): lower triangle, column-wise — what pdlp_problem_t takes
    P.q_dim = hessian->dim_;
    P.q_start = hessian->start_.data();
    P.q_index = hessian->index_.data();
    P.q_value = hessian->value_.data();
  }

  // hot start: the incumbent HighsSolution, used only when both parts are valid
  // (PDHG_PreSolve semantics; tests pdlp-restart*, check/TestPdlp.cpp:241-327)
  std::vector<double> start_col, start_row, start_dual;
  if (highs_solution.value_valid && highs_solution.dual_valid &&
      (HighsInt)highs_solution.col_value.size() >= lp.num_col_ &&
      (HighsInt)highs_solution.row_value.size() >= lp.num_row_ &&
      (HighsInt)highs_solution.row_dual.size() >= lp.num_row_) {
    start_col = highs_solution.col_value;
    start_row = highs_solution.row_value;
    start_dual = highs_solution.row_dual;
    P.start_col_value = start_col.data();
    P.start_row_value = start_row.data();
    P.start_row_dual = start_dual.data();
    P.start_value_valid = 1;
    P.start_dual_valid = 1;
  }

  // options -> pdlp_params_t (getUserParamsFromOptions)
  pdlp_params_t opt;
  pdlp_mi355x_default_params(&opt);
  // the library's log lines go where HiGHS's own do (log file, callbacks, output_flag)
  opt.log_callback = [](void* ctx, int /*level*/, const char* text) {
    highsLogUser(*static_cast<const HighsLogOptions*>(ctx), HighsLogType::kInfo, "%s", text);
  };
  opt.log_ctx = const_cast<HighsLogOptions*>(&options.log_options);
  // num_devices stays 0: PDLP_MI355X_DEVICES=G in the environment shards the LP over G GPUs of this process
  // (HiGHS has no option for it; a maintainer adding one would forward it here)
  opt.iter_limit = (int32_t)std::min<int64_t>((int64_t)options.pdlp_iteration_limit,
                                              (int64_t)std::numeric_limits<int32_t>::max());
  opt.log_level = options.output_flag ? (options.log_dev_level ? 2 : 1) : 0;
  opt.features_off = options.pdlp_features_off;
  opt.restart_method = options.pdlp_cupdlpc_restart_method;
  opt.primal_tol = options.primal_feasibility_tolerance;
  opt.dual_tol = options.dual_feasibility_tolerance;
  opt.gap_tol = options.pdlp_optimality_tolerance;
  if (options.kkt_tolerance != kDefaultKktTolerance)
    opt.primal_tol = opt.dual_tol = opt.gap_tol = options.kkt_tolerance;
  opt.time_limit = options.time_limit;  // raw option, like the reference (:707)
  if (opt.features_off & PDLP_FEATURE_SCALING_OFF)
    highsLogUser(options.log_options, HighsLogType::kInfo, "PDLP: Scaling off\n");
  if (opt.features_off & PDLP_FEATURE_ADAPTIVE_STEP_OFF)
    highsLogUser(options.log_options, HighsLogType::kInfo, "PDLP: Adaptive line search off\n");
  if ((opt.features_off & PDLP_FEATURE_RESTART_OFF) || opt.restart_method == 0)
    highsLogUser(options.log_options, HighsLogType::kInfo, "PDLP: Restart off\n");

  // outputs are HiGHS-owned vectors, resized before the call (CupdlpWrapper.cpp:190-193)
  highs_solution.col_value.resize(lp.num_col_);
  highs_solution.row_value.resize(lp.num_row_);
  highs_solution.col_dual.resize(lp.num_col_);
  highs_solution.row_dual.resize(lp.num_row_);
  pdlp_result_t R{};
  R.col_value = highs_solution.col_value.data();
  R.col_dual = highs_solution.col_dual.data();
  R.row_value = highs_solution.row_value.data();
  R.row_dual = highs_solution.row_dual.data();

  const int rc = pdlp_mi355x_solve(&P, &opt, &R);

  highs_info.pdlp_iteration_count = R.num_iter;
  highs_solution.value_valid = R.value_valid != 0;
  highs_solution.dual_valid = R.dual_valid != 0;
  highs_basis.valid = false;
  model_status = HighsModelStatus::kUnknown;
  if (rc != 0) {
    highsLogUser(options.log_options, HighsLogType::kError, "PDLP (MI355X): %s\n", pdlp_mi355x_last_error());
    model_status = HighsModelStatus::kSolveError;
    return HighsStatus::kError;
  }
  switch (R.term_code) {
    case PDLP_TERM_OPTIMAL: model_status = HighsModelStatus::kOptimal; break;
    case PDLP_TERM_INFEASIBLE: model_status = HighsModelStatus::kInfeasible; break;
    case PDLP_TERM_UNBOUNDED: model_status = HighsModelStatus::kUnbounded; break;
    case PDLP_TERM_INFEASIBLE_OR_UNBOUNDED: model_status = HighsModelStatus::kUnboundedOrInfeasible; break;
    case PDLP_TERM_TIMELIMIT_OR_ITERLIMIT:
      model_status = R.num_iter >= opt.iter_limit - 1 ? HighsModelStatus::kIterationLimit
                                                       : HighsModelStatus::kTimeLimit;
      break;
    default: model_status = HighsModelStatus::kUnknown; break;
  }
  return HighsStatus::kOk;
}
