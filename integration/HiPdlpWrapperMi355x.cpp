// HiPdlpWrapperMi355x.cpp — the reference-side binding for the SECOND PDLP path, solver="hipdlp":
// a replacement translation unit for highs/pdlp/HiPdlpWrapper.cpp (+ highs/pdlp/hipdlp/*.cc) that
// keeps HiGHS' own entry point
//
//     HighsStatus solveLpHiPdlp(HighsLpSolverObject& solver_object);            // HiPdlpWrapper.h
//     HighsStatus solveLpHiPdlp(const HighsOptions&, HighsTimer&, const HighsLp&, HighsBasis&,
//                               HighsSolution&, HighsModelStatus&, HighsInfo&, HighsCallback&);
//
// (call site HighsSolve.cpp:105-107) and forwards to libpdlp_mi355x.so with algorithm = 1.
// Only HiGHS public headers are used; the option map is that of PDLPSolver::setup
// (hipdlp/pdhg.cc:1783-1874), the status map that of HiPdlpWrapper.cpp:99-128.
#include <algorithm>
#include <cmath>
#include <limits>

#include "lp_data/HighsLpSolverObject.h"
#include "lp_data/HighsSolution.h"
#include "pdlp_mi355x.h"

HighsStatus solveLpHiPdlp(const HighsOptions& options, HighsTimer& timer, const HighsLp& lp,
                          HighsBasis& highs_basis, HighsSolution& highs_solution,
                          HighsModelStatus& model_status, HighsInfo& highs_info, HighsCallback& callback);

HighsStatus solveLpHiPdlp(HighsLpSolverObject& solver_object) {
  return solveLpHiPdlp(solver_object.options_, solver_object.timer_, solver_object.lp_, solver_object.basis_,
                       solver_object.solution_, solver_object.model_status_, solver_object.highs_info_,
                       solver_object.callback_);
}

HighsStatus solveLpHiPdlp(const HighsOptions& options, HighsTimer& timer, const HighsLp& lp,
                          HighsBasis& highs_basis, HighsSolution& highs_solution,
                          HighsModelStatus& model_status, HighsInfo& highs_info, HighsCallback& callback) {
  (void)callback;
  resetModelStatusAndHighsInfo(model_status, highs_info);
  highsLogUser(options.log_options, HighsLogType::kInfo, "Using HiPDLP first order PDLP solver on a GPU (MI355X)\n");

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

  pdlp_params_t opt;
  pdlp_mi355x_default_params(&opt);
  opt.algorithm = 1;
  opt.log_callback = [](void* ctx, int /*level*/, const char* text) {
    highsLogUser(*static_cast<const HighsLogOptions*>(ctx), HighsLogType::kInfo, "%s", text);
  };
  opt.log_ctx = const_cast<HighsLogOptions*>(&options.log_options);
  opt.gap_tol = options.pdlp_optimality_tolerance;  // params_.tolerance
  if (options.kkt_tolerance != kDefaultKktTolerance) opt.gap_tol = options.kkt_tolerance;
  opt.primal_tol = opt.dual_tol = opt.gap_tol;
  opt.iter_limit = (int32_t)std::min<int64_t>((int64_t)options.pdlp_iteration_limit,
                                              (int64_t)std::numeric_limits<int32_t>::max());
  // the reference compares the HiGHS run clock with the raw option; hand over what is left of it
  opt.time_limit = options.time_limit - timer.read();
  opt.features_off = options.pdlp_features_off;
  opt.scaling_mode = options.pdlp_scaling_mode;
  opt.ruiz_iterations = options.pdlp_ruiz_iterations;
  opt.step_size_strategy = options.pdlp_step_size_strategy == kPdlpStepSizeStrategyFixed ? 0 : 1;
  opt.log_level = options.output_flag ? (options.log_dev_level ? 2 : 1) : 0;

  highs_solution.clear();
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
  highs_basis.valid = false;
  model_status = HighsModelStatus::kUnknown;
  if (rc != 0) {
    highsLogUser(options.log_options, HighsLogType::kError, "HiPDLP (MI355X): %s\n", pdlp_mi355x_last_error());
    return HighsStatus::kError;
  }
  if (R.term_code == PDLP_TERM_OPTIMAL) model_status = HighsModelStatus::kOptimal;
  else if (R.term_code == PDLP_TERM_TIMELIMIT_OR_ITERLIMIT)
    model_status = R.reserved_i == 1 ? HighsModelStatus::kTimeLimit : HighsModelStatus::kIterationLimit;
  else return HighsStatus::kError;
  highs_solution.value_valid = true;
  highs_solution.dual_valid = true;
  return HighsStatus::kOk;
}
