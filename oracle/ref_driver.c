/*
 * ref_driver.c — TEST INFRASTRUCTURE ONLY.
 *
 * Thin driver that runs the REAL cuPDLP-C CPU core, compiled in place from
 * /root/reference/highs/pdlp/cupdlp/*.c (see Makefile, target _ref), behind
 * the same pdlp_problem_t/pdlp_params_t/pdlp_result_t structs as the product
 * and the restated oracle.  It plays the role of the C++ glue in
 * highs/pdlp/CupdlpWrapper.cpp:30-278 (which needs the whole HiGHS C++ tree
 * and is therefore not compiled here): it formulates the LP with the oracle's
 * restatement of formulateLP_highs, then hands over to the reference's own
 * Init_Scaling / PDHG_Scale_Data / PDHG_Alloc / LP_SolvePDHG / PDHG_Destroy.
 *
 * Nothing in here is copied from the reference; it only CALLS it.  The
 * resulting oracle/_ref/libpdlp_ref.so is git-ignored, travels to the GPU box
 * and is used (a) to pin pdlp_oracle.c and (b) as bench.py's
 * cpu_baseline.kind == "reference".
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pdlp/cupdlp/cupdlp.h"
#include "pdlp_oracle.h"

/* The reference defines these three in CupdlpWrapper.cpp:590-640 (C++ TU). */
void cupdlp_haslb(cupdlp_float* haslb, const cupdlp_float* lb, const cupdlp_float bound, const cupdlp_int len) {
  for (int i = 0; i < len; i++) haslb[i] = lb[i] > bound ? 1.0 : 0.0;
}
void cupdlp_hasub(cupdlp_float* hasub, const cupdlp_float* ub, const cupdlp_float bound, const cupdlp_int len) {
  for (int i = 0; i < len; i++) hasub[i] = ub[i] < bound ? 1.0 : 0.0;
}

static double mono_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int pdlp_ref_solve(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R) {
  if (!P || !opt || !R) return 1;
  const double t0 = mono_now();
  /* formulated, UNSCALED problem from the oracle's formulate() */
  pdlp_params_t noscale = *opt;
  noscale.features_off |= PDLP_FEATURE_SCALING_OFF;
  pdlp_oracle_formulated_t F;
  if (pdlp_oracle_formulate_scale(P, &noscale, &F)) return 1;
  const int n = F.n, m = F.m;
  const int nnz = (int)F.nnz;

  cupdlp_bool chgI[N_INT_USER_PARAM] = {0};
  cupdlp_int iPar[N_INT_USER_PARAM] = {0};
  cupdlp_bool chgF[N_FLOAT_USER_PARAM] = {0};
  cupdlp_float fPar[N_FLOAT_USER_PARAM] = {0.0};
  /* same mapping as getUserParamsFromOptions, CupdlpWrapper.cpp:642-717 */
  chgI[N_ITER_LIM] = 1; iPar[N_ITER_LIM] = opt->iter_limit;
  chgI[N_LOG_LEVEL] = 1; iPar[N_LOG_LEVEL] = opt->log_level;
  chgI[IF_SCALING] = 1; iPar[IF_SCALING] = (opt->features_off & PDLP_FEATURE_SCALING_OFF) ? 0 : 1;
  chgI[E_LINE_SEARCH_METHOD] = 1;
  iPar[E_LINE_SEARCH_METHOD] = (opt->features_off & PDLP_FEATURE_ADAPTIVE_STEP_OFF) ? PDHG_FIXED_LINESEARCH : PDHG_ADAPTIVE_LINESEARCH;
  chgF[D_PRIMAL_TOL] = 1; fPar[D_PRIMAL_TOL] = opt->primal_tol;
  chgF[D_DUAL_TOL] = 1; fPar[D_DUAL_TOL] = opt->dual_tol;
  chgF[D_GAP_TOL] = 1; fPar[D_GAP_TOL] = opt->gap_tol;
  chgF[D_TIME_LIM] = 1; fPar[D_TIME_LIM] = opt->time_limit;
  chgI[E_RESTART_METHOD] = 1;
  iPar[E_RESTART_METHOD] = ((opt->features_off & PDLP_FEATURE_RESTART_OFF) || opt->restart_method == 0) ? 0 : 1;

  CUPDLPscaling* scaling = (CUPDLPscaling*)malloc(sizeof(CUPDLPscaling));
  Init_Scaling(opt->log_level, scaling, n, m, F.cost, F.rhs);

  CUPDLPwork* w = (CUPDLPwork*)malloc(sizeof(CUPDLPwork));
  CUPDLPproblem* prob = (CUPDLPproblem*)malloc(sizeof(CUPDLPproblem));
  CUPDLPcsc* csc = NULL;
  csc_create(&csc);
  csc->nRows = m; csc->nCols = n; csc->nMatElem = nnz;
  csc->colMatBeg = (int*)malloc((size_t)(n + 1) * sizeof(int));
  csc->colMatIdx = (int*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
  csc->colMatElem = (double*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
  memcpy(csc->colMatBeg, F.csc_beg, (size_t)(n + 1) * sizeof(int));
  memcpy(csc->colMatIdx, F.csc_idx, (size_t)nnz * sizeof(int));
  memcpy(csc->colMatElem, F.csc_val, (size_t)nnz * sizeof(double));

  PDHG_Scale_Data(opt->log_level, csc, iPar[IF_SCALING], scaling, F.cost, F.lower, F.upper, F.rhs);

  /* what problem_alloc/data_alloc do, CupdlpWrapper.cpp:466-585 */
  prob->nRows = m; prob->nCols = n; prob->nEqs = F.n_eqs;
  prob->offset = P->offset; prob->sense_origin = P->sense < 0 ? -1.0 : 1.0;
  prob->data = (CUPDLPdata*)malloc(sizeof(CUPDLPdata));
  prob->cost = (double*)malloc((size_t)n * sizeof(double));
  prob->rhs = (double*)malloc((size_t)(m > 0 ? m : 1) * sizeof(double));
  prob->lower = (double*)malloc((size_t)n * sizeof(double));
  prob->upper = (double*)malloc((size_t)n * sizeof(double));
  prob->hasLower = (double*)calloc((size_t)n, sizeof(double));
  prob->hasUpper = (double*)calloc((size_t)n, sizeof(double));
  prob->data->nRows = m; prob->data->nCols = n;
  prob->data->matrix_format = CSR_CSC;
  prob->data->dense_matrix = NULL; prob->data->csr_matrix = NULL; prob->data->csc_matrix = NULL;
  prob->data->device = CPU;
  csc_create(&prob->data->csc_matrix);
  csc_alloc_matrix(prob->data->csc_matrix, m, n, csc, CSC);
  csr_create(&prob->data->csr_matrix);
  csr_alloc_matrix(prob->data->csr_matrix, m, n, csc, CSC);
  prob->data->csc_matrix->MatElemNormInf = infNorm(csc->colMatElem, csc->nMatElem);
  memcpy(prob->cost, F.cost, (size_t)n * sizeof(double));
  memcpy(prob->rhs, F.rhs, (size_t)m * sizeof(double));
  memcpy(prob->lower, F.lower, (size_t)n * sizeof(double));
  memcpy(prob->upper, F.upper, (size_t)n * sizeof(double));
  cupdlp_haslb(prob->hasLower, prob->lower, -INFINITY, n);
  cupdlp_hasub(prob->hasUpper, prob->upper, +INFINITY, n);

  w->problem = prob;
  w->scaling = scaling;
  PDHG_Alloc(w);
  w->timers->dScalingTime = 0;
  w->timers->dPresolveTime = 0;
  memcpy(w->rowScale, scaling->rowScale, (size_t)m * sizeof(double));
  memcpy(w->colScale, scaling->colScale, (size_t)n * sizeof(double));

  /* hot start arrays live in the result vectors, as in the reference */
  int value_valid = 0, dual_valid = 0;
  double* col_value = R->col_value ? R->col_value : (double*)calloc((size_t)P->num_col, sizeof(double));
  double* col_dual = R->col_dual ? R->col_dual : (double*)calloc((size_t)P->num_col, sizeof(double));
  double* row_value = R->row_value ? R->row_value : (double*)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
  double* row_dual = R->row_dual ? R->row_dual : (double*)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
  if (P->start_value_valid && P->start_dual_valid && P->start_col_value && P->start_row_value && P->start_row_dual) {
    memcpy(col_value, P->start_col_value, (size_t)P->num_col * sizeof(double));
    memcpy(row_value, P->start_row_value, (size_t)m * sizeof(double));
    memcpy(row_dual, P->start_row_dual, (size_t)m * sizeof(double));
    value_valid = 1; dual_valid = 1;
  }
  int model_status = 0;
  cupdlp_int num_iter = 0;
  const double t1 = mono_now();
  cupdlp_retcode rc = LP_SolvePDHG(w, chgI, iPar, chgF, fPar, NULL, P->num_col, col_value, col_dual,
                                   row_value, row_dual, &value_valid, &dual_valid, 0, NULL,
                                   F.row_new_idx, F.row_type, &model_status, &num_iter);
  const double t2 = mono_now();
  R->term_code = model_status;
  R->term_iterate = (int)w->resobj->termIterate;
  R->num_iter = num_iter;
  R->num_trials = w->stepsize->nStepSizeIter;
  R->num_restarts = -1;
  const int avg = (w->resobj->termCode == OPTIMAL && w->resobj->termIterate == AVERAGE_ITERATE);
  R->primal_obj = avg ? w->resobj->dPrimalObjAverage : w->resobj->dPrimalObj;
  R->dual_obj = avg ? w->resobj->dDualObjAverage : w->resobj->dDualObj;
  R->primal_feas = avg ? w->resobj->dPrimalFeasAverage : w->resobj->dPrimalFeas;
  R->dual_feas = avg ? w->resobj->dDualFeasAverage : w->resobj->dDualFeas;
  R->rel_gap = avg ? w->resobj->dRelObjGapAverage : w->resobj->dRelObjGap;
  R->norm_rhs = scaling->dNormRhs;
  R->norm_cost = scaling->dNormCost;
  R->value_valid = value_valid;
  R->dual_valid = dual_valid;
  R->setup_seconds = t1 - t0;
  R->solve_seconds = t2 - t1;

  PDHG_Destroy(&w);
  scaling_clear(scaling);
  csc_clear_host(csc);
  problem_clear(prob);
  if (!R->col_value) free(col_value);
  if (!R->col_dual) free(col_dual);
  if (!R->row_value) free(row_value);
  if (!R->row_dual) free(row_dual);
  pdlp_oracle_free_formulated(&F);
  return rc == RETCODE_OK ? 0 : 1;
}

/* Reference scaling alone (Init_Scaling + PDHG_Scale_Data) on a formulated
 * problem, to pin the oracle's scale_ruiz/scale_pc bit for bit. Arrays are
 * scaled in place; col_scale/row_scale receive the cumulative factors. */
int pdlp_ref_scale(int n, int m, int* csc_beg, int* csc_idx, double* csc_val, double* cost,
                   double* lower, double* upper, double* rhs, double* col_scale, double* row_scale) {
  CUPDLPscaling* scaling = (CUPDLPscaling*)malloc(sizeof(CUPDLPscaling));
  Init_Scaling(0, scaling, n, m, cost, rhs);
  CUPDLPcsc csc;
  memset(&csc, 0, sizeof(csc));
  csc.nRows = m; csc.nCols = n; csc.nMatElem = csc_beg[n];
  csc.colMatBeg = csc_beg; csc.colMatIdx = csc_idx; csc.colMatElem = csc_val;
  PDHG_Scale_Data(0, &csc, 1, scaling, cost, lower, upper, rhs);
  memcpy(col_scale, scaling->colScale, (size_t)n * sizeof(double));
  memcpy(row_scale, scaling->rowScale, (size_t)m * sizeof(double));
  scaling_clear(scaling);
  return 0;
}
