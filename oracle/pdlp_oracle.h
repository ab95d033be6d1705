/*
 * pdlp_oracle.h — TEST INFRASTRUCTURE ONLY (see pdlp_oracle.c).
 * CPU restatement of HiGHS' cuPDLP-C path; shares the problem/params/result
 * structs of the product's C ABI so a test can hand the same ctypes objects to
 * both sides.
 */
#ifndef PDLP_ORACLE_H_
#define PDLP_ORACLE_H_

#include "../include/pdlp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What the reference prints on a check iteration (cupdlp_solver.c:851-892) plus
 * the step-size state its PDLP_DEBUG_LOG hook records (cupdlp_utils.c:1785-1850). */
typedef struct pdlp_oracle_trace {
  int iter;
  int trials;
  double beta, primal_step, dual_step;
  double primal_obj, dual_obj, primal_feas, dual_feas;
  double primal_obj_avg, dual_obj_avg, primal_feas_avg, dual_feas_avg;
} pdlp_oracle_trace_t;
typedef void (*pdlp_oracle_trace_fn)(void* ctx, const pdlp_oracle_trace_t* t);

int pdlp_oracle_solve(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R);
int pdlp_oracle_solve_traced(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R,
                             pdlp_oracle_trace_fn trace, void* trace_ctx);

/* block boundaries of the slab layout as the device-order mode models them; blockBeg needs 256 + nMajor / 16 + 3 ints */
int pdlp_oracle_slab_blocks(int nMajor, int nMinor, const int* beg, const int* idx, int longLimit, int which, int* blockBeg);
/* exp(x[i]) and log(x[i]) of the oracle's own plain-arithmetic functions (det_math.h) */
void pdlp_oracle_det_exp_log(int n, const double* x, double* expOut, double* logOut);

/* Formulated + scaled problem (CupdlpWrapper.cpp:104-176), arrays owned by the struct. */
typedef struct pdlp_oracle_formulated {
  int n, m, n_eqs;
  long nnz;
  int *csc_beg, *csc_idx;
  double* csc_val;
  int *csr_beg, *csr_idx;
  double* csr_val;
  double *cost, *rhs, *lower, *upper, *col_scale, *row_scale;
  int *row_type, *row_new_idx;
  double norm_cost, norm_rhs, mat_norm_inf;
} pdlp_oracle_formulated_t;

int pdlp_oracle_formulate_scale(const pdlp_problem_t* P, const pdlp_params_t* opt,
                                pdlp_oracle_formulated_t* F);
void pdlp_oracle_free_formulated(pdlp_oracle_formulated_t* F);

void pdlp_oracle_spmv_csr(int m, const int* beg, const int* idx, const double* val,
                          const double* x, double* out);
void pdlp_oracle_spmv_csr_device_order(int m, const int* beg, const int* idx, const double* val,
                                       const double* x, double* out, int long_limit);
void pdlp_oracle_trial_step(const pdlp_oracle_formulated_t* F, double tau, double sigma,
                            const double* x, const double* y, const double* ax, const double* aty,
                            double* xU, double* yU, double* axU, double* atyU, double* out3);

#ifdef __cplusplus
}
#endif
#endif
