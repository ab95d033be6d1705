/* capi_check.c — the reference's C API (highs/interfaces/highs_c_api.h: Highs_create, Highs_passLp,
 * Highs_setStringOptionValue, Highs_run, Highs_getSolution, ...) driving the MI355X PDLP path through the
 * drop-in libhighs (integration/build_dropin.sh).  Nothing here knows about the GPU: it is what an
 * unmodified C client of HiGHS does.  LP = the "distillation" LP of the reference's own PDLP unit test
 * (check/TestPdlp.cpp:22-61, check/SpecialLps.h:278-296; optimal objective 31.2).
 * Usage: capi_check [solver]     solver = pdlp (default) | hipdlp
 *        capi_check solver model-file [kkt_tolerance]     Highs_readModel + Highs_run on a model file (LP or QP: the
 *                                         QP case needs libhighs_qp.so.1, see integration/Makefile) — prints the line
 *                                         the tests parse, exit code 0 = model status optimal
 * Exit code 0 = optimal with the expected objective and a primal-feasible solution. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "interfaces/highs_c_api.h"

int main(int argc, char** argv) {
  const char* solver = argc > 1 ? argv[1] : "pdlp";
  const HighsInt num_col = 2, num_row = 3, num_nz = 6;
  const double col_cost[2] = {8.0, 10.0};
  const double col_lower[2] = {0.0, 0.0};
  const double col_upper[2] = {1e30, 1e30};
  const double row_lower[3] = {7.0, 12.0, 6.0};
  const double row_upper[3] = {1e30, 1e30, 1e30};
  const HighsInt a_start[3] = {0, 3, 6};
  const HighsInt a_index[6] = {0, 1, 2, 0, 1, 2};
  const double a_value[6] = {2.0, 3.0, 2.0, 2.0, 4.0, 1.0};

  if (argc > 2) { /* a model file through the reference's reader entry point */
    void* h = Highs_create();
    if (!h) return 2;
    Highs_setBoolOptionValue(h, "output_flag", 1);
    if (Highs_setStringOptionValue(h, "solver", solver) != kHighsStatusOk) return 3;
    Highs_setStringOptionValue(h, "presolve", "off");
    Highs_setDoubleOptionValue(h, "kkt_tolerance", argc > 3 ? atof(argv[3]) : 1e-6);
    if (Highs_readModel(h, argv[2]) == kHighsStatusError) return 4;
    const HighsInt rs = Highs_run(h);
    const HighsInt ms = Highs_getModelStatus(h);
    HighsInt it = -1, qn = -1;
    Highs_getIntInfoValue(h, "pdlp_iteration_count", &it);
    Highs_getIntInfoValue(h, "qp_iteration_count", &qn);
    printf("capi_check: solver=%s file=%s run_status=%d model_status=%d objective=%.12g pdlp_iteration_count=%d qp_iteration_count=%d hessian_nz=%d\n",
           solver, argv[2], (int)rs, (int)ms, Highs_getObjectiveValue(h), (int)it, (int)qn, (int)Highs_getHessianNumNz(h));
    Highs_destroy(h);
    return ms == kHighsModelStatusOptimal ? 0 : 10;
  }
  void* highs = Highs_create();
  if (!highs) return 2;
  Highs_setBoolOptionValue(highs, "output_flag", 1);
  if (Highs_setStringOptionValue(highs, "solver", solver) != kHighsStatusOk) return 3;
  Highs_setStringOptionValue(highs, "presolve", "off");
  Highs_setDoubleOptionValue(highs, "kkt_tolerance", 1e-4);
  if (Highs_passLp(highs, num_col, num_row, num_nz, kHighsMatrixFormatColwise, kHighsObjSenseMinimize, 0.0, col_cost,
                   col_lower, col_upper, row_lower, row_upper, a_start, a_index, a_value) != kHighsStatusOk)
    return 4;
  const HighsInt run_status = Highs_run(highs);
  const HighsInt model_status = Highs_getModelStatus(highs);
  double col_value[2], col_dual[2], row_value[3], row_dual[3];
  Highs_getSolution(highs, col_value, col_dual, row_value, row_dual);
  const double obj = Highs_getObjectiveValue(highs);
  HighsInt iters = -1;
  Highs_getIntInfoValue(highs, "pdlp_iteration_count", &iters);
  printf("capi_check: solver=%s run_status=%d model_status=%d objective=%.10g pdlp_iteration_count=%d x=(%.8g, %.8g)\n",
         solver, (int)run_status, (int)model_status, obj, (int)iters, col_value[0], col_value[1]);
  int rc = 0;
  if (run_status != kHighsStatusOk || model_status != kHighsModelStatusOptimal) rc = 10;
  /* check/TestPdlp.cpp:32-33 asserts 1e-3 for the cuPDLP-C path; the Halpern path stops on its own relative criteria at kkt 1e-4 */
  if (fabs(obj - 31.2) > (strcmp(solver, "hipdlp") == 0 ? 5e-3 : 1e-3)) rc = 11;
  for (int i = 0; i < 3 && rc == 0; ++i)
    if (row_value[i] < row_lower[i] - 1e-3) rc = 12;
  if (iters <= 0) rc = 13;
  /* one-call form: Highs_lpCall picks HiGHS's default LP solver (simplex), so it is only checked for linkage */
  Highs_destroy(highs);
  return rc;
}
