/*
 * pdlp_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the reference algorithm on the
 * PDLP hot path of HiGHS v1.15.1 (solver="pdlp": highs/pdlp/CupdlpWrapper.cpp
 * + vendored cuPDLP-C under highs/pdlp/cupdlp/).  It exists to CHECK the HIP
 * implementation; nothing in the product path (highs_amd/, the C-ABI library)
 * may include, link, import or execute it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Parity pinning: with its left-to-right fp64 sums (the order of the
 * reference's CPU loops, cupdlp_linalg.c:111-125,318-333 and the CSC/CSR
 * scatter order of AxCPU/ATyCPU :35-109) this file reproduces the reference's
 * pinned iteration counts (160 / 79 on the distillation LP, check/TestPdlp.cpp:29,61;
 * 76 240 on 25fv47) and the 13 instance objectives of check/CMakeLists.txt:321-335
 * — see tests/test_oracle_golden.py — and is cross-checked against the real
 * cuPDLP-C core compiled from /root/reference into oracle/_ref (Makefile).
 *
 * Every function cites the reference file:line it follows.  All paths are
 * relative to /root/reference/highs/pdlp/.
 */
#include "pdlp_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

enum { ROW_EQ = 0, ROW_LEQ = 1, ROW_GEQ = 2, ROW_BOUND = 3 }; /* cupdlp_defs.h constraint types */

/* ------------------------------------------------------------------ */
/* level-1 helpers: cupdlp_linalg.c:111-125 (nrm2), :318-333 (dot)     */
/* ------------------------------------------------------------------ */
static double o_dot(int n, const double* x, const double* y) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}
static double o_nrm2(int n, const double* x) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i] * x[i];
  return sqrt(s);
}
static void o_axpy(int n, double a, const double* x, double* y) { /* y += a x, linalg.c:345-356 */
  for (int i = 0; i < n; ++i) y[i] += a * x[i];
}
static void o_scal(int n, double a, double* x) { /* linalg.c:359-370 */
  for (int i = 0; i < n; ++i) x[i] *= a;
}
static void o_proj_lb_vec(int n, double* x, const double* lb) { /* linalg.c:215-220 */
  for (int i = 0; i < n; ++i) x[i] = x[i] > lb[i] ? x[i] : lb[i];
}
static void o_proj_ub_vec(int n, double* x, const double* ub) { /* linalg.c:223-228 */
  for (int i = 0; i < n; ++i) x[i] = x[i] < ub[i] ? x[i] : ub[i];
}
static void o_proj_pos(int n, double* x) { /* linalg.c:231-236 with lb = 0 */
  for (int i = 0; i < n; ++i) x[i] = x[i] > 0.0 ? x[i] : 0.0;
}
static void o_proj_neg(int n, double* x) { /* linalg.c:239-244 with ub = 0 */
  for (int i = 0; i < n; ++i) x[i] = x[i] < 0.0 ? x[i] : 0.0;
}
static void o_emul(int n, double* x, const double* y) { /* cupdlp_cdot linalg.c:188-192 */
  for (int i = 0; i < n; ++i) x[i] *= y[i];
}
static void o_ediv(int n, double* x, const double* y) { /* cupdlp_cdiv linalg.c:195-199 */
  for (int i = 0; i < n; ++i) x[i] /= y[i];
}

static double o_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ */
/* the work area                                                       */
/* ------------------------------------------------------------------ */
typedef struct {
  /* formulated problem */
  int n, m, nEqs, n0; /* n0 = original column count */
  long nnz;
  int *cscBeg, *cscIdx;
  double* cscVal;
  int *csrBeg, *csrIdx;
  double* csrVal;
  double *cost, *rhs, *lower, *upper, *hasLower, *hasUpper, *lowerF, *upperF;
  double* qdiag; /* QP extension (no reference counterpart on this path): diagonal of Q, NULL for an LP */
  /* ... and its off-diagonal part N (symmetric, both triangles, by rows with ascending column; NULL when Q is
   * diagonal): the diagonal goes into the proximal step, N x is an explicit gradient term (a third SpMV per trial) */
  int *qnBeg, *qnIdx;
  double* qnVal;
  double *nx[2], *nxAvg; /* N x by parity (like A'y), N xAvg at checks */
  int slabQ, chunkQ, *planQ, nPlanQ; /* device-order layout of N (g_setup) */
  int *rowType, *rowNewIdx;
  double offset, sense;
  /* scaling */
  double *colScale, *rowScale;
  int ifScaled;
  double normCost, normRhs, matNormInf;
  /* iterates (double-buffered by nIter%2, cupdlp_defs.h:334-337) */
  double *x[2], *y[2], *ax[2], *aty[2];
  double *xAvg, *yAvg, *axAvg, *atyAvg, *xSum, *ySum, *xLast, *yLast;
  double *slackPos, *slackNeg, *slackPosAvg, *slackNegAvg;
  double *bufN, *bufN2, *bufM, *bufMax2, *bufMax3;
  /* step size state (CUPDLPstepsize) */
  int adaptive;
  int nStepSizeIter;
  double primalStep, dualStep, sumPrimalStep, sumDualStep, beta;
  /* residuals (CUPDLPresobj) */
  double feasTol;
  double pObj, dObj, gap, pFeas, dFeas, relGap;
  double pObjA, dObjA, gapA, pFeasA, dFeasA, relGapA;
  double pFeasLR, dFeasLR, gapLR, pFeasLC, dFeasLC, gapLC;
  double pInfObj, pInfRes, dInfObj, dInfRes, pInfObjA, pInfResA, dInfObjA, dInfResA;
  int termCode, termIterate;
  /* settings */
  int iterLim, restartOn, logLevel;
  double tolP, tolD, tolG, timeLim;
  /* counters */
  int nIter, iLastRestartIter, nRestarts;
  double solveBeg, solveTime;
  /* optional "device reduction order" mode (see the GPU-ORDER section below) */
  int gpuOrder;
  int *cssBeg, *cssIdx; /* A' by columns with ascending row index (what the device streams for A'y) */
  double* cssVal;
  int *planA, nPlanA, *planAt, nPlanAt; /* CSR-adaptive work plans, identical to the product's */
  /* slab layout (operands whose gathered vector has >= 2^18 entries): blocks of R consecutive majors,
   * majors longer than 256 entries go to a CSR side plan over the compacted long majors */
  int slabA, slabAt, nLongA, nLongAt, chunkA, chunkAt;
  int *longMapA, *longMapAt, *longBegA, *longBegAt;
  double *gPartA, *gPartB, *gStat;
} Work;

static double* dalloc(long n) { return (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }
static int* ialloc(long n) { return (int*)calloc((size_t)(n > 0 ? n : 1), sizeof(int)); }

static void work_free(Work* w) {
  free(w->cscBeg); free(w->cscIdx); free(w->cscVal);
  free(w->csrBeg); free(w->csrIdx); free(w->csrVal);
  free(w->cost); free(w->rhs); free(w->lower); free(w->upper);
  free(w->hasLower); free(w->hasUpper); free(w->lowerF); free(w->upperF); free(w->qdiag);
  free(w->qnBeg); free(w->qnIdx); free(w->qnVal); free(w->nx[0]); free(w->nx[1]); free(w->nxAvg); free(w->planQ);
  free(w->rowType); free(w->rowNewIdx); free(w->colScale); free(w->rowScale);
  for (int k = 0; k < 2; ++k) { free(w->x[k]); free(w->y[k]); free(w->ax[k]); free(w->aty[k]); }
  free(w->xAvg); free(w->yAvg); free(w->axAvg); free(w->atyAvg);
  free(w->xSum); free(w->ySum); free(w->xLast); free(w->yLast);
  free(w->slackPos); free(w->slackNeg); free(w->slackPosAvg); free(w->slackNegAvg);
  free(w->bufN); free(w->bufN2); free(w->bufM); free(w->bufMax2); free(w->bufMax3);
  free(w->cssBeg); free(w->cssIdx); free(w->cssVal); free(w->planA); free(w->planAt);
  free(w->gPartA); free(w->gPartB); free(w->gStat);
  free(w->longMapA); free(w->longMapAt); free(w->longBegA); free(w->longBegAt);
}

/* ------------------------------------------------------------------ */
/* formulateLP_highs — CupdlpWrapper.cpp:280-448                       */
/* ------------------------------------------------------------------ */
static int formulate(Work* w, const pdlp_problem_t* P) {
  const int n0 = P->num_col, m = P->num_row;
  const long nnz0 = P->a_start[n0];
  w->n0 = n0;
  w->m = m;
  w->offset = P->offset;
  w->sense = P->sense < 0 ? -1.0 : 1.0;
  w->rowType = ialloc(m);
  w->rowNewIdx = ialloc(m);
  int n = n0, nEqs = 0;
  long nnz = nnz0;
  /* row classification, thresholds +-1e20 (:315-344); free rows are treated as BOUND */
  for (int i = 0; i < m; ++i) {
    const int hasLo = P->row_lower[i] > -1e20, hasUp = P->row_upper[i] < 1e20;
    if (hasLo && hasUp && P->row_lower[i] == P->row_upper[i]) {
      w->rowType[i] = ROW_EQ; ++nEqs;
    } else if (hasLo && !hasUp) {
      w->rowType[i] = ROW_GEQ;
    } else if (!hasLo && hasUp) {
      w->rowType[i] = ROW_LEQ;
    } else { /* ranged, or free */
      w->rowType[i] = ROW_BOUND; ++n; ++nnz; ++nEqs;
    }
  }
  w->n = n; w->nEqs = nEqs; w->nnz = nnz;
  w->cost = dalloc(n); w->lower = dalloc(n); w->upper = dalloc(n); w->rhs = dalloc(m);
  w->cscBeg = ialloc(n + 1); w->cscIdx = ialloc(nnz); w->cscVal = dalloc(nnz);
  for (int j = 0; j < n0; ++j) { /* :356-361 */
    w->cost[j] = P->col_cost[j] * w->sense;
    w->lower[j] = P->col_lower[j];
    w->upper[j] = P->col_upper[j];
  }
  for (int i = 0, j = n0; i < m; ++i) /* slack bounds :367-373 (slack cost stays 0) */
    if (w->rowType[i] == ROW_BOUND) { w->lower[j] = P->row_lower[i]; w->upper[j] = P->row_upper[i]; ++j; }
  for (int j = 0; j < n; ++j) { /* :375-378 */
    if (w->lower[j] < -1e20) w->lower[j] = -INFINITY;
    if (w->upper[j] > 1e20) w->upper[j] = INFINITY;
  }
  /* QP extension: + 1/2 x'Qx with Q given as a lower-triangular column-wise HighsHessian (model/HighsHessian.h:22-34),
   * split into its diagonal and its off-diagonal part N exactly as the product does (pdlp_host.cpp extractHessian):
   * both triangles of N by rows with ascending column, repeated entries added up left to right */
  w->qdiag = NULL;
  w->qnBeg = NULL; w->qnIdx = NULL; w->qnVal = NULL;
  if (P->q_dim > 0 && P->q_start && P->q_start[P->q_dim] > 0) {
    int any = 0;
    long nOff = 0;
    double* q = dalloc(n);
    int* cnt = ialloc(n + 1);
    for (int j = 0; j < P->q_dim; ++j)
      for (int p = P->q_start[j]; p < P->q_start[j + 1]; ++p) {
        const int i = P->q_index[p];
        if (P->q_value[p] == 0.0) continue;
        any = 1;
        if (i == j) { q[j] += P->q_value[p] * w->sense; continue; }
        if (i < j || i >= P->q_dim) { free(q); free(cnt); return 1; }
        ++cnt[i + 1]; ++cnt[j + 1];
        nOff += 2;
      }
    if (any) w->qdiag = q; else free(q);
    if (nOff > 0) {
      for (int r = 0; r < n; ++r) cnt[r + 1] += cnt[r];
      int* pos = ialloc(n);
      int* tailBeg = ialloc(n);
      int* idx = ialloc(nOff);
      double* val = dalloc(nOff);
      for (int r = 0; r < n; ++r) pos[r] = cnt[r];
      for (int j = 0; j < P->q_dim; ++j)
        for (int p = P->q_start[j]; p < P->q_start[j + 1]; ++p) {
          const int i = P->q_index[p];
          if (i == j || P->q_value[p] == 0.0) continue;
          idx[pos[i]] = j; val[pos[i]++] = P->q_value[p] * w->sense;
        }
      for (int r = 0; r < n; ++r) tailBeg[r] = pos[r];
      for (int j = 0; j < P->q_dim; ++j)
        for (int p = P->q_start[j]; p < P->q_start[j + 1]; ++p) {
          const int i = P->q_index[p];
          if (i == j || P->q_value[p] == 0.0) continue;
          idx[pos[j]] = i; val[pos[j]++] = P->q_value[p] * w->sense;
        }
      for (int r = 0; r < n; ++r)
        for (int a = tailBeg[r] + 1; a < cnt[r + 1]; ++a) {
          const int ci = idx[a];
          const double cv = val[a];
          int b = a - 1;
          while (b >= tailBeg[r] && idx[b] > ci) { idx[b + 1] = idx[b]; val[b + 1] = val[b]; --b; }
          idx[b + 1] = ci; val[b + 1] = cv;
        }
      w->qnBeg = ialloc(n + 1); w->qnIdx = ialloc(nOff); w->qnVal = dalloc(nOff);
      long k = 0;
      for (int r = 0; r < n; ++r) {
        w->qnBeg[r] = (int)k;
        for (int a = cnt[r]; a < cnt[r + 1]; ++a) {
          if (a > cnt[r] && idx[a] == w->qnIdx[k - 1]) w->qnVal[k - 1] += val[a];
          else { w->qnIdx[k] = idx[a]; w->qnVal[k] = val[a]; ++k; }
        }
      }
      w->qnBeg[n] = (int)k;
      free(pos); free(tailBeg); free(idx); free(val);
      w->nx[0] = dalloc(n); w->nx[1] = dalloc(n); w->nxAvg = dalloc(n);
    }
    free(cnt);
  }
  /* row permutation: EQ/BOUND first (:382-392), then LEQ (negated) / GEQ (:394-404) */
  for (int i = 0, k = 0; i < m; ++i) {
    if (w->rowType[i] == ROW_EQ) { w->rhs[k] = P->row_lower[i]; w->rowNewIdx[i] = k++; }
    else if (w->rowType[i] == ROW_BOUND) { w->rhs[k] = 0.0; w->rowNewIdx[i] = k++; }
  }
  for (int i = 0, k = nEqs; i < m; ++i) {
    if (w->rowType[i] == ROW_LEQ) { w->rhs[k] = -P->row_upper[i]; w->rowNewIdx[i] = k++; }
    else if (w->rowType[i] == ROW_GEQ) { w->rhs[k] = P->row_lower[i]; w->rowNewIdx[i] = k++; }
  }
  /* matrix: column starts unchanged, slack columns appended with one entry (:408-410) */
  for (int j = 0; j <= n0; ++j) w->cscBeg[j] = P->a_start[j];
  for (int j = n0 + 1; j <= n; ++j) w->cscBeg[j] = w->cscBeg[j - 1] + 1;
  /* within a column: EQ/BOUND entries first, then LEQ (value negated) / GEQ (:413-436) */
  long k = 0;
  for (int j = 0; j < n0; ++j) {
    for (int p = P->a_start[j]; p < P->a_start[j + 1]; ++p) {
      const int t = w->rowType[P->a_index[p]];
      if (t == ROW_EQ || t == ROW_BOUND) { w->cscIdx[k] = w->rowNewIdx[P->a_index[p]]; w->cscVal[k] = P->a_value[p]; ++k; }
    }
    for (int p = P->a_start[j]; p < P->a_start[j + 1]; ++p) {
      const int t = w->rowType[P->a_index[p]];
      if (t == ROW_LEQ) { w->cscIdx[k] = w->rowNewIdx[P->a_index[p]]; w->cscVal[k] = -P->a_value[p]; ++k; }
      else if (t == ROW_GEQ) { w->cscIdx[k] = w->rowNewIdx[P->a_index[p]]; w->cscVal[k] = P->a_value[p]; ++k; }
    }
  }
  for (int i = 0, j = n0; i < m; ++i) /* slack entries -1 (:439-445) */
    if (w->rowType[i] == ROW_BOUND) { w->cscIdx[w->cscBeg[j]] = w->rowNewIdx[i]; w->cscVal[w->cscBeg[j]] = -1.0; ++j; }
  return 0;
}

/* ------------------------------------------------------------------ */
/* scaling — cupdlp_scaling.c                                          */
/* ------------------------------------------------------------------ */
/* scale_problem :17-45 */
static void scale_apply(Work* w, const double* cs, const double* rs) {
  const int n = w->n, m = w->m;
  o_ediv(n, w->cost, cs);
  if (w->qdiag) { o_ediv(n, w->qdiag, cs); o_ediv(n, w->qdiag, cs); } /* 1/2 q x^2 with x = x'/cs */
  if (w->qnBeg)
    for (int r = 0; r < n; ++r)
      for (int p = w->qnBeg[r]; p < w->qnBeg[r + 1]; ++p) w->qnVal[p] = (w->qnVal[p] / cs[r]) / cs[w->qnIdx[p]];
  o_emul(n, w->lower, cs);
  o_emul(n, w->upper, cs);
  o_ediv(m, w->rhs, rs);
  for (long p = 0; p < w->cscBeg[n]; ++p) w->cscVal[p] /= rs[w->cscIdx[p]];
  for (int j = 0; j < n; ++j)
    for (int p = w->cscBeg[j]; p < w->cscBeg[j + 1]; ++p) w->cscVal[p] /= cs[j];
  o_emul(n, w->colScale, cs);
  o_emul(m, w->rowScale, rs);
}
/* cupdlp_ruiz_scaling :47-120, infinity norm, RuizTimes = 10 (Init_Scaling :407-408) */
static void scale_ruiz(Work* w, int times) {
  const int n = w->n, m = w->m;
  double* cs = dalloc(n);
  double* rs = dalloc(m);
  for (int it = 0; it < times; ++it) {
    memset(cs, 0, sizeof(double) * (size_t)n);
    memset(rs, 0, sizeof(double) * (size_t)m);
    for (int j = 0; j < n; ++j) {
      double mx = 0.0;
      for (int p = w->cscBeg[j]; p < w->cscBeg[j + 1]; ++p) {
        const double a = fabs(w->cscVal[p]);
        if (a > mx) mx = a;
      }
      cs[j] = (w->cscBeg[j] == w->cscBeg[j + 1]) ? 0.0 : sqrt(mx);
      if (cs[j] == 0.0) cs[j] = 1.0;
    }
    if (m > 0) {
      for (long p = 0; p < w->cscBeg[n]; ++p) {
        const double a = fabs(w->cscVal[p]);
        if (rs[w->cscIdx[p]] < a) rs[w->cscIdx[p]] = a;
      }
      for (int i = 0; i < m; ++i) rs[i] = (rs[i] == 0.0) ? 1.0 : sqrt(rs[i]);
    }
    scale_apply(w, cs, rs);
  }
  free(cs); free(rs);
}
/* cupdlp_pc_scaling :174-231 with alpha = 1 (Init_Scaling :409) */
static void scale_pc(Work* w, double alpha) {
  const int n = w->n, m = w->m;
  double* cs = dalloc(n);
  double* rs = dalloc(m);
  if (m > 0) {
    for (int j = 0; j < n; ++j) {
      for (int p = w->cscBeg[j]; p < w->cscBeg[j + 1]; ++p) cs[j] += pow(fabs(w->cscVal[p]), alpha);
      cs[j] = sqrt(pow(cs[j], 1.0 / alpha));
      if (cs[j] == 0.0) cs[j] = 1.0;
    }
    for (long p = 0; p < w->cscBeg[n]; ++p) rs[w->cscIdx[p]] += pow(fabs(w->cscVal[p]), 2.0 - alpha);
    for (int i = 0; i < m; ++i) {
      rs[i] = sqrt(pow(rs[i], 1.0 / (2.0 - alpha)));
      if (rs[i] == 0.0) rs[i] = 1.0;
    }
  }
  scale_apply(w, cs, rs);
  free(cs); free(rs);
}

/* cupdlp_dcs_transpose cupdlp_cs.c:189-214: counting transpose; CSR rows come
 * out with ascending column index. */
static void build_csr(Work* w) {
  const int n = w->n, m = w->m;
  w->csrBeg = ialloc(m + 1); w->csrIdx = ialloc(w->nnz); w->csrVal = dalloc(w->nnz);
  int* cnt = ialloc(m);
  for (long p = 0; p < w->cscBeg[n]; ++p) cnt[w->cscIdx[p]]++;
  int acc = 0;
  for (int i = 0; i < m; ++i) { w->csrBeg[i] = acc; acc += cnt[i]; cnt[i] = w->csrBeg[i]; }
  w->csrBeg[m] = acc;
  for (int j = 0; j < n; ++j)
    for (int p = w->cscBeg[j]; p < w->cscBeg[j + 1]; ++p) {
      const int q = cnt[w->cscIdx[p]]++;
      w->csrIdx[q] = j; w->csrVal[q] = w->cscVal[p];
    }
  free(cnt);
}

/* AxCPU linalg.c:35-71 scatters CSC columns in ascending j, so ax[i] is the
 * left-to-right sum over ascending column index — identical to a row gather
 * over the ascending-column CSR built above. */
static void g_spmv(const int* beg, const int* idx, const double* val, int nMajor, const double* in, double* out, int chunk);
static void o_Ax(const Work* w, double* ax, const double* x) {
  if (w->gpuOrder) { g_spmv(w->csrBeg, w->csrIdx, w->csrVal, w->m, x, ax, w->chunkA); return; }
  for (int i = 0; i < w->m; ++i) {
    double s = 0.0;
    for (int p = w->csrBeg[i]; p < w->csrBeg[i + 1]; ++p) s += w->csrVal[p] * x[w->csrIdx[p]];
    ax[i] = s;
  }
}
/* ATyCPU linalg.c:73-109 scatters CSR rows in ascending i: aty[j] is the sum
 * over ascending (permuted) row index.  The CSC column of the formulated
 * matrix is NOT sorted by row (EQ/BOUND entries first), so do the scatter. */
static void o_ATy(const Work* w, double* aty, const double* y) {
  if (w->gpuOrder) { g_spmv(w->cssBeg, w->cssIdx, w->cssVal, w->n, y, aty, w->chunkAt); return; }
  memset(aty, 0, sizeof(double) * (size_t)w->n);
  for (int i = 0; i < w->m; ++i) {
    const double yi = y[i];
    for (int p = w->csrBeg[i]; p < w->csrBeg[i + 1]; ++p) aty[w->csrIdx[p]] += w->csrVal[p] * yi;
  }
}


/* N x for the off-diagonal part of a QP's Hessian (rows summed left to right, like A x on the device) */
static void g_spmv(const int* beg, const int* idx, const double* val, int nMajor, const double* in, double* out, int longLimit);
static void o_Nx(const Work* w, double* nx, const double* x) {
  if (w->gpuOrder) { g_spmv(w->qnBeg, w->qnIdx, w->qnVal, w->n, x, nx, w->chunkQ); return; }
  for (int r = 0; r < w->n; ++r) {
    double s = 0.0;
    for (int p = w->qnBeg[r]; p < w->qnBeg[r + 1]; ++p) s += w->qnVal[p] * x[w->qnIdx[p]];
    nx[r] = s;
  }
}

/* ------------------------------------------------------------------ */
/* GPU-ORDER mode (opt->reserved[0] == 1)                              */
/*                                                                    */
/* The HIP path computes x+, y+, A x+, A' y+ bit-identically to the    */
/* serial loops above; the ONLY arithmetic difference is the summation */
/* order of its reductions (per-lane strided accumulation, 64-lane     */
/* shuffle tree, fixed-order sum of 4 wave results, 4-chain sum of     */
/* per-block partials) — and, since round 4, the exp / log of the      */
/* restart's primal-weight update, which the device computes with the  */
/* plain-arithmetic functions (restated in det_math.h, < 1 ulp         */
/* from libm) instead of the host's libm.                              */
/* This section restates those orders exactly                          */
/* (highs_amd/csrc/pdlp_kernels.hip: waveSum, blockSum, reducePartials,*/
/* k_spmv epilogues, k_row_stats, k_col_stats, k_diff_norm2, k_dot)    */
/* for the CSR-stream layout, so that a whole GPU solve can be checked */
/* BIT FOR BIT against this oracle (tests/test_gpu_bitexact.py).       */
/* Constants mirror pdlp_kernels.hpp: 256 lanes per block, 2048        */
/* nonzeros / 2048 majors per work block, vector grids capped at 2048. */
/* ------------------------------------------------------------------ */
#include "gpu_order.h"
#include "det_math.h" /* the oracle's own exp / log in plain IEEE arithmetic: nothing of the product is included here */
/* planStream (pdlp_host.cpp): blocks of whole majors with at most `chunk` entries in total; a major longer than
 * chunk belongs to no block.  Returns [2*nBlocks] (first, end) pairs. */
static int* g_plan(const int* beg, int nMajor, int chunk, int* nBlocksOut) {
  int* plan = ialloc(2 * (long)nMajor + 2);
  int nb = 0, start = 0;
  while (start < nMajor) {
    if (beg[start + 1] - beg[start] > chunk) { ++start; continue; }
    const int base = beg[start];
    int end = start;
    while (end < nMajor && end - start < G_MAXMAJ && beg[end + 1] - base <= chunk) ++end;
    plan[2 * nb] = start; plan[2 * nb + 1] = end; ++nb;
    start = end;
  }
  *nBlocksOut = nb;
  return plan;
}
static void g_spmv(const int* beg, const int* idx, const double* val, int nMajor, const double* in, double* out, int longLimit) {
  for (int r = 0; r < nMajor; ++r) out[r] = g_major_sum(beg, idx, val, in, r, longLimit);
}
/* per-block partial of a per-major quantity: lane t accumulates majors r0+t, r0+t+256, ... */
static double g_block_partial(const int* plan, int blk, const double* perMajor) {
  double lane[G_T];
  const int r0 = plan[2 * blk], r1 = plan[2 * blk + 1];
  for (int t = 0; t < G_T; ++t) {
    double a = 0.0;
    for (int r = r0 + t; r < r1; r += G_T) a += perMajor[r];
    lane[t] = a;
  }
  return g_block_sum(lane);
}
enum { G_SLAB_AUTO_MINOR = 1 << 18 };

static void g_setup(Work* w, int layoutMode) {
  const int n = w->n, m = w->m;
  /* A' by columns with ascending row (transpose of the CSR) */
  w->cssBeg = ialloc(n + 1); w->cssIdx = ialloc(w->nnz); w->cssVal = dalloc(w->nnz);
  int* cnt = ialloc(n);
  for (long p = 0; p < w->csrBeg[m]; ++p) cnt[w->csrIdx[p]]++;
  int acc = 0;
  for (int j = 0; j < n; ++j) { w->cssBeg[j] = acc; acc += cnt[j]; cnt[j] = w->cssBeg[j]; }
  w->cssBeg[n] = acc;
  for (int i = 0; i < m; ++i)
    for (int p = w->csrBeg[i]; p < w->csrBeg[i + 1]; ++p) { const int q = cnt[w->csrIdx[p]]++; w->cssIdx[q] = i; w->cssVal[q] = w->csrVal[p]; }
  free(cnt);
  /* layoutMode: 0 = the product's automatic rule, 1 = CSR stream, 2 = slab */
  w->slabA = layoutMode == 2 || (layoutMode == 0 && n >= G_SLAB_AUTO_MINOR);   /* A gathers x (n) */
  w->slabAt = layoutMode == 2 || (layoutMode == 0 && m >= G_SLAB_AUTO_MINOR);  /* A' gathers y (m) */
  if (w->slabA && (m <= 0 || !g_slab_fits(m, n))) w->slabA = 0;   /* minors do not fit the entry packing */
  if (w->slabAt && (n <= 0 || !g_slab_fits(n, m))) w->slabAt = 0;
  /* the longest major that is summed left to right (longer ones: segment tasks, gpu_order.h g_long_major_sum) */
  w->chunkA = w->slabA ? G_SLAB_LONG : g_chunk_for(w->nnz);
  w->chunkAt = w->slabAt ? G_SLAB_LONG : g_chunk_for(w->nnz);
  /* slab layout: planA / planAt / planQ hold the block boundaries of the work partition, nPlan* = -(number of blocks) */
  int* gCold = ialloc((long)(n > m ? n : m) + 1);
  int* gCount = ialloc((long)(n > m ? n : m) + 1);
  if (w->slabA) {
    g_slab_cold(w->csrBeg, w->csrIdx, m, n, w->chunkA, gCold, gCount);
    w->planA = ialloc(g_slab_blocks_room(m)); w->nPlanA = -g_slab_blocks(w->csrBeg, gCold, m, n, w->chunkA, G_SLAB_MAJOR_COST_ROWS, w->planA);
  }
  else w->planA = g_plan(w->csrBeg, m, w->chunkA, &w->nPlanA);
  if (w->slabAt) {
    g_slab_cold(w->cssBeg, w->cssIdx, n, m, w->chunkAt, gCold, gCount);
    w->planAt = ialloc(g_slab_blocks_room(n)); w->nPlanAt = -g_slab_blocks(w->cssBeg, gCold, n, m, w->chunkAt, G_SLAB_MAJOR_COST_COLS, w->planAt);
  }
  else w->planAt = g_plan(w->cssBeg, n, w->chunkAt, &w->nPlanAt);
  if (w->qnBeg) { /* N gathers x (n), majors = n */
    w->slabQ = layoutMode == 2 || (layoutMode == 0 && n >= G_SLAB_AUTO_MINOR);
    if (w->slabQ && (n <= 0 || !g_slab_fits(n, n))) w->slabQ = 0;
    w->chunkQ = w->slabQ ? G_SLAB_LONG : g_chunk_for(w->qnBeg[n]);
    if (w->slabQ) {
      g_slab_cold(w->qnBeg, w->qnIdx, n, n, w->chunkQ, gCold, gCount);
      w->planQ = ialloc(g_slab_blocks_room(n)); w->nPlanQ = -g_slab_blocks(w->qnBeg, gCold, n, n, w->chunkQ, G_SLAB_MAJOR_COST_ROWS, w->planQ);
    }
    else w->planQ = g_plan(w->qnBeg, n, w->chunkQ, &w->nPlanQ);
  }
  free(gCold); free(gCount);
  const long mx = 2L * (n > m ? n : m) + G_MAXGRID + 8;
  w->gPartA = dalloc(mx); w->gPartB = dalloc(mx); w->gStat = dalloc(mx);
}

/* Fixed-order total of a per-major quantity exactly as the SpMV epilogues + k_decide add it up.  Partial slots:
 * one per work block of the stream (CSR layout) or per 1024-thread block of R majors (slab layout; long majors
 * skipped), then the long majors in ascending order — one slot each, or, beyond 2048 of them, one slot per group
 * of G consecutive ones added left to right (k_long_groups). */
static double g_epilogue_total(Work* w, int isAt /* 0: A, 1: A', 2: N (off-diagonal Hessian) */, const double* perMajor) {
  const int nMajor = isAt ? w->n : w->m;
  const int slab = isAt == 2 ? w->slabQ : isAt ? w->slabAt : w->slabA;
  const int* plan = isAt == 2 ? w->planQ : isAt ? w->planAt : w->planA;
  const int nPlan = isAt == 2 ? w->nPlanQ : isAt ? w->nPlanAt : w->nPlanA;
  const int limit = isAt == 2 ? w->chunkQ : isAt ? w->chunkAt : w->chunkA;
  const int* beg = isAt == 2 ? w->qnBeg : isAt ? w->cssBeg : w->csrBeg;
  double* part = w->gPartA;
  int np = 0;
  if (!slab) {
    for (int b = 0; b < nPlan; ++b) part[np++] = g_block_partial(plan, b, perMajor);
  } else {
    const int nBlocks = -nPlan;
    for (int b = 0; b < nBlocks; ++b) { /* k_spmv_slab: 1024 threads, thread t owns majors plan[b] + t, + 1024, ... */
      double lane[G_SLAB_T];
      const int rEnd = plan[b + 1];
      for (int t = 0; t < G_SLAB_T; ++t) {
        double a = 0.0;
        for (int r = plan[b] + t; r < rEnd; r += G_SLAB_T)
          if (beg[r + 1] - beg[r] <= limit) a += perMajor[r];
        lane[t] = a;
      }
      part[np++] = g_block_sum_n(lane, G_SLAB_T);
    }
  }
  int nLong = 0;
  for (int r = 0; r < nMajor; ++r) if (beg[r + 1] - beg[r] > limit) ++nLong;
  if (nLong > 0) {
    const int G = nLong > G_LONG_SLOT_CAP ? (nLong + G_LONG_SLOT_CAP - 1) / G_LONG_SLOT_CAP : 1;
    int c = 0;
    double acc = 0.0;
    for (int r = 0; r < nMajor; ++r) {
      if (beg[r + 1] - beg[r] <= limit) continue;
      if (c % G == 0) acc = 0.0;
      acc += perMajor[r];
      ++c;
      if (c % G == 0 || c == nLong) part[np++] = acc;
    }
  }
  return g_reduce_partials(part, np);
}

/* movement / interaction sums of one trial in device order */
static void g_trial_sums(Work* w, double* dX2, double* dY2, double* inter, double* qint) {
  const int n = w->n, m = w->m, c = w->nIter % 2, u = (w->nIter + 1) % 2;
  double* perM = w->bufMax2;
  double* perN = w->bufMax3;
  for (int i = 0; i < m; ++i) { const double d = w->y[c][i] - w->y[u][i]; perM[i] = d * d; }
  *dY2 = g_epilogue_total(w, 0, perM);
  for (int j = 0; j < n; ++j) { const double d = w->x[c][j] - w->x[u][j]; perN[j] = d * d; }
  *dX2 = g_epilogue_total(w, 1, perN);
  for (int j = 0; j < n; ++j) { const double dx = w->x[c][j] - w->x[u][j]; const double da = w->aty[c][j] - w->aty[u][j]; perN[j] = dx * da; }
  *inter = g_epilogue_total(w, 1, perN);
  *qint = 0.0;
  if (w->qnBeg) {
    for (int j = 0; j < n; ++j) { const double dx = w->x[c][j] - w->x[u][j]; const double dq = w->nx[c][j] - w->nx[u][j]; perN[j] = dx * dq; }
    *qint = g_epilogue_total(w, 2, perN);
  }
}

/* k_row_stats / k_col_stats element functions */
typedef struct { const Work* w; const double *ax, *y, *aty, *x; int q; const double* nx; } GStatCtx;
static double g_row_elem(const void* vc, int i) {
  const GStatCtx* c = (const GStatCtx*)vc;
  const Work* w = c->w;
  const int ineq = i >= w->nEqs;
  const double axv = c->ax[i], yv = c->y[i], b = w->rhs[i], rs = w->ifScaled ? w->rowScale[i] : 1.0;
  if (c->q == 0) { double r = axv + (-1.0) * b; if (ineq) r = r < 0.0 ? r : 0.0; r *= rs; return r * r; }
  if (c->q == 1) return yv * b;
  if (c->q == 2) return yv * yv;
  double cc = axv; if (ineq) cc = cc < 0.0 ? cc : 0.0; cc *= rs; return cc * cc;
}
static double g_col_elem(const void* vc, int j) {
  const GStatCtx* c = (const GStatCtx*)vc;
  const Work* w = c->w;
  const double xv = c->x[j], cj = w->cost[j], l = w->lower[j], u = w->upper[j];
  const double cs = w->ifScaled ? w->colScale[j] : 1.0;
  const double hasL = l > -INFINITY ? 1.0 : 0.0, hasU = u < INFINITY ? 1.0 : 0.0;
  const double lF = l > -INFINITY ? l : 0.0, uF = u < INFINITY ? u : 0.0;
  const double atyv = c->aty[j];
  double r = -atyv + cj;
  const double qj = w->qdiag ? w->qdiag[j] : 0.0;
  if (w->qdiag) r += qj * xv; /* reduced cost c + Qx - A'y */
  const double nj = c->nx ? c->nx[j] : 0.0;
  if (c->nx) r += nj;
  const double sp = (r > 0.0 ? r : 0.0) * hasL;
  const double sn = (-(r < 0.0 ? r : 0.0)) * hasU;
  switch (c->q) {
    case 0: return xv * cj;
    case 1: return sp * lF;
    case 2: return sn * uF;
    case 3: { double rd = r + (-1.0) * sp; rd += sn; rd *= cs; return rd * rd; }
    case 4: return sp * sp;
    case 5: return sn * sn;
    case 6: { double pc = (atyv + sp) - sn; pc *= cs; return pc * pc; }
    case 7: return xv * xv;
    case 8: { double lb = (xv < 0.0 ? xv : 0.0) * hasL; if (w->ifScaled) lb /= cs; return lb * lb; }
    case 10: { double h = (0.5 * qj * xv) * xv; if (c->nx) h += (0.5 * nj) * xv; return h; }
    default: { double ub = (xv > 0.0 ? xv : 0.0) * hasU; if (w->ifScaled) ub /= cs; return ub * ub; }
  }
}
/* residuals + infeasibility numbers of one iterate, as Solver::computeResiduals derives them */
static void g_residuals(Work* w, const double* x, const double* y, const double* ax, const double* aty, const double* nx,
                        double* sp, double* sn, double* pObj, double* dObj, double* pFeas, double* dFeas,
                        double* gap, double* relGap, double* pInfObj, double* pInfRes, double* dInfObj,
                        double* dInfRes) {
  GStatCtx c = {w, ax, y, aty, x, 0, nx};
  double rs[4], cs[11];
  for (int q = 0; q < 4; ++q) { c.q = q; rs[q] = g_grid_sum(w->m, g_row_elem, &c, w->gStat); }
  for (int q = 0; q < 10; ++q) { c.q = q; cs[q] = g_grid_sum(w->n, g_col_elem, &c, w->gStat); }
  cs[10] = 0.0;
  if (w->qdiag) { c.q = 10; cs[10] = g_grid_sum(w->n, g_col_elem, &c, w->gStat); }
  for (int j = 0; j < w->n; ++j) { /* slacks as k_col_stats stores them */
    const double l = w->lower[j], u = w->upper[j];
    double r = -aty[j] + w->cost[j];
    if (w->qdiag) r += w->qdiag[j] * x[j];
    if (nx) r += nx[j];
    sp[j] = (r > 0.0 ? r : 0.0) * (l > -INFINITY ? 1.0 : 0.0);
    sn[j] = (-(r < 0.0 ? r : 0.0)) * (u < INFINITY ? 1.0 : 0.0);
  }
  *pObj = (w->qdiag ? cs[0] + cs[10] : cs[0]) * w->sense + w->offset;
  *pFeas = sqrt(rs[0]);
  *dObj = (w->qdiag ? ((rs[1] + cs[1]) - cs[2]) - cs[10] : (rs[1] + cs[1] - cs[2])) * w->sense + w->offset;
  *dFeas = sqrt(cs[3]);
  *gap = *pObj - *dObj;
  *relGap = fabs(*pObj - *dObj) / (1.0 + fabs(*pObj) + fabs(*dObj));
  double dScale = sqrt(rs[2] + cs[4] + cs[5]);
  if (dScale < 1e-8) dScale = 1.0;
  *pInfObj = (*dObj - w->offset) / w->sense / dScale;
  *pInfRes = sqrt(cs[6]) / dScale;
  double pScale = sqrt(cs[7]);
  if (pScale < 1e-8) pScale = 1.0;
  *dInfObj = (*pObj - w->offset) / w->sense / pScale;
  *dInfRes = sqrt(rs[3] + cs[8] + cs[9]) / pScale;
}
typedef struct { const double *a, *b; } GDiffCtx;
static double g_diff_elem(const void* vc, int i) { const GDiffCtx* c = (const GDiffCtx*)vc; const double d = c->a[i] - c->b[i]; return d * d; }
static double g_dot_elem(const void* vc, int i) { const GDiffCtx* c = (const GDiffCtx*)vc; return c->a[i] * c->b[i]; }

/* ------------------------------------------------------------------ */
/* residuals — cupdlp_solver.c:12-204 (CPU branches)                   */
/* ------------------------------------------------------------------ */
static void primal_feasibility(Work* w, const double* ax, const double* x, const double* nx, double* feas, double* obj) {
  const int n = w->n, m = w->m;
  double cx = o_dot(n, x, w->cost);
  if (w->qdiag) {
    double h = 0.0;
    for (int j = 0; j < n; ++j) { double t = (0.5 * w->qdiag[j] * x[j]) * x[j]; if (nx) t += (0.5 * nx[j]) * x[j]; h += t; }
    cx += h;
  }
  *obj = cx * w->sense + w->offset;
  double* r = w->bufM;
  memcpy(r, ax, sizeof(double) * (size_t)m);
  o_axpy(m, -1.0, w->rhs, r);
  o_proj_neg(m - w->nEqs, r + w->nEqs);
  if (w->ifScaled) o_emul(m, r, w->rowScale);
  *feas = o_nrm2(m, r);
}
static void dual_feasibility(Work* w, const double* aty, const double* y, const double* x, const double* nx, double* feas,
                             double* obj, double* sp, double* sn) {
  const int n = w->n, m = w->m;
  double d = o_dot(m, y, w->rhs);
  double* r = w->bufN;
  memcpy(r, aty, sizeof(double) * (size_t)n);
  o_scal(n, -1.0, r);
  o_axpy(n, 1.0, w->cost, r);
  double qh = 0.0;
  if (w->qdiag)
    for (int j = 0; j < n; ++j) {
      r[j] += w->qdiag[j] * x[j];
      double t = (0.5 * w->qdiag[j] * x[j]) * x[j];
      if (nx) { r[j] += nx[j]; t += (0.5 * nx[j]) * x[j]; }
      qh += t;
    }
  memcpy(sp, r, sizeof(double) * (size_t)n);
  o_proj_pos(n, sp);
  o_emul(n, sp, w->hasLower);
  d += o_dot(n, sp, w->lowerF);
  memcpy(sn, r, sizeof(double) * (size_t)n);
  o_proj_neg(n, sn);
  o_scal(n, -1.0, sn);
  o_emul(n, sn, w->hasUpper);
  d -= o_dot(n, sn, w->upperF);
  d -= qh;
  *obj = d * w->sense + w->offset;
  o_axpy(n, -1.0, sp, r);
  o_axpy(n, 1.0, sn, r);
  if (w->ifScaled) o_emul(n, r, w->colScale);
  *feas = o_nrm2(n, r);
}
/* PDHG_Compute_Residuals :473-529 */
static void compute_residuals(Work* w) {
  const int c = w->nIter % 2;
  if (w->gpuOrder) {
    g_residuals(w, w->x[c], w->y[c], w->ax[c], w->aty[c], w->qnBeg ? w->nx[c] : NULL, w->slackPos, w->slackNeg, &w->pObj, &w->dObj, &w->pFeas,
                &w->dFeas, &w->gap, &w->relGap, &w->pInfObj, &w->pInfRes, &w->dInfObj, &w->dInfRes);
    g_residuals(w, w->xAvg, w->yAvg, w->axAvg, w->atyAvg, w->qnBeg ? w->nxAvg : NULL, w->slackPosAvg, w->slackNegAvg, &w->pObjA, &w->dObjA,
                &w->pFeasA, &w->dFeasA, &w->gapA, &w->relGapA, &w->pInfObjA, &w->pInfResA, &w->dInfObjA,
                &w->dInfResA);
    return;
  }
  const double* nxc = w->qnBeg ? w->nx[c] : NULL;
  const double* nxa = w->qnBeg ? w->nxAvg : NULL;
  primal_feasibility(w, w->ax[c], w->x[c], nxc, &w->pFeas, &w->pObj);
  dual_feasibility(w, w->aty[c], w->y[c], w->x[c], nxc, &w->dFeas, &w->dObj, w->slackPos, w->slackNeg);
  primal_feasibility(w, w->axAvg, w->xAvg, nxa, &w->pFeasA, &w->pObjA);
  dual_feasibility(w, w->atyAvg, w->yAvg, w->xAvg, nxa, &w->dFeasA, &w->dObjA, w->slackPosAvg, w->slackNegAvg);
  w->gap = w->pObj - w->dObj;
  w->relGap = fabs(w->pObj - w->dObj) / (1.0 + fabs(w->pObj) + fabs(w->dObj));
  w->gapA = w->pObjA - w->dObjA;
  w->relGapA = fabs(w->pObjA - w->dObjA) / (1.0 + fabs(w->pObjA) + fabs(w->dObjA));
}
/* PDHG_Compute_Primal_Infeasibility :206-311 (CPU branch) */
static void primal_infeasibility(Work* w, const double* y, const double* sp, const double* sn,
                                 const double* aty, double dualObj, double* obj, double* res) {
  const int n = w->n, m = w->m;
  double scale = sqrt(o_dot(m, y, y) + o_dot(n, sp, sp) + o_dot(n, sn, sn));
  if (scale < 1e-8) scale = 1.0;
  *obj = (dualObj - w->offset) / w->sense / scale;
  double* lb = w->bufN;  /* scaled lb ray */
  double* ub = w->bufN2; /* scaled ub ray */
  double* c = w->bufMax2;
  memcpy(lb, sp, sizeof(double) * (size_t)n); o_scal(n, 1 / scale, lb);
  memcpy(ub, sn, sizeof(double) * (size_t)n); o_scal(n, 1 / scale, ub);
  memcpy(c, aty, sizeof(double) * (size_t)n); o_scal(n, 1.0 / scale, c);
  o_axpy(n, 1.0, lb, c);
  o_axpy(n, -1.0, ub, c);
  if (w->ifScaled) o_emul(n, c, w->colScale);
  *res = o_nrm2(n, c);
}
/* PDHG_Compute_Dual_Infeasibility :313-429 (CPU branch) */
static void dual_infeasibility(Work* w, const double* x, const double* ax, double primalObj,
                               double* obj, double* res) {
  const int n = w->n, m = w->m;
  double* ray = w->bufN;
  memcpy(ray, x, sizeof(double) * (size_t)n);
  double scale = o_nrm2(n, ray);
  if (scale < 1e-8) scale = 1.0;
  o_scal(n, 1.0 / scale, ray);
  *obj = (primalObj - w->offset) / w->sense / scale;
  double* c = w->bufM;
  memcpy(c, ax, sizeof(double) * (size_t)m);
  o_scal(m, 1.0 / scale, c);
  o_proj_neg(m - w->nEqs, c + w->nEqs);
  if (w->ifScaled) o_emul(m, c, w->rowScale);
  const double cSq = o_dot(m, c, c);
  double* b = w->bufN2;
  memcpy(b, ray, sizeof(double) * (size_t)n);
  o_proj_neg(n, b); o_emul(n, b, w->hasLower);
  if (w->ifScaled) o_ediv(n, b, w->colScale);
  const double lbSq = o_dot(n, b, b);
  memcpy(b, ray, sizeof(double) * (size_t)n);
  o_proj_pos(n, b); o_emul(n, b, w->hasUpper);
  if (w->ifScaled) o_ediv(n, b, w->colScale);
  const double ubSq = o_dot(n, b, b);
  *res = sqrt(cSq + lbSq + ubSq);
}
/* PDHG_Compute_Infeas_Residuals :433-471 */
static void compute_infeas_residuals(Work* w) {
  const int c = w->nIter % 2;
  if (w->gpuOrder) return; /* already produced by g_residuals */
  primal_infeasibility(w, w->y[c], w->slackPos, w->slackNeg, w->aty[c], w->dObj, &w->pInfObj, &w->pInfRes);
  dual_infeasibility(w, w->x[c], w->ax[c], w->pObj, &w->dInfObj, &w->dInfRes);
  primal_infeasibility(w, w->yAvg, w->slackPosAvg, w->slackNegAvg, w->atyAvg, w->dObjA, &w->pInfObjA, &w->pInfResA);
  dual_infeasibility(w, w->xAvg, w->axAvg, w->pObjA, &w->dInfObjA, &w->dInfResA);
}

/* ------------------------------------------------------------------ */
/* steps — cupdlp_step.c                                               */
/* ------------------------------------------------------------------ */
/* PDHG_primalGradientStep :16-40 (CPU branch: copy, 2 axpy, projub, projlb) */
static void primal_step(Work* w, double* xU, const double* x, const double* aty, const double* nx, double tau) {
  const int n = w->n;
  memcpy(xU, x, sizeof(double) * (size_t)n);
  o_axpy(n, -tau, w->cost, xU);
  o_axpy(n, tau, aty, xU);
  if (nx) o_axpy(n, -tau, nx, xU); /* explicit gradient term of the off-diagonal part of Q */
  if (w->qdiag) /* prox of the separable quadratic: argmin <c - A'y, x> + 1/2 q x^2 + (x - x_k)^2 / (2 tau) */
    for (int j = 0; j < n; ++j) xU[j] = xU[j] / (1.0 + tau * w->qdiag[j]);
  o_proj_ub_vec(n, xU, w->upper);
  o_proj_lb_vec(n, xU, w->lower);
}
/* PDHG_dualGradientStep :43-69 */
static void dual_step(Work* w, double* yU, const double* y, const double* ax, const double* axU, double sigma) {
  const int m = w->m;
  memcpy(yU, y, sizeof(double) * (size_t)m);
  o_axpy(m, sigma, w->rhs, yU);
  o_axpy(m, -2.0 * sigma, axU, yU);
  o_axpy(m, sigma, ax, yU);
  o_proj_pos(m - w->nEqs, yU + w->nEqs);
}
/* cupdlp_compute_interaction_and_movement linalg.c:772-801 (CPU branch) */
static void movement_interaction(Work* w, double* movement, double* interaction, double* qint) {
  const int n = w->n, m = w->m, c = w->nIter % 2, u = (w->nIter + 1) % 2;
  const double sb = sqrt(w->beta);
  *qint = 0.0;
  if (w->gpuOrder) {
    double dX2, dY2;
    g_trial_sums(w, &dX2, &dY2, interaction, qint);
    *movement = dX2 * 0.5 * sb + dY2 / (2.0 * sb);
    return;
  }
  double* d2 = w->bufMax2;
  double* d3 = w->bufMax3;
  memcpy(d2, w->x[c], sizeof(double) * (size_t)n); o_axpy(n, -1.0, w->x[u], d2);
  const double dX = o_dot(n, d2, d2);
  memcpy(d2, w->y[c], sizeof(double) * (size_t)m); o_axpy(m, -1.0, w->y[u], d2);
  const double dY = o_dot(m, d2, d2);
  memcpy(d2, w->x[c], sizeof(double) * (size_t)n); o_axpy(n, -1.0, w->x[u], d2);
  memcpy(d3, w->aty[c], sizeof(double) * (size_t)n); o_axpy(n, -1.0, w->aty[u], d3);
  *interaction = o_dot(n, d2, d3);
  *movement = dX * 0.5 * sb + dY / (2.0 * sb);
  if (w->qnBeg) { /* dx . N dx (d2 still holds x - x+) */
    memcpy(d3, w->nx[c], sizeof(double) * (size_t)n); o_axpy(n, -1.0, w->nx[u], d3);
    *qint = o_dot(n, d2, d3);
  }
}
/* PDHG_Update_Iterate_Adaptive_Step_Size :215-310; returns 1 on time-out */
static int update_adaptive(Work* w) {
  const int c = w->nIter % 2, u = (w->nIter + 1) % 2;
  double eta = sqrt(w->primalStep * w->dualStep);
  int done = 0;
  while (!done) {
    ++w->nStepSizeIter;
    const double tau = eta / sqrt(w->beta), sigma = eta * sqrt(w->beta);
    primal_step(w, w->x[u], w->x[c], w->aty[c], w->qnBeg ? w->nx[c] : NULL, tau);
    o_Ax(w, w->ax[u], w->x[u]);
    dual_step(w, w->y[u], w->y[c], w->ax[c], w->ax[u], sigma);
    o_ATy(w, w->aty[u], w->y[u]);
    if (w->qnBeg) o_Nx(w, w->nx[u], w->x[u]);
    double mov = 0.0, inter = 0.0, qint = 0.0;
    movement_interaction(w, &mov, &inter, &qint);
    /* QP with off-diagonal Hessian entries: the forward step on N also needs tau <= |dx|^2 / |dx . N dx| (pdlp_devfn.hpp decideCore) */
    const double den = fabs(inter) + 0.5 * fabs(qint);
    const double limit = (den != 0.0) ? mov / den : INFINITY;
    if (eta <= limit) {
      done = 1;
    } else {
      w->solveTime = o_now() - w->solveBeg; /* CUPDLP_CHECK_TIMEOUT cupdlp_solver.h:14-21 */
      if (w->solveTime > w->timeLim) return 1;
    }
    const double first = (1.0 - pow(w->nStepSizeIter + 1.0, -0.3)) * limit;  /* cupdlp_defs.h:29-31 */
    const double second = (1.0 + pow(w->nStepSizeIter + 1.0, -0.6)) * eta;
    eta = fmin(first, second);
  }
  w->primalStep = eta / sqrt(w->beta);
  w->dualStep = eta * sqrt(w->beta);
  return 0;
}
/* PDHG_Update_Iterate_Constant_Step_Size :178-206 */
static void update_constant(Work* w) {
  const int c = w->nIter % 2, u = (w->nIter + 1) % 2;
  o_Ax(w, w->ax[c], w->x[c]);
  o_ATy(w, w->aty[c], w->y[c]);
  if (w->qnBeg) o_Nx(w, w->nx[c], w->x[c]);
  primal_step(w, w->x[u], w->x[c], w->aty[c], w->qnBeg ? w->nx[c] : NULL, w->primalStep);
  o_Ax(w, w->ax[u], w->x[u]);
  dual_step(w, w->y[u], w->y[c], w->ax[c], w->ax[u], w->dualStep);
  o_ATy(w, w->aty[u], w->y[u]);
  if (w->qnBeg) o_Nx(w, w->nx[u], w->x[u]);
}
/* PDHG_Update_Average :422-442 — uses the step sizes ALREADY overwritten with the next eta */
static void update_average(Work* w) {
  const int u = (w->nIter + 1) % 2;
  const double wgt = sqrt(w->primalStep * w->dualStep);
  o_axpy(w->n, wgt, w->x[u], w->xSum);
  o_axpy(w->m, wgt, w->y[u], w->ySum);
  w->sumPrimalStep += wgt;
  w->sumDualStep += wgt;
}
/* PDHG_Compute_Average_Iterate :377-420 */
static void compute_average(Work* w) {
  const double ps = w->sumPrimalStep > 0.0 ? 1.0 / w->sumPrimalStep : 1.0;
  const double ds = w->sumDualStep > 0.0 ? 1.0 / w->sumDualStep : 1.0;
  memcpy(w->xAvg, w->xSum, sizeof(double) * (size_t)w->n);
  memcpy(w->yAvg, w->ySum, sizeof(double) * (size_t)w->m);
  o_scal(w->n, ps, w->xAvg);
  o_scal(w->m, ds, w->yAvg);
  o_Ax(w, w->axAvg, w->xAvg);
  o_ATy(w, w->atyAvg, w->yAvg);
  if (w->qnBeg) o_Nx(w, w->nxAvg, w->xAvg);
}
/* PDHG_Power_Method :71-145 (20 iterations; the logged residual is not restated) */
static double power_method(Work* w) {
  const int n = w->n, m = w->m, c = w->nIter % 2;
  double* q = w->bufM;
  double lambda = 0.0;
  for (int i = 0; i < m; ++i) q[i] = 1.0;
  for (int it = 0; it < 20; ++it) {
    o_ATy(w, w->aty[c], q);
    o_Ax(w, w->ax[c], w->aty[c]);
    memcpy(q, w->ax[c], sizeof(double) * (size_t)m);
    double qn;
    if (w->gpuOrder) { GDiffCtx cq = {q, q}; qn = sqrt(g_grid_sum(m, g_dot_elem, &cq, w->gStat)); }
    else qn = o_nrm2(m, q);
    o_scal(m, 1.0 / qn, q);
    o_ATy(w, w->aty[c], q);
    if (w->gpuOrder) { GDiffCtx ca = {w->aty[c], w->aty[c]}; lambda = g_grid_sum(n, g_dot_elem, &ca, w->gStat); }
    else lambda = o_dot(n, w->aty[c], w->aty[c]);
    o_axpy(m, -lambda, q, w->ax[c]);
  }
  return lambda;
}
/* PDHG_Init_Step_Sizes :312-375 */
static void init_step_sizes(Work* w) {
  double lambda = 0.0;
  if (!w->adaptive) lambda = power_method(w);
  const double a = o_dot(w->n, w->cost, w->cost), b = o_dot(w->m, w->rhs, w->rhs);
  w->beta = (fmin(a, b) > 1e-6) ? a / b : 1.0;
  if (!w->adaptive) {
    w->primalStep = 0.8 / sqrt(lambda);
    w->dualStep = w->primalStep;
    w->primalStep /= sqrt(w->beta);
    w->dualStep *= sqrt(w->beta);
  } else {
    w->primalStep = (1.0 / w->matNormInf) / sqrt(w->beta);
    w->dualStep = w->primalStep * w->beta;
  }
  w->iLastRestartIter = 0;
  w->sumPrimalStep = 0.0;
  w->sumDualStep = 0.0;
}
/* PDHG_Init_Variables cupdlp_solver.c:531-591 */
static void init_variables(Work* w, int hasStart) {
  const int n = w->n, m = w->m, c = w->nIter % 2;
  if (!hasStart) memset(w->x[c], 0, sizeof(double) * (size_t)n);
  o_proj_ub_vec(n, w->x[c], w->upper);
  o_proj_lb_vec(n, w->x[c], w->lower);
  if (!hasStart) memset(w->y[c], 0, sizeof(double) * (size_t)m);
  o_Ax(w, w->ax[c], w->x[c]);
  o_ATy(w, w->aty[c], w->y[c]);
  if (w->qnBeg) o_Nx(w, w->nx[c], w->x[c]);
  memset(w->xSum, 0, sizeof(double) * (size_t)n);
  memset(w->ySum, 0, sizeof(double) * (size_t)m);
  memset(w->xAvg, 0, sizeof(double) * (size_t)n);
  memset(w->yAvg, 0, sizeof(double) * (size_t)m);
  o_proj_ub_vec(n, w->xSum, w->upper); o_proj_lb_vec(n, w->xSum, w->lower); /* :583-584 */
  o_proj_ub_vec(n, w->xAvg, w->upper); o_proj_lb_vec(n, w->xAvg, w->lower);
  w->sumPrimalStep = 0.0;
  w->sumDualStep = 0.0;
  memset(w->xLast, 0, sizeof(double) * (size_t)n);
  memset(w->yLast, 0, sizeof(double) * (size_t)m);
}

/* ------------------------------------------------------------------ */
/* restart — cupdlp_restart.c, cupdlp_proj.c:88-148, cupdlp_step.c:147-176 */
/* ------------------------------------------------------------------ */
static double restart_score(double beta, double p, double d, double g) { /* restart.c:113-124 */
  return sqrt(beta * p * p + d * d / beta + g * g);
}
enum { NO_RESTART = 0, RESTART_TO_CURRENT = 1, RESTART_TO_AVERAGE = 2 };
static int check_restart(Work* w) { /* restart.c:3-99 */
  if (w->nIter == w->iLastRestartIter) {
    w->pFeasLR = w->pFeas; w->dFeasLR = w->dFeas; w->gapLR = w->gap;
    w->pFeasLC = w->pFeas; w->dFeasLC = w->dFeas; w->gapLC = w->gap;
    return NO_RESTART;
  }
  const double muCur = restart_score(w->beta, w->pFeas, w->dFeas, w->gap);
  const double muAvg = restart_score(w->beta, w->pFeasA, w->dFeasA, w->gapA);
  int choice = RESTART_TO_AVERAGE;
  double muCand;
  if (muCur < muAvg) { choice = RESTART_TO_CURRENT; muCand = muCur; } else { muCand = muAvg; }
  if ((w->nIter - w->iLastRestartIter) >= 0.36 * w->nIter) {
    /* artificial restart */
  } else {
    const double muLR = restart_score(w->beta, w->pFeasLR, w->dFeasLR, w->gapLR);
    if (muCand < 0.2 * muLR) {
      /* sufficient decay */
    } else {
      const double muLC = restart_score(w->beta, w->pFeasLC, w->dFeasLC, w->gapLC);
      if (muCand < 0.8 * muLR && muCand > muLC) {
        /* necessary decay */
      } else {
        choice = NO_RESTART;
      }
    }
  }
  if (muCur < muAvg) { w->pFeasLC = w->pFeas; w->dFeasLC = w->dFeas; w->gapLC = w->gap; }
  else { w->pFeasLC = w->pFeasA; w->dFeasLC = w->dFeasA; w->gapLC = w->gapA; }
  return choice;
}
/* PDHG_Compute_Step_Size_Ratio step.c:147-176 */
static void step_size_ratio(Work* w) {
  const int n = w->n, m = w->m, c = w->nIter % 2;
  const double mean = sqrt(w->primalStep * w->dualStep);
  double* d = w->bufMax2;
  double dP, dD;
  if (w->gpuOrder) {
    GDiffCtx cx = {w->x[c], w->xLast}, cy = {w->y[c], w->yLast};
    dP = sqrt(g_grid_sum(n, g_diff_elem, &cx, w->gStat));
    dD = sqrt(g_grid_sum(m, g_diff_elem, &cy, w->gStat));
  } else {
    memcpy(d, w->x[c], sizeof(double) * (size_t)n); o_axpy(n, -1.0, w->xLast, d);
    dP = o_nrm2(n, d);
    memcpy(d, w->y[c], sizeof(double) * (size_t)m); o_axpy(m, -1.0, w->yLast, d);
    dD = o_nrm2(m, d);
  }
  if (fmin(dP, dD) > 1e-10) {
    if (w->gpuOrder) { /* exp / log in plain IEEE arithmetic as the device computes them (det_math.h: the oracle's own restatement) */
      const double lg = 0.5 * o_det_log(dD / dP) + 0.5 * o_det_log(sqrt(w->beta));
      w->beta = o_det_exp(lg) * o_det_exp(lg);
    } else { /* the reference: libm (cupdlp_step.c:165-170) */
      const double lg = 0.5 * log(dD / dP) + 0.5 * log(sqrt(w->beta));
      w->beta = exp(lg) * exp(lg);
    }
  }
  w->primalStep = mean / sqrt(w->beta);
  w->dualStep = w->primalStep * w->beta;
}
/* PDHG_Restart_Iterate_GPU proj.c:88-148 */
static void restart_iterate(Work* w) {
  if (!w->restartOn) return;
  const int choice = check_restart(w);
  if (choice == NO_RESTART) return;
  const int n = w->n, m = w->m, c = w->nIter % 2;
  w->sumPrimalStep = 0.0; w->sumDualStep = 0.0;
  memset(w->xSum, 0, sizeof(double) * (size_t)n);
  memset(w->ySum, 0, sizeof(double) * (size_t)m);
  if (choice == RESTART_TO_AVERAGE) {
    w->pFeasLR = w->pFeasA; w->dFeasLR = w->dFeasA; w->gapLR = w->gapA;
    memcpy(w->x[c], w->xAvg, sizeof(double) * (size_t)n);
    memcpy(w->y[c], w->yAvg, sizeof(double) * (size_t)m);
    memcpy(w->ax[c], w->axAvg, sizeof(double) * (size_t)m);
    memcpy(w->aty[c], w->atyAvg, sizeof(double) * (size_t)n);
    if (w->qnBeg) memcpy(w->nx[c], w->nxAvg, sizeof(double) * (size_t)n);
  } else {
    w->pFeasLR = w->pFeas; w->dFeasLR = w->dFeas; w->gapLR = w->gap;
  }
  step_size_ratio(w);
  memcpy(w->xLast, w->x[c], sizeof(double) * (size_t)n);
  memcpy(w->yLast, w->y[c], sizeof(double) * (size_t)m);
  w->iLastRestartIter = w->nIter;
  ++w->nRestarts;
  compute_residuals(w);
}

/* ------------------------------------------------------------------ */
/* termination — cupdlp_solver.c:710-841                               */
/* ------------------------------------------------------------------ */
static int check_termination(const Work* w, int avg) {
  const double pf = avg ? w->pFeasA : w->pFeas, df = avg ? w->dFeasA : w->dFeas, rg = avg ? w->relGapA : w->relGap;
  return (pf < w->tolP * (1.0 + w->normRhs)) && (df < w->tolD * (1.0 + w->normCost)) && (rg < w->tolG);
}
static int check_infeasibility(const Work* w) { /* :710-795 */
  int t = 0;
  if (w->pInfObj > 0.0 && w->pInfRes < w->feasTol * w->pInfObj) t = 1;
  if (w->dInfObj < 0.0 && w->dInfRes < -w->feasTol * w->dInfObj) t = 1;
  if (w->pInfObjA > 0.0 && w->pInfResA < w->feasTol * w->pInfObjA) t = 1;
  if (w->dInfObjA < 0.0 && w->dInfResA < -w->feasTol * w->dInfObjA) t = 1;
  return t;
}

/* PDHG_Solve cupdlp_solver.c:899-1215 */
static void pdhg_solve(Work* w, int hasStart, pdlp_oracle_trace_fn trace, void* trace_ctx) {
  w->nIter = 0;
  w->solveBeg = o_now();
  w->termCode = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT;
  w->termIterate = 0;
  init_step_sizes(w);
  init_variables(w, hasStart);
  for (w->nIter = 0; w->nIter < w->iterLim; ++w->nIter) {
    w->solveTime = o_now() - w->solveBeg;
    int checking = (w->nIter < 10) || (w->nIter == w->iterLim - 1) || (w->solveTime > w->timeLim);
    checking = checking || (w->nIter % 40 == 0); /* CUPDLP_RELEASE_INTERVAL cupdlp_defs.h:39 */
    if (checking) {
      compute_average(w);
      compute_residuals(w);
      compute_infeas_residuals(w);
      if (trace) {
        pdlp_oracle_trace_t t = {w->nIter, w->nStepSizeIter, w->beta, w->primalStep, w->dualStep,
                                 w->pObj, w->dObj, w->pFeas, w->dFeas, w->pObjA, w->dObjA, w->pFeasA, w->dFeasA};
        trace(trace_ctx, &t);
      }
      if (check_termination(w, 0)) { w->termIterate = 0; w->termCode = PDLP_TERM_OPTIMAL; break; }
      if (check_termination(w, 1)) { /* :1022-1041 copy the average into the current slot */
        const int c = w->nIter % 2;
        memcpy(w->x[c], w->xAvg, sizeof(double) * (size_t)w->n);
        memcpy(w->y[c], w->yAvg, sizeof(double) * (size_t)w->m);
        memcpy(w->ax[c], w->axAvg, sizeof(double) * (size_t)w->m);
        memcpy(w->aty[c], w->atyAvg, sizeof(double) * (size_t)w->n);
        memcpy(w->slackPos, w->slackPosAvg, sizeof(double) * (size_t)w->n);
        memcpy(w->slackNeg, w->slackNegAvg, sizeof(double) * (size_t)w->n);
        w->termIterate = 1; w->termCode = PDLP_TERM_OPTIMAL; break;
      }
      if (check_infeasibility(w)) { w->termCode = PDLP_TERM_INFEASIBLE_OR_UNBOUNDED; break; }
      if (w->solveTime > w->timeLim) { w->termCode = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT; break; }
      if (w->nIter >= w->iterLim - 1) { w->termCode = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT; break; }
      restart_iterate(w);
    }
    /* PDHG_Update_Iterate step.c:444-478 */
    if (w->adaptive) {
      if (update_adaptive(w)) { w->termCode = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT; break; }
    } else {
      update_constant(w);
    }
    update_average(w);
  }
  w->solveTime = o_now() - w->solveBeg;
}

/* PDHG_PreSolve cupdlp_solver.c:1217-1279 */
static int presolve_hot_start(Work* w, const pdlp_problem_t* P) {
  if (!P->start_value_valid || !P->start_dual_valid) return 0;
  if (!P->start_col_value || !P->start_row_value || !P->start_row_dual) return 0;
  double* x = w->x[0];
  double* y = w->y[0];
  memset(x, 0, sizeof(double) * (size_t)w->n);
  memset(y, 0, sizeof(double) * (size_t)w->m);
  int j = 0;
  for (; j < w->n0; ++j) x[j] = P->start_col_value[j];
  for (int i = 0; i < w->m; ++i) {
    const double mu = w->rowType[i] == ROW_LEQ ? -1.0 : 1.0;
    y[w->rowNewIdx[i]] = w->sense * mu * P->start_row_dual[i];
    if (w->rowType[i] == ROW_BOUND) x[j++] = P->start_row_value[i];
  }
  if (w->ifScaled) { o_emul(w->n, x, w->colScale); o_emul(w->m, y, w->rowScale); }
  return 1;
}

/* PDHG_PostSolve cupdlp_solver.c:1281-1435 */
static void postsolve(Work* w, pdlp_result_t* R) {
  const int n = w->n, m = w->m, n0 = w->n0, c = w->nIter % 2;
  double *x = w->x[c], *y = w->y[c], *ax = w->ax[c], *aty = w->aty[c];
  if (w->ifScaled) {
    o_ediv(n, x, w->colScale);
    o_ediv(m, y, w->rowScale);
    o_emul(n, w->slackPos, w->colScale);
    o_emul(n, w->slackNeg, w->colScale);
    o_emul(m, ax, w->rowScale);
    o_emul(n, aty, w->colScale);
  }
  int cv = 0, cd = 0, rv = 0, rd = 0;
  if (R->col_value) { memcpy(R->col_value, x, sizeof(double) * (size_t)n0); cv = 1; }
  if (R->row_value) {
    for (int i = 0; i < m; ++i) R->row_value[i] = ax[w->rowNewIdx[i]];
    for (int i = 0, j = 0; i < m; ++i) {
      if (w->rowType[i] == ROW_LEQ) R->row_value[i] = -R->row_value[i];
      else if (w->rowType[i] == ROW_BOUND) { R->row_value[i] = R->row_value[i] + x[n0 + j]; ++j; }
    }
    rv = 1;
  }
  if (R->col_dual) {
    for (int j = 0; j < n0; ++j) R->col_dual[j] = w->slackPos[j] - w->slackNeg[j];
    o_scal(n0, w->sense, R->col_dual);
    cd = 1;
  }
  if (R->row_dual) {
    for (int i = 0; i < m; ++i) R->row_dual[i] = y[w->rowNewIdx[i]];
    o_scal(m, w->sense, R->row_dual);
    for (int i = 0; i < m; ++i)
      if (w->rowType[i] == ROW_LEQ) R->row_dual[i] = -R->row_dual[i];
    rd = 1;
  }
  R->value_valid = cv && rv;
  R->dual_valid = cd && rd;
}

/* Build everything up to and including PDHG_Alloc (CupdlpWrapper.cpp:104-176). */
static int work_setup(Work* w, const pdlp_problem_t* P, const pdlp_params_t* opt) {
  memset(w, 0, sizeof(*w));
  /* HiGHS answers LPs without rows / columns / nonzeros itself (solveUnconstrainedLp, HighsSolve.cpp:61-66); the
   * PDHG loop has nothing to iterate on there (1/max|a_ij| is the initial step size): refuse, like the product */
  {
    int any = 0;
    const long nnz0 = (P->num_col > 0 && P->a_start) ? P->a_start[P->num_col] : 0;
    for (long p = 0; p < nnz0 && !any; ++p) any = P->a_value[p] != 0.0;
    if (P->num_row == 0 || P->num_col == 0 || !any) return 1;
  }
  if (formulate(w, P)) return 1;
  const int n = w->n, m = w->m;
  /* Init_Scaling cupdlp_scaling.c:395-425: norms of the UNSCALED formulated data */
  w->colScale = dalloc(n); w->rowScale = dalloc(m);
  for (int j = 0; j < n; ++j) w->colScale[j] = 1.0;
  for (int i = 0; i < m; ++i) w->rowScale[i] = 1.0;
  w->normCost = o_nrm2(n, w->cost);
  w->normRhs = o_nrm2(m, w->rhs);
  /* PDHG_Scale_Data :233-393 */
  if (!(opt->features_off & PDLP_FEATURE_SCALING_OFF)) {
    scale_ruiz(w, 10);
    scale_pc(w, 1.0);
    w->ifScaled = 1;
  }
  /* problem_alloc CupdlpWrapper.cpp:517-585 */
  build_csr(w);
  w->matNormInf = 0.0;
  for (long p = 0; p < w->nnz; ++p) { const double a = fabs(w->cscVal[p]); if (a > w->matNormInf) w->matNormInf = a; }
  w->hasLower = dalloc(n); w->hasUpper = dalloc(n); w->lowerF = dalloc(n); w->upperF = dalloc(n);
  for (int j = 0; j < n; ++j) {
    w->hasLower[j] = w->lower[j] > -INFINITY ? 1.0 : 0.0;
    w->hasUpper[j] = w->upper[j] < INFINITY ? 1.0 : 0.0;
    w->lowerF[j] = w->lower[j] > -INFINITY ? w->lower[j] : 0.0; /* cupdlp_utils.c:885-886 */
    w->upperF[j] = w->upper[j] < INFINITY ? w->upper[j] : 0.0;
  }
  /* PDHG_Alloc cupdlp_utils.c:1045-1096 */
  const long mx = n > m ? n : m;
  for (int k = 0; k < 2; ++k) { w->x[k] = dalloc(n); w->y[k] = dalloc(m); w->ax[k] = dalloc(m); w->aty[k] = dalloc(n); }
  w->xAvg = dalloc(n); w->yAvg = dalloc(m); w->axAvg = dalloc(m); w->atyAvg = dalloc(n);
  w->xSum = dalloc(n); w->ySum = dalloc(m); w->xLast = dalloc(n); w->yLast = dalloc(m);
  w->slackPos = dalloc(n); w->slackNeg = dalloc(n); w->slackPosAvg = dalloc(n); w->slackNegAvg = dalloc(n);
  w->bufN = dalloc(n); w->bufN2 = dalloc(n); w->bufM = dalloc(m);
  w->bufMax2 = dalloc(mx < 2048 ? 2048 : mx); w->bufMax3 = dalloc(mx < 2048 ? 2048 : mx);
  /* defaults + user parameters: cupdlp_utils.c:813-832,889,981-997; CupdlpWrapper.cpp:642-717 */
  w->feasTol = 1e-8;
  w->pInfRes = w->dInfRes = w->pInfResA = w->dInfResA = 1.0;
  w->iterLim = opt->iter_limit;
  w->tolP = opt->primal_tol; w->tolD = opt->dual_tol; w->tolG = opt->gap_tol;
  w->timeLim = opt->time_limit;
  w->logLevel = opt->log_level;
  w->adaptive = (opt->features_off & PDLP_FEATURE_ADAPTIVE_STEP_OFF) ? 0 : 1;
  w->restartOn = (opt->features_off & PDLP_FEATURE_RESTART_OFF) ? 0 : 1;
  if (opt->restart_method == 0) w->restartOn = 0;
  w->gpuOrder = opt->reserved[0] == 1;
  if (w->gpuOrder) g_setup(w, opt->reserved[1]);
  return 0;
}

int pdlp_oracle_solve_traced(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R,
                             pdlp_oracle_trace_fn trace, void* trace_ctx) {
  if (!P || !opt || !R) return 1;
  Work w;
  const double t0 = o_now();
  if (work_setup(&w, P, opt)) { work_free(&w); return 1; }
  const double t1 = o_now();
  /* LP_SolvePDHG cupdlp_solver.c:1437-1498 */
  presolve_hot_start(&w, P);
  const int hasStart = (P->start_value_valid + P->start_dual_valid) != 0; /* :1465 */
  pdhg_solve(&w, hasStart, trace, trace_ctx);
  R->term_code = w.termCode;
  R->term_iterate = w.termIterate;
  R->num_iter = w.nIter;
  R->num_trials = w.nStepSizeIter;
  R->num_restarts = w.nRestarts;
  const int avg = (w.termCode == PDLP_TERM_OPTIMAL && w.termIterate == 1);
  R->primal_obj = avg ? w.pObjA : w.pObj;
  R->dual_obj = avg ? w.dObjA : w.dObj;
  R->primal_feas = avg ? w.pFeasA : w.pFeas;
  R->dual_feas = avg ? w.dFeasA : w.dFeas;
  R->rel_gap = avg ? w.relGapA : w.relGap;
  R->norm_rhs = w.normRhs;
  R->norm_cost = w.normCost;
  R->setup_seconds = t1 - t0;
  R->solve_seconds = w.solveTime;
  postsolve(&w, R);
  work_free(&w);
  return 0;
}

int pdlp_oracle_solve(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R) {
  return pdlp_oracle_solve_traced(P, opt, R, NULL, NULL);
}

/* ---- piecewise entry points for kernel-level parity tests --------------- */

int pdlp_oracle_formulate_scale(const pdlp_problem_t* P, const pdlp_params_t* opt,
                                pdlp_oracle_formulated_t* F) {
  Work w;
  if (work_setup(&w, P, opt)) { work_free(&w); return 1; }
  memset(F, 0, sizeof(*F));
  F->n = w.n; F->m = w.m; F->n_eqs = w.nEqs; F->nnz = w.nnz;
  F->csc_beg = w.cscBeg; F->csc_idx = w.cscIdx; F->csc_val = w.cscVal;
  F->csr_beg = w.csrBeg; F->csr_idx = w.csrIdx; F->csr_val = w.csrVal;
  F->cost = w.cost; F->rhs = w.rhs; F->lower = w.lower; F->upper = w.upper;
  F->col_scale = w.colScale; F->row_scale = w.rowScale;
  F->row_type = w.rowType; F->row_new_idx = w.rowNewIdx;
  F->norm_cost = w.normCost; F->norm_rhs = w.normRhs; F->mat_norm_inf = w.matNormInf;
  /* ownership of the arrays above moves to F; release the rest */
  w.cscBeg = w.cscIdx = w.csrBeg = w.csrIdx = w.rowType = w.rowNewIdx = NULL;
  w.cscVal = w.csrVal = w.cost = w.rhs = w.lower = w.upper = w.colScale = w.rowScale = NULL;
  work_free(&w);
  return 0;
}
void pdlp_oracle_free_formulated(pdlp_oracle_formulated_t* F) {
  free(F->csc_beg); free(F->csc_idx); free(F->csc_val);
  free(F->csr_beg); free(F->csr_idx); free(F->csr_val);
  free(F->cost); free(F->rhs); free(F->lower); free(F->upper);
  free(F->col_scale); free(F->row_scale); free(F->row_type); free(F->row_new_idx);
  memset(F, 0, sizeof(*F));
}

/* ax = A x over a CSR (left-to-right row sums): AxCPU semantics */
void pdlp_oracle_spmv_csr(int m, const int* beg, const int* idx, const double* val,
                          const double* x, double* out) {
  for (int i = 0; i < m; ++i) {
    double s = 0.0;
    for (int p = beg[i]; p < beg[i + 1]; ++p) s += val[p] * x[idx[p]];
    out[i] = s;
  }
}

/* The block boundaries of the slab layout as the device-order mode models them (gpu_order.h g_slab_cold + g_slab_blocks),
 * for the CPU test that compares them with the product's partition: blockBeg[0 .. nBlocks], returns nBlocks.
 * which = 0: the operand by rows (majorCost 2), 1: the transposed one (10). */
int pdlp_oracle_slab_blocks(int nMajor, int nMinor, const int* beg, const int* idx, int longLimit, int which, int* blockBeg) {
  int* cold = ialloc((long)nMajor + 1);
  int* count = ialloc((long)nMinor + 1);
  g_slab_cold(beg, idx, nMajor, nMinor, longLimit, cold, count);
  const int nB = g_slab_blocks(beg, cold, nMajor, nMinor, longLimit, which ? G_SLAB_MAJOR_COST_COLS : G_SLAB_MAJOR_COST_ROWS, blockBeg);
  free(cold); free(count);
  return nB;
}

/* exp and log of the oracle's device-order mode (det_math.h), for the accuracy test against libm and the comparison
 * with the product's functions (pdlp_mi355x_det_exp_log) */
void pdlp_oracle_det_exp_log(int n, const double* x, double* expOut, double* logOut) {
  for (int i = 0; i < n; ++i) { expOut[i] = o_det_exp(x[i]); logOut[i] = o_det_log(x[i]); }
}

/* the same in the product's summation order: majors with more than long_limit entries are cut into segment tasks
 * (gpu_order.h g_long_major_sum); long_limit = the stream's chunk (512 below 2^18 nonzeros, else 2048) or 256 in
 * the slab layout */
void pdlp_oracle_spmv_csr_device_order(int m, const int* beg, const int* idx, const double* val,
                                       const double* x, double* out, int long_limit) {
  for (int i = 0; i < m; ++i) out[i] = g_major_sum(beg, idx, val, x, i, long_limit);
}

/* One trial step of cupdlp_step.c:241-257 + linalg.c:772-801 on caller vectors
 * (all in the scaled, formulated space).  out3 = {dX^2, dY^2, interaction}. */
void pdlp_oracle_trial_step(const pdlp_oracle_formulated_t* F, double tau, double sigma,
                            const double* x, const double* y, const double* ax, const double* aty,
                            double* xU, double* yU, double* axU, double* atyU, double* out3) {
  const int n = F->n, m = F->m;
  for (int j = 0; j < n; ++j) {
    double v = x[j];
    v += -tau * F->cost[j];
    v += tau * aty[j];
    v = v < F->upper[j] ? v : F->upper[j];
    v = v > F->lower[j] ? v : F->lower[j];
    xU[j] = v;
  }
  pdlp_oracle_spmv_csr(m, F->csr_beg, F->csr_idx, F->csr_val, xU, axU);
  for (int i = 0; i < m; ++i) {
    double v = y[i];
    v += sigma * F->rhs[i];
    v += -2.0 * sigma * axU[i];
    v += sigma * ax[i];
    if (i >= F->n_eqs) v = v > 0.0 ? v : 0.0;
    yU[i] = v;
  }
  memset(atyU, 0, sizeof(double) * (size_t)n);
  for (int i = 0; i < m; ++i)
    for (int p = F->csr_beg[i]; p < F->csr_beg[i + 1]; ++p) atyU[F->csr_idx[p]] += F->csr_val[p] * yU[i];
  double dX = 0.0, dY = 0.0, in = 0.0;
  for (int j = 0; j < n; ++j) { const double d = x[j] - xU[j]; dX += d * d; }
  for (int i = 0; i < m; ++i) { const double d = y[i] - yU[i]; dY += d * d; }
  for (int j = 0; j < n; ++j) in += (x[j] - xU[j]) * (aty[j] - atyU[j]);
  out3[0] = dX; out3[1] = dY; out3[2] = in;
}
