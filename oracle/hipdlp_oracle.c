/*
 * hipdlp_oracle.c — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or
 * called from the product (highs_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the reference's SECOND PDLP path, solver="hipdlp"
 * (SURVEY §8(f)-2): restarted Halpern PDHG with reflection, fixed step sizes from
 * a power method and a PID-controlled primal weight.  Function by function it
 * follows the reference's CPU branch (the #else of every #ifdef CUPDLP_GPU):
 *
 *   solveLpHiPdlp            highs/pdlp/HiPdlpWrapper.cpp:26-141
 *   PDLPSolver::setup        highs/pdlp/hipdlp/pdhg.cc:1783-1874
 *   preprocessLp             pdhg.cc:152-357
 *   Scaling::*               highs/pdlp/hipdlp/scaling.cc:20-307
 *   initializeStepSizes      pdhg.cc:1944-1977, powerMethod :1529-1670 (AA' variant)
 *   solve                    pdhg.cc:494-707
 *   performHalpernPdhgStep   pdhg.cc:961-1018
 *   computeFixedPointError   pdhg.cc:709-739
 *   runConvergenceCheck      pdhg.cc:784-899, checkConvergence :1474-1527,
 *                            computePrimalFeasibility :1297-1320, computeDualSlacks :1322-1378,
 *                            computeDualFeasibility :1380-1412, computeDualObjective :1447-1472
 *   checkRestartCriteria     pdhg.cc:901-927 (factors restart.hpp:91-93)
 *   updatePrimalWeightAtRestart pdhg.cc:1979-2049
 *   unscaleSolution          pdhg.cc:1883-1897, scaling.cc:264-278
 *   postprocess              pdhg.cc:359-492
 *   linalg::ax / aTy / dot   highs/pdlp/hipdlp/linalg.cc:36-74
 *
 * PINNED against the reference binary run in the build container
 * (tests/golden/make_golden_hipdlp.py -> tests/golden/reference_hipdlp.json:
 * iteration counts, objectives and full solutions of the check instances).
 *
 * Second summation mode (opt->reserved[0] == 1, "device reduction order"): the HIP path's Halpern steps
 * are bit-identical to the loops below; its only arithmetic difference is the ORDER in which the
 * power method's dot products, the fixed-point error sums, the check statistics and the two restart
 * norms are added (vector-kernel grid sums, gpu_order.h).  In that mode this oracle follows a whole GPU
 * solve bit for bit (tests/test_gpu_hipdlp.py); majors longer than the SpMV chunk are summed block-strided as the
 * GPU does (h_dev_setup, gpu_order.h g_major_sum).
 *
 * Reference behaviours reproduced on purpose (they are what a drop-in must match):
 *  - the objective sense is NOT applied to the costs (pdhg.cc:171 only stores it, :481 uses
 *    it for col_dual): a maximisation LP is minimised;
 *  - on an iteration / time limit the returned x, y are the zero start (only a converged
 *    check writes the output vectors, pdhg.cc:866-877);
 *  - restart-off and the restart-strategy option are ignored (pdhg.cc:1844-1852).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "det_math.h"
#include "gpu_order.h"
#include "pdlp_oracle.h"

enum { H_EQ = 0, H_LEQ = 1, H_GEQ = 2, H_BOUND = 3, H_FREE = 4 };
#define H_CHECK_INTERVAL 40 /* PDHG_CHECK_INTERVAL, pdhg.cc:32 */

typedef struct {
  int n, m, n0, nEqs;
  long nnz;
  int *beg, *idx; /* CSC, rows ascending within a column (pdhg.cc:311) */
  double* val;
  double *cost, *lower, *upper, *rl, *ru, *colScale, *rowScale;
  int *ctype, *newIdx;
  unsigned char* isEq;
  int scaled;
  double cNorm, bNorm, offset;
  int sense;
  /* device reduction order only (h_dev_setup): row-wise copy with ascending columns and the chunk above which
   * the GPU sums a major block-strided (gpu_order.h g_major_sum); rBeg == NULL otherwise */
  int *rBeg, *rIdx;
  double* rVal;
  int chunkA, chunkAt;
} HLp;

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) abort(); return p; }
static double* dvec(long n) { double* p = (double*)xmalloc(sizeof(double) * (size_t)(n > 0 ? n : 1)); memset(p, 0, sizeof(double) * (size_t)(n > 0 ? n : 1)); return p; }

/* linalg::ax, linalg.cc:36-47 (scatter over the columns) */
static void h_ax(const HLp* L, const double* x, double* out) {
  if (L->rBeg) { /* same left-to-right order per row; rows longer than the chunk as the GPU sums them */
    for (int i = 0; i < L->m; ++i) out[i] = g_major_sum(L->rBeg, L->rIdx, L->rVal, x, i, L->chunkA);
    return;
  }
  for (int i = 0; i < L->m; ++i) out[i] = 0.0;
  for (int c = 0; c < L->n; ++c)
    for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) out[L->idx[p]] += L->val[p] * x[c];
}
/* linalg::aTy, linalg.cc:49-61 */
static void h_aty(const HLp* L, const double* y, double* out) {
  if (L->rBeg) {
    for (int c = 0; c < L->n; ++c) out[c] = g_major_sum(L->beg, L->idx, L->val, y, c, L->chunkAt);
    return;
  }
  for (int c = 0; c < L->n; ++c) {
    double s = 0.0;
    for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) s += L->val[p] * y[L->idx[p]];
    out[c] = s;
  }
}
/* device reduction order: row-wise copy of the (scaled) matrix and the long-major chunks of both operands */
static void h_dev_setup(HLp* L, int layoutMode) {
  const int n = L->n, m = L->m;
  const long nnz = n > 0 ? L->beg[n] : 0;
  L->rBeg = (int*)xmalloc(sizeof(int) * (size_t)(m + 1));
  L->rIdx = (int*)xmalloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  L->rVal = dvec(nnz);
  int* fill = (int*)xmalloc(sizeof(int) * (size_t)(m + 1));
  memset(fill, 0, sizeof(int) * (size_t)(m + 1));
  for (long p = 0; p < nnz; ++p) fill[L->idx[p]]++;
  int acc = 0;
  for (int i = 0; i < m; ++i) { L->rBeg[i] = acc; acc += fill[i]; fill[i] = L->rBeg[i]; }
  L->rBeg[m] = acc;
  for (int c = 0; c < n; ++c)
    for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) { const int q = fill[L->idx[p]]++; L->rIdx[q] = c; L->rVal[q] = L->val[p]; }
  free(fill);
  L->chunkA = g_long_major_chunk(L->rBeg, m, n, layoutMode);
  L->chunkAt = g_long_major_chunk(L->beg, n, m, layoutMode);
}
static double h_dot(const double* a, const double* b, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* ---- device reduction order (gpu_order.h) ------------------------------------------------ */
typedef struct { const double *a, *b, *c, *d, *e, *f; const unsigned char* flag; int scaled; } HCtx;
static double e_dot(const void* v, int i) { const HCtx* c = (const HCtx*)v; return c->a[i] * c->b[i]; }
static double e_diff2(const void* v, int i) { const HCtx* c = (const HCtx*)v; const double d = c->a[i] - c->b[i]; return d * d; }
static double e_diffdot(const void* v, int i) { const HCtx* c = (const HCtx*)v; return (c->a[i] - c->b[i]) * c->c[i]; }
static double g_sum(int len, g_elem_fn f, const HCtx* c) {
  double* scratch = (double*)xmalloc(sizeof(double) * (size_t)G_MAXGRID);
  const double r = g_grid_sum(len, f, c, scratch);
  free(scratch);
  return r;
}
static double x_dot(int dev, const double* a, const double* b, int n) {
  if (!dev) return h_dot(a, b, n);
  HCtx c = {a, b, 0, 0, 0, 0, 0, 0};
  return g_sum(n, e_dot, &c);
}
static double x_diff2(int dev, const double* a, const double* b, int n) {
  if (dev) { HCtx c = {a, b, 0, 0, 0, 0, 0, 0}; return g_sum(n, e_diff2, &c); }
  double s = 0.0;
  for (int i = 0; i < n; ++i) { const double d = a[i] - b[i]; s += d * d; }
  return s;
}

typedef struct { int row; double val; } Ent;
static int entCmp(const void* a, const void* b) {
  const Ent *x = (const Ent*)a, *y = (const Ent*)b;
  if (x->row != y->row) return x->row < y->row ? -1 : 1;
  return x->val < y->val ? -1 : (x->val > y->val ? 1 : 0);
}

/* preprocessLp, pdhg.cc:152-357 */
static void h_preprocess(const pdlp_problem_t* P, HLp* L) {
  const int m = P->num_row, n0 = P->num_col;
  memset(L, 0, sizeof(*L));
  L->offset = P->offset;
  L->sense = P->sense >= 0 ? 1 : -1;
  L->ctype = (int*)xmalloc(sizeof(int) * (size_t)m);
  L->newIdx = (int*)xmalloc(sizeof(int) * (size_t)m);
  int newCols = 0, nEqs = 0;
  for (int i = 0; i < m; ++i) {
    const int hasL = P->row_lower[i] > -INFINITY, hasU = P->row_upper[i] < INFINITY;
    if (hasL && hasU) {
      if (P->row_lower[i] == P->row_upper[i]) { L->ctype[i] = H_EQ; ++nEqs; }
      else { L->ctype[i] = H_BOUND; ++newCols; ++nEqs; }
    } else if (hasL) L->ctype[i] = H_GEQ;
    else if (hasU) L->ctype[i] = H_LEQ;
    else { L->ctype[i] = H_FREE; ++newCols; ++nEqs; }
  }
  const int n = n0 + newCols;
  L->n = n; L->m = m; L->n0 = n0; L->nEqs = nEqs;
  int e = 0, q = nEqs;
  for (int i = 0; i < m; ++i) {
    const int t = L->ctype[i];
    L->newIdx[i] = (t == H_EQ || t == H_BOUND || t == H_FREE) ? e++ : q++;
  }
  L->isEq = (unsigned char*)xmalloc((size_t)m + 1);
  for (int i = 0; i < m; ++i) {
    const int t = L->ctype[i];
    L->isEq[L->newIdx[i]] = (t == H_EQ || t == H_BOUND || t == H_FREE);
  }
  L->cost = dvec(n); L->lower = dvec(n); L->upper = dvec(n); L->rl = dvec(m); L->ru = dvec(m);
  for (int j = 0; j < n0; ++j) { L->cost[j] = P->col_cost[j]; L->lower[j] = P->col_lower[j]; L->upper[j] = P->col_upper[j]; }
  for (int i = 0, s = n0; i < m; ++i)
    if (L->ctype[i] == H_BOUND || L->ctype[i] == H_FREE) {
      L->cost[s] = 0.0; L->lower[s] = P->row_lower[i]; L->upper[s] = P->row_upper[i]; ++s;
    }
  for (int i = 0; i < m; ++i) {
    const int r = L->newIdx[i];
    switch (L->ctype[i]) {
      case H_EQ: L->rl[r] = P->row_lower[i]; L->ru[r] = P->row_upper[i]; break;
      case H_GEQ: L->rl[r] = P->row_lower[i]; L->ru[r] = INFINITY; break;
      case H_LEQ: L->rl[r] = -P->row_upper[i]; L->ru[r] = INFINITY; break;
      default: L->rl[r] = 0.0; L->ru[r] = 0.0; break;
    }
  }
  const long nnz0 = n0 > 0 ? P->a_start[n0] : 0;
  L->nnz = nnz0 + newCols;
  L->beg = (int*)xmalloc(sizeof(int) * ((size_t)n + 1));
  L->idx = (int*)xmalloc(sizeof(int) * (size_t)(L->nnz + 1));
  L->val = dvec(L->nnz + 1);
  long w = 0;
  L->beg[0] = 0;
  int maxLen = 1;
  for (int c = 0; c < n0; ++c) if (P->a_start[c + 1] - P->a_start[c] > maxLen) maxLen = P->a_start[c + 1] - P->a_start[c];
  Ent* tmp = (Ent*)xmalloc(sizeof(Ent) * (size_t)maxLen);
  for (int c = 0; c < n0; ++c) {
    int k = 0;
    for (int p = P->a_start[c]; p < P->a_start[c + 1]; ++p) {
      const int orow = P->a_index[p];
      double v = P->a_value[p];
      if (L->ctype[orow] == H_LEQ) v = -v;
      tmp[k].row = L->newIdx[orow]; tmp[k].val = v; ++k;
    }
    qsort(tmp, (size_t)k, sizeof(Ent), entCmp);
    for (int t = 0; t < k; ++t) { L->idx[w] = tmp[t].row; L->val[w] = tmp[t].val; ++w; }
    L->beg[c + 1] = (int)w;
  }
  free(tmp);
  for (int i = 0, s = n0; i < m; ++i)
    if (L->ctype[i] == H_BOUND || L->ctype[i] == H_FREE) {
      L->idx[w] = L->newIdx[i]; L->val[w] = -1.0; ++w;
      L->beg[++s] = (int)w;
    }
  L->cNorm = sqrt(h_dot(L->cost, L->cost, n));
  L->bNorm = sqrt(h_dot(L->rl, L->rl, m));
  L->colScale = dvec(n); L->rowScale = dvec(m);
  for (int j = 0; j < n; ++j) L->colScale[j] = 1.0;
  for (int i = 0; i < m; ++i) L->rowScale[i] = 1.0;
}

static void h_free(HLp* L) {
  free(L->beg); free(L->idx); free(L->val); free(L->cost); free(L->lower); free(L->upper); free(L->rl); free(L->ru);
  free(L->colScale); free(L->rowScale); free(L->ctype); free(L->newIdx); free(L->isEq);
  free(L->rBeg); free(L->rIdx); free(L->rVal);
}

/* Scaling::applyScaling, scaling.cc:222-262, + the cumulative update */
static void h_apply_scaling(HLp* L, const double* cs, const double* rs) {
  for (int j = 0; j < L->n; ++j) L->cost[j] /= cs[j];
  for (int j = 0; j < L->n; ++j) {
    if (L->lower[j] > -INFINITY) L->lower[j] *= cs[j];
    if (L->upper[j] < INFINITY) L->upper[j] *= cs[j];
  }
  for (int i = 0; i < L->m; ++i) {
    if (L->rl[i] > -INFINITY) L->rl[i] /= rs[i];
    if (L->ru[i] < INFINITY) L->ru[i] /= rs[i];
  }
  for (int c = 0; c < L->n; ++c)
    for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) L->val[p] /= (rs[L->idx[p]] * cs[c]);
  for (int j = 0; j < L->n; ++j) L->colScale[j] *= cs[j];
  for (int i = 0; i < L->m; ++i) L->rowScale[i] *= rs[i];
}

/* Scaling::scaleProblem, scaling.cc:31-57 */
static void h_scale(HLp* L, int ruiz, int pc, int l2, int ruizIters) {
  double* cs = dvec(L->n);
  double* rs = dvec(L->m);
  L->scaled = 0;
  if (ruiz) { /* applyRuizScaling, :59-125 (infinity norm) */
    for (int it = 0; it < ruizIters; ++it) {
      for (int i = 0; i < L->m; ++i) rs[i] = 0.0;
      for (int c = 0; c < L->n; ++c) {
        double mx = 0.0;
        for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) mx = fmax(mx, fabs(L->val[p]));
        cs[c] = (L->beg[c + 1] > L->beg[c]) ? sqrt(mx) : 0.0;
        if (cs[c] == 0.0) cs[c] = 1.0;
      }
      for (int c = 0; c < L->n; ++c)
        for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) rs[L->idx[p]] = fmax(rs[L->idx[p]], fabs(L->val[p]));
      for (int i = 0; i < L->m; ++i) rs[i] = rs[i] == 0.0 ? 1.0 : sqrt(rs[i]);
      h_apply_scaling(L, cs, rs);
    }
    L->scaled = 1;
  }
  if (pc) { /* applyPockChambolleScaling, :127-178 with alpha = 1 (pow(v,1) == v) */
    for (int i = 0; i < L->m; ++i) rs[i] = 0.0;
    for (int c = 0; c < L->n; ++c) {
      double s = 0.0;
      for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) s += fabs(L->val[p]);
      cs[c] = s > 0.0 ? sqrt(s) : 1.0;
    }
    for (int c = 0; c < L->n; ++c)
      for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) rs[L->idx[p]] += fabs(L->val[p]);
    for (int i = 0; i < L->m; ++i) rs[i] = rs[i] > 0.0 ? sqrt(rs[i]) : 1.0;
    h_apply_scaling(L, cs, rs);
    L->scaled = 1;
  }
  if (l2) { /* applyL2Scaling, :180-220 */
    for (int i = 0; i < L->m; ++i) rs[i] = 0.0;
    for (int c = 0; c < L->n; ++c) {
      double s = 0.0;
      for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) s += L->val[p] * L->val[p];
      cs[c] = s > 0.0 ? sqrt(sqrt(s)) : 1.0;
    }
    for (int c = 0; c < L->n; ++c)
      for (int p = L->beg[c]; p < L->beg[c + 1]; ++p) rs[L->idx[p]] += L->val[p] * L->val[p];
    for (int i = 0; i < L->m; ++i) rs[i] = rs[i] > 0.0 ? sqrt(sqrt(rs[i])) : 1.0;
    h_apply_scaling(L, cs, rs);
    L->scaled = 1;
  }
  free(cs); free(rs);
}

/* powerMethod, pdhg.cc:1529-1670, kCuPdlpAATPowerMethod */
static double h_power_method(const HLp* L, int dev) {
  if (L->n == 0 || L->m == 0) return 1.0;
  double* x = dvec(L->m);
  double* y = dvec(L->n);
  double* z = dvec(L->m);
  for (int i = 0; i < L->m; ++i) x[i] = 1.0;
  double lambda = 0.0;
  for (int it = 0; it < 20; ++it) {
    h_aty(L, x, y);
    h_ax(L, y, z);
    const double zn = sqrt(x_dot(dev, z, z, L->m));
    for (int i = 0; i < L->m; ++i) z[i] /= zn;
    h_aty(L, z, y);
    lambda = x_dot(dev, y, y, L->n);
    memcpy(x, z, sizeof(double) * (size_t)L->m);
  }
  free(x); free(y); free(z);
  return lambda;
}

typedef struct {
  double pObj, dObj, gap, relGap, pFeas, dFeas;
} HRes;

/* checkConvergence, pdhg.cc:1474-1527.  cachedSlack != NULL: the major-step dual slack
 * (computeDualSlacks :1336-1346), else the sign projection. */
/* element functions of the device-order check (k_h_row_stats / k_h_col_stats, pdlp_halpern.hip) */
static double e_row0(const void* v, int i) {
  const HCtx* c = (const HCtx*)v; /* a=ax b=rl c=rowScale flag=isEq */
  double r = c->a[i] - c->b[i];
  if (!c->flag[i]) r = r < 0.0 ? r : 0.0;
  if (c->scaled) r *= c->c[i];
  return r * r;
}
static double e_row1(const void* v, int i) { const HCtx* c = (const HCtx*)v; return c->b[i] * c->d[i]; } /* rl*y */
typedef struct { const HLp* L; const double *x, *aty, *sp, *sn; } HColCtx;
static double e_col0(const void* v, int j) {
  const HColCtx* c = (const HColCtx*)v;
  double t = (c->L->cost[j] - c->aty[j]) - c->sp[j] + c->sn[j];
  if (c->L->scaled) t *= c->L->colScale[j];
  return t * t;
}
static double e_col1(const void* v, int j) { const HColCtx* c = (const HColCtx*)v; return c->L->cost[j] * c->x[j]; }
static double e_col2(const void* v, int j) { const HColCtx* c = (const HColCtx*)v; return c->L->lower[j] > -INFINITY ? c->L->lower[j] * c->sp[j] : 0.0; }
static double e_col3(const void* v, int j) { const HColCtx* c = (const HColCtx*)v; return c->L->upper[j] < INFINITY ? c->L->upper[j] * c->sn[j] : 0.0; }

static int h_check(const HLp* L, const double* x, const double* y, const double* ax, const double* aty,
                   const double* cachedSlack, double eps, HRes* r, double* sp, double* sn, int dev) {
  if (dev) { /* same quantities, sums in the kernels' order and the host's combination (pdlp_halpern.cpp check()) */
    for (int j = 0; j < L->n; ++j) {
      const double dr = L->cost[j] - aty[j];
      double ds = 0.0;
      if (cachedSlack) ds = cachedSlack[j];
      else {
        const int hasL = L->lower[j] > -INFINITY, hasU = L->upper[j] < INFINITY;
        if (hasL && hasU) ds = dr;
        else if (hasL) ds = dr > 0.0 ? dr : 0.0;
        else if (hasU) ds = dr < 0.0 ? dr : 0.0;
      }
      sp[j] = ds > 0.0 ? ds : 0.0;
      sn[j] = -ds > 0.0 ? -ds : 0.0;
    }
    HCtx rc = {ax, L->rl, L->rowScale, y, 0, 0, L->isEq, L->scaled};
    HColCtx cc = {L, x, aty, sp, sn};
    double* scratch = (double*)xmalloc(sizeof(double) * (size_t)G_MAXGRID);
    const double rs0 = g_grid_sum(L->m, e_row0, &rc, scratch), rs1 = g_grid_sum(L->m, e_row1, &rc, scratch);
    const double cs0 = g_grid_sum(L->n, e_col0, &cc, scratch), cs1 = g_grid_sum(L->n, e_col1, &cc, scratch);
    const double cs2 = g_grid_sum(L->n, e_col2, &cc, scratch), cs3 = g_grid_sum(L->n, e_col3, &cc, scratch);
    free(scratch);
    r->pFeas = sqrt(rs0);
    r->dFeas = sqrt(cs0);
    r->pObj = L->offset + cs1;
    r->dObj = ((L->offset + rs1) + cs2) - cs3;
    const double g = r->pObj - r->dObj;
    r->gap = fabs(g);
    r->relGap = fabs(g) / (1.0 + fabs(r->pObj) + fabs(r->dObj));
    return r->pFeas < eps * (1.0 + L->bNorm) && r->dFeas < eps * (1.0 + L->cNorm) && r->relGap < eps;
  }
  double s = 0.0;
  for (int i = 0; i < L->m; ++i) {
    double v = ax[i] - L->rl[i];
    if (!L->isEq[i]) v = fmin(0.0, v);
    if (L->scaled) v *= L->rowScale[i];
    s += v * v;
  }
  r->pFeas = sqrt(s);
  s = 0.0;
  for (int j = 0; j < L->n; ++j) {
    const double dr = L->cost[j] - aty[j];
    double ds = 0.0;
    if (cachedSlack) ds = cachedSlack[j];
    else {
      const int hasL = L->lower[j] > -INFINITY, hasU = L->upper[j] < INFINITY;
      if (hasL && hasU) ds = dr;
      else if (hasL) ds = fmax(0.0, dr);
      else if (hasU) ds = fmin(0.0, dr);
    }
    sp[j] = fmax(0.0, ds);
    sn[j] = fmax(0.0, -ds);
    double v = dr - sp[j] + sn[j];
    if (L->scaled) v *= L->colScale[j];
    s += v * v;
  }
  r->dFeas = sqrt(s);
  double po = L->offset;
  for (int j = 0; j < L->n; ++j) po += L->cost[j] * x[j];
  double dobj = L->offset;
  for (int i = 0; i < L->m; ++i) dobj += L->rl[i] * y[i];
  for (int j = 0; j < L->n; ++j) if (L->lower[j] > -INFINITY) dobj += L->lower[j] * sp[j];
  for (int j = 0; j < L->n; ++j) if (L->upper[j] < INFINITY) dobj -= L->upper[j] * sn[j];
  r->pObj = po; r->dObj = dobj;
  const double g = po - dobj;
  r->gap = fabs(g);
  r->relGap = fabs(g) / (1.0 + fabs(po) + fabs(dobj));
  return r->pFeas < eps * (1.0 + L->bNorm) && r->dFeas < eps * (1.0 + L->cNorm) && r->relGap < eps;
}

typedef struct {
  HLp L;
  double *xc, *yc, *xn, *yn, *rx, *ry, *xa, *ya, *aty, *axn, *slack, *sp, *sn;
  double tau, sigma, eta, omega, beta, pw, bestPw, bestGap, errSum, lastErr;
  int hIter, slackValid, dev;
} HState;

/* performHalpernPdhgStep, pdhg.cc:961-1018 */
static void h_step(HState* S, int major, int kOff) {
  const HLp* L = &S->L;
  const double ps = S->tau, ds = S->sigma, rho = 1.0; /* halpern_gamma, defs.hpp:71 / pdhg.cc:1913 */
  const int k = S->hIter + kOff;
  const double w = (double)k / (k + 1.0);
  if (major) S->slackValid = 1;
  for (int j = 0; j < L->n; ++j) {
    const double temp = S->xc[j] - ps * (L->cost[j] - S->aty[j]);
    const double proj = fmax(L->lower[j], fmin(temp, L->upper[j]));
    if (major) { S->xn[j] = proj; S->slack[j] = (proj - temp) / ps; }
    S->rx[j] = 2.0 * proj - S->xc[j];
  }
  h_ax(L, S->rx, S->axn);
  for (int i = 0; i < L->m; ++i) {
    const double temp = S->yc[i] / ds - S->axn[i];
    const double lo = -L->ru[i], up = -L->rl[i];
    const double proj = fmax(lo, fmin(temp, up));
    const double pd = (temp - proj) * ds;
    if (major) S->yn[i] = pd;
    S->ry[i] = 2.0 * pd - S->yc[i];
  }
  for (int j = 0; j < L->n; ++j) {
    const double bl = rho * S->rx[j] + (1.0 - rho) * S->xc[j];
    S->xc[j] = w * bl + (1.0 - w) * S->xa[j];
  }
  for (int i = 0; i < L->m; ++i) {
    const double bl = rho * S->ry[i] + (1.0 - rho) * S->yc[i];
    S->yc[i] = w * bl + (1.0 - w) * S->ya[i];
  }
  h_aty(L, S->yc, S->aty);
}

/* computeFixedPointError, pdhg.cc:709-739 */
static double h_fpe(const HState* S) {
  const HLp* L = &S->L;
  double pn = 0.0, dn = 0.0, cross = 0.0;
  double* dx = dvec(L->n);
  double* dy = dvec(L->m);
  double* atd = dvec(L->n);
  for (int j = 0; j < L->n; ++j) { dx[j] = S->xn[j] - S->rx[j]; pn += dx[j] * dx[j]; }
  for (int i = 0; i < L->m; ++i) { dy[i] = S->yn[i] - S->ry[i]; dn += dy[i] * dy[i]; }
  h_aty(L, dy, atd);
  for (int j = 0; j < L->n; ++j) cross += dx[j] * atd[j];
  if (S->dev) { /* k_h_fpe_rows / k_h_fpe_cols */
    HCtx c = {S->xn, S->rx, atd, 0, 0, 0, 0, 0};
    dn = x_diff2(1, S->yn, S->ry, L->m);
    pn = x_diff2(1, S->xn, S->rx, L->n);
    cross = g_sum(L->n, e_diffdot, &c);
  }
  free(dx); free(dy); free(atd);
  const double movement = pn * S->omega + dn / S->omega;
  const double interaction = 2.0 * S->eta * cross;
  return sqrt(fmax(0.0, movement + interaction));
}

/* updatePrimalWeightAtRestart, pdhg.cc:1979-2049 (k_p 0.99, k_i 0.01, k_d 0, i_smooth 0.3) */
static void h_update_weight(HState* S, const HRes* res) {
  const HLp* L = &S->L;
  double pd = 0.0, dd = 0.0;
  pd = x_diff2(S->dev, S->xn, S->xa, L->n);
  dd = x_diff2(S->dev, S->yn, S->ya, L->m);
  pd = sqrt(pd); dd = sqrt(dd);
  const double relP = res->pFeas / (1.0 + L->bNorm), relD = res->dFeas / (1.0 + L->cNorm);
  const double ratio = relP > 0.0 ? relD / relP : 1e300;
  if (pd > 1e-16 && dd > 1e-16 && pd < 1e12 && dd < 1e12 && ratio > 1e-8 && ratio < 1e8) {
    /* device-order mode: the product's check iteration runs on the device and computes log / exp in plain arithmetic
     * (det_math.h is this oracle's own restatement of those functions); the reference, and the serial mode, call libm */
    const double err = S->dev ? o_det_log(dd) - o_det_log(pd) - o_det_log(S->pw) : log(dd) - log(pd) - log(S->pw);
    S->errSum = 0.3 * S->errSum + err;
    const double dErr = err - S->lastErr;
    const double arg = 0.99 * err + 0.01 * S->errSum + 0.0 * dErr;
    S->pw *= S->dev ? o_det_exp(arg) : exp(arg);
    S->lastErr = err;
  } else {
    S->pw = S->bestPw; S->errSum = 0.0; S->lastErr = 0.0;
  }
  const double gap = (relP > 0.0 && relD > 0.0)
                         ? (S->dev ? fabs(o_det_log(relD / relP) * 0.4342944819032518 /* 1 / ln 10 */) : fabs(log10(relD / relP)))
                         : S->bestGap;
  if (gap < S->bestGap) { S->bestGap = gap; S->bestPw = S->pw; }
  const double eta = sqrt(S->tau * S->sigma);
  S->beta = S->pw * S->pw;
  S->tau = eta / S->pw;
  S->sigma = eta * S->pw;
  S->omega = sqrt(S->beta); /* params_.omega = primal_weight_, then updateBeta: omega = sqrt(beta) */
}

static double nowSec(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Probe for the parity tests: state after `steps` Halpern steps of the first block (no check,
 * no restart), optionally with imposed step sizes. */
typedef struct hipdlp_oracle_probe {
  int steps;             /* 1..40 */
  double tau, sigma;     /* > 0: override the power-method step sizes */
  double *x_cur, *y_cur, *x_next, *y_next; /* [n], [m] out (may be NULL) */
  double out_tau, out_sigma, out_fpe, out_lambda;
  int n, m;
} hipdlp_oracle_probe_t;

static int h_solve(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R, hipdlp_oracle_probe_t* probe) {
  const double t0 = nowSec();
  HState S;
  memset(&S, 0, sizeof(S));
  HLp* L = &S.L;
  h_preprocess(P, L);
  const int scalingOn = !(opt->features_off & PDLP_FEATURE_SCALING_OFF);
  const int mode = opt->scaling_mode;
  if (scalingOn) h_scale(L, mode & 1, mode & 4, mode & 2, opt->ruiz_iterations);
  const int n = L->n, m = L->m;
  const double eps = opt->gap_tol;
  const int pid = opt->step_size_strategy != 0;
  S.dev = opt->reserved[0] == 1;
  if (S.dev) h_dev_setup(L, opt->reserved[1]);
  /* initializeStepSizes */
  S.omega = (L->cNorm + 1.0) / (L->bNorm + 1.0);
  S.pw = S.omega; S.bestPw = S.pw; S.beta = S.pw * S.pw;
  const double lambda = h_power_method(L, S.dev);
  const double base = 0.998 / sqrt(lambda);
  S.eta = base; S.tau = base / S.omega; S.sigma = base * S.omega;
  S.bestGap = INFINITY;
  if (probe && probe->tau > 0.0) { S.tau = probe->tau; S.sigma = probe->sigma; }
  S.xc = dvec(n); S.yc = dvec(m); S.xn = dvec(n); S.yn = dvec(m); S.rx = dvec(n); S.ry = dvec(m);
  S.xa = dvec(n); S.ya = dvec(m); S.aty = dvec(n); S.axn = dvec(m); S.slack = dvec(n); S.sp = dvec(n); S.sn = dvec(n);
  double* outX = dvec(n);
  double* outY = dvec(m);
  double* axT = dvec(m);
  double* atyT = dvec(n);
  /* projectBounds(x = 0), linalg.cc:23-34 */
  for (int j = 0; j < n; ++j) {
    if (S.xc[j] > L->upper[j]) S.xc[j] = L->upper[j];
    if (S.xc[j] < L->lower[j]) S.xc[j] = L->lower[j];
  }
  memcpy(S.xa, S.xc, sizeof(double) * (size_t)n);
  memcpy(S.ya, S.yc, sizeof(double) * (size_t)m);
  h_ax(L, S.xc, axT);
  h_aty(L, S.yc, S.aty);
  const double tLoop = nowSec();
  int term = -1; /* -1 not set; 0 optimal; 1 maxiter; 2 timeout */
  int iters = 0, restarts = 0, doRestart = 0;
  double fpe = 0.0, initFpe = 0.0, lastFpe = INFINITY;
  HRes res;
  memset(&res, 0, sizeof(res));
  if (h_check(L, S.xc, S.yc, axT, S.aty, NULL, eps, &res, S.sp, S.sn, S.dev)) {
    memcpy(outX, S.xc, sizeof(double) * (size_t)n);
    memcpy(outY, S.yc, sizeof(double) * (size_t)m);
    term = 0;
  }
  while (term < 0 && iters < opt->iter_limit) {
    if (nowSec() - t0 > opt->time_limit) { term = 2; break; }
    h_step(&S, 1, 1);
    if (doRestart) { fpe = h_fpe(&S); initFpe = fpe; doRestart = 0; }
    if (probe && probe->steps < H_CHECK_INTERVAL) {
      for (int i = 2; i <= probe->steps; ++i) h_step(&S, i == probe->steps, i);
      break;
    }
    for (int i = 2; i <= H_CHECK_INTERVAL - 1; ++i) h_step(&S, 0, i);
    h_step(&S, 1, H_CHECK_INTERVAL);
    fpe = h_fpe(&S);
    if (probe) break;
    S.hIter += H_CHECK_INTERVAL;
    iters += H_CHECK_INTERVAL;
    h_ax(L, S.xn, axT);
    h_aty(L, S.yn, atyT);
    if (h_check(L, S.xn, S.yn, axT, atyT, S.slackValid ? S.slack : NULL, eps, &res, S.sp, S.sn, S.dev)) {
      memcpy(outX, S.xn, sizeof(double) * (size_t)n);
      memcpy(outY, S.yn, sizeof(double) * (size_t)m);
      term = 0;
      break;
    }
    /* checkRestartCriteria, pdhg.cc:901-927 */
    doRestart = 0;
    if (iters == H_CHECK_INTERVAL) doRestart = 1;
    else if (iters > H_CHECK_INTERVAL) {
      if (fpe <= 0.2 * initFpe) doRestart = 1;
      else if (fpe <= 0.8 * initFpe && fpe > lastFpe) doRestart = 1;
      else if (S.hIter >= 0.36 * iters) doRestart = 1;
    }
    lastFpe = fpe;
    if (doRestart) {
      if (pid) h_update_weight(&S, &res);
      memcpy(S.xa, S.xn, sizeof(double) * (size_t)n);
      memcpy(S.ya, S.yn, sizeof(double) * (size_t)m);
      memcpy(S.xc, S.xn, sizeof(double) * (size_t)n);
      memcpy(S.yc, S.yn, sizeof(double) * (size_t)m);
      h_aty(L, S.yc, S.aty);
      S.hIter = 0;
      lastFpe = INFINITY;
      ++restarts;
    }
  }
  if (probe) {
    probe->n = n; probe->m = m;
    if (probe->x_cur) memcpy(probe->x_cur, S.xc, sizeof(double) * (size_t)n);
    if (probe->y_cur) memcpy(probe->y_cur, S.yc, sizeof(double) * (size_t)m);
    if (probe->x_next) memcpy(probe->x_next, S.xn, sizeof(double) * (size_t)n);
    if (probe->y_next) memcpy(probe->y_next, S.yn, sizeof(double) * (size_t)m);
    probe->out_tau = S.tau; probe->out_sigma = S.sigma; probe->out_fpe = fpe; probe->out_lambda = lambda;
  }
  if (term < 0) term = 1;
  /* unscaleSolution + postprocess */
  if (R) {
    if (L->scaled) {
      for (int j = 0; j < n; ++j) outX[j] /= L->colScale[j];
      for (int i = 0; i < m; ++i) outY[i] /= L->rowScale[i];
    }
    for (int j = 0; j < n; ++j) { S.sp[j] *= L->colScale[j]; S.sn[j] *= L->colScale[j]; }
    double pobj = P->offset;
    for (int j = 0; j < L->n0; ++j) pobj += P->col_cost[j] * outX[j];
    if (R->col_value) for (int j = 0; j < L->n0; ++j) R->col_value[j] = outX[j];
    if (R->row_dual)
      for (int i = 0; i < m; ++i) {
        const double v = outY[L->newIdx[i]];
        R->row_dual[i] = L->ctype[i] == H_LEQ ? -v : v;
      }
    if (R->row_value) {
      for (int i = 0; i < m; ++i) R->row_value[i] = 0.0;
      for (int c = 0; c < L->n0; ++c)
        for (int p = P->a_start[c]; p < P->a_start[c + 1]; ++p) R->row_value[P->a_index[p]] += P->a_value[p] * outX[c];
    }
    if (R->col_dual) for (int j = 0; j < L->n0; ++j) R->col_dual[j] = (S.sp[j] - S.sn[j]) * (double)L->sense;
    R->value_valid = 1; R->dual_valid = 1;
    R->term_code = term == 0 ? PDLP_TERM_OPTIMAL : PDLP_TERM_TIMELIMIT_OR_ITERLIMIT;
    R->reserved_i = term == 2 ? 1 : 0; /* 1 = time limit */
    R->term_iterate = 0;
    R->num_iter = iters; R->num_trials = 0; R->num_restarts = restarts;
    R->primal_obj = pobj; R->dual_obj = res.dObj; R->primal_feas = res.pFeas; R->dual_feas = res.dFeas;
    R->rel_gap = res.relGap; R->norm_rhs = L->bNorm; R->norm_cost = L->cNorm;
    R->setup_seconds = tLoop - t0; R->solve_seconds = nowSec() - tLoop;
  }
  free(S.xc); free(S.yc); free(S.xn); free(S.yn); free(S.rx); free(S.ry); free(S.xa); free(S.ya); free(S.aty);
  free(S.axn); free(S.slack); free(S.sp); free(S.sn); free(outX); free(outY); free(axT); free(atyT);
  h_free(L);
  return 0;
}

int hipdlp_oracle_solve(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R) {
  if (!P || !opt || !R) return 1;
  if (P->num_row == 0 || P->num_col == 0 || !P->a_start || P->a_start[P->num_col] == 0) return 1; /* HiGHS: solveUnconstrainedLp */
  return h_solve(P, opt, R, NULL);
}

int hipdlp_oracle_probe(const pdlp_problem_t* P, const pdlp_params_t* opt, hipdlp_oracle_probe_t* probe) {
  if (!P || !opt || !probe || probe->steps < 1 || probe->steps > H_CHECK_INTERVAL) return 1;
  return h_solve(P, opt, NULL, probe);
}

/* Preprocessed + scaled problem for the host-side parity tests (arrays malloc'ed; caller frees each). */
typedef struct hipdlp_oracle_prepared {
  int n, m, n_eqs;
  long nnz;
  int *beg, *idx;
  double* val;
  double *cost, *lower, *upper, *row_lower, *row_upper, *col_scale, *row_scale;
  double norm_cost, norm_rhs;
} hipdlp_oracle_prepared_t;

int hipdlp_oracle_prepare(const pdlp_problem_t* P, const pdlp_params_t* opt, hipdlp_oracle_prepared_t* out) {
  HLp L;
  h_preprocess(P, &L);
  if (!(opt->features_off & PDLP_FEATURE_SCALING_OFF))
    h_scale(&L, opt->scaling_mode & 1, opt->scaling_mode & 4, opt->scaling_mode & 2, opt->ruiz_iterations);
  out->n = L.n; out->m = L.m; out->n_eqs = L.nEqs; out->nnz = L.nnz;
  out->beg = L.beg; out->idx = L.idx; out->val = L.val;
  out->cost = L.cost; out->lower = L.lower; out->upper = L.upper; out->row_lower = L.rl; out->row_upper = L.ru;
  out->col_scale = L.colScale; out->row_scale = L.rowScale;
  out->norm_cost = L.cNorm; out->norm_rhs = L.bNorm;
  free(L.ctype); free(L.newIdx); free(L.isEq);
  return 0;
}
void hipdlp_oracle_free_prepared(hipdlp_oracle_prepared_t* p) {
  free(p->beg); free(p->idx); free(p->val); free(p->cost); free(p->lower); free(p->upper); free(p->row_lower);
  free(p->row_upper); free(p->col_scale); free(p->row_scale);
  memset(p, 0, sizeof(*p));
}
