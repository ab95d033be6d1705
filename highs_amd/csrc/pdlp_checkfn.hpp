// pdlp_checkfn.hpp — device functions of the check iteration shared by its two forms: the kernels of the launch
// sequence (pdlp_kernels.hip k_row_stats2 / k_col_stats2 / k_final_reduce2, pdlp_check.hip k_check_decide /
// k_restart_vec / k_restart_finish) and the ONE-launch check of the persistent small-LP loop (pdlp_check.hip
// k_check_small).  One definition of every per-element expression and of the scalar logic: the two forms cannot differ
// in a bit.  Reference: cupdlp_solver.c:12-204, 229-256, 326-366 (statistics), :433-529, :710-841 (residuals,
// termination), cupdlp_restart.c:3-124, cupdlp_step.c:147-176 (restart, primal weight).
#pragma once
#include "pdlp_devfn.hpp"

namespace pdlp {
namespace {

// AGENT: the vector was written earlier in the SAME launch by another workgroup (agent-scope load); else streamed
template <bool AGENT>
__device__ __forceinline__ double ldChk(const double* p) { return AGENT ? ldAgent(p) : ldStream(p); }

// Row i of the row pass, both iterates (quantities 0..3 current, 4..7 average; pdlp_kernels.hpp kRowStats)
template <bool AGENT_AVG>
__device__ __forceinline__ void rowStatsElem(double (&a)[2 * kRowStats], int i, const double* __restrict__ axC, const double* __restrict__ yC,
                                             const double* __restrict__ axA, const double* __restrict__ yA, const double* __restrict__ rhs,
                                             const double* __restrict__ rowScale, int scaled, bool ineq) {
  const double b = ldStream(rhs + i);
  const double rs = scaled ? ldStream(rowScale + i) : 1.0;
  const double axv[2] = {ldStream(axC + i), ldChk<AGENT_AVG>(axA + i)}, yv[2] = {ldStream(yC + i), ldChk<AGENT_AVG>(yA + i)};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double r = axv[k] + (-1.0) * b;
    if (ineq) r = r < 0.0 ? r : 0.0;
    r *= rs;
    a[4 * k + 0] += r * r;
    a[4 * k + 1] += yv[k] * b;
    a[4 * k + 2] += yv[k] * yv[k];
    double c = axv[k];
    if (ineq) c = c < 0.0 ? c : 0.0;
    c *= rs;
    a[4 * k + 3] += c * c;
  }
}

struct ColStatPtrs {
  const double* atyC; const double* xC; const double* atyA; const double* xA;
  const double* cost; const double* lower; const double* upper; const double* colScale; const double* qdiag;
  const double* nxC; const double* nxA;
  double* spC; double* snC; double* spA; double* snA;
};
// Column j of the column pass, both iterates (quantities 0..10 current, 11..21 average; kColStats); stores the slacks
template <bool AGENT_AVG>
__device__ __forceinline__ void colStatsElem(double (&a)[2 * kColStats], int j, const ColStatPtrs& p, int scaled) {
  const double c = ldStream(p.cost + j), l = ldStream(p.lower + j), u = ldStream(p.upper + j);
  const double cs = scaled ? ldStream(p.colScale + j) : 1.0;
  const double hasL = l > -INFINITY ? 1.0 : 0.0, hasU = u < INFINITY ? 1.0 : 0.0;
  const double lF = l > -INFINITY ? l : 0.0, uF = u < INFINITY ? u : 0.0;
  const double qj = p.qdiag ? p.qdiag[j] : 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double xv = k ? ldChk<AGENT_AVG>(p.xA + j) : ldStream(p.xC + j);
    const double atyv = k ? ldChk<AGENT_AVG>(p.atyA + j) : ldStream(p.atyC + j);
    const double* __restrict__ nx = k ? p.nxA : p.nxC;
    double* a_ = a + kColStats * k;
    double r = -atyv + c;
    double half = 0.0;
    if (p.qdiag) { r += qj * xv; half = (0.5 * qj * xv) * xv; }
    if (nx) { const double nj = k ? ldChk<AGENT_AVG>(nx + j) : ldStream(nx + j); r += nj; half += (0.5 * nj) * xv; }
    a_[10] += half;
    const double sp = (r > 0.0 ? r : 0.0) * hasL;
    const double sn = (-(r < 0.0 ? r : 0.0)) * hasU;
    stStream((k ? p.spA : p.spC) + j, sp);
    stStream((k ? p.snA : p.snC) + j, sn);
    a_[0] += xv * c;
    a_[1] += sp * lF;
    a_[2] += sn * uF;
    double rd = r + (-1.0) * sp;
    rd += sn;
    rd *= cs;
    a_[3] += rd * rd;
    a_[4] += sp * sp;
    a_[5] += sn * sn;
    double pc = (atyv + sp) - sn;
    pc *= cs;
    a_[6] += pc * pc;
    a_[7] += xv * xv;
    double lb = (xv < 0.0 ? xv : 0.0) * hasL;
    double ub = (xv > 0.0 ? xv : 0.0) * hasU;
    if (scaled) { lb /= cs; ub /= cs; }
    a_[8] += lb * lb;
    a_[9] += ub * ub;
  }
}

// NQ block sums with ONE barrier: every wave shuffles its NQ values down, lane 0 parks them, thread q adds the wave
// results of quantity q in order — the same tree as blockSum per quantity (bit-identical), without 2 NQ barriers.
// blk: the (virtual) block the sums belong to.  AGENT: the partials are read by other workgroups of the same launch.
template <int NQ, bool AGENT>
__device__ __forceinline__ void blockSumManyAt(double (&a)[NQ], double (*scratch)[kVecThreads / kWave], double* partials, int pstride, int blk) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double t = waveSum(a[q]);
    if (lane == 0) scratch[q][w] = t;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < kVecThreads / kWave; ++i) r += scratch[threadIdx.x][i];
    if (AGENT) stAgent(partials + threadIdx.x * pstride + blk, r);
    else partials[threadIdx.x * pstride + blk] = r;
  }
}

// reducePartials (pdlp_devfn.hpp) with agent-scope loads of the partials
__device__ __attribute__((unused)) double reducePartialsAgent(const double* __restrict__ p, int count, double* scratch) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * kVecThreads < count; i += 4 * kVecThreads) {
    const double a0 = ldAgent(p + i), a1 = ldAgent(p + i + kVecThreads), a2 = ldAgent(p + i + 2 * kVecThreads), a3 = ldAgent(p + i + 3 * kVecThreads);
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; i < count; i += kVecThreads) s0 += ldAgent(p + i);
  return blockSum<kVecThreads>((s0 + s1) + (s2 + s3), scratch);
}

// ---- the scalar side ------------------------------------------------------------------------------------------------
constexpr int kStatRowCur = 0, kStatRowAvg = kRowStats, kStatColCur = 2 * kRowStats, kStatColAvg = 2 * kRowStats + kColStats,
              kStatTotal = 2 * kRowStats + 2 * kColStats;

// cuPDLP's resobj numbers of one iterate from its 4 row and 11 column statistics (Solver::computeResiduals `fill`)
__device__ __forceinline__ void fillResiduals(ResidualsDev& r, const double* rs, const double* cs, const CheckCtl& c) {
  // QP (cs[10] = 1/2 x'Qx): primal c'x + 1/2 x'Qx, dual b'y + l's+ - u's- - 1/2 x'Qx
  r.pObj = (c.qp ? cs[0] + cs[10] : cs[0]) * c.sense + c.offset;
  r.pFeas = sqrt(rs[0]);
  r.dObj = (c.qp ? ((rs[1] + cs[1]) - cs[2]) - cs[10] : (rs[1] + cs[1] - cs[2])) * c.sense + c.offset;
  r.dFeas = sqrt(cs[3]);
  r.gap = r.pObj - r.dObj;
  r.relGap = fabs(r.pObj - r.dObj) / (1.0 + fabs(r.pObj) + fabs(r.dObj));
  double dScale = sqrt(rs[2] + cs[4] + cs[5]);  // ||(y, s+, s-)||, cupdlp_solver.c:230-237
  if (dScale < 1e-8) dScale = 1.0;
  r.pInfObj = (r.dObj - c.offset) / c.sense / dScale;
  r.pInfRes = sqrt(cs[6]) / dScale;
  double pScale = sqrt(cs[7]);  // ||x||, :328-332
  if (pScale < 1e-8) pScale = 1.0;
  r.dInfObj = (r.pObj - c.offset) / c.sense / pScale;
  r.dInfRes = sqrt(rs[3] + cs[8] + cs[9]) / pScale;
}
__device__ __forceinline__ bool converged(const ResidualsDev& r, const CheckCtl& c) {  // cupdlp_solver.c:797-841
  return (r.pFeas < c.primalTolAbs) && (r.dFeas < c.dualTolAbs) && (r.relGap < c.gapTol);
}
__device__ __forceinline__ bool certificate(const ResidualsDev& r, double feasTol) {  // cupdlp_solver.c:710-795
  const bool primalInf = r.pInfObj > 0.0 && r.pInfRes < feasTol * r.pInfObj;
  const bool dualInf = r.dInfObj < 0.0 && r.dInfRes < -feasTol * r.dInfObj;
  return primalInf || dualInf;
}
__device__ __forceinline__ double restartScore(double beta, double p, double d, double g) {  // cupdlp_restart.c:113-124
  return sqrt(beta * p * p + d * d / beta + g * g);
}
// next halt of the reference's schedule (Solver::nextCheckIter), clipped to the fixed-work target
__device__ __forceinline__ int nextHalt(int it, const CheckCtl& c) {
  long long next;
  if (it + 1 < 10) next = it + 1;
  else next = ((long long)it / c.interval + 1) * c.interval;
  const long long last = (long long)c.optIterLimit - 1;
  if (last > it && last < next) next = last;
  if (!c.terminate && next > c.iterLimit) next = c.iterLimit;
  if (next > 2147483647LL) next = 2147483647LL;
  return (int)next;
}
__device__ __forceinline__ void writeRecord(CheckRecord* rec, const DevState& s, const CheckCtl& c) {
  if (!rec) return;
  rec->it = c.lastCheckIter; rec->terminated = c.terminated; rec->termCode = c.termCode; rec->termIterate = c.termIterate;
  rec->restartKind = c.restartKind; rec->nRestarts = c.nRestarts; rec->nChecks = c.nChecks; rec->nTrials = s.nTrials;
  rec->beta = s.beta;
  rec->cur = c.cur; rec->avg = c.avg;
  __threadfence_system();
  rec->ran = 1;
}

// Residuals of both iterates, the termination tests and the restart decision of one check (one thread; s and c may
// live in HBM or in LDS).  Returns true when the solve has ended.  Sets c.restartKind (0 none, 1 current, 2 average).
__device__ __forceinline__ bool checkDecideCore(DevState& s, CheckCtl& c, const double* stat) {
  const int it = s.nIter;
  fillResiduals(c.cur, stat + kStatRowCur, stat + kStatColCur, c);
  fillResiduals(c.avg, stat + kStatRowAvg, stat + kStatColAvg, c);
  c.nChecks += 1;
  c.lastCheckIter = it;
  c.restartKind = 0;
  s.avgW = 0.0;  // the flush of this check has added the pending averages
  s.avgWx = 0.0;
  if (c.terminate) {
    bool term = true;
    if (converged(c.cur, c)) { c.termIterate = 0; c.termCode = 0 /* PDLP_TERM_OPTIMAL */; }
    else if (converged(c.avg, c)) { c.termIterate = 1; c.termCode = 0; }
    else if (certificate(c.cur, c.feasTol) || certificate(c.avg, c.feasTol)) c.termCode = 3 /* PDLP_TERM_INFEASIBLE_OR_UNBOUNDED */;
    else if (it >= c.iterLimit - 1) c.termCode = 4 /* PDLP_TERM_TIMELIMIT_OR_ITERLIMIT */;
    else term = false;
    if (term) {
      c.terminated = 1;  // the device stays halted: everything queued behind is a no-op
      return true;
    }
  }
  // ---- PDHG_Check_Restart_GPU (cupdlp_restart.c:3-124) ----
  if (!c.restartOn) return false;
  if (it == c.iLastRestartIter) {
    c.pFeasLR = c.cur.pFeas; c.dFeasLR = c.cur.dFeas; c.gapLR = c.cur.gap;
    c.pFeasLC = c.cur.pFeas; c.dFeasLC = c.cur.dFeas; c.gapLC = c.cur.gap;
    return false;
  }
  const double muCur = restartScore(s.beta, c.cur.pFeas, c.cur.dFeas, c.cur.gap);
  const double muAvg = restartScore(s.beta, c.avg.pFeas, c.avg.dFeas, c.avg.gap);
  const bool toCurrent = muCur < muAvg;
  const double muCand = toCurrent ? muCur : muAvg;
  bool restart = true;
  if ((it - c.iLastRestartIter) >= 0.36 * it) {
    // artificial restart
  } else {
    const double muLR = restartScore(s.beta, c.pFeasLR, c.dFeasLR, c.gapLR);
    if (muCand < 0.2 * muLR) {
      // sufficient decay
    } else {
      const double muLC = restartScore(s.beta, c.pFeasLC, c.dFeasLC, c.gapLC);
      if (!(muCand < 0.8 * muLR && muCand > muLC)) restart = false;  // necessary decay
    }
  }
  const ResidualsDev& cand = toCurrent ? c.cur : c.avg;
  c.pFeasLC = cand.pFeas; c.dFeasLC = cand.dFeas; c.gapLC = cand.gap;
  if (!restart) return false;
  c.pFeasLR = cand.pFeas; c.dFeasLR = cand.dFeas; c.gapLR = cand.gap;
  c.restartKind = toCurrent ? 1 : 2;
  return false;
}

// After the vector side of a restart: primal weight and step sizes (PDHG_Compute_Step_Size_Ratio, cupdlp_step.c:147-176)
// from ||x - xLast||^2, ||y - yLast||^2; then — restart or not — the next halt of the schedule and the device runs on.
__device__ __forceinline__ void restartFinishCore(DevState& s, CheckCtl& c, double dP2, double dD2) {
  const int it = s.nIter;
  if (c.restartKind) {
    s.sumPrimalStep = 0.0;
    s.sumDualStep = 0.0;
    const double mean = sqrt(s.primalStep * s.dualStep);
    const double dP = sqrt(dP2), dD = sqrt(dD2);
    if (fmin(dP, dD) > 1e-10) {
      const double lg = 0.5 * pdlp_det_log(dD / dP) + 0.5 * pdlp_det_log(sqrt(s.beta));
      s.beta = pdlp_det_exp(lg) * pdlp_det_exp(lg);
    }
    s.primalStep = mean / sqrt(s.beta);
    s.dualStep = s.primalStep * s.beta;
    s.eta = sqrt(s.primalStep * s.dualStep);
    if (c.adaptive) {
      s.tau = s.eta / sqrt(s.beta);
      s.sigma = s.eta * sqrt(s.beta);
    } else {
      s.tau = s.primalStep;
      s.sigma = s.dualStep;
    }
    c.iLastRestartIter = it;
    c.nRestarts += 1;
  }
  s.haltIter = nextHalt(it, c);
  s.halted = 0;
  s.pending = 0;
}

}  // namespace
}  // namespace pdlp
