// pdlp_check.hip — the scalar side of a check iteration, on the device (round 4).
//
// Reference: PDHG_Solve's check block (cupdlp_solver.c:975-1069) = residuals of the current and the average iterate
// (PDHG_Compute_Residuals / _Infeas_Residuals, :433-529), the termination tests (:797-841, :710-795), then
// PDHG_Restart_Iterate (cupdlp_proj.c:88-148) with PDHG_Check_Restart_GPU (cupdlp_restart.c:3-124) and the
// primal-weight update PDHG_Compute_Step_Size_Ratio (cupdlp_step.c:147-176).  Until round 3 this ran on the host
// between two device stops (pdlp_solver.cpp computeResiduals / restartIterate, which stay for the sharded paths);
// here the same arithmetic, operation for operation, runs in three small kernels behind the statistics kernels of the
// check, so that the host does not have to look at a check before the next trial batch starts:
//     k_check_decide    30 statistics -> residuals, termination, restart decision            (one thread)
//     k_restart_vec     restart: sums cleared, average -> current, ||x - xLast||^2 partials   (vector grid)
//     k_restart_finish  beta, step sizes, next halt iteration; the device runs on             (one block)
// exp / log of the weight update: pdlp_detmath.h (the same bits on host, device and in the oracle's device-order mode).
#include <hip/hip_runtime.h>

#include <cmath>

#include "pdlp_devfn.hpp"
#include "pdlp_kernels.hpp"

namespace pdlp {

namespace {

// cuPDLP's resobj numbers of one iterate from its 4 row and 11 column statistics (Solver::computeResiduals `fill`)
__device__ void fillResiduals(ResidualsDev& r, const double* rs, const double* cs, const CheckCtl& c) {
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
__device__ bool converged(const ResidualsDev& r, const CheckCtl& c) {  // cupdlp_solver.c:797-841
  return (r.pFeas < c.primalTolAbs) && (r.dFeas < c.dualTolAbs) && (r.relGap < c.gapTol);
}
__device__ bool certificate(const ResidualsDev& r, double feasTol) {  // cupdlp_solver.c:710-795
  const bool primalInf = r.pInfObj > 0.0 && r.pInfRes < feasTol * r.pInfObj;
  const bool dualInf = r.dInfObj < 0.0 && r.dInfRes < -feasTol * r.dInfObj;
  return primalInf || dualInf;
}
__device__ double restartScore(double beta, double p, double d, double g) {  // cupdlp_restart.c:113-124
  return sqrt(beta * p * p + d * d / beta + g * g);
}
// next halt of the reference's schedule (Solver::nextCheckIter), clipped to the fixed-work target
__device__ int nextHalt(int it, const CheckCtl& c) {
  long long next;
  if (it + 1 < 10) next = it + 1;
  else next = ((long long)it / c.interval + 1) * c.interval;
  const long long last = (long long)c.optIterLimit - 1;
  if (last > it && last < next) next = last;
  if (!c.terminate && next > c.iterLimit) next = c.iterLimit;
  if (next > 2147483647LL) next = 2147483647LL;
  return (int)next;
}
__device__ void writeRecord(CheckRecord* rec, const DevState& s, const CheckCtl& c) {
  if (!rec) return;
  rec->it = c.lastCheckIter; rec->terminated = c.terminated; rec->termCode = c.termCode; rec->termIterate = c.termIterate;
  rec->restartKind = c.restartKind; rec->nRestarts = c.nRestarts; rec->nChecks = c.nChecks; rec->nTrials = s.nTrials;
  rec->beta = s.beta;
  rec->cur = c.cur; rec->avg = c.avg;
  __threadfence_system();
  rec->ran = 1;
}

constexpr int kStatRowCur = 0, kStatRowAvg = kRowStats, kStatColCur = 2 * kRowStats, kStatColAvg = 2 * kRowStats + kColStats;

__global__ __launch_bounds__(kWave) void k_check_decide(DevState* st, CheckCtl* cc, const double* __restrict__ stat, CheckRecord* rec) {
  if (threadIdx.x != 0) return;
  if (!checkDue(st, cc)) return;
  CheckCtl& c = *cc;
  DevState& s = *st;
  const int it = s.nIter;
  fillResiduals(c.cur, stat + kStatRowCur, stat + kStatColCur, c);
  fillResiduals(c.avg, stat + kStatRowAvg, stat + kStatColAvg, c);
  c.nChecks += 1;
  c.lastCheckIter = it;
  c.restartKind = 0;
  s.avgW = 0.0;  // the flush kernel of this check has added the pending averages
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
      writeRecord(rec, s, c);
      return;
    }
  }
  // ---- PDHG_Check_Restart_GPU (cupdlp_restart.c:3-124) ----
  if (!c.restartOn) return;
  if (it == c.iLastRestartIter) {
    c.pFeasLR = c.cur.pFeas; c.dFeasLR = c.cur.dFeas; c.gapLR = c.cur.gap;
    c.pFeasLC = c.cur.pFeas; c.dFeasLC = c.cur.dFeas; c.gapLC = c.cur.gap;
    return;
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
  if (!restart) return;
  c.pFeasLR = cand.pFeas; c.dFeasLR = cand.dFeas; c.gapLR = cand.gap;
  c.restartKind = toCurrent ? 1 : 2;
}

// Grid: nbX blocks over the columns, then nbY blocks over the rows — each part with the lanes, strides and block sums
// of k_diff_norm2 on its own grid (launchDiffNorm2 with vecBlocks(len) blocks), so that the two norms have the bits
// of the host-driven restart.
__global__ __launch_bounds__(kVecThreads) void k_restart_vec(const IterVecs v, const DevState* st, const CheckCtl* cc, const RestartVecs r,
                                                             double* partX, int nbX, double* partY, int nbY) {
  if (!checkDue(st, cc)) return;
  const int kind = cc->restartKind;
  if (kind == 0) return;
  __shared__ double scratch[kVecThreads / kWave];
  const int c = st->cur;
  double acc = 0.0;
  if ((int)blockIdx.x < nbX) {
    double* __restrict__ x = v.x[c];
    const int stride = nbX * kVecThreads;
    for (int j = blockIdx.x * kVecThreads + threadIdx.x; j < v.n; j += stride) {
      v.xSum[j] = 0.0;
      double xv;
      if (kind == 2) {
        xv = r.xAvg[j];
        x[j] = xv;
        v.aty[c][j] = r.atyAvg[j];
        if (v.nx[0]) v.nx[c][j] = r.nxAvg[j];
      } else {
        xv = x[j];
      }
      const double d = xv - r.xLast[j];
      acc += d * d;
      r.xLast[j] = xv;
    }
    const double t = blockSum<kVecThreads>(acc, scratch);
    if (threadIdx.x == 0) partX[blockIdx.x] = t;
  } else {
    const int b = (int)blockIdx.x - nbX;
    double* __restrict__ y = v.y[c];
    const int stride = nbY * kVecThreads;
    for (int i = b * kVecThreads + threadIdx.x; i < v.m; i += stride) {
      v.ySum[i] = 0.0;
      double yv;
      if (kind == 2) {
        yv = r.yAvg[i];
        y[i] = yv;
        v.ax[c][i] = r.axAvg[i];
      } else {
        yv = y[i];
      }
      const double d = yv - r.yLast[i];
      acc += d * d;
      r.yLast[i] = yv;
    }
    const double t = blockSum<kVecThreads>(acc, scratch);
    if (threadIdx.x == 0) partY[b] = t;
  }
}

__global__ __launch_bounds__(kVecThreads) void k_restart_finish(DevState* st, CheckCtl* cc, const double* __restrict__ partX, int nbX,
                                                                const double* __restrict__ partY, int nbY, CheckRecord* rec) {
  if (!checkDue(st, cc)) return;
  __shared__ double scratch[kVecThreads / kWave];
  const int kind = cc->restartKind;
  double dP2 = 0.0, dD2 = 0.0;
  if (kind) {  // the sums of k_final_reduce over each partial array
    dP2 = reducePartials(partX, nbX, scratch);
    dD2 = reducePartials(partY, nbY, scratch);
  }
  if (threadIdx.x != 0) return;
  CheckCtl& c = *cc;
  DevState& s = *st;
  const int it = s.nIter;
  if (kind) {
    s.sumPrimalStep = 0.0;
    s.sumDualStep = 0.0;
    // PDHG_Compute_Step_Size_Ratio, cupdlp_step.c:147-176
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
  writeRecord(rec, s, c);
}

}  // namespace

void launchCheckDecide(DevState* st, CheckCtl* cc, const double* stat, CheckRecord* rec, hipStream_t s) {
  hipLaunchKernelGGL(k_check_decide, dim3(1), dim3(kWave), 0, s, st, cc, stat, rec);
}
void launchRestartVec(const IterVecs& v, const DevState* st, const CheckCtl* cc, const RestartVecs& r, double* partX, int32_t nbX,
                      double* partY, int32_t nbY, hipStream_t s) {
  hipLaunchKernelGGL(k_restart_vec, dim3(nbX + nbY), dim3(kVecThreads), 0, s, v, st, cc, r, partX, nbX, partY, nbY);
}
void launchRestartFinish(DevState* st, CheckCtl* cc, const double* partX, int32_t nbX, const double* partY, int32_t nbY,
                         CheckRecord* rec, hipStream_t s) {
  hipLaunchKernelGGL(k_restart_finish, dim3(1), dim3(kVecThreads), 0, s, st, cc, partX, nbX, partY, nbY, rec);
}

}  // namespace pdlp
