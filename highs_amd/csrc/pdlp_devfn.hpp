// pdlp_devfn.hpp — device functions shared by the kernel translation units
// (pdlp_kernels.hip, pdlp_mesh.hip): deterministic wave/block sums and the
// accept/reject + step-size update of the adaptive rule.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "pdlp_kernels.hpp"

namespace pdlp {
namespace {

constexpr int kWave = 64;

// Streamed vector traffic (iterates, costs, bounds, sums: everything that is touched once per kernel)
// is loaded and stored NON-TEMPORALLY: it then does not displace the two matrix copies (192 MB at the
// bench size) from the 256 MB Infinity Cache, and the entry/value streams of both SpMVs are served from
// there instead of HBM — measured -15 us per SpMV inside the iteration.  The two GATHERED vectors
// (x+ for A x+, y+ for A' y+; HiPDLP: y_current, reflected x) are written with ordinary stores: every
// XCD reads them right after.
template <typename T>
__device__ __forceinline__ T ldStream(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void stStream(T* p, T v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ double waveSum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Deterministic block sum for blocks of NT threads; result valid in thread 0.
template <int NT>
__device__ __forceinline__ double blockSum(double v, double* scratch /* [NT/64] */) {
  v = waveSum(v);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NT / kWave; ++i) r += scratch[i];
  }
  __syncthreads();
  return r;
}

// LDS slot of the q-th staged product: one pad double per 8 keeps the
// thread-per-row read-back (stride = row length, typically 8..16 doubles)
// off a single bank pair.
__device__ __forceinline__ int slot(int q) { return q + (q >> 3); }

// Fixed-order sum of `count` partials by one block of 256 threads (4 loads in flight per lane).
__device__ __attribute__((unused)) double reducePartials(const double* __restrict__ p, int count, double* scratch) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * kVecThreads < count; i += 4 * kVecThreads) {
    const double a0 = p[i], a1 = p[i + kVecThreads], a2 = p[i + 2 * kVecThreads], a3 = p[i + 3 * kVecThreads];
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; i < count; i += kVecThreads) s0 += p[i];
  return blockSum<kVecThreads>((s0 + s1) + (s2 + s3), scratch);
}

// Accept/reject and step-size update of PDHG_Update_Iterate_Adaptive_Step_Size
// (cupdlp_step.c:237-306), the bookkeeping of PDHG_Update_Average (:433-441)
// and the parity flip that the reference gets from ++nIter.  One thread.
// TABLE_ONLY (the fused trial kernel, which has no registers to spare for an inlined pow): the host-tabulated powers
// must cover the trial counter — Solver::refreshPowTable keeps >= 1024 entries ahead at every stop; if they ever
// do not, the state is flagged (commError) instead of silently taking the device's own pow.
// QP with off-diagonal Hessian entries (no reference counterpart): the explicit N x term of the primal step is a
// forward step on a smooth function, so the trial must also satisfy tau <= |dx|^2 / |dx . N dx|; with tau = eta/w the
// two conditions combine to eta <= movement / (|interaction| + |dx . N dx| / 2), which is the LP rule for qint = 0.
template <bool TABLE_ONLY>
__device__ __forceinline__ void decideCore(DevState& s, double dX2, double dY2, double inter, double qint) {
  const double sb = sqrt(s.beta);
  const double movement = dX2 * 0.5 * sb + dY2 / (2.0 * sb);
  s.nTrials += 1;
  bool accept = true;
  double etaNew = s.eta;
  double limit = INFINITY;
  if (s.adaptive) {
    const double den = fabs(inter) + 0.5 * fabs(qint);
    limit = (den != 0.0) ? movement / den : INFINITY;
    accept = s.eta <= limit;
    const double k1 = (double)s.nTrials + 1.0;
    const int ti = s.nTrials - s.powBase;
    const bool tab = s.powRed != nullptr && ti >= 0 && ti < s.powCount;
    double pRed, pGrow;
    if (TABLE_ONLY) {
      pRed = tab ? s.powRed[ti] : 0.0;
      pGrow = tab ? s.powGrow[ti] : 0.0;
      if (!tab) s.commError = 1;
    } else {
      pRed = tab ? s.powRed[ti] : pow(k1, -0.3);    // PDHG_STEPSIZE_REDUCTION_EXP
      pGrow = tab ? s.powGrow[ti] : pow(k1, -0.6);  // PDHG_STEPSIZE_GROWTH_EXP
    }
    const double first = (1.0 - pRed) * limit;
    const double second = (1.0 + pGrow) * s.eta;
    etaNew = fmin(first, second);
  }
  s.dX2 = dX2; s.dY2 = dY2; s.inter = inter; s.movement = movement; s.limit = limit; s.qint = qint;
  s.lastAccepted = accept ? 1 : 0;
  s.pending = 0;
  if (accept) {
    if (s.adaptive) {
      s.primalStep = etaNew / sqrt(s.beta);
      s.dualStep = etaNew * sqrt(s.beta);
    }
    const double w = sqrt(s.primalStep * s.dualStep);  // uses the NEXT step sizes (step.c:433)
    s.sumPrimalStep += w;
    s.sumDualStep += w;
    s.avgW = w;
    s.avgWx = w;
    s.cur ^= 1;
    s.nIter += 1;
    s.eta = w;  // next iteration starts from sqrt(primalStep*dualStep) (step.c:231)
    if (s.nIter >= s.haltIter) s.halted = 1;
  } else {
    s.eta = etaNew;
    s.avgW = 0.0;
    s.avgWx = 0.0;
  }
  if (s.adaptive) {
    s.tau = s.eta / sqrt(s.beta);
    s.sigma = s.eta * sqrt(s.beta);
  }
}
template <bool TABLE_ONLY = false>
__device__ __attribute__((unused)) void decideUpdate(DevState* st, double dX2, double dY2, double inter, double qint = 0.0) {
  if (TABLE_ONLY) {  // the state sits in LDS: updated in place (a register copy of the record costs 50 VGPRs)
    decideCore<true>(*st, dX2, dY2, inter, qint);
  } else {
    DevState s = *st;
    decideCore<false>(s, dX2, dY2, inter, qint);
    *st = s;
  }
}

// Epilogue operands that do not depend on the SpMV result.
struct Pre { double a, b, c, d, e; };

// HiPDLP step, column side (pdhg.cc:975-990 and :1008-1011): s = (A'y)_j.
__device__ __forceinline__ __attribute__((unused)) void halpernPrimal(const HalpernVecs& h, int j, double s, const Pre& p, double tau,
                                              double rho, double w) {
  const double xc = p.a, cost = p.b, xa = p.c, l = p.d, u = p.e;
  const double temp = xc - tau * (cost - s);
  const double t = (u < temp) ? u : temp;  // std::min(temp, u)
  const double proj = (l < t) ? t : l;     // std::max(l, .)   (linalg::projectBox)
  if (h.major) {
    stStream(h.xn + j, proj);
    stStream(h.slack + j, (proj - temp) / tau);
  }
  const double rx = 2.0 * proj - xc;
  h.rx[j] = rx;  // gathered by the A x kernel: ordinary store
  const double blended = rho * rx + (1.0 - rho) * xc;
  stStream(h.xc + j, w * blended + (1.0 - w) * xa);
}
// HiPDLP step, row side (pdhg.cc:995-1006 and :1012-1015): s = (A reflected_x)_i.
__device__ __forceinline__ __attribute__((unused)) void halpernDual(const HalpernVecs& h, int i, double s, const Pre& p, double sigma,
                                            double rho, double w) {
  const double yc = p.a, ya = p.b, rl = p.c, ru = p.d;
  const double temp = yc / sigma - s;
  const double lo = -ru, up = -rl;
  const double t = (up < temp) ? up : temp;
  const double proj = (lo < t) ? t : lo;
  const double pd = (temp - proj) * sigma;
  const double ry = 2.0 * pd - yc;
  if (h.major) {
    stStream(h.yn + i, pd);
    stStream(h.ry + i, ry);
  }
  const double blended = rho * ry + (1.0 - rho) * yc;
  h.yc[i] = w * blended + (1.0 - w) * ya;  // gathered by the A' y kernel: ordinary store
}


}  // namespace
}  // namespace pdlp
