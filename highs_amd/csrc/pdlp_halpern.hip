// pdlp_halpern.hip — vector kernels of the HiPDLP path's check iterations (fixed-point error,
// convergence statistics).  The Halpern step itself lives in the SpMV epilogues (pdlp_kernels.hip).
// Reductions: per-lane strided sums -> wave shuffle tree -> fixed-order block sum -> fixed-order
// final reduce (deterministic, no atomics).
#include "pdlp_halpern.hpp"

#include <algorithm>
#include <cmath>

#include "pdlp_devfn.hpp"
#include "pdlp_halpernfn.hpp"

namespace pdlp {

namespace {

// computeFixedPointError, pdhg.cc:709-739: delta_y = y_next - reflected_y and its squared norm
__global__ __launch_bounds__(kVecThreads) void k_h_fpe_rows(const double* __restrict__ yn, const double* __restrict__ ry,
                                                            double* __restrict__ dy, int m, double* part, const int32_t* gate) {
  if (gate && *gate == 0) return;
  __shared__ double scratch[kVecThreads / kWave];
  double s = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const double d = yn[i] - ry[i];
    dy[i] = d;
    s += d * d;
  }
  const double t = blockSum<kVecThreads>(s, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// ... delta_x = x_next - reflected_x: squared norm and <delta_x, A' delta_y>
__global__ __launch_bounds__(kVecThreads) void k_h_fpe_cols(const double* __restrict__ xn, const double* __restrict__ rx,
                                                            const double* __restrict__ atd, int n, double* partDx2,
                                                            double* partCross, const int32_t* gate) {
  if (gate && *gate == 0) return;
  __shared__ double scratch[2][kVecThreads / kWave];
  double s0 = 0.0, s1 = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const double d = xn[j] - rx[j];
    s0 += d * d;
    s1 += d * atd[j];
  }
  const double t0 = blockSum<kVecThreads>(s0, scratch[0]);
  const double t1 = blockSum<kVecThreads>(s1, scratch[1]);
  if (threadIdx.x == 0) { partDx2[blockIdx.x] = t0; partCross[blockIdx.x] = t1; }
}

// computePrimalFeasibility (pdhg.cc:1297-1320) + the b'y term of computeDualObjective (:1452-1455)
__global__ __launch_bounds__(kVecThreads) void k_h_row_stats(const double* __restrict__ ax, const double* __restrict__ y,
                                                             const double* __restrict__ rl,
                                                             const double* __restrict__ rowScale,
                                                             const uint8_t* __restrict__ isEq, int m, int scaled,
                                                             double* part, int pstride, const int32_t* gate) {
  if (gate && *gate == 0) return;
  __shared__ double scratch[kVecThreads / kWave];
  double a0 = 0.0, a1 = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const double b = rl[i];
    double r = ax[i] - b;
    if (!isEq[i]) r = r < 0.0 ? r : 0.0;  // std::min(0.0, r)
    if (scaled) r *= rowScale[i];
    a0 += r * r;
    a1 += b * y[i];
  }
  const double t0 = blockSum<kVecThreads>(a0, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = t0;
  const double t1 = blockSum<kVecThreads>(a1, scratch);
  if (threadIdx.x == 0) part[pstride + blockIdx.x] = t1;
}

// computeDualSlacks (pdhg.cc:1322-1378, Halpern branch), computeDualFeasibility (:1380-1412), the
// objective sums of checkConvergence (:1490-1499) and computeDualObjective (:1457-1470)
__global__ __launch_bounds__(kVecThreads) void k_h_col_stats(const double* __restrict__ aty, const double* __restrict__ x,
                                                             const double* __restrict__ cost,
                                                             const double* __restrict__ lower,
                                                             const double* __restrict__ upper,
                                                             const double* __restrict__ colScale,
                                                             const double* __restrict__ cachedSlack, int n, int scaled,
                                                             double* __restrict__ sp, double* __restrict__ sn,
                                                             double* part, int pstride, const int32_t* gate) {
  if (gate && *gate == 0) return;
  __shared__ double scratch[kVecThreads / kWave];
  double a[kHColStats] = {0.0, 0.0, 0.0, 0.0};
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const double c = cost[j], l = lower[j], u = upper[j];
    const double dr = c - aty[j];
    const bool hasL = l > -INFINITY, hasU = u < INFINITY;
    double ds = 0.0;
    if (cachedSlack) ds = cachedSlack[j];
    else if (hasL && hasU) ds = dr;
    else if (hasL) ds = dr > 0.0 ? dr : 0.0;
    else if (hasU) ds = dr < 0.0 ? dr : 0.0;
    const double p = ds > 0.0 ? ds : 0.0;
    const double q = -ds > 0.0 ? -ds : 0.0;
    sp[j] = p;
    sn[j] = q;
    double v = dr - p + q;
    if (scaled) v *= colScale[j];
    a[0] += v * v;
    a[1] += c * x[j];
    if (hasL) a[2] += l * p;
    if (hasU) a[3] += u * q;
  }
#pragma unroll
  for (int k = 0; k < kHColStats; ++k) {
    const double t = blockSum<kVecThreads>(a[k], scratch);
    if (threadIdx.x == 0) part[k * pstride + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kVecThreads) void k_div_scalar(double* v, double denom, int len) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) v[i] /= denom;
}

// The end of a block in the device-driven loop: one thread takes the decision of pdlp_halpernfn.hpp halpernDecide on the
// check's statistics, in place on the state record, and writes the check's line into the pinned ring.
__global__ void k_h_decide(HalpernState* st, const double* __restrict__ stat, HalpernRecord* ring) {
  if (st->halted) return;
  HalpernState s = *st;
  const HalpernRecord r = halpernDecide(s, stat);
  *st = s;
  ring[(s.nChecks - 1) % kHalpernRing] = r;
}
// restart (pdhg.cc:663-692): anchor and current iterate <- pdhg iterate of the last major step; gated by doRestart
__global__ __launch_bounds__(kVecThreads) void k_h_restart_copy(const HalpernState* st, double* __restrict__ xa, double* __restrict__ xc,
                                                                const double* __restrict__ xn, int n, double* __restrict__ ya,
                                                                double* __restrict__ yc, const double* __restrict__ yn, int m) {
  if (!st->doRestart) return;
  const int stride = gridDim.x * blockDim.x, tot = n + m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
    if (i < n) { const double v = xn[i]; xa[i] = v; xc[i] = v; }
    else { const double v = yn[i - n]; ya[i - n] = v; yc[i - n] = v; }
  }
}
// a converged check keeps its iterate (pdhg.cc:866-877); gated by converged
__global__ __launch_bounds__(kVecThreads) void k_h_keep_output(const HalpernState* st, double* __restrict__ outX, const double* __restrict__ xn,
                                                               int n, double* __restrict__ outY, const double* __restrict__ yn, int m) {
  if (!st->converged) return;
  const int stride = gridDim.x * blockDim.x, tot = n + m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
    if (i < n) outX[i] = xn[i];
    else outY[i - n] = yn[i - n];
  }
}

}  // namespace

void launchHalpernDecide(HalpernState* st, const double* stat, HalpernRecord* ring, hipStream_t s) {
  hipLaunchKernelGGL(k_h_decide, dim3(1), dim3(1), 0, s, st, stat, ring);
}
void launchHalpernRestartCopy(const HalpernState* st, double* xa, double* xc, const double* xn, int32_t n, double* ya, double* yc,
                              const double* yn, int32_t m, hipStream_t s) {
  hipLaunchKernelGGL(k_h_restart_copy, dim3(vecBlocks(std::max(n + m, 1))), dim3(kVecThreads), 0, s, st, xa, xc, xn, n, ya, yc, yn, m);
}
void launchHalpernKeepOutput(const HalpernState* st, double* outX, const double* xn, int32_t n, double* outY, const double* yn, int32_t m,
                             hipStream_t s) {
  hipLaunchKernelGGL(k_h_keep_output, dim3(vecBlocks(std::max(n + m, 1))), dim3(kVecThreads), 0, s, st, outX, xn, n, outY, yn, m);
}

void launchHalpernFpeRows(const double* yn, const double* ry, double* dy, int32_t m, double* part, int32_t nBlocks,
                          hipStream_t s, const int32_t* gate) {
  hipLaunchKernelGGL(k_h_fpe_rows, dim3(nBlocks), dim3(kVecThreads), 0, s, yn, ry, dy, m, part, gate);
}
void launchHalpernFpeCols(const double* xn, const double* rx, const double* atd, int32_t n, double* partDx2,
                          double* partCross, int32_t nBlocks, hipStream_t s, const int32_t* gate) {
  hipLaunchKernelGGL(k_h_fpe_cols, dim3(nBlocks), dim3(kVecThreads), 0, s, xn, rx, atd, n, partDx2, partCross, gate);
}
void launchHalpernRowStats(const double* ax, const double* y, const double* rl, const double* rowScale,
                           const uint8_t* isEq, int32_t m, int scaled, double* part, int32_t stride, int32_t nBlocks,
                           hipStream_t s, const int32_t* gate) {
  hipLaunchKernelGGL(k_h_row_stats, dim3(nBlocks), dim3(kVecThreads), 0, s, ax, y, rl, rowScale, isEq, m, scaled, part,
                     stride, gate);
}
void launchHalpernColStats(const double* aty, const double* x, const double* cost, const double* lower,
                           const double* upper, const double* colScale, const double* cachedSlack, int32_t n,
                           int scaled, double* sp, double* sn, double* part, int32_t stride, int32_t nBlocks,
                           hipStream_t s, const int32_t* gate) {
  hipLaunchKernelGGL(k_h_col_stats, dim3(nBlocks), dim3(kVecThreads), 0, s, aty, x, cost, lower, upper, colScale,
                     cachedSlack, n, scaled, sp, sn, part, stride, gate);
}
void launchDivScalar(double* v, double denom, int32_t len, hipStream_t s) {
  if (len <= 0) return;
  hipLaunchKernelGGL(k_div_scalar, dim3(vecBlocks(len)), dim3(kVecThreads), 0, s, v, denom, len);
}

}  // namespace pdlp
