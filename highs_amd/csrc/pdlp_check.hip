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

#include "pdlp_checkfn.hpp"
#include "pdlp_devfn.hpp"
#include "pdlp_kernels.hpp"

namespace pdlp {

namespace {

__global__ __launch_bounds__(kWave) void k_check_decide(DevState* st, CheckCtl* cc, const double* __restrict__ stat, CheckRecord* rec) {
  if (threadIdx.x != 0) return;
  if (!checkDue(st, cc)) return;
  if (checkDecideCore(*st, *cc, stat)) writeRecord(rec, *st, *cc);  // (the solve has ended; otherwise k_restart_finish writes the record)
}

// The vector side of a restart for one (virtual) block of the grid nbX + nbY of k_restart_vec: blocks < nbX work on the
// columns, the others on the rows, each part with the lanes, strides and block sums of k_diff_norm2 on its own grid
// (launchDiffNorm2 with vecBlocks(len) blocks), so that the two norms have the bits of the host-driven restart.
// AGENT: the average vectors were written, and the partials are read, by other workgroups of the same launch.
template <bool AGENT>
__device__ __forceinline__ void restartVecBlock(const IterVecs& v, int c, int kind, const RestartVecs& r, int vb, double* partX, int nbX,
                                                double* partY, int nbY, double* scratch) {
  double acc = 0.0;
  if (vb < nbX) {
    double* __restrict__ x = v.x[c];
    const int stride = nbX * kVecThreads;
    for (int j = vb * kVecThreads + threadIdx.x; j < v.n; j += stride) {
      if (AGENT) stAgent(v.xSum + j, 0.0); else v.xSum[j] = 0.0;  // (one launch: phase F of another workgroup wrote it, see k_check_small)
      double xv;
      if (kind == 2) {
        xv = ldChk<AGENT>(r.xAvg + j);
        x[j] = xv;
        v.aty[c][j] = ldChk<AGENT>(r.atyAvg + j);
        if (v.nx[0]) v.nx[c][j] = r.nxAvg[j];
      } else {
        xv = x[j];
      }
      const double d = xv - r.xLast[j];
      acc += d * d;
      r.xLast[j] = xv;
    }
    const double t = blockSum<kVecThreads>(acc, scratch);
    if (threadIdx.x == 0) { if (AGENT) stAgent(partX + vb, t); else partX[vb] = t; }
  } else {
    const int b = vb - nbX;
    double* __restrict__ y = v.y[c];
    const int stride = nbY * kVecThreads;
    for (int i = b * kVecThreads + threadIdx.x; i < v.m; i += stride) {
      if (AGENT) stAgent(v.ySum + i, 0.0); else v.ySum[i] = 0.0;
      double yv;
      if (kind == 2) {
        yv = ldChk<AGENT>(r.yAvg + i);
        y[i] = yv;
        v.ax[c][i] = ldChk<AGENT>(r.axAvg + i);
      } else {
        yv = y[i];
      }
      const double d = yv - r.yLast[i];
      acc += d * d;
      r.yLast[i] = yv;
    }
    const double t = blockSum<kVecThreads>(acc, scratch);
    if (threadIdx.x == 0) { if (AGENT) stAgent(partY + b, t); else partY[b] = t; }
  }
}

__global__ __launch_bounds__(kVecThreads) void k_restart_vec(const IterVecs v, const DevState* st, const CheckCtl* cc, const RestartVecs r,
                                                             double* partX, int nbX, double* partY, int nbY) {
  if (!checkDue(st, cc)) return;
  const int kind = cc->restartKind;
  if (kind == 0) return;
  __shared__ double scratch[kVecThreads / kWave];
  restartVecBlock<false>(v, st->cur, kind, r, (int)blockIdx.x, partX, nbX, partY, nbY, scratch);
}

__global__ __launch_bounds__(kVecThreads) void k_restart_copy_full(const DevState* st, const CheckCtl* cc, double* x0, double* x1,
                                                                   double* aty0, double* aty1, double* y0, double* y1,
                                                                   const double* __restrict__ xAvg, const double* __restrict__ atyAvg,
                                                                   const double* __restrict__ yAvg, int n, int yLen) {
  if (!checkDue(st, cc)) return;
  if (cc->restartKind != 2) return;
  const int c = st->cur;
  double* __restrict__ x = c ? x1 : x0;
  double* __restrict__ aty = c ? aty1 : aty0;
  double* __restrict__ y = c ? y1 : y0;
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) { x[j] = xAvg[j]; aty[j] = atyAvg[j]; }
  if (y)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < yLen; i += stride) y[i] = yAvg[i];
}

__global__ __launch_bounds__(kVecThreads) void k_restart_finish(DevState* st, CheckCtl* cc, const double* __restrict__ partX, int nbX,
                                                                const double* __restrict__ partY, int nbY, CheckRecord* rec) {
  if (!checkDue(st, cc)) return;
  __shared__ double scratch[kVecThreads / kWave];
  double dP2 = 0.0, dD2 = 0.0;
  if (cc->restartKind) {  // the sums of k_final_reduce over each partial array
    dP2 = reducePartials(partX, nbX, scratch);
    dD2 = reducePartials(partY, nbY, scratch);
  }
  if (threadIdx.x != 0) return;
  restartFinishCore(*st, *cc, dP2, dD2);
  writeRecord(rec, *st, *cc);
}

// ---- the whole check of a small LP as ONE launch --------------------------------------------------------------------
// Netlib-class LPs run their trials in one persistent launch per check period (pdlp_small.hip); the check behind it was
// ten launches of a few microseconds each — 48 us per period on 25fv47, a tenth of the loop.  Here the same phases run
// inside one launch of the same few dozen resident workgroups, separated by grid barriers (sweep barrier, agent-scope
// hand-overs, roll call first — this is a barrier launch like the trial loop):
//     F  pending averages + xAvg, yAvg                      | barrier
//     S  A xAvg, A' yAvg (stream blocks + segment tasks)    | barrier
//     R  row / column statistics of both iterates           | barrier     virtual blocks: block vb of the grid of
//     Q  the 30 fixed-order sums of the block partials      | barrier     k_row_stats2 / k_col_stats2 / k_restart_vec is
//     D  residuals, termination, restart decision — in every workgroup      run by workgroup vb % G with the same lanes,
//     V  (restart) sums cleared, average -> current, norms  | barrier     strides and trees: the same bits
//     W  primal weight, next halt; workgroup 0 writes the state, the control record and the host's record
// Every per-element expression and the scalar logic are the functions of pdlp_checkfn.hpp that the launch sequence uses.
struct CheckSmallArgs {
  SpmvMat A, At;
  LongMat LA, LAt;
  IterVecs v;
  DevState* st;
  CheckCtl* cc;
  CheckRecord* rec;
  RestartVecs r;  // xAvg, yAvg, axAvg, atyAvg, xLast, yLast
  double* xAvg; double* yAvg; double* axAvg; double* atyAvg;
  const double* rowScale; const double* colScale;
  double* spC; double* snC; double* spA; double* snA;
  double* statPart; double* statOut; double* partX; double* partY;
  unsigned long long* bar;   // G arrival words, the timeout flag, the roll-call word
  unsigned long long seq;    // number of this launch since the words were zeroed (1, 2, ...)
  unsigned long long limit;
  int32_t statStride, scaled;
};

__device__ __forceinline__ int vecBlocksDev(int len) {
  long long b = ((long long)len + kVecThreads - 1) / kVecThreads;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// out[r] = sum of the major's products, left to right (k_spmv's stream path with the plain epilogue), for one work block
template <int CHUNK>
__device__ __forceinline__ void plainSpmvBlock(const SpmvMat& M, int blk, const double* in, double* out, double* prod) {
  constexpr int kPer = CHUNK / kSpmvThreads;
  const int tid = threadIdx.x;
  const int4 bb = *reinterpret_cast<const int4*>(M.blockBeg + 4 * blk);
  const int r0 = bb.x, r1 = bb.y, p0 = bb.z, cnt = bb.w - bb.z;
  const int last = cnt > 0 ? cnt - 1 : 0;
  int32_t ci[kPer];
  double va[kPer], xg[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int q = tid + k * kSpmvThreads;
    const int qq = q < last ? q : last;
    ci[k] = M.idx[p0 + qq];
    va[k] = M.val[p0 + qq];
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) xg[k] = ldAgent(in + ci[k]);
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int q = tid + k * kSpmvThreads;
    if (q < cnt) prod[slot(q)] = va[k] * xg[k];
  }
  __syncthreads();
  for (int r = r0 + tid; r < r1; r += kSpmvThreads) {
    const int qb = M.beg[r] - p0, qe = M.beg[r + 1] - p0;
    stAgent(out + r, majorSum(prod, qb, qe));
  }
  __syncthreads();
}
// a group of four segment tasks of the long majors, plain epilogue (pdlp_kernels.hip longBlock: same lanes and sums)
__device__ __forceinline__ void plainLongBlock(const LongMat& L, int tb, const double* in, double* out, double* lds /* [4] */) {
  constexpr int W = kSpmvThreads / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
  const int t = tb * W + wave;
  LongTask T;
  T.c = -1; T.pBeg = T.pEnd = 0; T.first = t; T.nSeg = 1; T.major = 0; T.contained = 1; T.seg = 0;
  if (t < L.nTasks) {
    const int32_t* q = reinterpret_cast<const int32_t*>(L.tasks + t);
    T.pBeg = ldUniform(q); T.pEnd = ldUniform(q + 1); T.c = ldUniform(q + 2); T.first = ldUniform(q + 3);
    T.nSeg = ldUniform(q + 4); T.major = ldUniform(q + 5); T.contained = ldUniform(q + 6); T.seg = ldUniform(q + 7);
  }
  const bool active = T.c >= 0;
  const int seg = T.seg;
  constexpr int kPer = kLongSegment / kWave;
  double s = 0.0;
  for (int base = T.pBeg; base < T.pEnd; base += kLongSegment) {
    int32_t ci[kPer];
    double va[kPer], xg[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int q = base + k * kWave + lane;
      const int qq = q < T.pEnd ? q : T.pEnd - 1;
      ci[k] = L.idx[qq];
      va[k] = L.val[qq];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) xg[k] = ldAgent(in + ci[k]);
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (base + k * kWave + lane < T.pEnd) s += va[k] * xg[k];
  }
  s = waveSum(s);
  int last = 0;
  if (lane == 0) {
    lds[wave] = s;
    if (active && !T.contained) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(L.segSum + T.first + seg), (unsigned long long)__double_as_longlong(s),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(L.ticket + T.c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = old == (unsigned)(T.nSeg - 1) ? 1 : 0;
    }
  }
  last = __builtin_amdgcn_readfirstlane(last);
  __syncthreads();
  double total = 0.0;
  bool finish = false;
  if (active && T.contained && seg == 0) {
    for (int k = 0; k < T.nSeg; ++k) total += lds[wave + k];
    finish = true;
  } else if (last) {
    if (lane == 0) __hip_atomic_store(L.ticket + T.c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double v = 0.0;
    if (lane < T.nSeg)
      v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(L.segSum + T.first + lane),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int k = 0; k < T.nSeg; ++k) total += __shfl(v, k, kWave);
    finish = true;
  }
  if (finish && lane == 0) stAgent(out + T.major, total);
  __syncthreads();
}

template <int CHUNK_A, int CHUNK_AT>
__global__ __launch_bounds__(kVecThreads) void k_check_small(const CheckSmallArgs a) {
  constexpr int kMaxChunk = CHUNK_A > CHUNK_AT ? CHUNK_A : CHUNK_AT;
  __shared__ double prod[kMaxChunk + kMaxChunk / 8 + 8];
  __shared__ double scratch[2 * kColStats][kVecThreads / kWave];
  __shared__ double stat[kStatTotal];
  __shared__ DevState sh;
  __shared__ CheckCtl ctl;
  __shared__ int flag;
  __shared__ int barBad;  // a grid barrier of this launch did not hold in this workgroup (gridBarrier's verdict)
  const int tid = threadIdx.x, lb = blockIdx.x, G = gridDim.x;
  if (tid == 0) barBad = 0;
  // roll call: every launch of the sequence takes part, due or not (the count is cumulative)
  if (tid < kWave) {
    const bool here = rollCall(a.bar + G + 1, a.seq * (unsigned long long)G, a.limit, tid);
    if (tid == 0) flag = here ? 1 : 0;
  }
  __syncthreads();
  if (!flag) {
    if (tid == 0) __hip_atomic_store(&a.st->commError, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (!checkDue(a.st, a.cc)) return;  // (nobody writes the two records before the last barrier of the launch)
  for (int w = tid; w < (int)(sizeof(DevState) / 4); w += kVecThreads) reinterpret_cast<uint32_t*>(&sh)[w] = reinterpret_cast<const uint32_t*>(a.st)[w];
  for (int w = tid; w < (int)(sizeof(CheckCtl) / 4); w += kVecThreads) reinterpret_cast<uint32_t*>(&ctl)[w] = reinterpret_cast<const uint32_t*>(a.cc)[w];
  padSlots(prod, kMaxChunk + kMaxChunk / 8 + 8, tid, kVecThreads);
  __syncthreads();
  unsigned long long epoch = 8ull * a.seq;
  // A barrier that does not hold (a resident workgroup stalled for longer than the timeout, a poisoned word) must not let
  // the phases behind it pass for a check: the verdict is kept, and before anything is handed on (phase W) workgroup 0
  // also reads the timeout flag behind the arrival words — set by ANY workgroup that gave up — as the trial loop does.
  // Then the records are not written; commError = 1 and halted = 1 instead (syncState throws).
  auto meet = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < kWave) {
      const int verdict = gridBarrier<false>(a.bar, lb, G, ++epoch, tid, a.limit);
      if (tid == 0 && verdict != kBarOk) barBad = 1;
    } else {
      ++epoch;
    }
    __syncthreads();
  };
  const IterVecs& v = a.v;
  const int cur = sh.cur, n = v.n, m = v.m;
  // ---- F: pending average update and the averages (k_flush_scale) ----
  {
    const double w = sh.avgW, wx = sh.avgWx;
    const double ps = sh.sumPrimalStep > 0.0 ? 1.0 / sh.sumPrimalStep : 1.0;
    const double ds = sh.sumDualStep > 0.0 ? 1.0 / sh.sumDualStep : 1.0;
    for (int i = lb * kVecThreads + tid; i < n + m; i += G * kVecThreads) {
      if (i < n) {
        double sx = ldStream(v.xSum + i);
        // (xSum / ySum are written again in phase V, by another workgroup on maybe another XCD: two dirty copies of a line
        // in two L2s would be written back in no particular order — both writers store through to memory)
        if (wx != 0.0) { sx = sx + wx * ldStream(v.x[cur] + i); stAgent(v.xSum + i, sx); }
        stAgent(a.xAvg + i, sx * ps);
      } else {
        const int k = i - n;
        double sy = ldStream(v.ySum + k);
        if (w != 0.0) { sy = sy + w * ldStream(v.y[cur] + k); stAgent(v.ySum + k, sy); }
        stAgent(a.yAvg + k, sy * ds);
      }
    }
  }
  meet();
  // ---- S: A xAvg and A' yAvg ----
  for (int b = lb; b < a.A.nBlocks; b += G) plainSpmvBlock<CHUNK_A>(a.A, b, a.xAvg, a.axAvg, prod);
  for (int tb = lb; tb * (kSpmvThreads / kWave) < a.LA.nTasks; tb += G) plainLongBlock(a.LA, tb, a.xAvg, a.axAvg, scratch[0]);
  for (int b = lb; b < a.At.nBlocks; b += G) plainSpmvBlock<CHUNK_AT>(a.At, b, a.yAvg, a.atyAvg, prod);
  for (int tb = lb; tb * (kSpmvThreads / kWave) < a.LAt.nTasks; tb += G) plainLongBlock(a.LAt, tb, a.yAvg, a.atyAvg, scratch[0]);
  meet();
  // ---- R: statistics of both iterates on the grids of k_row_stats2 / k_col_stats2 ----
  const int nbM = vecBlocksDev(m > 0 ? m : 1), nbN = vecBlocksDev(n > 0 ? n : 1);
  for (int vb = lb; vb < nbM; vb += G) {
    double acc[2 * kRowStats];
#pragma unroll
    for (int q = 0; q < 2 * kRowStats; ++q) acc[q] = 0.0;
    for (int i = vb * kVecThreads + tid; i < m; i += nbM * kVecThreads)
      rowStatsElem<true>(acc, i, v.ax[cur], v.y[cur], a.axAvg, a.yAvg, v.rhs, a.rowScale, a.scaled, (i + v.rowOffset) >= v.nEqs);
    blockSumManyAt<2 * kRowStats, true>(acc, scratch, a.statPart + (size_t)kStatRowCur * a.statStride, a.statStride, vb);
    __syncthreads();
  }
  {
    const ColStatPtrs p{v.aty[cur], v.x[cur], a.atyAvg, a.xAvg, v.cost, v.lower, v.upper, a.colScale, v.qdiag, nullptr, nullptr,
                        a.spC, a.snC, a.spA, a.snA};
    for (int vb = lb; vb < nbN; vb += G) {
      double acc[2 * kColStats];
#pragma unroll
      for (int q = 0; q < 2 * kColStats; ++q) acc[q] = 0.0;
      for (int j = vb * kVecThreads + tid; j < n; j += nbN * kVecThreads) colStatsElem<true>(acc, j, p, a.scaled);
      blockSumManyAt<2 * kColStats, true>(acc, scratch, a.statPart + (size_t)kStatColCur * a.statStride, a.statStride, vb);
      __syncthreads();
    }
  }
  meet();
  // ---- Q: the 30 fixed-order sums (k_final_reduce2) ----
  for (int q = lb; q < kStatTotal; q += G) {
    const double r = reducePartialsAgent(a.statPart + (size_t)q * a.statStride, q < 2 * kRowStats ? nbM : nbN, scratch[0]);
    if (tid == 0) stAgent(a.statOut + q, r);
  }
  meet();
  // ---- D: residuals, termination, restart decision: the same in every workgroup ----
  if (tid < kStatTotal) stat[tid] = ldAgent(a.statOut + tid);
  __syncthreads();
  if (tid == 0) flag = checkDecideCore(sh, ctl, stat) ? 1 : 0;
  __syncthreads();
  const bool over = flag != 0;
  const int kind = ctl.restartKind;
  double dP2 = 0.0, dD2 = 0.0;
  if (!over && kind) {
    // ---- V: the vector side of the restart on the grid of k_restart_vec ----
    for (int vb = lb; vb < nbN + nbM; vb += G) restartVecBlock<true>(v, cur, kind, a.r, vb, a.partX, nbN, a.partY, nbM, scratch[0]);
    meet();
    dP2 = reducePartialsAgent(a.partX, nbN, scratch[0]);
    dD2 = reducePartialsAgent(a.partY, nbM, scratch[0]);
  }
  // ---- W: primal weight, step sizes, next halt; workgroup 0 hands the records on ----
  if (lb != 0) {
    // (a workgroup whose own barrier failed may be the only one that knows: it raises the flag workgroup 0 reads — it
    // is already set when the failure was a timeout, this covers a poisoned word met on the way)
    if (tid == 0 && barBad) __hip_atomic_store(a.bar + G, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (tid == 0) {
    const unsigned long long timedOut = __hip_atomic_load(a.bar + G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (timedOut) barBad = 1;
  }
  __syncthreads();
  if (barBad) {  // the phases above may have run unsynchronised: nothing of this check is handed on
    if (tid == 0) {
      __hip_atomic_store(&a.st->commError, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.st->halted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (tid == 0 && !over) restartFinishCore(sh, ctl, dP2, dD2);
  __syncthreads();
  for (int w = tid; w < (int)(sizeof(DevState) / 4); w += kVecThreads) reinterpret_cast<uint32_t*>(a.st)[w] = reinterpret_cast<const uint32_t*>(&sh)[w];
  for (int w = tid; w < (int)(sizeof(CheckCtl) / 4); w += kVecThreads) reinterpret_cast<uint32_t*>(a.cc)[w] = reinterpret_cast<const uint32_t*>(&ctl)[w];
  if (tid == 0) writeRecord(a.rec, sh, ctl);
}

using CheckSmallKernel = void (*)(const CheckSmallArgs);
CheckSmallKernel pickCheckSmall(int chunkA, int chunkAt) {
  if (chunkA == kChunkSmall && chunkAt == kChunkSmall) return k_check_small<kChunkSmall, kChunkSmall>;
  if (chunkA == kChunk && chunkAt == kChunk) return k_check_small<kChunk, kChunk>;
  if (chunkA == kChunk && chunkAt == kChunkSmall) return k_check_small<kChunk, kChunkSmall>;
  if (chunkA == kChunkSmall && chunkAt == kChunk) return k_check_small<kChunkSmall, kChunk>;
  return nullptr;
}

}  // namespace

void launchCheckDecide(DevState* st, CheckCtl* cc, const double* stat, CheckRecord* rec, hipStream_t s) {
  hipLaunchKernelGGL(k_check_decide, dim3(1), dim3(kWave), 0, s, st, cc, stat, rec);
}
void launchRestartVec(const IterVecs& v, const DevState* st, const CheckCtl* cc, const RestartVecs& r, double* partX, int32_t nbX,
                      double* partY, int32_t nbY, hipStream_t s) {
  hipLaunchKernelGGL(k_restart_vec, dim3(nbX + nbY), dim3(kVecThreads), 0, s, v, st, cc, r, partX, nbX, partY, nbY);
}
void launchRestartCopyFull(const DevState* st, const CheckCtl* cc, double* const x[2], double* const aty[2], double* const yFull[2],
                           const double* xAvg, const double* atyAvg, const double* yAvgFull, int32_t n, int32_t yLen, hipStream_t s) {
  const int32_t len = n > yLen ? n : yLen;
  hipLaunchKernelGGL(k_restart_copy_full, dim3(vecBlocks(len > 0 ? len : 1)), dim3(kVecThreads), 0, s, st, cc, x[0], x[1], aty[0], aty[1],
                     yFull ? yFull[0] : (double*)nullptr, yFull ? yFull[1] : (double*)nullptr, xAvg, atyAvg, yAvgFull, n, yLen);
}
void launchRestartFinish(DevState* st, CheckCtl* cc, const double* partX, int32_t nbX, const double* partY, int32_t nbY,
                         CheckRecord* rec, hipStream_t s) {
  hipLaunchKernelGGL(k_restart_finish, dim3(1), dim3(kVecThreads), 0, s, st, cc, partX, nbX, partY, nbY, rec);
}

// Workgroups the device keeps resident of the one-launch check (0: this pair of operands does not qualify)
int checkSmallResident(const MatView& A, const MatView& At, int device) {
  if (A.useSlab || At.useSlab || A.lng.contrib != nullptr || At.lng.contrib != nullptr) return 0;
  CheckSmallKernel k = pickCheckSmall(A.csr.chunk, At.csr.chunk);
  if (!k) return 0;
  int perCu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k, kVecThreads, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  return (perCu < 4 ? perCu : 4) * cus;
}
void launchCheckSmall(const MatView& A, const MatView& At, const IterVecs& v, DevState* st, CheckCtl* cc, CheckRecord* rec,
                      const RestartVecs& r, const double* rowScale, const double* colScale, int scaled, double* spC, double* snC,
                      double* spA, double* snA, double* statPart, int32_t statStride, double* statOut, double* partX, double* partY,
                      unsigned long long* bar, int32_t grid, unsigned long long seq, int32_t timeoutMs, hipStream_t s) {
  CheckSmallArgs a{};
  a.A = A.csr; a.At = At.csr; a.LA = A.lng; a.LAt = At.lng; a.v = v; a.st = st; a.cc = cc; a.rec = rec; a.r = r;
  a.xAvg = const_cast<double*>(r.xAvg); a.yAvg = const_cast<double*>(r.yAvg);
  a.axAvg = const_cast<double*>(r.axAvg); a.atyAvg = const_cast<double*>(r.atyAvg);
  a.rowScale = rowScale; a.colScale = colScale; a.scaled = scaled;
  a.spC = spC; a.snC = snC; a.spA = spA; a.snA = snA;
  a.statPart = statPart; a.statStride = statStride; a.statOut = statOut; a.partX = partX; a.partY = partY;
  a.bar = bar; a.seq = seq;
  a.limit = (unsigned long long)(timeoutMs > 0 ? timeoutMs : 1000) * 100000ull;
  hipLaunchKernelGGL(pickCheckSmall(A.csr.chunk, At.csr.chunk), dim3(grid), dim3(kVecThreads), 0, s, a);
}

}  // namespace pdlp
