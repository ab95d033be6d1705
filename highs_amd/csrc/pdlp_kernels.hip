// pdlp_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the PDLP hot loop.
//
// The loop is HBM-bound (two fp64 SpMVs + level-1 passes, ~0.1 flop/byte), so
// the rules that matter are: coalesced streaming of the CSR/CSC arrays, many
// independent loads in flight per lane, no re-reads (every level-1 pass is
// fused into the SpMV that produces or consumes the vector), deterministic
// two-stage reductions (no float atomics), and no host round trip per trial.
// MFMA is not used: nothing here is a dense contraction.
//
// Compiled with -ffp-contract=off: the element-wise updates then round exactly
// like the reference's scalar CPU loops (mul, then add), which makes x+, y+,
// A x+ and A' y+ of a trial step bit-identical to the oracle; only the
// reductions (tree order here, left-to-right there) differ in the last bits.
#include "pdlp_kernels.hpp"

#include <cmath>

#include "pdlp_devfn.hpp"

namespace pdlp {

namespace {

enum Epilogue { kPlain = 0, kDualStep = 1, kAtyInteract = 2, kAtyPartial = 3, kHalpernPrimal = 4, kHalpernDual = 5 };
constexpr bool usesDevState(int epi) { return epi == kDualStep || epi == kAtyInteract || epi == kAtyPartial; }

struct SpmvArgs {
  SpmvMat A;
  SlabMat S;
  const DevState* st;  // nullptr for kPlain
  // kPlain / kAtyPartial
  const double* in;
  double* out;
  // iteration vectors (kDualStep / kAtyInteract / kAtyPartial)
  IterVecs v;
  double* part0;  // dY^2 (dual) | dX^2 (aty)
  double* part1;  // interaction (aty)
  HalpernVecs h;  // kHalpernPrimal / kHalpernDual
};

// CSR-adaptive SpMV (stream + long-row paths) with a fused, major-local epilogue.
// One work block = up to kChunk consecutive nonzeros belonging to whole majors.
//   phase 1: all lanes stream val[]/idx[] with unit stride (coalesced), gather
//            the input vector, and park the products in LDS;
//   phase 2: one lane per major adds its products left to right (the
//            reference's summation order) and runs the epilogue.
// Block-uniform read through the scalar (constant) path: s_load counts on lgkmcnt,
// so it never forces a wait on the vector-memory prefetches in flight.
template <typename T>
__device__ __forceinline__ T ldUniform(const T* p) {
  return *(const __attribute__((address_space(4))) T*)(p);
}

template <int EPI, bool MAPPED>
__global__ __launch_bounds__(kSpmvThreads) void k_spmv(const SpmvArgs a) {
  const DevState* st = a.st;
  if (usesDevState(EPI) && st->halted) return;
  __shared__ double prod[kChunk + kChunk / 8 + 8];
  __shared__ double scratch[2][kSpmvThreads / kWave];

  const int tid = threadIdx.x;
  const int blk = blockIdx.x;
  const int r0 = a.A.blockBeg[blk], r1 = a.A.blockBeg[blk + 1];
  const int p0 = a.A.beg[r0], p1 = a.A.beg[r1];
  const int32_t* __restrict__ idx = a.A.idx;
  const double* __restrict__ val = a.A.val;

  int cur = 0, nxt = 1;
  double sigma = 0.0, avgW = 0.0;
  if (usesDevState(EPI)) {
    cur = st->cur;
    nxt = cur ^ 1;
    sigma = st->sigma;
    avgW = st->avgW;
  }
  double hTau = 0.0, hRho = 1.0, hW = 0.0;
  if (EPI == kHalpernPrimal || EPI == kHalpernDual) {
    const HalpernState hs = *a.h.hs;
    hTau = hs.tau; sigma = hs.sigma; hRho = hs.rho;
    const int k = hs.hIter + a.h.kOff;
    hW = (double)k / ((double)k + 1.0);
  }
  const double* __restrict__ in;
  if (EPI == kPlain) in = a.in;
  else if (EPI == kDualStep) in = a.v.x[nxt];
  else if (EPI == kHalpernPrimal) in = a.h.yc;
  else if (EPI == kHalpernDual) in = a.h.rx;
  else in = a.v.y[nxt];

  double acc0 = 0.0, acc1 = 0.0;  // per-thread epilogue partials

  // Epilogue operands that do not depend on the SpMV result are fetched early
  // (before the products are staged) so their latency overlaps the stream.
  auto prefetch = [&](int r) -> Pre {
    Pre p{0.0, 0.0, 0.0, 0.0, 0.0};
    if (MAPPED) r = a.A.majorMap[r];
    if (EPI == kDualStep) {
      p.a = a.v.y[cur][r]; p.b = a.v.rhs[r]; p.c = a.v.ax[cur][r];
    } else if (EPI == kAtyInteract) {
      p.a = a.v.x[cur][r]; p.b = a.v.x[nxt][r]; p.c = a.v.aty[cur][r];
    } else if (EPI == kHalpernPrimal) {
      p.a = a.h.xc[r]; p.b = a.h.cost[r]; p.c = a.h.xa[r]; p.d = a.h.lower[r]; p.e = a.h.upper[r];
    } else if (EPI == kHalpernDual) {
      p.a = a.h.yc[r]; p.b = a.h.ya[r]; p.c = a.h.rowLower[r]; p.d = a.h.rowUpper[r];
    }
    return p;
  };
  auto epilogue = [&](int r, double s, const Pre& p) {
    if (MAPPED) r = a.A.majorMap[r];
    if (EPI == kPlain || EPI == kAtyPartial) {
      a.out[r] = s;
    } else if (EPI == kHalpernPrimal) {
      halpernPrimal(a.h, r, s, p, hTau, hRho, hW);
    } else if (EPI == kHalpernDual) {
      halpernDual(a.h, r, s, p, sigma, hRho, hW);
    } else if (EPI == kDualStep) {
      // y+ = proj(y + sigma*(b - 2 A x+ + A x)), cupdlp_step.c:43-69
      const double yv = p.a;
      if (avgW != 0.0) a.v.ySum[r] += avgW * yv;  // deferred PDHG_Update_Average (step.c:438)
      double t = yv;
      t += sigma * p.b;
      t += (-2.0 * sigma) * s;
      t += sigma * p.c;
      if (r + a.v.rowOffset >= a.v.nEqs) t = t > 0.0 ? t : 0.0;
      a.v.ax[nxt][r] = s;
      a.v.y[nxt][r] = t;
      const double d = yv - t;
      acc0 += d * d;
    } else {  // kAtyInteract: cupdlp_linalg.c:772-801
      const double dx = p.a - p.b;
      const double da = p.c - s;
      a.v.aty[nxt][r] = s;
      acc0 += dx * dx;
      acc1 += dx * da;
    }
  };

  if (r1 - r0 == 1 && p1 - p0 > kChunk) {
    // long major: the whole block strides over it; tree-reduced (deterministic)
    double s = 0.0;
    for (int p = p0 + tid; p < p1; p += kSpmvThreads) s += val[p] * in[idx[p]];
    s = blockSum<kSpmvThreads>(s, scratch[0]);
    if (tid == 0) epilogue(r0, s, prefetch(r0));
  } else {
    constexpr int kPer = kChunk / kSpmvThreads;
    const int cnt = p1 - p0;
    // Bookkeeping of this lane's first major, issued ahead of the stream.  All
    // loads below are unconditional with clamped indices: a load inside an
    // exec-masked branch makes hipcc drain vmcnt at the join, which serialises
    // the stream (one idx/val pair in flight instead of 2*kPer).
    const int rFirst = r0 + tid;
    const int rr = rFirst < r1 ? rFirst : r1 - 1;
    int qb = a.A.beg[rr] - p0;
    int qe = a.A.beg[rr + 1] - p0;
    Pre pre = prefetch(rr);
    // phase 1: kPer unit-stride loads of idx/val per lane, all issued before the
    // dependent gathers, so a wave keeps 3*kPer memory operations in flight
    const int last = cnt > 0 ? cnt - 1 : 0;  // idx/val carry one pad element
    int32_t ci[kPer];
    double va[kPer], xg[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int q = tid + k * kSpmvThreads;
      const int qq = q < last ? q : last;
      ci[k] = idx[p0 + qq];
      va[k] = val[p0 + qq];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) xg[k] = in[ci[k]];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int q = tid + k * kSpmvThreads;
      if (q < cnt) prod[slot(q)] = va[k] * xg[k];
    }
    __syncthreads();
    // phase 2: one lane per major, products added left to right
    for (int r = rFirst; r < r1; r += kSpmvThreads) {
      if (r != rFirst) {
        qb = a.A.beg[r] - p0;
        qe = a.A.beg[r + 1] - p0;
        pre = prefetch(r);
      }
      double s = 0.0;
      int q = qb;
      for (; q + 4 <= qe; q += 4) {
        const double t0 = prod[slot(q)], t1 = prod[slot(q + 1)], t2 = prod[slot(q + 2)], t3 = prod[slot(q + 3)];
        s += t0; s += t1; s += t2; s += t3;
      }
      for (; q < qe; ++q) s += prod[slot(q)];
      epilogue(r, s, pre);
    }
  }

  if (EPI == kDualStep) {
    const double t = blockSum<kSpmvThreads>(acc0, scratch[0]);
    if (tid == 0) a.part0[a.A.partOffset + blk] = t;
  } else if (EPI == kAtyInteract) {
    const double t0 = blockSum<kSpmvThreads>(acc0, scratch[0]);
    const double t1 = blockSum<kSpmvThreads>(acc1, scratch[1]);
    if (tid == 0) { a.part0[a.A.partOffset + blk] = t0; a.part1[a.A.partOffset + blk] = t1; }
  }
}

// Slab SpMV: one block owns rowsPerBlock consecutive majors and streams ITS
// nonzeros, which the host sorted by (slab of the gathered vector, local major,
// minor).  All resident blocks walk the slabs in the same order, so the 512 KB
// slab currently gathered from stays in every XCD's L2 instead of costing one
// 64-byte fabric request per 8-byte gather.  Per 256-entry window: products go
// to LDS, the first lane of each run of equal majors adds the run, left to
// right, onto the major's LDS accumulator.  Because slabs and the minors inside
// a slab ascend, every major is still summed in ascending minor order — the
// reference's order — and the result is bit-identical to the CSR path.
template <int EPI>
__global__ __launch_bounds__(kSlabThreads, 8) void k_spmv_slab(const SpmvArgs a) {
  const DevState* st = a.st;
  if (usesDevState(EPI) && st->halted) return;
  // dynamic LDS (all carve offsets are multiples of 16 bytes; no static __shared__ in this kernel):
  //   acc[R] f64 | stage[2][256] f64 | scratch[2][4] f64 | srow[2][256] u16
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* acc = reinterpret_cast<double*>(smem);
  double(*stage)[kSlabThreads] = reinterpret_cast<double(*)[kSlabThreads]>(acc + a.S.rowsPerBlock);
  double(*scratch)[kSlabThreads / kWave] = reinterpret_cast<double(*)[kSlabThreads / kWave]>(&stage[2][0]);
  uint16_t(*srow)[kSlabThreads] = reinterpret_cast<uint16_t(*)[kSlabThreads]>(&scratch[2][0]);

  const int tid = threadIdx.x;
  const int blk = blockIdx.x;
  const int R = a.S.rowsPerBlock;
  const int rBase = blk * R;
  const int rEnd = (rBase + R < a.S.nMajor) ? rBase + R : a.S.nMajor;
  const uint32_t* __restrict__ ent = a.S.ent;
  const double* __restrict__ val = a.S.val;
  // this block's static window list (block-uniform -> scalar loads, which do not
  // touch the vector-memory counter the prefetches below depend on)
  const int wBeg = a.S.winPtr[blk], wEnd = a.S.winPtr[blk + 1];
  const int32_t* __restrict__ winBeg = a.S.winBeg;
  const uint32_t* __restrict__ winInfo = a.S.winInfo;

  int cur = 0, nxt = 1;
  double sigma = 0.0, avgW = 0.0;
  if (usesDevState(EPI)) {
    cur = st->cur;
    nxt = cur ^ 1;
    sigma = st->sigma;
    avgW = st->avgW;
  }
  double hTau = 0.0, hRho = 1.0, hW = 0.0;
  if (EPI == kHalpernPrimal || EPI == kHalpernDual) {
    const HalpernState hs = *a.h.hs;
    hTau = hs.tau; sigma = hs.sigma; hRho = hs.rho;
    const int k = hs.hIter + a.h.kOff;
    hW = (double)k / ((double)k + 1.0);
  }
  const double* __restrict__ in;
  if (EPI == kPlain) in = a.in;
  else if (EPI == kDualStep) in = a.v.x[nxt];
  else if (EPI == kHalpernPrimal) in = a.h.yc;
  else if (EPI == kHalpernDual) in = a.h.rx;
  else in = a.v.y[nxt];

  for (int r = tid; r < R; r += kSlabThreads) acc[r] = 0.0;

  double acc0 = 0.0, acc1 = 0.0;
  auto prefetch = [&](int r) -> Pre {
    Pre p{0.0, 0.0, 0.0, 0.0, 0.0};
    if (EPI == kDualStep) { p.a = a.v.y[cur][r]; p.b = a.v.rhs[r]; p.c = a.v.ax[cur][r]; }
    else if (EPI == kAtyInteract) { p.a = a.v.x[cur][r]; p.b = a.v.x[nxt][r]; p.c = a.v.aty[cur][r]; }
    else if (EPI == kHalpernPrimal) {
      p.a = a.h.xc[r]; p.b = a.h.cost[r]; p.c = a.h.xa[r]; p.d = a.h.lower[r]; p.e = a.h.upper[r];
    } else if (EPI == kHalpernDual) {
      p.a = a.h.yc[r]; p.b = a.h.ya[r]; p.c = a.h.rowLower[r]; p.d = a.h.rowUpper[r];
    }
    return p;
  };
  // operands of this lane's first two majors, fetched ahead of the stream (clamped, unconditional)
  const int rA = rBase + tid < rEnd ? rBase + tid : rEnd - 1;
  const int rB = rBase + tid + kSlabThreads < rEnd ? rBase + tid + kSlabThreads : rEnd - 1;
  const Pre preA = prefetch(rA), preB = prefetch(rB);

  __syncthreads();  // acc[] is zeroed
  // Windows of up to 256 entries that never straddle a slab boundary: inside a
  // window a major forms ONE run (entries are sorted by major within the slab),
  // so each accumulator has a single writer per window and the barrier between
  // windows orders the runs of a major slab after slab.
  struct Win { int beg, cnt, slab; };
  auto getWin = [&](int i) -> Win {  // block-uniform; windows past the end are empty
    int ic = i < wEnd ? i : wEnd - 1;  // winBeg/winInfo carry one pad element
    ic = __builtin_amdgcn_readfirstlane(ic < 0 ? 0 : ic);  // SGPR index -> s_load (lgkmcnt, not vmcnt)
    const uint32_t info = ldUniform(winInfo + ic);
    const int beg = ldUniform(winBeg + ic);
    return Win{beg, i < wEnd ? (int)(info & 0xffffu) : 0, (int)(info >> 16)};
  };
  // All loads are unconditional (lanes past the window read the next entries, the
  // arrays carry a pad element) so that hipcc never drains vmcnt at a branch join.
  auto gatherIdx = [&](const Win& w, uint32_t en) -> size_t {
    return tid < w.cnt ? (((size_t)w.slab << 16) + (en & 0xffffu)) : 0;
  };
  // One window per iteration, nothing carried across iterations: entries -> gather -> LDS.
  // More memory-level parallelism was measured to HURT: batching the gathers of 2/4/8 windows
  // (69/79/86 us vs 54 us) or prefetching the entry stream 4/8/12 windows ahead (62/64/67 us)
  // lets the resident blocks drift over more slabs than the L2 holds.  Locality beats MLP here.
  for (int wi = wBeg; wi < wEnd; ++wi) {
    const Win w = getWin(wi);
    const uint32_t en = ent[w.beg + tid];
    const double vv = val[w.beg + tid];
    const double xg = in[gatherIdx(w, en)];
    const int buf = (wi - wBeg) & 1;  // alternate LDS staging buffers: one barrier per window
    const bool valid = tid < w.cnt;
    const uint32_t lrow = en >> 16;
    const double prod = vv * xg;
    stage[buf][tid] = prod;
    srow[buf][tid] = valid ? (uint16_t)lrow : (uint16_t)0xffff;
    __syncthreads();
    if (valid && (tid == 0 || srow[buf][tid - 1] != (uint16_t)lrow)) {
      double s = acc[lrow];
      s += prod;
      for (int j = tid + 1; j < kSlabThreads && srow[buf][j] == (uint16_t)lrow; ++j) s += stage[buf][j];
      acc[lrow] = s;
    }
  }
  __syncthreads();

  auto epilogue = [&](int r, double s, const Pre& p) {
    if (EPI == kPlain || EPI == kAtyPartial) {
      a.out[r] = s;
    } else if (EPI == kHalpernPrimal) {
      halpernPrimal(a.h, r, s, p, hTau, hRho, hW);
    } else if (EPI == kHalpernDual) {
      halpernDual(a.h, r, s, p, sigma, hRho, hW);
    } else if (EPI == kDualStep) {
      const double yv = p.a;
      if (avgW != 0.0) a.v.ySum[r] += avgW * yv;
      double t = yv;
      t += sigma * p.b;
      t += (-2.0 * sigma) * s;
      t += sigma * p.c;
      if (r + a.v.rowOffset >= a.v.nEqs) t = t > 0.0 ? t : 0.0;
      a.v.ax[nxt][r] = s;
      a.v.y[nxt][r] = t;
      const double d = yv - t;
      acc0 += d * d;
    } else {
      const double dx = p.a - p.b;
      const double da = p.c - s;
      a.v.aty[nxt][r] = s;
      acc0 += dx * dx;
      acc1 += dx * da;
    }
  };
  const uint32_t* __restrict__ mask = a.S.longMask + (size_t)blk * (R / 32);
  for (int lr = tid, it = 0; rBase + lr < rEnd; lr += kSlabThreads, ++it) {
    if ((mask[lr >> 5] >> (lr & 31)) & 1u) continue;  // long major: the CSR side kernel owns it
    const int r = rBase + lr;
    const Pre p = it == 0 ? preA : (it == 1 ? preB : prefetch(r));
    epilogue(r, acc[lr], p);
  }

  if (EPI == kDualStep) {
    const double t = blockSum<kSlabThreads>(acc0, scratch[0]);
    if (tid == 0) a.part0[blk] = t;
  } else if (EPI == kAtyInteract) {
    const double t0 = blockSum<kSlabThreads>(acc0, scratch[0]);
    const double t1 = blockSum<kSlabThreads>(acc1, scratch[1]);
    if (tid == 0) { a.part0[blk] = t0; a.part1[blk] = t1; }
  }
}

// x+ = clamp(x - tau (c - A'y), l, u): cupdlp_step.c:16-40, rounding as the CPU branch.
__global__ __launch_bounds__(kVecThreads) void k_primal_step(const IterVecs v, const DevState* st) {
  if (st->halted) return;
  const int cur = st->cur, nxt = cur ^ 1;
  const double tau = st->tau, avgW = st->avgW;
  const double* __restrict__ x = v.x[cur];
  const double* __restrict__ aty = v.aty[cur];
  double* __restrict__ xn = v.x[nxt];
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
    const double xv = x[j];
    if (avgW != 0.0) v.xSum[j] += avgW * xv;  // deferred PDHG_Update_Average (step.c:437)
    double t = xv;
    t += (-tau) * v.cost[j];
    t += tau * aty[j];
    const double u = v.upper[j], l = v.lower[j];
    t = t < u ? t : u;
    t = t > l ? t : l;
    xn[j] = t;
  }
}

// Sharded: movement/interaction after the all-reduce of the A_g' y partials.
__global__ __launch_bounds__(kVecThreads) void k_interact(const IterVecs v, const DevState* st,
                                                          const double* __restrict__ atyReduced, double* partDX,
                                                          double* partInter) {
  if (st->halted) return;
  __shared__ double scratch[2][kVecThreads / kWave];
  const int cur = st->cur, nxt = cur ^ 1;
  double a0 = 0.0, a1 = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
    const double s = atyReduced[j];
    const double dx = v.x[cur][j] - v.x[nxt][j];
    const double da = v.aty[cur][j] - s;
    v.aty[nxt][j] = s;
    a0 += dx * dx;
    a1 += dx * da;
  }
  const double t0 = blockSum<kVecThreads>(a0, scratch[0]);
  const double t1 = blockSum<kVecThreads>(a1, scratch[1]);
  if (threadIdx.x == 0) { partDX[blockIdx.x] = t0; partInter[blockIdx.x] = t1; }
}

__global__ __launch_bounds__(kVecThreads) void k_reduce_to(const double* partials, int count, double* out,
                                                           const DevState* st) {
  if (st && st->halted) return;
  __shared__ double scratch[kVecThreads / kWave];
  const double s = reducePartials(partials, count, scratch);
  if (threadIdx.x == 0) *out = s;
}

// k_decide: the decision kernel (one block).  All partial loads of the three sums are issued
// together and reduced in one pass — the kernel is pure latency (it sits between two trials).
__global__ __launch_bounds__(kVecThreads) void k_decide(DevState* st, const double* __restrict__ partDY, int nDY,
                                                        const double* __restrict__ partDX,
                                                        const double* __restrict__ partInter, int nDX,
                                                        const double* dyGlobal) {
  if (st->halted) return;
  __shared__ double scratch[3][kVecThreads / kWave];
  const int tid = threadIdx.x;
  // fixed order: lane t sums elements t, t+256, ... (4 independent chains), then wave/LDS tree
  auto laneSum = [&](const double* __restrict__ p, int count) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = tid;
    for (; i + 3 * kVecThreads < count; i += 4 * kVecThreads) {
      const double a0 = p[i], a1 = p[i + kVecThreads], a2 = p[i + 2 * kVecThreads], a3 = p[i + 3 * kVecThreads];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; i < count; i += kVecThreads) s0 += p[i];
    return (s0 + s1) + (s2 + s3);
  };
  double vY = dyGlobal ? 0.0 : laneSum(partDY, nDY);
  double vX = laneSum(partDX, nDX);
  double vI = laneSum(partInter, nDX);
  vY = waveSum(vY); vX = waveSum(vX); vI = waveSum(vI);
  const int lane = tid & (kWave - 1), w = tid / kWave;
  if (lane == 0) { scratch[0][w] = vY; scratch[1][w] = vX; scratch[2][w] = vI; }
  __syncthreads();
  if (tid != 0) return;
  double dY2 = 0.0, dX2 = 0.0, inter = 0.0;
#pragma unroll
  for (int i = 0; i < kVecThreads / kWave; ++i) { dY2 += scratch[0][i]; dX2 += scratch[1][i]; inter += scratch[2][i]; }
  decideUpdate(st, dX2, dyGlobal ? *dyGlobal : dY2, inter);
}

// Apply a pending average update (before a check iteration reads xSum/ySum).
__global__ __launch_bounds__(kVecThreads) void k_flush_average(const IterVecs v, const DevState* st) {
  const double w = st->avgW;
  if (w == 0.0) return;
  const int cur = st->cur;
  const int stride = gridDim.x * blockDim.x;
  const int tot = v.n + v.m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
    if (i < v.n) v.xSum[i] += w * v.x[cur][i];
    else v.ySum[i - v.n] += w * v.y[cur][i - v.n];
  }
}
__global__ void k_clear_avgw(DevState* st) { st->avgW = 0.0; }

__global__ __launch_bounds__(kVecThreads) void k_scale_copy(double* __restrict__ dst, const double* __restrict__ src,
                                                            double a, int len) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) dst[i] = src[i] * a;
}
__global__ __launch_bounds__(kVecThreads) void k_fill(double* dst, double value, int len) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) dst[i] = value;
}
__global__ __launch_bounds__(kVecThreads) void k_project(double* x, const double* lower, const double* upper, int n) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double t = x[i];
    t = t < upper[i] ? t : upper[i];  // projub then projlb, cupdlp_proj.c:17-25
    t = t > lower[i] ? t : lower[i];
    x[i] = t;
  }
}
__global__ __launch_bounds__(kVecThreads) void k_mul(double* x, const double* y, int len) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) x[i] *= y[i];
}
__global__ __launch_bounds__(kVecThreads) void k_div(double* x, const double* y, int len) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) x[i] /= y[i];
}

// Row pass of PDHG_Compute_Primal_Feasibility / the y-side of the dual
// objective and of the infeasibility certificates (cupdlp_solver.c:12-67,80,230,339-345).
__global__ __launch_bounds__(kVecThreads) void k_row_stats(const double* __restrict__ ax, const double* __restrict__ y,
                                                           const double* __restrict__ rhs,
                                                           const double* __restrict__ rowScale, int m, int nEqs,
                                                           int rowOffset, int scaled, double* partials, int pstride) {
  __shared__ double scratch[kVecThreads / kWave];
  double a[kRowStats] = {0.0, 0.0, 0.0, 0.0};
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const bool ineq = (i + rowOffset) >= nEqs;
    const double axv = ax[i], yv = y[i], b = rhs[i];
    const double rs = scaled ? rowScale[i] : 1.0;
    double r = axv + (-1.0) * b;
    if (ineq) r = r < 0.0 ? r : 0.0;
    r *= rs;
    a[0] += r * r;
    a[1] += yv * b;
    a[2] += yv * yv;
    double c = axv;
    if (ineq) c = c < 0.0 ? c : 0.0;
    c *= rs;
    a[3] += c * c;
  }
#pragma unroll
  for (int q = 0; q < kRowStats; ++q) {
    const double t = blockSum<kVecThreads>(a[q], scratch);
    if (threadIdx.x == 0) partials[q * pstride + blockIdx.x] = t;
  }
}

// Column pass of PDHG_Compute_Dual_Feasibility and the x-side certificates
// (cupdlp_solver.c:69-204, 229-256, 326-366); also stores the slacks s+, s-.
__global__ __launch_bounds__(kVecThreads) void k_col_stats(const double* __restrict__ aty, const double* __restrict__ x,
                                                           const double* __restrict__ cost,
                                                           const double* __restrict__ lower,
                                                           const double* __restrict__ upper,
                                                           const double* __restrict__ colScale, int n, int scaled,
                                                           double* slackPos, double* slackNeg, double* partials,
                                                           int pstride) {
  __shared__ double scratch[kVecThreads / kWave];
  double a[kColStats];
#pragma unroll
  for (int q = 0; q < kColStats; ++q) a[q] = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const double xv = x[j], c = cost[j], l = lower[j], u = upper[j];
    const double cs = scaled ? colScale[j] : 1.0;
    const double hasL = l > -INFINITY ? 1.0 : 0.0, hasU = u < INFINITY ? 1.0 : 0.0;
    const double lF = l > -INFINITY ? l : 0.0, uF = u < INFINITY ? u : 0.0;
    const double atyv = aty[j];
    double r = -atyv + c;                       // c - A'y
    double sp = (r > 0.0 ? r : 0.0) * hasL;     // s+ (:157-159)
    double sn = (-(r < 0.0 ? r : 0.0)) * hasU;  // s- (:171-175)
    slackPos[j] = sp;
    slackNeg[j] = sn;
    a[0] += xv * c;
    a[1] += sp * lF;
    a[2] += sn * uF;
    double rd = r + (-1.0) * sp;
    rd += sn;
    rd *= cs;
    a[3] += rd * rd;
    a[4] += sp * sp;
    a[5] += sn * sn;
    double pc = (atyv + sp) - sn;
    pc *= cs;
    a[6] += pc * pc;
    a[7] += xv * xv;
    double lb = (xv < 0.0 ? xv : 0.0) * hasL;
    double ub = (xv > 0.0 ? xv : 0.0) * hasU;
    if (scaled) { lb /= cs; ub /= cs; }
    a[8] += lb * lb;
    a[9] += ub * ub;
  }
#pragma unroll
  for (int q = 0; q < kColStats; ++q) {
    const double t = blockSum<kVecThreads>(a[q], scratch);
    if (threadIdx.x == 0) partials[q * pstride + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kVecThreads) void k_final_reduce(const double* partials, int pstride, int nBlocks,
                                                              double* out) {
  __shared__ double scratch[kVecThreads / kWave];
  const double s = reducePartials(partials + (size_t)blockIdx.x * pstride, nBlocks, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(kVecThreads) void k_diff_norm2(const double* __restrict__ a, const double* __restrict__ b,
                                                            int len, double* partials) {
  __shared__ double scratch[kVecThreads / kWave];
  double s = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
    const double d = a[i] - b[i];
    s += d * d;
  }
  const double t = blockSum<kVecThreads>(s, scratch);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
__global__ __launch_bounds__(kVecThreads) void k_dot(const double* __restrict__ a, const double* __restrict__ b,
                                                     int len, double* partials) {
  __shared__ double scratch[kVecThreads / kWave];
  double s = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) s += a[i] * b[i];
  const double t = blockSum<kVecThreads>(s, scratch);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

}  // namespace

// Memory-bound vector kernels: cap the grid at 2048 blocks (8 per CU) and
// grid-stride the rest.
int32_t vecBlocks(int32_t len) {
  int64_t b = ((int64_t)len + kVecThreads - 1) / kVecThreads;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int32_t)b;
}

void launchPrimalStep(const IterVecs& v, const DevState* st, hipStream_t s) {
  hipLaunchKernelGGL(k_primal_step, dim3(vecBlocks(v.n)), dim3(kVecThreads), 0, s, v, st);
}

namespace {
template <int EPI>
void launchSpmv(const MatView& M, SpmvArgs a, hipStream_t s) {
  if (M.useSlab && M.slab.nBlocks > 0) {
    a.S = M.slab;
    const size_t lds = (size_t)M.slab.rowsPerBlock * 8 + 2 * kSlabThreads * 8 + 2 * (kSlabThreads / kWave) * 8 +
                       2 * kSlabThreads * 2;
    hipLaunchKernelGGL((k_spmv_slab<EPI>), dim3(M.slab.nBlocks), dim3(kSlabThreads), lds, s, a);
  }
  if (M.csr.nBlocks > 0) {
    a.A = M.csr;
    if (M.csr.majorMap) hipLaunchKernelGGL((k_spmv<EPI, true>), dim3(M.csr.nBlocks), dim3(kSpmvThreads), 0, s, a);
    else hipLaunchKernelGGL((k_spmv<EPI, false>), dim3(M.csr.nBlocks), dim3(kSpmvThreads), 0, s, a);
  }
}
}  // namespace

void launchSpmvAxDual(const MatView& A, const IterVecs& v, const DevState* st, double* partDY, hipStream_t s) {
  SpmvArgs a{};
  a.st = st; a.v = v; a.part0 = partDY;
  launchSpmv<kDualStep>(A, a, s);
}
void launchSpmvAtyInteract(const MatView& At, const IterVecs& v, const DevState* st, double* partDX,
                           double* partInter, hipStream_t s) {
  SpmvArgs a{};
  a.st = st; a.v = v; a.part0 = partDX; a.part1 = partInter;
  launchSpmv<kAtyInteract>(At, a, s);
}
void launchSpmvAtyPartial(const MatView& At, const IterVecs& v, const DevState* st, double* out, hipStream_t s) {
  SpmvArgs a{};
  a.st = st; a.v = v; a.out = out;
  launchSpmv<kAtyPartial>(At, a, s);
}
void launchHalpernPrimal(const MatView& At, const HalpernVecs& h, hipStream_t s) {
  SpmvArgs a{};
  a.h = h;
  launchSpmv<kHalpernPrimal>(At, a, s);
}
void launchHalpernDual(const MatView& A, const HalpernVecs& h, hipStream_t s) {
  SpmvArgs a{};
  a.h = h;
  launchSpmv<kHalpernDual>(A, a, s);
}
void launchSpmvPlain(const MatView& A, const double* in, double* out, hipStream_t s) {
  SpmvArgs a{};
  a.st = nullptr; a.in = in; a.out = out;
  launchSpmv<kPlain>(A, a, s);
}
void launchInteract(const IterVecs& v, const DevState* st, const double* atyReduced, double* partDX,
                    double* partInter, int32_t nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(k_interact, dim3(nBlocks), dim3(kVecThreads), 0, s, v, st, atyReduced, partDX, partInter);
}
void launchReduceTo(const double* partials, int32_t count, double* out, const DevState* st, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(kVecThreads), 0, s, partials, count, out, st);
}
void launchDecide(DevState* st, const double* partDY, int32_t nDY, const double* partDX, const double* partInter,
                  int32_t nDX, const double* dyGlobal, hipStream_t s) {
  hipLaunchKernelGGL(k_decide, dim3(1), dim3(kVecThreads), 0, s, st, partDY, nDY, partDX, partInter, nDX, dyGlobal);
}
void launchFlushAverage(const IterVecs& v, DevState* st, hipStream_t s) {
  hipLaunchKernelGGL(k_flush_average, dim3(vecBlocks(v.n + v.m)), dim3(kVecThreads), 0, s, v, st);
  hipLaunchKernelGGL(k_clear_avgw, dim3(1), dim3(1), 0, s, st);
}
void launchScaleCopy(double* dst, const double* src, double a, int32_t len, hipStream_t s) {
  if (len <= 0) return;
  hipLaunchKernelGGL(k_scale_copy, dim3(vecBlocks(len)), dim3(kVecThreads), 0, s, dst, src, a, len);
}
void launchFill(double* dst, double value, int32_t len, hipStream_t s) {
  if (len <= 0) return;
  hipLaunchKernelGGL(k_fill, dim3(vecBlocks(len)), dim3(kVecThreads), 0, s, dst, value, len);
}
void launchProjectBounds(double* x, const double* lower, const double* upper, int32_t n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_project, dim3(vecBlocks(n)), dim3(kVecThreads), 0, s, x, lower, upper, n);
}
void launchMulInPlace(double* x, const double* y, int32_t len, hipStream_t s) {
  if (len <= 0) return;
  hipLaunchKernelGGL(k_mul, dim3(vecBlocks(len)), dim3(kVecThreads), 0, s, x, y, len);
}
void launchDivInPlace(double* x, const double* y, int32_t len, hipStream_t s) {
  if (len <= 0) return;
  hipLaunchKernelGGL(k_div, dim3(vecBlocks(len)), dim3(kVecThreads), 0, s, x, y, len);
}
void launchRowStats(const double* ax, const double* y, const double* rhs, const double* rowScale, int32_t m,
                    int32_t nEqs, int32_t rowOffset, int scaled, double* partials, int32_t stride, int32_t nBlocks,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_row_stats, dim3(nBlocks), dim3(kVecThreads), 0, s, ax, y, rhs, rowScale, m, nEqs, rowOffset,
                     scaled, partials, stride);
}
void launchColStats(const double* aty, const double* x, const double* cost, const double* lower,
                    const double* upper, const double* colScale, int32_t n, int scaled, double* slackPos,
                    double* slackNeg, double* partials, int32_t stride, int32_t nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(k_col_stats, dim3(nBlocks), dim3(kVecThreads), 0, s, aty, x, cost, lower, upper, colScale, n,
                     scaled, slackPos, slackNeg, partials, stride);
}
void launchFinalReduce(const double* partials, int32_t stride, int32_t nBlocks, int32_t nQ, double* out,
                       hipStream_t s) {
  hipLaunchKernelGGL(k_final_reduce, dim3(nQ), dim3(kVecThreads), 0, s, partials, stride, nBlocks, out);
}
void launchDiffNorm2(const double* a, const double* b, int32_t len, double* partials, int32_t nBlocks,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_diff_norm2, dim3(nBlocks), dim3(kVecThreads), 0, s, a, b, len, partials);
}
void launchDot(const double* a, const double* b, int32_t len, double* partials, int32_t nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(k_dot, dim3(nBlocks), dim3(kVecThreads), 0, s, a, b, len, partials);
}

}  // namespace pdlp
