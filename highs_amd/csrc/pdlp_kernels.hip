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

#include <type_traits>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstddef>

#include "pdlp_checkfn.hpp"
#include "pdlp_devfn.hpp"

namespace pdlp {

namespace {

// kAtyFused = kAtyInteract + grid barrier + decision + the next trial's primal step (slab kernel only)
// kQxInteract: N x+ for the off-diagonal part N of a QP's Hessian, with the partials of dx . N dx
enum Epilogue { kPlain = 0, kDualStep = 1, kAtyInteract = 2, kAtyPartial = 3, kHalpernPrimal = 4, kHalpernDual = 5, kAtyFused = 6,
                kQxInteract = 7 };
constexpr bool usesDevState(int epi) {
  return epi == kDualStep || epi == kAtyInteract || epi == kAtyPartial || epi == kAtyFused || epi == kQxInteract;
}
constexpr bool isInteract(int epi) { return epi == kAtyInteract || epi == kAtyFused; }

struct SpmvArgs {
  SpmvMat A;
  SlabMat S;
  LongMat L;
  int32_t xcdMap;  // 1: XCD x owns a contiguous range of work blocks (see xcdContiguousBlock)
  const DevState* st;  // nullptr for kPlain
  // kPlain / kAtyPartial
  const double* in;
  double* out;
  // iteration vectors (kDualStep / kAtyInteract / kAtyPartial)
  IterVecs v;
  double* part0;  // dY^2 (dual) | dX^2 (aty)
  double* part1;  // interaction (aty)
  HalpernVecs h;  // kHalpernPrimal / kHalpernDual
  // kAtyFused
  DevState* stOut;
  const double* partDY;
  int32_t nDY, nDX;
  unsigned long long* bar;
  unsigned long long barLimit;  // 100 MHz ticks the grid barrier may wait for a missing block
  int32_t faultTrial;           // tests: the barrier of the trial that raises the trial counter to this value expects one block too many
  int32_t inlineTasks;          // kAtyFused: the streaming blocks run the segment tasks of the long majors themselves (no extra blocks)
  int32_t touchTail;            // kAtyFused: columns beyond the register-held ones are touched before the barrier (k_spmv_slab)
  int32_t coTaskBlocks;         // kAtyFused: ... or that many extra workgroups run them, resident next to the streaming blocks, and arrive at the barrier
  CheckGate gate;  // kPlain inside a device-driven check: the launch is a no-op unless the check is due
  // development (PDLP_MI355X_SLAB_PROF=1): per block {launches, ticks to the end of the stream, to the end of the epilogue, to
  // the barrier's end, to the kernel's end} of the slab launches kDualStep / kAtyFused, 100 MHz wall clock
  unsigned long long* prof;
};

// The major-local epilogue fused into both SpMV kernels: what happens to (A v)_r once it is known.
//   kDualStep     y+ = proj(y + sigma (b - 2 A x+ + A x)), sum (dy)^2        cupdlp_step.c:43-69
//   kAtyInteract  sum (dx)^2, sum dx . d(A'y)                                cupdlp_linalg.c:772-801
//   kHalpern*     the Halpern PDHG step of the HiPDLP path                   hipdlp/pdhg.cc:961-1018
// The problem's CONSTANT vectors of the primal step (c, l, u: 24 bytes per column and iteration) are read with ordinary
// loads: next to the two matrix copies (192 MB at 1M x 1M) they stay in the 256 MB Infinity Cache, and the tail of the
// fused trial behind its stream — operand loads and stores of every column at once, bandwidth-bound — moves that much
// less through HBM (round 6: config c fused launch 55.2 -> 49.8 us, qp 40.4 -> 39.8; d, e unchanged) — where the cache has
// room to spare (IterVecs::constCached, pdlp_kernels.hpp constCached(): not at 1M x 1M / 8M nonzeros).  The
// iterates and running sums, read AND written once per iteration, stay non-temporal (xSum as ordinary traffic: no gain), and
// so do the constants of the epilogues that travel with the stream (rhs; HiPDLP's c, l, u, row bounds: 40 more bytes per
// row / column pushed the matrices out of the cache — HiPDLP 120 -> 130-135 us per iteration).
// (A compile-time choice: a run-time select between the two loads of one address is merged into ONE ordinary load by the
// compiler — the non-temporal hint is metadata — so the policy is a template parameter of the fused slab kernel.)
template <bool CACHED, class T>
__device__ __forceinline__ T ldConst(const T* p) { return CACHED ? *p : ldStream(p); }

template <int EPI>
struct Epi {
  const SpmvArgs& a;
  int cur = 0, nxt = 1;
  double sigma = 0.0, avgW = 0.0, hTau = 0.0, hRho = 1.0, hW = 0.0;
  double acc0 = 0.0, acc1 = 0.0;  // per-thread reduction partials

  __device__ __forceinline__ explicit Epi(const SpmvArgs& args) : a(args) {
    if (usesDevState(EPI)) {
      cur = a.st->cur;
      nxt = cur ^ 1;
      sigma = a.st->sigma;
      avgW = a.st->avgW;
    }
    if (EPI == kHalpernPrimal || EPI == kHalpernDual) {
      const HalpernState hs = *a.h.hs;
      hTau = hs.tau; sigma = hs.sigma; hRho = hs.rho;
      const int k = hs.hIter + a.h.kOff;
      hW = (double)k / ((double)k + 1.0);
    }
  }
  // the gathered vector
  __device__ __forceinline__ const double* input() const {
    if (EPI == kPlain) return a.in;
    if (EPI == kDualStep || EPI == kQxInteract) return a.v.x[nxt];
    if (EPI == kHalpernPrimal) return a.h.yc;
    if (EPI == kHalpernDual) return a.h.rx;
    return a.v.y[nxt];
  }
  // operands that do not depend on the SpMV result: fetched early, so their latency overlaps the stream
  __device__ __forceinline__ Pre prefetch(int r) const {
    Pre p{0.0, 0.0, 0.0, 0.0, 0.0};
    if (EPI == kDualStep) {
      p.a = ldStream(a.v.y[cur] + r); p.b = ldStream(a.v.rhs + r); p.c = ldStream(a.v.ax[cur] + r);
      if (avgW != 0.0) p.d = ldStream(a.v.ySum + r);  // (the running sum the deferred average update adds to: not a round trip behind the stream)
    } else if (isInteract(EPI)) {
      p.a = ldStream(a.v.x[cur] + r); p.b = ldStream(a.v.x[nxt] + r); p.c = ldStream(a.v.aty[cur] + r);
    } else if (EPI == kQxInteract) {
      p.a = ldStream(a.v.x[cur] + r); p.b = ldStream(a.v.x[nxt] + r); p.c = ldStream(a.v.nx[cur] + r);
    } else if (EPI == kHalpernPrimal) {
      p.a = ldStream(a.h.xc + r); p.b = ldStream(a.h.cost + r); p.c = ldStream(a.h.xa + r);
      p.d = ldStream(a.h.lower + r); p.e = ldStream(a.h.upper + r);
    } else if (EPI == kHalpernDual) {
      p.a = ldStream(a.h.yc + r); p.b = ldStream(a.h.ya + r); p.c = ldStream(a.h.rowLower + r);
      p.d = ldStream(a.h.rowUpper + r);
    }
    return p;
  }
  __device__ __forceinline__ void apply(int r, double s, const Pre& p) {
    if (EPI == kPlain || EPI == kAtyPartial) {
      a.out[r] = s;
    } else if (EPI == kHalpernPrimal) {
      halpernPrimal(a.h, r, s, p, hTau, hRho, hW);
    } else if (EPI == kHalpernDual) {
      halpernDual(a.h, r, s, p, sigma, hRho, hW);
    } else if (EPI == kQxInteract) {
      const double dx = p.a - p.b;
      const double dq = p.c - s;
      stStream(a.v.nx[nxt] + r, s);
      acc0 += dx * dq;
    } else if (EPI == kDualStep) {
      const double yv = p.a;
      if (avgW != 0.0) stStream(a.v.ySum + r, p.d + avgW * yv);  // deferred PDHG_Update_Average (step.c:438)
      double t = yv;
      t += sigma * p.b;
      t += (-2.0 * sigma) * s;
      t += sigma * p.c;
      if (r + a.v.rowOffset >= a.v.nEqs) t = t > 0.0 ? t : 0.0;
      stStream(a.v.ax[nxt] + r, s);
      a.v.y[nxt][r] = t;  // gathered by the next kernel: ordinary store
      const double d = yv - t;
      acc0 += d * d;
    } else {  // kAtyInteract, kAtyFused
      const double dx = p.a - p.b;
      const double da = p.c - s;
      stStream(a.v.aty[nxt] + r, s);
      acc0 += dx * dx;
      acc1 += dx * da;
    }
  }
  // per-block partials of the reductions (deterministic: wave shuffle tree -> fixed-order sum of the waves)
  template <int NT>
  __device__ __forceinline__ void finish(int slot, double (*scratch)[NT / 64]) {
    if (EPI == kDualStep || EPI == kQxInteract) {
      const double t = blockSum<NT>(acc0, scratch[0]);
      if (threadIdx.x == 0) a.part0[slot] = t;
    } else if (EPI == kAtyInteract) {
      const double t0 = blockSum<NT>(acc0, scratch[0]);
      const double t1 = blockSum<NT>(acc1, scratch[1]);
      if (threadIdx.x == 0) { a.part0[slot] = t0; a.part1[slot] = t1; }
    } else if (EPI == kAtyFused) {  // read by every other block after the grid barrier: write-through stores
      const double t0 = blockSum<NT>(acc0, scratch[0]);
      const double t1 = blockSum<NT>(acc1, scratch[1]);
      if (threadIdx.x == 0) { stAgent(a.part0 + slot, t0); stAgent(a.part1 + slot, t1); }
    }
  }
};

// The segment tasks of the long majors (pdlp_kernels.hpp LongMat): workgroup lb of the extra blocks runs the tasks
// [lb*W, (lb+1)*W), one per wave.  Lane l adds the products of the entries l, l+64, ... of the segment in ascending
// order (8 unit-stride loads of idx/val per lane and pass, all issued before the 8 dependent gathers), 64-lane
// shuffle tree.  A major whose segments all sit in this workgroup is finished through LDS; a spanning one through
// HBM: segment sum stored, ticket taken, the wave with the last ticket adds the segment sums left to right.
// Cross-workgroup visibility: every shared word (segSum, ticket) is only touched with agent-scope relaxed atomics
// (sc1: write-through stores, L1-bypassing loads), and the segment sum has landed (s_waitcnt vmcnt(0)) before the
// ticket is taken.  The epilogue operands of the major are fetched with the first loads, not after the reduction.
template <int EPI, int W>
__device__ __forceinline__ void longBlock(const SpmvArgs& a, Epi<EPI>& epi, int lb, double* lds /* [W] */, bool many = false) {
  const LongMat& L = a.L;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
  // task group lb * (W / g) + wave / g, g = L.taskGroup tasks each: a workgroup of W waves takes W / g consecutive groups
  // when it runs tasks behind its stream (`many`), ONE group — waves beyond g idle — as an extra workgroup of a launch
  // (more, lighter task workgroups: every CU gets one next to its streaming block)
  const int g = L.taskGroup;
  const int sub = wave / g;
  const int per = W / g;  // whole groups per pass of W waves (`many`)
  const int grp = many ? lb * per + sub : lb;
  const bool mine = many ? sub < per : sub == 0;
  const int t = mine ? grp * g + (wave - sub * g) : L.nTasks;
  LongTask T;
  T.c = -1; T.pBeg = T.pEnd = 0; T.first = t; T.nSeg = 1; T.major = 0; T.contained = 1; T.seg = 0;
  if (t < L.nTasks) {  // (one 32-byte scalar load)
    const int32_t* q = reinterpret_cast<const int32_t*>(L.tasks + t);
    T.pBeg = ldUniform(q); T.pEnd = ldUniform(q + 1); T.c = ldUniform(q + 2); T.first = ldUniform(q + 3);
    T.nSeg = ldUniform(q + 4); T.major = ldUniform(q + 5); T.contained = ldUniform(q + 6); T.seg = ldUniform(q + 7);
  }
  const bool active = T.c >= 0;
  const int seg = T.seg;
  Pre pre{0.0, 0.0, 0.0, 0.0, 0.0};
  if (active && (seg == 0 || !T.contained)) pre = epi.prefetch(T.major);
  const int32_t* __restrict__ idx = L.idx;
  const double* __restrict__ val = L.val;
  const double* __restrict__ in = epi.input();
  constexpr int kPer = kLongSegment / kWave;
  double s = 0.0;
  for (int base = T.pBeg; base < T.pEnd; base += kLongSegment) {
    int32_t ci[kPer];
    double va[kPer], xg[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {  // unconditional, clamped (a load in an exec-masked branch drains vmcnt)
      const int q = base + k * kWave + lane;
      const int qq = q < T.pEnd ? q : T.pEnd - 1;
      ci[k] = idx[qq];
      va[k] = val[qq];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) xg[k] = in[ci[k]];
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the segment sum has landed before the ticket is taken
      const unsigned old = __hip_atomic_fetch_add(L.ticket + T.c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = old == (unsigned)(T.nSeg - 1) ? 1 : 0;
    }
  }
  last = __builtin_amdgcn_readfirstlane(last);
  __syncthreads();
  double total = 0.0;
  bool finish = false;
  if (active && T.contained && seg == 0) {
    for (int k = 0; k < T.nSeg; ++k) total += lds[wave + k];  // left to right
    finish = true;
  } else if (last) {  // last ticket of a spanning major: every segment sum is in HBM
    if (lane == 0) __hip_atomic_store(L.ticket + T.c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
    double v = 0.0;
    if (lane < T.nSeg)
      v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(L.segSum + T.first + lane),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int k = 0; k < T.nSeg; ++k) total += __shfl(v, k, kWave);  // left to right
    finish = true;
  }
  if (finish && lane == 0) {
    const double keep0 = epi.acc0, keep1 = epi.acc1;
    epi.acc0 = 0.0; epi.acc1 = 0.0;
    epi.apply(T.major, total, pre);
    // kAtyFused: the block that owns this column takes (A'y+)_c from memory for the next primal step, and every block
    // reads the major's contributions behind the grid barrier: write-through (agent-scope) stores
    if (EPI == kAtyFused) stAgent(a.v.aty[epi.nxt] + T.major, total);
    if (EPI == kDualStep || EPI == kQxInteract || isInteract(EPI)) {  // the major's own slot (or, beyond kLongSlotCap, its entry for k_long_groups)
      double* o0 = L.contrib ? L.contrib + T.c : a.part0 + L.slotBase + T.c;
      if (EPI == kAtyFused) stAgent(o0, epi.acc0); else *o0 = epi.acc0;
      if (isInteract(EPI)) {
        double* o1 = L.contrib ? L.contrib + L.nLong + T.c : a.part1 + L.slotBase + T.c;
        if (EPI == kAtyFused) stAgent(o1, epi.acc1); else *o1 = epi.acc1;
      }
    }
    epi.acc0 = keep0; epi.acc1 = keep1;
  }
}

// The grid barrier of a fused trial did not hold (pdlp_devfn.hpp gridBarrier: a block of the launch was not resident in
// time — the device is shared — or, kBarBroken, an earlier barrier was split).  No block takes the decision or does the
// primal step: the iterate buffers of the CURRENT parity and the state record are as they were before the trial, apart
// from the pending average weight of y, which the A x+ launch of this trial has already added to ySum (the host clears
// it).  Block 0 hands the record on with commError = 3 (fall back to plain launches) or 1 (error) and halted = 1, so
// that whatever is queued behind is a no-op.  words: the state record, in LDS.
__device__ __forceinline__ void fusedBarrierFailed(DevState* stOut, const uint32_t* words, int verdict, bool writer, int tid) {
  if (!writer || tid >= (int)(sizeof(DevState) / 4)) return;
  uint32_t w = words[tid];
  if (tid == (int)(offsetof(DevState, commError) / 4)) w = verdict == kBarFailed ? 3u : 1u;
  if (tid == (int)(offsetof(DevState, halted) / 4)) w = 1u;
  reinterpret_cast<uint32_t*>(stOut)[tid] = w;
}

// More long majors than kLongSlotCap: slot g of the partial arrays = contributions of the majors [g*G, (g+1)*G),
// added left to right.
__global__ __launch_bounds__(kVecThreads) void k_long_groups(const LongMat L, const DevState* st, double* part0, double* part1) {
  if (st && st->halted) return;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= L.nSlots) return;
  const int c0 = g * L.groupSize, c1 = c0 + L.groupSize < L.nLong ? c0 + L.groupSize : L.nLong;
  double s0 = 0.0, s1 = 0.0;
  for (int c = c0; c < c1; ++c) { s0 += L.contrib[c]; if (part1) s1 += L.contrib[L.nLong + c]; }
  part0[L.slotBase + g] = s0;
  if (part1) part1[L.slotBase + g] = s1;
}

// CSR-adaptive SpMV with the fused epilogue (long majors: segment tasks in the extra blocks at the end of the grid).
// One work block = up to kChunk consecutive nonzeros belonging to whole majors.
//   phase 1: all lanes stream val[]/idx[] with unit stride (coalesced), gather
//            the input vector, and park the products in LDS;
//   phase 2: one lane per major adds its products left to right (the
//            reference's summation order) and runs the epilogue.
template <int EPI, int CHUNK>
__global__ __launch_bounds__(kSpmvThreads) void k_spmv(const SpmvArgs a) {
  if (EPI == kAtyFused && a.st->halted) {  // keep the two state slots identical while the queue drains
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevState) / 4)
      reinterpret_cast<uint32_t*>(a.stOut)[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.st)[threadIdx.x];
    return;
  }
  if (usesDevState(EPI) && a.st->halted) return;
  if ((EPI == kHalpernPrimal || EPI == kHalpernDual) && a.h.hs->halted) return;  // (HiPDLP's device-driven loop has ended)
  if (EPI == kPlain && !gateOpen(a.gate)) return;
  __shared__ double prod[CHUNK + CHUNK / 8 + 8];
  __shared__ double scratch[2][kSpmvThreads / kWave];
  // kAtyFused (the 2-launch trial on the stream layout): reduction scratch of the decision and the state record
  __shared__ double tscr[EPI == kAtyFused ? 4 : 1][kVecThreads / kWave];
  __shared__ uint32_t shWords[EPI == kAtyFused ? (sizeof(DevState) + 3) / 4 : 1];
  __shared__ int barVerdict;

  const int tid = threadIdx.x;
  Epi<EPI> epi(a);
  if ((int)blockIdx.x >= a.A.nBlocks) {  // the extra blocks: one segment task of a long major per wave
    longBlock<EPI, kSpmvThreads / kWave>(a, epi, (int)blockIdx.x - a.A.nBlocks, scratch[0]);
    return;
  }
  const int blk = a.xcdMap ? xcdContiguousBlock(blockIdx.x, a.A.nBlocks) : (int)blockIdx.x;
  const int4 bb = *reinterpret_cast<const int4*>(a.A.blockBeg + 4 * blk);  // (first major, end major, first entry, end entry)
  const int r0 = bb.x, r1 = bb.y, p0 = bb.z, p1 = bb.w;
  padSlots(prod, CHUNK + CHUNK / 8 + 8, tid, kSpmvThreads);  // the pad slots of the strip: -0.0 (majorSum adds them)
  const int32_t* __restrict__ idx = a.A.idx;
  const double* __restrict__ val = a.A.val;
  const double* __restrict__ in = epi.input();
  auto vecIndex = [&](int r) { return r; };

  {
    constexpr int kPer = CHUNK / kSpmvThreads;
    const int cnt = p1 - p0;
    // Bookkeeping of this lane's first major, issued ahead of the stream.  All
    // loads below are unconditional with clamped indices: a load inside an
    // exec-masked branch makes hipcc drain vmcnt at the join, which serialises
    // the stream (one idx/val pair in flight instead of 2*kPer).
    const int rFirst = r0 + tid;
    const int rr = rFirst < r1 ? rFirst : r1 - 1;
    int qb = a.A.beg[rr] - p0;
    int qe = a.A.beg[rr + 1] - p0;
    Pre pre = epi.prefetch(vecIndex(rr));
    // kAtyFused: what the NEXT primal step needs of this lane's first major and no decision can change (c, l, u)
    Pre fix{0.0, 0.0, 0.0, 0.0, 0.0};
    double keepX = 0.0, keepS = 0.0;  // x+ and (A'y+) of that major, for the step after an accepted trial
    if (EPI == kAtyFused) { fix.a = ldStream(a.v.cost + rr); fix.b = ldStream(a.v.lower + rr); fix.c = ldStream(a.v.upper + rr); }
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
        pre = epi.prefetch(vecIndex(r));
      }
      const double s = majorSum(prod, qb, qe);
      epi.apply(vecIndex(r), s, pre);
      if (EPI == kAtyFused && r == rFirst) { keepX = pre.b; keepS = s; }
    }
    if (EPI == kAtyFused) fix.d = ldStream(a.v.xSum + rr);  // in flight across the barrier and the decision
    epi.template finish<kSpmvThreads>(a.A.partOffset + blk, scratch);
    if (EPI == kAtyFused) {
      // ---- every block's partials in HBM -> decision (identical in every block) -> the next trial's primal step on the
      // block's majors (pdlp_kernels.hip k_spmv_slab has the same tail) ----
      DevState* sh = reinterpret_cast<DevState*>(shWords);
      if (tid < kWave) {
        const int nExp = a.A.nBlocks + (a.st->nTrials + 1 == a.faultTrial ? 1 : 0);
        const int verdict = gridBarrier(a.bar, (int)blockIdx.x, nExp, (unsigned long long)a.st->nTrials + 1ull, tid, a.barLimit);
        if (tid == 0) barVerdict = verdict;
      }
      if (tid >= kWave && tid - kWave < (int)(sizeof(DevState) / 4)) shWords[tid - kWave] = reinterpret_cast<const uint32_t*>(a.st)[tid - kWave];
      __syncthreads();
      if (barVerdict != kBarOk) {  // not every block of this launch was resident in time: the trial stays undecided (fusedBarrierFailed)
        fusedBarrierFailed(a.stOut, shWords, barVerdict, blockIdx.x == 0, tid);
        return;
      }
      const unsigned long long timedOut = tid == 0 ? __hip_atomic_load(a.bar + a.A.nBlocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      double dY2, dX2, inter;
      trialSumsT<1>(a.partDY, a.nDY, a.part0, a.part1, a.nDX, tscr, dY2, dX2, inter);
      if (tid == 0) {
        decideUpdate<true>(sh, dX2, dY2, inter);
        if (timedOut) sh->commError = 1;
      }
      __syncthreads();
      const int halted = sh->halted, curN = sh->cur, accepted = sh->lastAccepted;
      const double tau = sh->tau, avgWx = sh->avgWx;
      if (blockIdx.x == 0 && tid < (int)(sizeof(DevState) / 4)) {  // pending = 0 (decideCore); avgWx: added to xSum below
        uint32_t w = shWords[tid];
        constexpr int kAvgWxWord = offsetof(DevState, avgWx) / 4;
        if (!halted && (tid == kAvgWxWord || tid == kAvgWxWord + 1)) w = 0u;
        reinterpret_cast<uint32_t*>(a.stOut)[tid] = w;
      }
      if (halted) return;
      const double* __restrict__ xBase = a.v.x[curN];
      const double* __restrict__ atyBase = a.v.aty[curN];
      double* __restrict__ xOut = a.v.x[curN ^ 1];
      for (int r = rFirst; r < r1; r += kSpmvThreads) {
        double xb, ab, c, l, u, xs;
        if (r == rFirst) {  // from registers (a rejected trial, 3 %, fetches x and A'y again)
          xb = accepted ? keepX : ldStream(xBase + r); ab = accepted ? keepS : ldStream(atyBase + r);
          c = fix.a; l = fix.b; u = fix.c; xs = fix.d;
        } else {            // (more than 256 majors in the block: short columns)
          xb = ldStream(xBase + r); ab = ldStream(atyBase + r);
          c = ldStream(a.v.cost + r); l = ldStream(a.v.lower + r); u = ldStream(a.v.upper + r); xs = ldStream(a.v.xSum + r);
        }
        if (avgWx != 0.0) stStream(a.v.xSum + r, xs + avgWx * xb);  // deferred PDHG_Update_Average (step.c:437)
        double t = xb;
        t += (-tau) * c;
        t += tau * ab;
        if (a.v.qdiag) t = t / (1.0 + tau * ldStream(a.v.qdiag + r));
        t = t < u ? t : u;
        t = t > l ? t : l;
        xOut[r] = t;  // gathered by the A x+ kernel: ordinary store
      }
      return;
    }
    return;
  }
}

// Slab SpMV (layout: pdlp_host.hpp SlabLayout) — for operands whose gathered vector does not fit an
// XCD's 4 MB L2.  One 1024-thread block per CU; each of its 16 WAVES owns a run of consecutive majors
// (blocks and waves are cut by work, pdlp_host.hpp slabPartition) and streams its own nonzeros — one dense list sorted by (slab of the gathered vector, local
// major, minor) — 64 at a time through a register pipeline: entry/value loads kSlabSlots groups ahead of
// the accumulation, the gather one group ahead, all counted on vmcnt by the compiler.  A workgroup
// barrier per group keeps the CU's waves on the same slab (pacing only: a wave touches nothing but its
// own accumulators), and since every CU has the same amount of work per slab, all CUs of an XCD sweep
// the gathered vector together: measured L2 misses = the compulsory ones (TCC_MISS 1.33 M per launch at
// the bench size, of which 0.5 M are the fill of x into 8 L2s and 0.75 M the matrix stream); free-running
// waves (no barrier, 8 blocks/CU) drift apart — the SIMDs issue oldest-first once the vector-memory
// queue is full — and miss 3.9 M times.
//   What bounds the kernel: every gathered double is its own L2 request.  Round-3 counters at the bench size
// (profiles/r03_development_measurements.md §13): TCC_REQ 8.98 M per launch over the 128 L2 channels = 0.67
// requests per channel per clock for the whole launch, mean L2 latency 331 cycles, VALU busy 25 %, waves at
// vector-memory wait counters half of their life; a deeper gather pipeline or 32 instead of 16 waves per CU
// change nothing.  The request rate of the L2, not latency, issue slots or bytes (0.35 of the HBM peak).
//   Accumulation: the 64 products go to a wave-private LDS strip; the first lane of each run of equal
// local majors adds the run, left to right, onto the major's LDS accumulator.  A 64-entry group may
// straddle slabs, so the same major can own two runs in it: those are in different ASCENDING stretches
// of the group (a slab boundary that matters shows up as a descent of the local major), and the
// stretches are applied one after the other.  Slabs ascend and minors ascend inside a slab, so every
// major is summed in ascending minor order — the reference's order: bit-identical to the CSR path.
constexpr int kSlabSlots = 3;  // register pipeline depth (groups of 64 entries per wave)
// TWO: register budget for two resident blocks per CU (8 waves per SIMD) — the extra blocks with the segment
// tasks of the long majors then run NEXT to the streaming blocks instead of after them.
// LATE (kAtyFused): the operands of the next primal step that no decision can change (c, l, u, xSum, q) are fetched behind
// the block's ARRIVAL at the grid barrier instead of travelling with the stream, and for TWICE as many columns per thread
// (the stream's registers are free by then): 8192 columns per block (4096 in the 64-register variant) are stepped from
// registers behind the barrier, nothing but the stores left there (round 6: config c spent 13 us behind its barrier on
// 8200 columns per block, half of them fetched there).
#ifndef PDLP_TWO_EXTRA
#define PDLP_TWO_EXTRA 0
#endif
template <int EPI, bool TWO, int NB, int GD, bool LATE = false, bool CC = false, bool ULO = false>
__global__ __launch_bounds__(kSlabThreads, TWO ? 2 * kSlabThreads / 256 : kSlabThreads / 256) void k_spmv_slab(const SpmvArgs a) {
  if (EPI == kAtyFused && a.st->halted) {  // keep the two state slots identical while the queue drains
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevState) / 4)
      reinterpret_cast<uint32_t*>(a.stOut)[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.st)[threadIdx.x];
    return;
  }
  if (usesDevState(EPI) && a.st->halted) return;
  if ((EPI == kHalpernPrimal || EPI == kHalpernDual) && a.h.hs->halted) return;  // (HiPDLP's device-driven loop has ended)
  if (EPI == kPlain && !gateOpen(a.gate)) return;
  static_assert(GD >= 1 && NB >= GD + 2, "entry loads need two steps, gathers GD steps");
  const unsigned long long tProf0 = a.prof ? wall_clock64() : 0ull;
  int profBlk = (int)blockIdx.x;  // (the LOGICAL block once it is known: the table is indexed like the partition)
  auto profStamp = [&](int k) {
    if (a.prof && threadIdx.x == 0 && (int)blockIdx.x < a.S.nBlocks) {
      unsigned long long* q = a.prof + ((EPI == kAtyFused ? 1024 : 0) + profBlk) * 8;
      if (k == 0) q[0] += 1;
      q[1 + k] += wall_clock64() - tProf0;
    }
  };
  constexpr int kWaves = kSlabThreads / kWave;
  constexpr int kSlabPre = TWO ? 2 : 4;  // majors per thread whose epilogue operands are fetched before the stream
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x >= a.S.nBlocks) {  // the extra blocks: one segment task of a long major per wave
    Epi<EPI> epiL(a);
    longBlock<EPI, kWaves>(a, epiL, (int)blockIdx.x - a.S.nBlocks, reinterpret_cast<double*>(smem));
    if (a.prof && threadIdx.x == 0 && (int)blockIdx.x - a.S.nBlocks < 512) {  // (development: when the task workgroups are done, rows 512.. of the table)
      unsigned long long* q = a.prof + ((EPI == kAtyFused ? 1024 : 0) + 512 + (int)blockIdx.x - a.S.nBlocks) * 8;
      q[0] += 1; q[1] += wall_clock64() - tProf0;
    }
    if (EPI == kAtyFused) {
      // a task workgroup of the fused launch: what it published (A'y+ of its long columns, their contributions) has landed;
      // it arrives at the grid barrier the streaming blocks wait at — and leaves (it needs nothing from behind the barrier)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_store(a.bar + blockIdx.x, (unsigned long long)a.st->nTrials + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // dynamic LDS: acc[R] f64 | stg[16][64] f64 | scratch[2][16] f64 | (kAtyFused) trial scratch[4][4] f64, DevState
  const int R = (a.S.rowsPerBlock + 1) & ~1;  // (the most majors any block owns; even: the strips behind stay 16-byte aligned)
  double* acc = reinterpret_cast<double*>(smem);
  double* stgAll = acc + R;
  double(*scratch)[kWaves] = reinterpret_cast<double(*)[kWaves]>(stgAll + kSlabThreads);

  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int blk = a.xcdMap ? xcdContiguousBlock(blockIdx.x, a.S.nBlocks) : (int)blockIdx.x;
  profBlk = blk;
  const int mb = a.S.minorBits;
  const uint32_t mmask = (1u << mb) - 1u;
  const int gw = blk * kWaves + wave;
  const int rBase = ldUniform(a.S.waveBeg + blk * kWaves), rEnd = ldUniform(a.S.waveBeg + blk * kWaves + kWaves);  // never empty
  const int wBeg = ldUniform(a.S.waveBeg + gw);
  const int Rw = ldUniform(a.S.waveBeg + gw + 1) - wBeg;
  const int e0 = ldUniform(a.S.wavePtr + gw), e1 = ldUniform(a.S.wavePtr + gw + 1);
  double* wacc = acc + (wBeg - rBase);
  double* stg = stgAll + wave * kWave;
  Epi<EPI> epi(a);
  const double* __restrict__ in = epi.input();

  for (int r = lane; r < Rw; r += kWave) wacc[r] = 0.0;
  // operands of this thread's first kSlabPre majors (all of them at the bench size: 3920 majors per block),
  // fetched ahead of the stream (clamped, unconditional) so that nothing is loaded in the kernel's tail
  Pre pre[kSlabPre];
#pragma unroll
  for (int k = 0; k < kSlabPre; ++k) {
    const int r = rBase + tid + k * kSlabThreads;
    pre[k] = epi.prefetch(r < rEnd ? r : rEnd - 1);
  }
  // kAtyFused: the operands of the NEXT primal step that no decision can change (c, l, u, xSum) travel with the stream
  // (not in the 64-register variant that leaves room for the task workgroups: there they are fetched behind the barrier)
  constexpr bool kFixEarly = !TWO && !LATE;
  // columns per thread stepped from registers: the kSlabPre whose epilogue operands travel with the stream, plus kExtra whose
  // operands are fetched behind the arrival (LATE: as many again; the 64-register variant: PDLP_TWO_EXTRA, measured)
  constexpr int kExtra = EPI != kAtyFused ? 0 : LATE ? kSlabPre : TWO ? PDLP_TWO_EXTRA : 0;
  constexpr int kFixN = EPI == kAtyFused ? kSlabPre + kExtra : 1;
  Pre fix[kFixN];
  double xbLate[kExtra > 0 ? kExtra : 1];  // x+ of the extra columns (the first ones': pre[k].b)
  // bounds that ALL columns of this block share (IterVecs::colBlockUni): taken from two scalars instead of 8 / 16 bytes per column
  const int uniB = (EPI == kAtyFused && a.v.colBlockUni) ? ldUniform(a.v.colBlockUni + blk) : 0;
  const double lo0 = uniB ? ldUniform(a.v.colBlockBounds + 2 * blk) : 0.0, up0 = uniB ? ldUniform(a.v.colBlockBounds + 2 * blk + 1) : 0.0;  // (scalar registers)
  if (EPI == kAtyFused && kFixEarly) {
#pragma unroll
    for (int k = 0; k < kSlabPre; ++k) {
      const int r0_ = rBase + tid + k * kSlabThreads;
      const int r = r0_ < rEnd ? r0_ : rEnd - 1;
      fix[k].a = ldStream(a.v.cost + r); fix[k].b = ldStream(a.v.lower + r); fix[k].c = ldStream(a.v.upper + r);
      fix[k].d = 0.0; fix[k].e = a.v.qdiag ? ldStream(a.v.qdiag + r) : 0.0;  // (QP: the diagonal of Q of the prox step)
    }
  }

  // Consume one 64-entry group of this wave: products to the wave's LDS strip, the first lane of each run of equal
  // local majors adds the run, left to right, onto the major's accumulator (ascending stretches one after the other
  // when the group straddles slabs).
  auto consumeGroup = [&](uint32_t e, double prod, int nValid) {
    const bool valid = lane < nValid;
    const uint32_t lrow = valid ? (e >> mb) : 0xffffffffu;
    stg[lane] = prod;
    const uint32_t prev = (uint32_t)__shfl_up((int)lrow, 1, kWave);
    const bool isHead = valid && (lane == 0 || lrow != prev);
    const bool desc = valid && lane != 0 && lrow < prev;
    const uint64_t heads = __ballot(isHead);
    const uint64_t descs = __ballot(desc);
    // end of this lane's run = next head above it (or the end of the group)
    const uint64_t above = (heads >> lane) >> 1;
    const int vEnd = nValid < kWave ? nValid : kWave;
    const int end = above ? lane + __ffsll((unsigned long long)above) : vEnd;
    __builtin_amdgcn_wave_barrier();
    auto addRun = [&]() {
      double s = wacc[lrow];
      s += prod;
      int j = lane + 1;
      for (; j + 4 <= end; j += 4) {  // long runs (clustered columns): four LDS reads in flight per step
        const double t0 = stg[j], t1 = stg[j + 1], t2 = stg[j + 2], t3 = stg[j + 3];
        s += t0; s += t1; s += t2; s += t3;
      }
      for (; j < end; ++j) s += stg[j];
      wacc[lrow] = s;
    };
    if (descs == 0) {
      if (isHead) addRun();
    } else {
      // the group straddles slabs: ascending stretches one after the other (a major may own a run in each)
      const int seg = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(descs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)descs, 0u)) + (desc ? 1 : 0);
      const int nSeg = __popcll(descs) + 1;
      for (int sg = 0; sg < nSeg; ++sg) {
        if (isHead && seg == sg) addRun();
        __builtin_amdgcn_wave_barrier();
      }
    }
  };
  // ---- the stream ----
  // Register pipeline over NB slots, unrolled NB times so that a slot is a fixed register (no moves of
  // values still in flight, which would drain vmcnt): step g gathers for group g+1, then consumes
  // group g (slot g % NB) and refills that slot with group g+NB.  All loads are unconditional; past the
  // wave's last entry they re-read that entry (same cache line), and groups past nG contribute nothing.
  const uint32_t* __restrict__ ent = a.S.ent + e0;
  const double* __restrict__ val = a.S.val + e0;
  const int cnt = e1 - e0;
  const int nG = (cnt + kWave - 1) / kWave;
  const int last = cnt > 0 ? cnt - 1 : 0;  // (an empty wave reads entry e0, which exists: ent/val carry one pad element)
  auto entryIndex = [&](int g) { const int q = g * kWave + lane; return q < last ? q : last; };
  auto gather = [&](uint32_t e) -> double {
    const uint32_t off = (e & mmask) << 3;  // byte offset: minor < 2^26
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(in) + off);
  };
  uint32_t E[NB];
  double V[NB], X[NB];
  // prologue = the steps -NB..-1 of the same schedule (same issue order as the steady state, so the
  // compiler's vmcnt bookkeeping at the loop header does not have to assume the worst)
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if (k + GD >= NB) X[(k + GD) % NB] = gather(E[(k + GD) % NB]);
    const int q = entryIndex(k);
    E[k] = ent[q];
    V[k] = val[q];
    __builtin_amdgcn_sched_barrier(0);  // keep this issue order
  }
  // block-uniform trip count: the per-step barrier must be reached by every wave
  int nRounds = (nG + NB - 1) / NB;
  {
    int* share = reinterpret_cast<int*>(scratch);
    if (lane == 0) share[wave] = nRounds;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kWaves; ++w) nRounds = share[w] > nRounds ? share[w] : nRounds;
    __syncthreads();
  }
  for (int o = 0; o < nRounds; ++o) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int g = o * NB + u;
      X[(u + GD) % NB] = gather(E[(u + GD) % NB]);  // group g+GD
      // consume group g
      const int nValid = cnt - g * kWave;  // lanes >= nValid hold nothing of this wave (<= 0: phantom group)
      const uint32_t eCur = E[u];
      const double prod = V[u] * X[u];
      {  // slot u is free: refill it with group g+NB
        const int q = entryIndex(g + NB);
        E[u] = ent[q];
        V[u] = val[q];
      }
      __builtin_amdgcn_sched_barrier(0);  // gather, then refill, then the LDS work: in this order
      consumeGroup(eCur, prod, nValid);
      if (!a.S.noPace) __syncthreads();  // pacing: the CU's waves stay on the same slab
    }
  }
  if (a.S.noPace) __syncthreads();  // (free-running waves: every wave's accumulators are final before the epilogue reads them)

  profStamp(0);
  const uint32_t* __restrict__ mask = a.S.longMask;  // bit r: major r is a long one (its segment tasks own it)
  auto longMajor = [&](int r) { return ((mask[r >> 5] >> (r & 31)) & 1u) != 0u; };
#pragma unroll
  for (int k = 0; k < kSlabPre; ++k) {
    const int lr = tid + k * kSlabThreads;
    if (rBase + lr < rEnd && !longMajor(rBase + lr)) epi.apply(rBase + lr, acc[lr], pre[k]);
  }
  // (more than kSlabPre majors per thread — 2.1 M columns in 256 blocks: the operands of the next kSlabPre majors are
  // fetched together, one memory round trip per batch instead of one per major; same majors in the same order)
  for (int lr0 = tid + kSlabPre * kSlabThreads; rBase + lr0 < rEnd; lr0 += kSlabPre * kSlabThreads) {
    Pre more[kSlabPre];
#pragma unroll
    for (int k = 0; k < kSlabPre; ++k) {
      const int r = rBase + lr0 + k * kSlabThreads;
      more[k] = epi.prefetch(r < rEnd ? r : rEnd - 1);
    }
#pragma unroll
    for (int k = 0; k < kSlabPre; ++k) {
      const int lr = lr0 + k * kSlabThreads;
      if (rBase + lr < rEnd && !longMajor(rBase + lr)) epi.apply(rBase + lr, acc[lr], more[k]);
    }
  }
  epi.template finish<kSlabThreads>(blk, scratch);
  profStamp(1);
  if (EPI == kAtyFused && !TWO && a.inlineTasks) {  // (the 64-register variant carries task workgroups instead)
    // Long columns in the fused trial: their segment tasks cannot be extra workgroups (those would have to be resident
    // next to the waiting blocks), so the streaming blocks take them — task group tb goes to block tb % nBlocks, one
    // task per wave, same lanes and sums as in the extra blocks of the other launches (longBlock).
    const int perPass = kWaves / a.L.taskGroup;  // whole task groups per pass of the block's waves
    const int nTB = ((a.L.nTasks + a.L.taskGroup - 1) / a.L.taskGroup + perPass - 1) / perPass;
    for (int tb = (int)blockIdx.x; tb < nTB; tb += a.S.nBlocks) {
      longBlock<EPI, kWaves>(a, epi, tb, &scratch[0][0], true);
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's published words have landed before the block arrives
    __syncthreads();
  }
  // ---- every block's partials in HBM -> decision (identical in every block) -> the next trial's primal step ----
  // The block ARRIVES at the grid barrier before it prefetches anything for the phase behind it: the arrival waits for this
  // wave's memory operations (vmcnt(0)), and with the operand loads of the next primal step issued in front of it that wait
  // was an HBM round trip under load — every block's, so the release came 2.7 us (config b) to 5 us (config d, ten loads
  // per thread) after the last epilogue (round 6).
  if (EPI == kAtyFused && wave == 0) {
    gridArrive(a.bar, (int)blockIdx.x, (unsigned long long)a.st->nTrials + 1ull, lane);
    profStamp(4);  // (the block's published words have landed, its arrival word is on its way)
  }
  if (EPI == kAtyFused && kFixEarly) {  // xSum of the own columns: in flight across the barrier and the decision
#pragma unroll
    for (int k = 0; k < kSlabPre; ++k) {
      const int r0_ = rBase + tid + k * kSlabThreads;
      fix[k].d = ldStream(a.v.xSum + (r0_ < rEnd ? r0_ : rEnd - 1));
    }
  }
  if (EPI == kAtyFused && !kFixEarly) {
    // (the stream's pipeline registers are free now — the operands of the next primal step are fetched here, in flight
    // across the grid barrier and the decision instead of behind them)
#pragma unroll
    for (int k = 0; k < kFixN; ++k) {
      const int r0_ = rBase + tid + k * kSlabThreads;
      const int r = r0_ < rEnd ? r0_ : rEnd - 1;
      fix[k].a = ldConst<CC>(a.v.cost + r);
      // (a bound that all columns of the block share: no load; ULO — ALL columns of the operand share the lower bound, the
      // rule in LPs: no register either, the step takes the scalar)
      if (!ULO) fix[k].b = (uniB & 1) ? lo0 : ldConst<CC>(a.v.lower + r);
      fix[k].c = (uniB & 2) ? up0 : ldConst<CC>(a.v.upper + r);
      fix[k].d = ldStream(a.v.xSum + r);
      // (the diagonal of Q of a QP's prox step: in a register for the first kSlabPre columns, fetched behind the barrier for the others)
      fix[k].e = (k < kSlabPre && a.v.qdiag) ? ldStream(a.v.qdiag + r) : 0.0;
      if (kExtra > 0 && k >= kSlabPre) xbLate[k - kSlabPre] = ldStream(a.v.x[epi.nxt] + r);  // (x+ of the trial: what the step starts from when it is accepted)
    }
  }
  // (round 6) Columns beyond the ones stepped from registers — the blocks that own thousands of one-entry slack columns —
  // used to fetch their operands behind the barrier, kTail columns per HBM round trip under load: those blocks ended
  // 4.5 us (config d) / 7 us (f) after the others.  Their lines are TOUCHED here instead (one dword per lane and operand,
  // result unused, all into one register), so that the loads behind the barrier hit the L2.
  uint32_t touched = 0u;
  if (EPI == kAtyFused && TWO && a.touchTail) {  // (the 128-register variant steps 8192 columns per block from registers: nothing to gain there, config c lost 2 us)
    const double* __restrict__ xNxt = a.v.x[epi.nxt];
    int n = 0;
    for (int lr0 = tid + kFixN * kSlabThreads; rBase + lr0 < rEnd && n < 8; lr0 += kSlabThreads, ++n) {
      const int r = rBase + lr0;
      asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(a.v.cost + r) : "memory");
      if (!(uniB & 1)) asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(a.v.lower + r) : "memory");
      if (!(uniB & 2)) asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(a.v.upper + r) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(a.v.xSum + r) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(xNxt + r) : "memory");
    }
  }
  if (EPI == kAtyFused) {
    double(*tscr)[kVecThreads / kWave] = reinterpret_cast<double(*)[kVecThreads / kWave]>(&scratch[2][0]);
    DevState* sh = reinterpret_cast<DevState*>(&tscr[4][0]);
    int* barVerdict = reinterpret_cast<int*>(reinterpret_cast<char*>(sh) + ((sizeof(DevState) + 7) / 8) * 8);
    if (wave == 0) {
      const int nExp = a.S.nBlocks + a.coTaskBlocks + (a.st->nTrials + 1 == a.faultTrial ? 1 : 0);
      const int verdict = gridWait(a.bar, (int)blockIdx.x, nExp, (unsigned long long)a.st->nTrials + 1ull, lane, a.barLimit);
      if (lane == 0) *barVerdict = verdict;
    }
    {  // the state record -> LDS, one word per thread (no register copy of the 50-word record)
      const uint32_t* src = reinterpret_cast<const uint32_t*>(a.st);
      uint32_t* dstw = reinterpret_cast<uint32_t*>(sh);
      if (tid >= kVecThreads && tid - kVecThreads < (int)(sizeof(DevState) / 4)) dstw[tid - kVecThreads] = src[tid - kVecThreads];
    }
    __syncthreads();
    profStamp(2);
    if (TWO && a.touchTail) asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched) : : "memory");  // (the register of the touches is free again)
    if (*barVerdict != kBarOk) {  // not every block of this launch was resident in time: the trial stays undecided (fusedBarrierFailed)
      fusedBarrierFailed(a.stOut, reinterpret_cast<const uint32_t*>(sh), *barVerdict, blockIdx.x == 0, tid);
      return;
    }
    // (the barrier's timeout flag: fetched together with the partials, looked at behind the decision — not a round trip of its own)
    const unsigned long long timedOut = tid == 0 ? __hip_atomic_load(a.bar + a.S.nBlocks + a.coTaskBlocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    double dY2, dX2, inter;
    trialSumsT<1>(a.partDY, a.nDY, a.part0, a.part1, a.nDX, tscr, dY2, dX2, inter);
    if (tid == 0) {
      decideUpdate<true>(sh, dX2, dY2, inter);
      if (timedOut) sh->commError = 1;
    }
    __syncthreads();
    const int halted = sh->halted, curN = sh->cur, accepted = sh->lastAccepted;
    const double tau = sh->tau, avgWx = sh->avgWx;
    if (blockIdx.x == 0 && tid < (int)(sizeof(DevState) / 4)) {  // pending = 0 (decideCore); avgWx: added to xSum below
      uint32_t w = reinterpret_cast<const uint32_t*>(sh)[tid];
      constexpr int kAvgWxWord = offsetof(DevState, avgWx) / 4;
      if (!halted && (tid == kAvgWxWord || tid == kAvgWxWord + 1)) w = 0u;
      reinterpret_cast<uint32_t*>(a.stOut)[tid] = w;
    }
    if (halted) return;
    // the iterate the step starts from: the trial's (x+, A'y+) when it was accepted, else (x, A'y) again
    const double* __restrict__ xBase = a.v.x[curN];
    const double* __restrict__ atyBase = a.v.aty[curN];
    double* __restrict__ xOut = a.v.x[curN ^ 1];
    auto step = [&](int r, double xb, double ab, double c, double l, double u, double xs, double q) {
      if (avgWx != 0.0) stStream(a.v.xSum + r, xs + avgWx * xb);  // deferred PDHG_Update_Average (step.c:437)
      double t = xb;
      t += (-tau) * c;
      t += tau * ab;
      if (a.v.qdiag) t = t / (1.0 + tau * q);
      t = t < u ? t : u;
      t = t > l ? t : l;
      xOut[r] = t;  // gathered by the A x+ kernel: ordinary store
    };
    // (a long column's A'y+ was computed by a segment task, maybe in another block: taken from memory, agent scope)
    auto isLong = [&](int lr) { return (a.inlineTasks || a.coTaskBlocks) && longMajor(rBase + lr); };
#pragma unroll
    for (int k = 0; k < kSlabPre; ++k) {
      const int lr = tid + k * kSlabThreads;
      if (rBase + lr < rEnd) {  // (a rejected trial, 3 %, fetches x and A'y again)
        const int r = rBase + lr;
        const double ab = isLong(lr) ? ldAgent(atyBase + r) : accepted ? acc[lr] : ldStream(atyBase + r);
        step(r, accepted ? pre[k].b : ldStream(xBase + r), ab, fix[k].a, ULO ? lo0 : fix[k].b, fix[k].c, fix[k].d, fix[k].e);
      }
    }
    if (kExtra > 0) {  // the extra columns per thread: from registers too
#pragma unroll
      for (int k = kSlabPre; k < kFixN; ++k) {
        const int lr = tid + k * kSlabThreads;
        if (rBase + lr < rEnd) {
          const int r = rBase + lr;
          const double ab = isLong(lr) ? ldAgent(atyBase + r) : accepted ? acc[lr] : ldStream(atyBase + r);
          step(r, accepted ? xbLate[k - kSlabPre] : ldStream(xBase + r), ab, fix[k].a, ULO ? lo0 : fix[k].b, fix[k].c, fix[k].d,
               a.v.qdiag ? ldStream(a.v.qdiag + r) : 0.0);
        }
      }
    }
    // (more columns per block than that: kTail columns' operands per round trip — two in the LATE variants, where this
    // loop only sees blocks beyond 8192 / 4096 columns and the registers hold twice as many columns across the barrier)
    constexpr int kTail = LATE ? 2 : TWO ? 3 : kSlabPre;
    for (int lr0 = tid + kFixN * kSlabThreads; rBase + lr0 < rEnd; lr0 += kTail * kSlabThreads) {
      double xb[kTail], ab[kTail], cc[kTail], ll[kTail], uu[kTail], xs[kTail], qq[kTail];
#pragma unroll
      for (int k = 0; k < kTail; ++k) {
        const int lr1 = lr0 + k * kSlabThreads;
        const int lr = rBase + lr1 < rEnd ? lr1 : rEnd - 1 - rBase;
        const int r = rBase + lr;
        xb[k] = ldStream(xBase + r);
        ab[k] = isLong(lr) ? ldAgent(atyBase + r) : accepted ? acc[lr] : ldStream(atyBase + r);
        cc[k] = ldConst<CC>(a.v.cost + r); ll[k] = (uniB & 1) ? lo0 : ldConst<CC>(a.v.lower + r); uu[k] = (uniB & 2) ? up0 : ldConst<CC>(a.v.upper + r);
        xs[k] = ldStream(a.v.xSum + r);
        qq[k] = a.v.qdiag ? ldStream(a.v.qdiag + r) : 0.0;
      }
#pragma unroll
      for (int k = 0; k < kTail; ++k) {
        const int lr = lr0 + k * kSlabThreads;
        if (rBase + lr < rEnd) step(rBase + lr, xb[k], ab[k], cc[k], ll[k], uu[k], xs[k], qq[k]);
      }
    }
    if (a.prof) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      profStamp(3);
    }
  }
}

// x+ = clamp(x - tau (c - A'y), l, u): cupdlp_step.c:16-40, rounding as the CPU branch.
__global__ __launch_bounds__(kVecThreads) void k_primal_step(const IterVecs v, const DevState* st) {
  if (st->halted) return;
  const int cur = st->cur, nxt = cur ^ 1;
  const double tau = st->tau, avgW = st->avgWx;
  const double* __restrict__ x = v.x[cur];
  const double* __restrict__ aty = v.aty[cur];
  double* __restrict__ xn = v.x[nxt];
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
    const double xv = ldStream(x + j);
    if (avgW != 0.0) stStream(v.xSum + j, ldStream(v.xSum + j) + avgW * xv);  // deferred PDHG_Update_Average (step.c:437)
    double t = xv;
    t += (-tau) * ldStream(v.cost + j);
    t += tau * ldStream(aty + j);
    if (v.nx[0]) t += (-tau) * ldStream(v.nx[cur] + j);  // explicit gradient term of the off-diagonal part of Q
    if (v.qdiag) t = t / (1.0 + tau * ldStream(v.qdiag + j));
    const double u = ldStream(v.upper + j), l = ldStream(v.lower + j);
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
                                                        const double* dyGlobal, int onlyIfPending,
                                                        const double* __restrict__ partQ, int nQ) {
  if (st->halted) return;
  if (onlyIfPending && !st->pending) return;
  __shared__ double scratch[4][kVecThreads / kWave];
  double dY2, dX2, inter, qint = 0.0;
  trialSums(dyGlobal ? nullptr : partDY, nDY, partDX, partInter, nDX, scratch, dY2, dX2, inter, partQ, nQ, &qint);
  if (threadIdx.x != 0) return;
  decideUpdate(st, dX2, dyGlobal ? *dyGlobal : dY2, inter, qint);
}

// Single-GPU loop: decision of the previous trial + primal step of this one in ONE launch (see
// launchDecidePrimal).  x+ = clamp(x - tau (c - A'y), l, u): cupdlp_step.c:16-40, rounding as the CPU branch.
// The operands of the primal step are fetched BEFORE the decision is known, assuming the pending trial gets
// accepted (97 % do): their HBM latency then covers the reduction of the partials; after a rejection the
// two iterate-dependent operands are fetched again from the other buffers.
__global__ __launch_bounds__(kVecThreads) void k_decide_primal(const IterVecs v, const DevState* __restrict__ stIn,
                                                               DevState* __restrict__ stOut,
                                                               const double* __restrict__ partDY, int nDY,
                                                               const double* __restrict__ partDX,
                                                               const double* __restrict__ partInter, int nDX,
                                                               const double* __restrict__ partQ, int nQ) {
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  if (stIn->halted) {  // keep the two slots identical while the queue drains
    if (writer) *stOut = *stIn;
    return;
  }
  __shared__ double scratch[4][kVecThreads / kWave];
  __shared__ DevState sh;
  const int pending = stIn->pending;
  const int guess = pending ? (stIn->cur ^ 1) : stIn->cur;  // parity of the iterate if the pending trial is accepted
  constexpr int kPer = 2;  // elements per thread and pass, all their loads in flight together
  const int stride = gridDim.x * blockDim.x;
  const int j0 = blockIdx.x * blockDim.x + threadIdx.x;
  double xv[kPer], av[kPer], cv[kPer], lv[kPer], uv[kPer], sv[kPer], qv[kPer], nv[kPer];
  auto fetchFixed = [&](int base) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int j = base + k * stride;
      const int jj = j < v.n ? j : (v.n > 0 ? v.n - 1 : 0);  // clamped, unconditional (every vector has >= 1 element)
      cv[k] = ldStream(v.cost + jj); lv[k] = ldStream(v.lower + jj); uv[k] = ldStream(v.upper + jj);
      sv[k] = ldStream(v.xSum + jj);
      qv[k] = v.qdiag ? ldStream(v.qdiag + jj) : 0.0;
    }
  };
  auto fetchIterate = [&](int base, int par) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int j = base + k * stride;
      const int jj = j < v.n ? j : (v.n > 0 ? v.n - 1 : 0);
      xv[k] = ldStream(v.x[par] + jj); av[k] = ldStream(v.aty[par] + jj);
      nv[k] = v.nx[0] ? ldStream(v.nx[par] + jj) : 0.0;
    }
  };
  fetchIterate(j0, guess);
  fetchFixed(j0);
  if (pending) {
    double dY2, dX2, inter, qint = 0.0;
    trialSums(partDY, nDY, partDX, partInter, nDX, scratch, dY2, dX2, inter, partQ, nQ, &qint);
    if (threadIdx.x == 0) {
      sh = *stIn;
      decideUpdate(&sh, dX2, dY2, inter, qint);
    }
  } else if (threadIdx.x == 0) {
    sh = *stIn;
  }
  __syncthreads();
  const int halted = sh.halted, cur = sh.cur, nxt = cur ^ 1;
  const double tau = sh.tau, avgW = sh.avgWx;
  if (writer) {
    DevState t = sh;
    t.pending = halted ? 0 : 1;
    if (!halted) t.avgWx = 0.0;  // added to xSum below
    *stOut = t;
  }
  if (halted) return;
  double* __restrict__ xn = v.x[nxt];
  for (int base = j0; base < v.n; base += kPer * stride) {
    if (base != j0) { fetchIterate(base, cur); fetchFixed(base); }
    else if (cur != guess) fetchIterate(base, cur);  // the pending trial was rejected
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int j = base + k * stride;
      if (j >= v.n) break;
      if (avgW != 0.0) stStream(v.xSum + j, sv[k] + avgW * xv[k]);  // deferred PDHG_Update_Average (step.c:437)
      double t = xv[k];
      t += (-tau) * cv[k];
      t += tau * av[k];
      if (v.nx[0]) t += (-tau) * nv[k];          // explicit gradient term of the off-diagonal part of Q
      if (v.qdiag) t = t / (1.0 + tau * qv[k]);  // prox of 1/2 q x^2: argmin <c - A'y, x> + q x^2/2 + (x - x_k)^2 / (2 tau)
      t = t < uv[k] ? t : uv[k];
      t = t > lv[k] ? t : lv[k];
      xn[j] = t;  // gathered by the A x+ kernel: ordinary store
    }
  }
}

// Apply a pending average update (before a check iteration reads xSum/ySum).
__global__ __launch_bounds__(kVecThreads) void k_flush_average(const IterVecs v, const DevState* st) {
  const double w = st->avgW, wx = st->avgWx;  // (the two sides are consumed by different kernels of a trial)
  if (w == 0.0 && wx == 0.0) return;
  const int cur = st->cur;
  const int stride = gridDim.x * blockDim.x;
  const int tot = v.n + v.m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
    // (check-iteration kernels stream their vectors non-temporally too: the matrices stay in the Infinity Cache)
    if (i < v.n) { if (wx != 0.0) stStream(v.xSum + i, ldStream(v.xSum + i) + wx * ldStream(v.x[cur] + i)); }
    else if (w != 0.0) stStream(v.ySum + (i - v.n), ldStream(v.ySum + (i - v.n)) + w * ldStream(v.y[cur] + (i - v.n)));
  }
}
__global__ void k_clear_avgw(DevState* st) { st->avgW = 0.0; st->avgWx = 0.0; }

// Check iteration, one pass instead of three (k_flush_average, k_clear_avgw, 2 x k_scale_copy): the pending average
// update and the average itself, xAvg = xSum / sum(tau), yAvg = ySum / sum(sigma) (PDHG_Compute_Average_Iterate,
// cupdlp_step.c:377-420).  Same operations, in the same order, as the separate kernels: bit-identical.  The pending
// weights come from the host's copy of the state (host-driven check: the host clears them there and pushes the state
// back before the loop resumes) or, in a device-driven check, from the state record itself (k_check_decide clears them).
__global__ __launch_bounds__(kVecThreads) void k_flush_scale(const IterVecs v, const CheckGate g, int cur, double w, double wx, double ps,
                                                             double ds, double* __restrict__ xAvg, double* __restrict__ yAvg) {
  if (g.st) {
    if (!checkDue(g.st, g.cc)) return;
    cur = g.st->cur; w = g.st->avgW; wx = g.st->avgWx;
    ps = g.st->sumPrimalStep > 0.0 ? 1.0 / g.st->sumPrimalStep : 1.0;
    ds = g.st->sumDualStep > 0.0 ? 1.0 / g.st->sumDualStep : 1.0;
  }
  const int stride = gridDim.x * blockDim.x;
  const int tot = v.n + v.m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
    if (i < v.n) {
      double sx = ldStream(v.xSum + i);
      if (wx != 0.0) { sx = sx + wx * ldStream(v.x[cur] + i); stStream(v.xSum + i, sx); }
      xAvg[i] = sx * ps;  // (gathered by the A xAvg kernel: ordinary store)
    } else {
      const int k = i - v.n;
      double sy = ldStream(v.ySum + k);
      if (w != 0.0) { sy = sy + w * ldStream(v.y[cur] + k); stStream(v.ySum + k, sy); }
      yAvg[k] = sy * ds;
    }
  }
}

__global__ __launch_bounds__(kVecThreads) void k_scale_copy(double* __restrict__ dst, const double* __restrict__ src,
                                                            double a, int len) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) dst[i] = ldStream(src + i) * a;
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

// Row pass of PDHG_Compute_Primal_Feasibility / the y-side of the dual objective and of the infeasibility
// certificates (cupdlp_solver.c:12-67,80,230,339-345) for the current AND the average iterate in one pass (rhs and
// rowScale are read once; one barrier for the eight block sums): quantities 0..3 = current, 4..7 = average.
__global__ __launch_bounds__(kVecThreads) void k_row_stats2(const IterVecs v, const CheckGate g, int cur, const double* __restrict__ axA,
                                                            const double* __restrict__ yA, const double* __restrict__ rowScale,
                                                            int scaled, double* partials, int pstride) {
  if (g.st) {
    if (!checkDue(g.st, g.cc)) return;
    cur = g.st->cur;
  }
  __shared__ double scratch[2 * kRowStats][kVecThreads / kWave];
  const double* __restrict__ axC = v.ax[cur];
  const double* __restrict__ yC = v.y[cur];
  const double* __restrict__ rhs = v.rhs;
  const int m = v.m, nEqs = v.nEqs, rowOffset = v.rowOffset;
  double a[2 * kRowStats];
#pragma unroll
  for (int q = 0; q < 2 * kRowStats; ++q) a[q] = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
    rowStatsElem<false>(a, i, axC, yC, axA, yA, rhs, rowScale, scaled, (i + rowOffset) >= nEqs);
  blockSumManyAt<2 * kRowStats, false>(a, scratch, partials, pstride, (int)blockIdx.x);
}

// Column pass of PDHG_Compute_Dual_Feasibility and the x-side certificates (cupdlp_solver.c:69-204, 229-256,
// 326-366) for the current AND the average iterate in one pass (cost, bounds and colScale are read once; one barrier
// for the 22 block sums): quantities 0..10 = current, 11..21 = average; also stores the slacks s+, s- of both.
__global__ __launch_bounds__(kVecThreads) void k_col_stats2(const IterVecs v, const CheckGate g, int cur, const double* __restrict__ atyA,
                                                            const double* __restrict__ xA, const double* __restrict__ colScale,
                                                            const double* __restrict__ nxA, int scaled, double* spC, double* snC,
                                                            double* spA, double* snA, double* partials, int pstride) {
  if (g.st) {
    if (!checkDue(g.st, g.cc)) return;
    cur = g.st->cur;
  }
  const ColStatPtrs p{v.aty[cur], v.x[cur], atyA, xA, v.cost, v.lower, v.upper, colScale, v.qdiag, v.nx[0] ? v.nx[cur] : nullptr, nxA,
                      spC, snC, spA, snA};
  const int n = v.n;
  __shared__ double scratch[2 * kColStats][kVecThreads / kWave];
  double a[2 * kColStats];
#pragma unroll
  for (int q = 0; q < 2 * kColStats; ++q) a[q] = 0.0;
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) colStatsElem<false>(a, j, p, scaled);
  blockSumManyAt<2 * kColStats, false>(a, scratch, partials, pstride, (int)blockIdx.x);
}

// out[q] = fixed-order sum of quantity q's per-block partials; the first nQ0 quantities have nBlocks0 partials each,
// the others nBlocks1 (row and column statistics of a check in one launch)
__global__ __launch_bounds__(kVecThreads) void k_final_reduce2(const double* partials, int pstride, int nQ0, int nBlocks0,
                                                               int nBlocks1, double* out, const CheckGate g) {
  if (!gateOpen(g)) return;
  __shared__ double scratch[kVecThreads / kWave];
  const double s = reducePartials(partials + (size_t)blockIdx.x * pstride, (int)blockIdx.x < nQ0 ? nBlocks0 : nBlocks1, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(kVecThreads) void k_final_reduce(const double* partials, int pstride, int nBlocks,
                                                              double* out, const int32_t* gate) {
  if (gate && *gate == 0) return;
  __shared__ double scratch[kVecThreads / kWave];
  const double s = reducePartials(partials + (size_t)blockIdx.x * pstride, nBlocks, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(kVecThreads) void k_diff_norm2(const double* __restrict__ a, const double* __restrict__ b,
                                                            int len, double* partials, const int32_t* gate) {
  if (gate && *gate == 0) return;
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
// PDLP_MI355X_SLAB_PROF=1 (development): per-block phase times of the two slab launches of a trial, printed at exit —
// mean / fastest / slowest block of {stream, + epilogue, + grid barrier, + decision and primal step}, us per launch.
unsigned long long* slabProf() {
  static unsigned long long* prof = [] {
    unsigned long long* p = nullptr;
    constexpr size_t kWords = 2 * 1024 * 8;
    if (devEnv("PDLP_MI355X_SLAB_PROF") && hipMalloc((void**)&p, kWords * 8) == hipSuccess) {
      (void)hipMemset(p, 0, kWords * 8);
      static unsigned long long* keep = p;
      atexit([] {
        std::vector<unsigned long long> h(kWords);
        if (hipMemcpy(h.data(), keep, kWords * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
        if (const char* path = devEnv("PDLP_MI355X_SLAB_PROF"); path && path[0] != '1') {  // a path: the raw table too
          if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, kWords, f); fclose(f); }
        }
        for (int half = 0; half < 2; ++half) {
          const int base = half * 1024;
          auto stat = [&](int b0, int b1, int k, double& mean, double& lo, double& hi) {
            mean = 0; lo = 1e300; hi = 0;
            int cnt = 0;
            for (int b = b0; b < b1; ++b) {
              const unsigned long long n = h[(size_t)(base + b) * 8];
              if (!n) continue;
              const double us = (double)h[(size_t)(base + b) * 8 + 1 + k] * 0.01 / (double)n;
              mean += us; lo = us < lo ? us : lo; hi = us > hi ? us : hi; ++cnt;
            }
            if (cnt) mean /= cnt;
            return cnt;
          };
          for (int k : {0, 1, 4, 2, 3}) {  // rows 0..511: the streaming blocks (logical block index)
            double mean, lo, hi;
            const int cnt = stat(0, 512, k, mean, lo, hi);
            if (cnt && hi > 0)
              fprintf(stderr, "slab launch %s, %d blocks, to the end of %s: mean %.2f us, fastest block %.2f, slowest %.2f\n",
                      half ? "A'y+ (fused)" : "A x+", cnt, k == 0 ? "the stream" : k == 1 ? "the epilogue" : k == 4 ? "the arrival" : k == 2 ? "the grid barrier" : "the kernel",
                      mean, lo, hi);
          }
          double mean, lo, hi;  // rows 512..: the task workgroups (segment tasks of the long majors), when each was done
          const int cnt = stat(512, 1024, 0, mean, lo, hi);
          if (cnt) fprintf(stderr, "slab launch %s, %d task workgroups done: mean %.2f us, first %.2f, last %.2f\n", half ? "A'y+ (fused)" : "A x+", cnt, mean, lo, hi);
        }
      });
    }
    return p;
  }();
  return prof;
}

template <int EPI>
void launchSpmv(const MatView& M, SpmvArgs a, hipStream_t s) {
  a.xcdMap = M.xcdMap;
  if (EPI == kDualStep && M.useSlab && M.slab.nBlocks <= 512) a.prof = slabProf();
  a.L = M.lng;
  a.A = M.csr;
  const int nTasks = M.lng.nTasks;
  if (M.useSlab && M.slab.nBlocks > 0) {
    a.S = M.slab;
    const size_t lds = (size_t)((M.slab.rowsPerBlock + 1) & ~1) * 8 + kSlabThreads * 8 + 2 * (kSlabThreads / kWave) * 8;
    const dim3 grid(M.slab.nBlocks + (nTasks + M.lng.taskGroup - 1) / M.lng.taskGroup);  // one task group per extra workgroup
    // (gather distance 2 / 3 with 4 / 6 slots measured the same as (3, 1) on the random and on the structured LP, round 3)
    // segment tasks ride along: register budget for two resident blocks per CU, so that a task block runs NEXT to a streaming one
    if (nTasks > 0) hipLaunchKernelGGL((k_spmv_slab<EPI, true, kSlabSlots, 1>), grid, dim3(kSlabThreads), lds, s, a);
    else hipLaunchKernelGGL((k_spmv_slab<EPI, false, kSlabSlots, 1>), grid, dim3(kSlabThreads), lds, s, a);
  } else if (M.csr.nBlocks > 0 || nTasks > 0) {
    const dim3 grid(M.csr.nBlocks + (nTasks + kSpmvThreads / kWave - 1) / (kSpmvThreads / kWave)), block(kSpmvThreads);
    if (M.csr.chunk == kChunkSmall) hipLaunchKernelGGL((k_spmv<EPI, kChunkSmall>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_spmv<EPI, kChunk>), grid, block, 0, s, a);
  }
  if (nTasks > 0 && M.lng.contrib && (EPI == kDualStep || EPI == kAtyInteract || EPI == kQxInteract))
    hipLaunchKernelGGL(k_long_groups, dim3((M.lng.nSlots + kVecThreads - 1) / kVecThreads), dim3(kVecThreads), 0, s, M.lng,
                       a.st, a.part0, EPI == kAtyInteract ? a.part1 : nullptr);
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
namespace {
size_t fusedLds(const MatView& At) {
  return (size_t)((At.slab.rowsPerBlock + 1) & ~1) * 8 + kSlabThreads * 8 + 2 * (kSlabThreads / kWave) * 8 + 4 * (kVecThreads / kWave) * 8 +
         sizeof(DevState) + 16;
}
}  // namespace
namespace {
int fusedTaskGroups(const MatView& At) { return (At.lng.nTasks + At.lng.taskGroup - 1) / At.lng.taskGroup; }
}  // namespace
int fusedCoTaskBlocks(const MatView& At, int device) {
  // long columns of a slab operand (not beyond kLongSlotCap: their contributions would need the k_long_groups launch):
  // the 64-register variant of the fused kernel holds two 1024-thread blocks per CU — every streaming block and every
  // task workgroup resident at once — where its LDS request fits twice
  if (!At.useSlab || At.lng.nTasks <= 0 || At.lng.contrib != nullptr || At.slab.nBlocks <= 0) return 0;
  int perCu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_spmv_slab<kAtyFused, true, kSlabSlots, 1>, kSlabThreads, fusedLds(At)) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  const int groups = fusedTaskGroups(At);
  if (perCu < 2 || At.slab.nBlocks > cus || groups > cus) return 0;  // (one streaming block and at most one task workgroup per CU)
  return groups;
}
int fusedAtyBlocksResident(const MatView& At, int device) {
  (void)slabProf();  // (development buffer: allocated at set-up, never inside a stream capture)
  // long columns: the slab kernel runs their segment tasks inside the fused launch (co-resident task workgroups,
  // MatView::coTaskBlocks, or the streaming blocks themselves, SpmvArgs::inlineTasks); not the stream-layout kernel, and
  // not beyond kLongSlotCap long columns (their contributions need the k_long_groups launch)
  if (At.lng.nTasks > 0 && (!At.useSlab || At.lng.contrib != nullptr)) return 0;
  int perCu = 0, cus = 0;
  hipError_t e;
  if (At.useSlab) {
    if (At.slab.nBlocks <= 0) return 0;
    e = At.coTaskBlocks > 0
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_spmv_slab<kAtyFused, true, kSlabSlots, 1>, kSlabThreads, fusedLds(At))
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_spmv_slab<kAtyFused, false, kSlabSlots, 1, true>, kSlabThreads, fusedLds(At));
  } else {
    if (At.csr.nBlocks <= 0) return 0;
    e = At.csr.chunk == kChunkSmall ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_spmv<kAtyFused, kChunkSmall>, kSpmvThreads, 0)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_spmv<kAtyFused, kChunk>, kSpmvThreads, 0);
  }
  if (e != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  return perCu * cus;
}
int fusedAtyBlocks(const MatView& At) { return At.useSlab ? At.slab.nBlocks + At.coTaskBlocks : At.csr.nBlocks; }
void launchSpmvAtyFusedPrimal(const MatView& At, const IterVecs& v, const DevState* stIn, DevState* stOut,
                              const double* partDY, int32_t nDY, double* partDX, double* partInter, unsigned long long* bar,
                              hipStream_t s, int32_t timeoutMs, int32_t faultTrial) {
  SpmvArgs a{};
  a.barLimit = (unsigned long long)(timeoutMs > 0 ? timeoutMs : 1000) * 100000ull;
  a.faultTrial = faultTrial;
  a.coTaskBlocks = At.useSlab && At.lng.nTasks > 0 ? At.coTaskBlocks : 0;
  a.inlineTasks = At.useSlab && At.lng.nTasks > 0 && a.coTaskBlocks == 0 ? 1 : 0;
  a.touchTail = At.touchTail;
  a.st = stIn; a.v = v; a.part0 = partDX; a.part1 = partInter;
  a.stOut = stOut; a.partDY = partDY; a.nDY = nDY; a.nDX = At.nPartials; a.bar = bar;
  if (At.useSlab && At.slab.nBlocks <= 512) a.prof = slabProf();
  a.xcdMap = At.xcdMap; a.L = At.lng; a.A = At.csr; a.S = At.slab;
  // (LATE — twice as many columns stepped from registers behind the barrier — in the 128-register variant only: in the
  // 64-register one, which carries the task workgroups, it spills and measured slower: config d 36.5 -> 41.2 us, round 6)
  // (CC — c, l, u of the primal step as ordinary loads where IterVecs::constCached says the Infinity Cache has room; ULO —
  // every column shares one lower bound, IterVecs::lowerUniform)
  const dim3 gridTwo(At.slab.nBlocks + a.coTaskBlocks), gridOne(At.slab.nBlocks), block(kSlabThreads);
  const size_t lds = At.useSlab ? fusedLds(At) : 0;
  auto launchTwo = [&](auto cc, auto ulo) {
    hipLaunchKernelGGL((k_spmv_slab<kAtyFused, true, kSlabSlots, 1, false, decltype(cc)::value, decltype(ulo)::value>), gridTwo, block, lds, s, a);
  };
  auto launchOne = [&](auto cc, auto ulo) {
    hipLaunchKernelGGL((k_spmv_slab<kAtyFused, false, kSlabSlots, 1, true, decltype(cc)::value, decltype(ulo)::value>), gridOne, block, lds, s, a);
  };
  auto pick = [&](auto&& launch) {
    using T = std::true_type; using F = std::false_type;
    if (v.constCached) { if (v.lowerUniform) launch(T{}, T{}); else launch(T{}, F{}); }
    else { if (v.lowerUniform) launch(F{}, T{}); else launch(F{}, F{}); }
  };
  if (At.useSlab && a.coTaskBlocks > 0) pick(launchTwo);
  else if (At.useSlab) pick(launchOne);
  else if (At.csr.chunk == kChunkSmall)
    hipLaunchKernelGGL((k_spmv<kAtyFused, kChunkSmall>), dim3(At.csr.nBlocks), dim3(kSpmvThreads), 0, s, a);
  else
    hipLaunchKernelGGL((k_spmv<kAtyFused, kChunk>), dim3(At.csr.nBlocks), dim3(kSpmvThreads), 0, s, a);
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
void launchSpmvPlain(const MatView& A, const double* in, double* out, hipStream_t s, CheckGate g) {
  SpmvArgs a{};
  a.st = nullptr; a.in = in; a.out = out; a.gate = g;
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
                  int32_t nDX, const double* dyGlobal, hipStream_t s, bool onlyIfPending, const double* partQ, int32_t nQ) {
  hipLaunchKernelGGL(k_decide, dim3(1), dim3(kVecThreads), 0, s, st, partDY, nDY, partDX, partInter, nDX, dyGlobal,
                     onlyIfPending ? 1 : 0, partQ, nQ);
}
void launchSpmvQxInteract(const MatView& N, const IterVecs& v, const DevState* st, double* partQ, hipStream_t s) {
  SpmvArgs a{};
  a.st = st; a.v = v; a.part0 = partQ;
  launchSpmv<kQxInteract>(N, a, s);
}
void launchDecidePrimal(const IterVecs& v, const DevState* stIn, DevState* stOut, const double* partDY, int32_t nDY,
                        const double* partDX, const double* partInter, int32_t nDX, hipStream_t s, const double* partQ,
                        int32_t nQ) {
  // 4 blocks per CU, several passes per thread: 14.1 us against 17.0 us with one pass per thread (2048 blocks) at
  // n = 1M — the loads of the next pass overlap the stores of the current one
  int nb = vecBlocks(v.n);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_decide_primal, dim3(nb), dim3(kVecThreads), 0, s, v, stIn, stOut, partDY, nDY, partDX,
                     partInter, nDX, partQ, nQ);
}
void launchFlushAverage(const IterVecs& v, DevState* st, hipStream_t s) {
  hipLaunchKernelGGL(k_flush_average, dim3(vecBlocks(v.n + v.m)), dim3(kVecThreads), 0, s, v, st);
  hipLaunchKernelGGL(k_clear_avgw, dim3(1), dim3(1), 0, s, st);
}
void launchClearAvgW(DevState* st, hipStream_t s) { hipLaunchKernelGGL(k_clear_avgw, dim3(1), dim3(1), 0, s, st); }
void launchFlushScale(const IterVecs& v, CheckGate g, int cur, double w, double wx, double ps, double ds, double* xAvg, double* yAvg,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_flush_scale, dim3(vecBlocks(v.n + v.m)), dim3(kVecThreads), 0, s, v, g, cur, w, wx, ps, ds, xAvg, yAvg);
}
void launchRowStats2(const IterVecs& v, CheckGate g, int cur, const double* axA, const double* yA, const double* rowScale, int scaled,
                     double* partials, int32_t stride, int32_t nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(k_row_stats2, dim3(nBlocks), dim3(kVecThreads), 0, s, v, g, cur, axA, yA, rowScale, scaled, partials, stride);
}
void launchColStats2(const IterVecs& v, CheckGate g, int cur, const double* atyA, const double* xA, const double* colScale,
                     const double* nxA, int scaled, double* spC, double* snC, double* spA, double* snA, double* partials,
                     int32_t stride, int32_t nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(k_col_stats2, dim3(nBlocks), dim3(kVecThreads), 0, s, v, g, cur, atyA, xA, colScale, nxA, scaled, spC, snC, spA,
                     snA, partials, stride);
}
void launchFinalReduce2(const double* partials, int32_t stride, int32_t nQ0, int32_t nBlocks0, int32_t nQ1, int32_t nBlocks1,
                        double* out, CheckGate g, hipStream_t s) {
  hipLaunchKernelGGL(k_final_reduce2, dim3(nQ0 + nQ1), dim3(kVecThreads), 0, s, partials, stride, nQ0, nBlocks0, nBlocks1, out, g);
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
void launchFinalReduce(const double* partials, int32_t stride, int32_t nBlocks, int32_t nQ, double* out,
                       hipStream_t s, const int32_t* gate) {
  hipLaunchKernelGGL(k_final_reduce, dim3(nQ), dim3(kVecThreads), 0, s, partials, stride, nBlocks, out, gate);
}
void launchDiffNorm2(const double* a, const double* b, int32_t len, double* partials, int32_t nBlocks,
                     hipStream_t s, const int32_t* gate) {
  hipLaunchKernelGGL(k_diff_norm2, dim3(nBlocks), dim3(kVecThreads), 0, s, a, b, len, partials, gate);
}
namespace {
// Per slab block b (majors [waveBeg[16 b], waveBeg[16 b + 16])), ONE workgroup each: smallest and largest minor index over
// its majors of at most `longLimit` entries (CSR with ascending minors: the first and the last entry of a major), their
// entry count, and — hist != nullptr — the entries per tile of 2^tileLog2 minors, added to the histogram row of the XCD
// that runs the block under the contiguous map (pdlp_host.hpp xcdOfLogicalBlock / xcdTileOwners).  Thread-local
// min / max / count, block reduction, one store per block (round 5 took three contended global atomics per major:
// 3.4 ms per call at 1M majors); the histogram is counted in LDS and flushed with one integer atomic per non-empty tile.
constexpr int kSpanMaxTiles = 4096;
__global__ __launch_bounds__(kVecThreads) void k_block_span(const int32_t* __restrict__ beg, const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ waveBeg, int nBlocks, int longLimit, int32_t* lo,
                                                            int32_t* hi, int32_t* cnt, int tileLog2, int nTiles, int32_t* hist) {
  __shared__ int32_t tile[kSpanMaxTiles];
  __shared__ int32_t red[3][kVecThreads / kWave];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r0 = waveBeg[b * 16], r1 = waveBeg[b * 16 + 16];
  if (hist) for (int t = tid; t < nTiles; t += kVecThreads) tile[t] = 0;
  __syncthreads();
  int32_t l = INT_MAX, h = -1, c = 0;
  for (int r = r0 + tid; r < r1; r += kVecThreads) {
    const int p0 = beg[r], p1 = beg[r + 1];
    if (p1 <= p0 || p1 - p0 > longLimit) continue;
    const int32_t a = idx[p0], z = idx[p1 - 1];
    l = a < l ? a : l; h = z > h ? z : h; c += p1 - p0;
    if (hist) for (int p = p0; p < p1; ++p) atomicAdd(&tile[idx[p] >> tileLog2], 1);
  }
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const int32_t l2 = __shfl_down(l, off, kWave), h2 = __shfl_down(h, off, kWave), c2 = __shfl_down(c, off, kWave);
    l = l2 < l ? l2 : l; h = h2 > h ? h2 : h; c += c2;
  }
  if ((tid & (kWave - 1)) == 0) { red[0][tid / kWave] = l; red[1][tid / kWave] = h; red[2][tid / kWave] = c; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kVecThreads / kWave; ++w) {
      l = red[0][w] < l ? red[0][w] : l; h = red[1][w] > h ? red[1][w] : h; c += red[2][w];
    }
    lo[b] = l; hi[b] = h; cnt[b] = c;
  }
  if (hist) {
    const int qlo = nBlocks / 8, rr = nBlocks % 8;  // = pdlp_host.hpp xcdOfLogicalBlock
    const int x = b < rr * (qlo + 1) ? b / (qlo + 1) : (qlo > 0 ? rr + (b - rr * (qlo + 1)) / qlo : 0);
    for (int t = tid; t < nTiles; t += kVecThreads)
      if (tile[t]) atomicAdd(hist + (size_t)x * nTiles + t, tile[t]);
  }
}
}  // namespace
void launchBlockSpan(const int32_t* beg, const int32_t* idx, const int32_t* waveBeg, int32_t nBlocks, int32_t longLimit, int32_t* lo,
                     int32_t* hi, int32_t* cnt, int32_t tileLog2, int32_t nTiles, int32_t* hist, hipStream_t s) {
  if (nBlocks <= 0) return;
  if (nTiles > kSpanMaxTiles) hist = nullptr;  // (xcdTileLog2 never asks for more)
  hipLaunchKernelGGL(k_block_span, dim3(nBlocks), dim3(kVecThreads), 0, s, beg, idx, waveBeg, nBlocks, longLimit, lo, hi, cnt, tileLog2,
                     nTiles, hist);
}
namespace {
__global__ __launch_bounds__(kVecThreads) void k_block_bounds(const double* __restrict__ lower, const double* __restrict__ upper,
                                                              const int32_t* __restrict__ waveBeg, int32_t* uni, double* bounds) {
  __shared__ int diff[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r0 = waveBeg[b * 16], r1 = waveBeg[b * 16 + 16];  // (never empty)
  if (tid < 2) diff[tid] = 0;
  __syncthreads();
  const long long l0 = __double_as_longlong(lower[r0]), u0 = __double_as_longlong(upper[r0]);
  bool dl = false, du = false;
  for (int r = r0 + tid; r < r1; r += kVecThreads) {
    dl = dl || __double_as_longlong(lower[r]) != l0;
    du = du || __double_as_longlong(upper[r]) != u0;
  }
  if (dl) diff[0] = 1;
  if (du) diff[1] = 1;
  __syncthreads();
  if (tid == 0) {
    uni[b] = (diff[0] ? 0 : 1) | (diff[1] ? 0 : 2);
    bounds[2 * b] = lower[r0];
    bounds[2 * b + 1] = upper[r0];
  }
}
}  // namespace
void launchBlockBounds(const double* lower, const double* upper, const int32_t* waveBeg, int32_t nBlocks, int32_t* uni, double* bounds,
                       hipStream_t s) {
  if (nBlocks <= 0) return;
  hipLaunchKernelGGL(k_block_bounds, dim3(nBlocks), dim3(kVecThreads), 0, s, lower, upper, waveBeg, uni, bounds);
}
namespace {
__global__ __launch_bounds__(kVecThreads) void k_add_int(int32_t* v, int32_t d, long long len) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) v[i] += d;
}
}  // namespace
void launchAddInt(int32_t* v, int32_t d, int64_t len, hipStream_t s) {
  if (len <= 0) return;
  hipLaunchKernelGGL(k_add_int, dim3(vecBlocks((int32_t)std::min<int64_t>(len, 1 << 30))), dim3(kVecThreads), 0, s, v, d, (long long)len);
}
void launchDot(const double* a, const double* b, int32_t len, double* partials, int32_t nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(k_dot, dim3(nBlocks), dim3(kVecThreads), 0, s, a, b, len, partials);
}

}  // namespace pdlp
