// pdlp_small.hip — the PDHG trial loop of SMALL and MID-SIZE LPs (every work block of both operands resident at once)
// as ONE persistent launch.
//
// Below ~10^5 nonzeros a trial step is latency, not bandwidth: three dependent launches of ~6 us each, of which
// ~2 us is the kernel boundary and the rest a chain of three or four dependent memory round trips.  Here a batch
// of trials runs inside one launch of a few dozen resident workgroups: the phases of a trial
//     P  x+ = clamp(x - tau (c - A'y), l, u) on a share of the columns        cupdlp_step.c:16-40
//     A  A x+ with the dual step and the (dy)^2 partials                       cupdlp_step.c:43-69
//     T  A'y+ with the (dx)^2 and dx.d(A'y) partials                           cupdlp_linalg.c:772-801
//     D  accept / reject and the step-size update, in EVERY workgroup          cupdlp_step.c:215-310
// are separated by grid barriers instead of kernel boundaries — one arrival word per workgroup and a sweep by one wave
// (pdlp_devfn.hpp gridBarrier) for a few dozen workgroups; per XCD first, then between the XCDs (hierBarrier) for the
// hundreds of workgroups of a mid-size LP (100k x 100k / 1M nonzeros: 490), where the sweep costs 4.8 us per barrier.
//   TWO barriers per trial where both operands have the small (512-entry) work blocks and no row is a segment task
// (template parameter PINA, round 4): x+_j = clamp(x_j - tau (c_j - (A'y)_j)) costs seven operations, so phase A does not
// wait for the owners of the columns it gathers — every lane recomputes x+ of its two entries' columns from (x, A'y) of
// the iterate the trial starts from (fetched for both parities while the previous decision is computed; cost and bounds
// sit in registers), with the operations of phase P in their order: the bits are the owner's.  The owner still stores
// x+ (phase T and the next trial read it) but nobody waits for that store: P | barrier | A becomes one phase.  Measured:
// 25fv47 12.4 -> 11.5, 80bau3b 14.2 -> 12.2 us per trial.  Mid-size blocks (eight entries per lane, five loads each,
// scratch) measured 26.8 -> 46.9 us per trial and LPs with long rows 11.2 -> 11.8: those keep the P phase.
// Work blocks, lane assignments and reduction trees are EXACTLY those of k_spmv / k_decide_primal (pdlp_kernels.hip),
// majors longer than a work block ride along as the segment tasks of longBlock (smallLongBlock, round 4), so iterates
// and decisions are bit-identical to the 3-launch loop and the oracle's device-order mode follows them unchanged.
//   Every launch starts with a ROLL CALL (pdlp_devfn.hpp rollCall): a plain launch promises no co-residency, and on a
// shared device a workgroup may not get its CU — the launch then changes nothing, reports commError = 3, and the solver
// goes on with plain launches (Solver::syncState).
//   Visibility inside a launch: per-CU L1s are never refreshed by other CUs' stores and the eight XCD L2s are not
// coherent with each other, so EVERY access to a vector that changes during the launch (iterates, sums, partials)
// is an agent-scope relaxed atomic (global_load/store sc1: write-through, L1-bypassing); a workgroup's stores have
// landed (s_waitcnt vmcnt(0) in every wave, then the block barrier) before its arrival word is written.  Matrix,
// plans, costs, bounds and right-hand sides never change: ordinary loads.
//   (Measured alternative for the gathers, round 3: ordinary cached loads behind an agent-scope acquire — buffer_inv sc1
// in every wave — after every barrier: phase A 5.8 -> 8.9 us, the decision 3.3 -> 6.2 us at 100k x 100k.  Round 4's
// tools/barrier_bench.hip prices the instruction itself: +1.5 us per barrier when ONE wave of a workgroup issues it,
// +28 us when all sixteen do.  The per-access agent-scope loads stay.)
//   XCD-LOCAL mode (the default): only every eighth workgroup of the launch works — under the dispatch order observed
// on this part those share ONE XCD, i.e. one coherent L2 — and then ordinary stores (the L1 is write-through) with
// non-temporal loads (served by the L2, never by a stale L1 line) are coherent without a trip to memory: a dependent
// round trip costs an L2 hit instead of an HBM access.  The placement is CHECKED, never assumed: every worker
// publishes the XCC id it runs on before the first trial; unless all agree the launch changes nothing and reports it,
// and the solver continues with agent-scope accesses on all XCDs (the mode above) from then on.
//   The state record lives in LDS of every workgroup (identical copies: every workgroup takes the same decision
// from the same partials); workgroup 0 writes it back when the batch ends or the device halts.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "pdlp_devfn.hpp"
#include "pdlp_kernels.hpp"

namespace pdlp {

namespace {

struct SmallArgs {
  SpmvMat A, At;
  LongMat LA, LAt;           // segment tasks of the long majors (nTasks = 0: none)
  int32_t nPartA, nPartAt;   // partial slots of each operand: stream blocks + long majors
  IterVecs v;
  DevState* st;
  double* partDY;
  double* partDX;
  double* partInter;
  unsigned long long* bar;
  int32_t xcdA, xcdAt;
  int32_t maxTrials;
  int32_t selfTest;          // XCD-local mode, first launch of a solver: check that nt loads see another CU's store behind an L1-warm line
  int32_t primalInA;         // the primal step is recomputed by the gathers of phase A (no P phase, no barrier behind it)
  unsigned long long expect; // workgroups the roll call waits for (= the working workgroups of the launch, unless a test asks for a failure)
  unsigned long long limit;  // 100 MHz ticks a roll call or barrier wait may last
  unsigned long long* prof;  // development: 100 MHz ticks per phase {P, barrier, A, barrier, T, barrier, D}, accumulated by workgroup 0
};

// (inside a launch that has passed its roll call a wait can only fail on a defect: the timeout raises the flag word,
// which the decision phase turns into commError = 1)
template <bool LOCAL>
__device__ __forceinline__ void arrive(unsigned long long* bar, int lb, int nBlocks, unsigned long long epoch, unsigned long long limit) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores (write-through, or into the shared L2) ...
  __syncthreads();                                    // ... before the block's arrival word is written
  if (threadIdx.x < kWave) (void)gridBarrier<LOCAL>(bar, lb, nBlocks, epoch, (int)threadIdx.x, limit);
  __syncthreads();
}
// the same through the XCD-hierarchical barrier (k-th barrier of the launch)
__device__ __forceinline__ void arriveHier(const HierBar& h, unsigned long long k) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < kWave) hierBarrier(h, k, (int)threadIdx.x);
  __syncthreads();
}
// accesses to the vectors that change during the launch: agent scope on all XCDs, or L2-coherent on one XCD
template <bool LOCAL>
__device__ __forceinline__ double ldM(const double* p) { return LOCAL ? ldStream(p) : ldAgent(p); }
template <bool LOCAL>
__device__ __forceinline__ void stM(double* p, double v) { if (LOCAL) *p = v; else stAgent(p, v); }

// The work block a workgroup owns in one operand, loaded ONCE per launch: a persistent workgroup runs the same
// block of A and of A' in every trial, and the matrix never changes — entries, values and the bookkeeping of the
// lane's first major stay in registers; a trial then only gathers, adds and stores.
template <int CHUNK>
struct OwnBlock {
  static constexpr int kPer = CHUNK / kSpmvThreads;
  int r0 = 0, r1 = 0, p0 = 0, cnt = 0, qb = 0, qe = 0, slot_ = 0;
  int32_t ci[kPer];
  double va[kPer];
  // primal step inside phase A (small blocks, two entries per lane): cost, bounds (and the diagonal of Q) of every entry's column
  static constexpr bool kKeepPrimal = kPer <= 2;
  double pc[kKeepPrimal ? kPer : 1], pl[kKeepPrimal ? kPer : 1], pu[kKeepPrimal ? kPer : 1], pq[kKeepPrimal ? kPer : 1];
  double fixed = 0.0;  // rhs of the lane's first row (phase A)
  bool have = false;
  __device__ __forceinline__ void loadPrimal(const IterVecs& v) {
    if (kKeepPrimal) {
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        pc[k] = v.cost[ci[k]]; pl[k] = v.lower[ci[k]]; pu[k] = v.upper[ci[k]]; pq[k] = v.qdiag ? v.qdiag[ci[k]] : 0.0;
      }
    }
  }
  __device__ __forceinline__ void load(const SpmvMat& M, int blk, bool dual, const double* rhs) {
    have = true;
    slot_ = M.partOffset + blk;
    const int4 bb = *reinterpret_cast<const int4*>(M.blockBeg + 4 * blk);  // (first major, end major, first entry, end entry)
    r0 = bb.x; r1 = bb.y; p0 = bb.z; cnt = bb.w - bb.z;
    const int tid = threadIdx.x;
    const int rFirst = r0 + tid;
    const int rr = rFirst < r1 ? rFirst : r1 - 1;
    qb = M.beg[rr] - p0;
    qe = M.beg[rr + 1] - p0;
    if (dual) fixed = rhs[rr];
    const int last = cnt > 0 ? cnt - 1 : 0;  // idx/val carry one pad element
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int q = tid + k * kSpmvThreads;
      const int qq = q < last ? q : last;
      ci[k] = M.idx[p0 + qq];
      va[k] = M.val[p0 + qq];
    }
  }
};

// One trial's pass over the owned block (same plan, lanes and sums as k_spmv's stream path) with the epilogue of
// phase A (DUAL) or T.  Per-thread reduction partials are added to acc0 / acc1.
// The primal step of a column from (x, A'y) of the iterate the trial starts from: the operations of phase P in their order,
// so every workgroup that needs x+_j gets the bits its owner stores (cupdlp_step.c:16-40; Q diagonal: the prox step of the QP path).
__device__ __forceinline__ double primalFrom(double x, double ay, double tau, double c, double l, double u, bool prox, double q) {
  double t = x;
  t += (-tau) * c;
  t += tau * ay;
  if (prox) t = t / (1.0 + tau * q);
  t = t < u ? t : u;
  t = t > l ? t : l;
  return t;
}

// xIn (PINA, phase A): the gathered vector's entries are handed in — x+ recomputed by the caller — instead of gathered here.
template <int CHUNK, bool DUAL, bool LOCAL, bool PINA = false>
__device__ __forceinline__ void smallSpmvBlock(const SmallArgs& a, const SpmvMat& M, const OwnBlock<CHUNK>& B, int cur, double sigma,
                                               double avgW, double* prod, double& acc0, double& acc1, const double* xIn = nullptr) {
  const int tid = threadIdx.x, nxt = cur ^ 1;
  const double* in = DUAL ? a.v.x[nxt] : a.v.y[nxt];
  constexpr int kPer = CHUNK / kSpmvThreads;
  const int r0 = B.r0, r1 = B.r1, p0 = B.p0, cnt = B.cnt;
  const int rFirst = r0 + tid;
  const int rr = rFirst < r1 ? rFirst : r1 - 1;
  int qb = B.qb, qe = B.qe;
  auto prefetch = [&](int r, bool first) {
    Pre p{0.0, 0.0, 0.0, 0.0, 0.0};
    if (DUAL) {
      p.a = ldM<LOCAL>(a.v.y[cur] + r); p.b = first ? B.fixed : a.v.rhs[r]; p.c = ldM<LOCAL>(a.v.ax[cur] + r);
      if (avgW != 0.0) p.d = ldM<LOCAL>(a.v.ySum + r);  // (the deferred average update's running sum: fetched with the rest, not behind the row sum)
    }
    else { p.a = ldM<LOCAL>(a.v.x[cur] + r); p.b = ldM<LOCAL>(a.v.x[nxt] + r); p.c = ldM<LOCAL>(a.v.aty[cur] + r); }
    return p;
  };
  double xg[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) xg[k] = DUAL && PINA ? xIn[k] : ldM<LOCAL>(in + B.ci[k]);
  Pre pre = prefetch(rr, true);
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int q = tid + k * kSpmvThreads;
    if (q < cnt) prod[slot(q)] = B.va[k] * xg[k];
  }
  __syncthreads();
  for (int r = rFirst; r < r1; r += kSpmvThreads) {
    if (r != rFirst) {
      qb = M.beg[r] - p0;
      qe = M.beg[r + 1] - p0;
      pre = prefetch(r, false);
    }
    const double s = majorSum(prod, qb, qe);
    if (DUAL) {
      const double yv = pre.a;
      if (avgW != 0.0) stM<LOCAL>(a.v.ySum + r, pre.d + avgW * yv);  // deferred PDHG_Update_Average (step.c:438)
      double t = yv;
      t += sigma * pre.b;
      t += (-2.0 * sigma) * s;
      t += sigma * pre.c;
      if (r + a.v.rowOffset >= a.v.nEqs) t = t > 0.0 ? t : 0.0;
      stM<LOCAL>(a.v.ax[nxt] + r, s);
      stM<LOCAL>(a.v.y[nxt] + r, t);
      const double d = yv - t;
      acc0 += d * d;
    } else {
      const double dx = pre.a - pre.b;
      const double da = pre.c - s;
      stM<LOCAL>(a.v.aty[nxt] + r, s);
      acc0 += dx * dx;
      acc1 += dx * da;
    }
  }
}


// The segment tasks of the long majors inside the persistent loop (majors longer than a stream block: standata, standgub,
// standmps, cplex1 of the reference's instances; dense rows / columns of structured LPs).  Task group tb = one task per
// wave, exactly the lanes, sums, LDS / ticket hand-overs and per-major partial slots of pdlp_kernels.hip longBlock — so
// the bits are those of the launch loop — with the loop's accesses to the changing vectors (ldM / stM).  Task groups are
// dealt to the workgroups round robin after their stream block; a major whose segments span task groups is finished by
// the wave with the last ticket, in whichever workgroup that is (agent-scope segment sums and tickets in every mode).
template <bool DUAL, bool LOCAL>
__device__ __forceinline__ void smallLongBlock(const SmallArgs& a, const LongMat& L, int tb, int cur, double sigma, double avgW,
                                               double* lds /* [4] */, double* part0, double* part1) {
  constexpr int W = kSpmvThreads / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
  const int nxt = cur ^ 1;
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
  const int r = T.major;
  double pa = 0.0, pb = 0.0, pc = 0.0, pd = 0.0;
  if (active && (seg == 0 || !T.contained)) {
    if (DUAL) {
      pa = ldM<LOCAL>(a.v.y[cur] + r); pb = a.v.rhs[r]; pc = ldM<LOCAL>(a.v.ax[cur] + r);
      if (avgW != 0.0) pd = ldM<LOCAL>(a.v.ySum + r);
    }
    else { pa = ldM<LOCAL>(a.v.x[cur] + r); pb = ldM<LOCAL>(a.v.x[nxt] + r); pc = ldM<LOCAL>(a.v.aty[cur] + r); }
  }
  const int32_t* __restrict__ idx = L.idx;
  const double* __restrict__ val = L.val;
  const double* in = DUAL ? a.v.x[nxt] : a.v.y[nxt];
  constexpr int kPer = kLongSegment / kWave;
  double s = 0.0;
  for (int base = T.pBeg; base < T.pEnd; base += kLongSegment) {
    int32_t ci[kPer];
    double va[kPer], xg[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {  // unconditional, clamped
      const int q = base + k * kWave + lane;
      const int qq = q < T.pEnd ? q : T.pEnd - 1;
      ci[k] = idx[qq];
      va[k] = val[qq];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) xg[k] = ldM<LOCAL>(in + ci[k]);
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
  } else if (last) {
    if (lane == 0) __hip_atomic_store(L.ticket + T.c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
    double v = 0.0;
    if (lane < T.nSeg)
      v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(L.segSum + T.first + lane),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int k = 0; k < T.nSeg; ++k) total += __shfl(v, k, kWave);  // left to right
    finish = true;
  }
  if (finish && lane == 0) {
    if (DUAL) {
      const double yv = pa;
      if (avgW != 0.0) stM<LOCAL>(a.v.ySum + r, pd + avgW * yv);  // deferred PDHG_Update_Average (step.c:438)
      double tt = yv;
      tt += sigma * pb;
      tt += (-2.0 * sigma) * total;
      tt += sigma * pc;
      if (r + a.v.rowOffset >= a.v.nEqs) tt = tt > 0.0 ? tt : 0.0;
      stM<LOCAL>(a.v.ax[nxt] + r, total);
      stM<LOCAL>(a.v.y[nxt] + r, tt);
      const double d = yv - tt;
      stM<LOCAL>(part0 + L.slotBase + T.c, 0.0 + d * d);
    } else {
      const double dx = pa - pb;
      const double da = pc - total;
      stM<LOCAL>(a.v.aty[nxt] + r, total);
      stM<LOCAL>(part0 + L.slotBase + T.c, 0.0 + dx * dx);
      stM<LOCAL>(part1 + L.slotBase + T.c, 0.0 + dx * da);
    }
  }
}

// MODE 0: agent-scope accesses, sweep barrier; 1: XCD-local; 2: agent-scope accesses, XCD-hierarchical barrier
// PINA: no P phase — the gathers of phase A recompute x+ of their columns (primalOf), the owner of a column stores it
// alongside; two barriers per trial instead of three.  A workgroup may then enter phase A of the next trial while another
// still sums the (dy)^2 partials of this one: those alternate between two halves of partDY with the trial's parity.
template <int CHUNK_A, int CHUNK_AT, int MODE, bool PINA>
__global__ __launch_bounds__(kSpmvThreads) void k_trials_small(const SmallArgs a) {
  constexpr bool LOCAL = MODE == 1;
  constexpr int kMaxChunk = CHUNK_A > CHUNK_AT ? CHUNK_A : CHUNK_AT;
  __shared__ double prod[kMaxChunk + kMaxChunk / 8 + 8];
  __shared__ double scratch[2][kSpmvThreads / kWave];
  __shared__ double tscr[4][kVecThreads / kWave];
  __shared__ DevState sh;
  __shared__ int placementOk;
  __shared__ int hierN;
  __shared__ uint32_t hierActive;
  const int tid = threadIdx.x;
  if (LOCAL && (blockIdx.x & 7) != 0) return;  // XCD-local: every eighth workgroup works
  const int lb = LOCAL ? (int)blockIdx.x >> 3 : (int)blockIdx.x;      // logical workgroup
  const int G = LOCAL ? (int)gridDim.x >> 3 : (int)gridDim.x;
  if (tid < (int)(sizeof(DevState) / 4)) reinterpret_cast<uint32_t*>(&sh)[tid] = reinterpret_cast<const uint32_t*>(a.st)[tid];
  padSlots(prod, kMaxChunk + kMaxChunk / 8 + 8, tid, kSpmvThreads);  // (never written again)
  __syncthreads();
  if (sh.halted) return;
  // Roll call (pdlp_devfn.hpp rollCall): nothing is written before every working workgroup of the launch is known to be
  // resident.  If they are not all there in time (another tenant holds CUs), the launch changes nothing but commError = 3
  // and the solver goes on with plain launches.
  if (tid < kWave) {
    const bool here = rollCall(a.bar + G + 1, a.expect, a.limit, tid);
    if (tid == 0) placementOk = here ? 1 : 0;
  }
  __syncthreads();
  if (!placementOk) {
    if (tid == 0) __hip_atomic_store(&a.st->commError, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  __syncthreads();
  // the blocks this workgroup owns (the grid has at least as many workgroups as either operand has blocks)
  const int nA = a.A.nBlocks, nAt = a.At.nBlocks;
  const int nTBA = (a.LA.nTasks + kSpmvThreads / kWave - 1) / (kSpmvThreads / kWave);    // task groups of 4 (one task per wave)
  const int nTBAt = (a.LAt.nTasks + kSpmvThreads / kWave - 1) / (kSpmvThreads / kWave);
  OwnBlock<CHUNK_A> bA;
  OwnBlock<CHUNK_AT> bAt;
  if (lb < nA) {
    bA.load(a.A, a.xcdA ? xcdContiguousBlock(lb, nA) : lb, true, a.v.rhs);
    if (PINA) bA.loadPrimal(a.v);
  }
  if (lb < nAt) bAt.load(a.At, a.xcdAt ? xcdContiguousBlock(lb, nAt) : lb, false, nullptr);
  if (LOCAL) {
    // the placement check: XCC ids of all workers (words behind the arrival words and the timeout flag)
    unsigned long long* ids = a.bar + G + 8;
    if (tid == 0) __hip_atomic_store(ids + lb, (unsigned long long)xccId() + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrive<false>(a.bar, lb, G, 4ull * (unsigned long long)sh.nTrials + 1ull, a.limit);
    if (tid == 0) {
      const unsigned long long mine = __hip_atomic_load(ids + lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int ok = 1;
      for (int i = 0; i < G; ++i) ok &= __hip_atomic_load(ids + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine;
      placementOk = ok;
    }
    __syncthreads();
    if (!placementOk) {  // nothing has been touched: report, and let the host continue on all XCDs
      if (lb == 0 && tid == 0) a.st->commError = 2;
      return;
    }
    if (a.selfTest) {
      // What this mode relies on, checked once per solver on the placement it actually got: a `global_load ... nt` is served
      // by the XCD's L2, never by a line this CU's L1 still holds from an earlier read (measured so on this part:
      // MI355X_MICROARCH.md, "sc1 / sc0 sc1 / nt loads bypass L1 only"; the ISA does not promise it for nt).  Every
      // workgroup warms its L1 with a plain read of its neighbour's word, the neighbour then stores a new value (plain
      // store, as the loop's stores), and an nt load must return the new value.  If any workgroup reads the old one the
      // launch changes nothing and reports the placement as unusable: the solver goes on with agent-scope accesses.
      unsigned long long* tw = a.bar + smallBarWords(G) - (size_t)(2 * G + 8);  // G test words, G arrival words, flag, failure word
      unsigned long long* tbar = tw + G;
      unsigned long long* fail = tbar + G + 1;
      const int nb = (lb + 1) % G;
      auto plainLoad = [&](const unsigned long long* p) {
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
      };
      auto ntLoad = [&](const unsigned long long* p) {
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
      };
      if (tid == 0) *reinterpret_cast<volatile unsigned long long*>(tw + lb) = 1ull;
      arrive<true>(tbar, lb, G, 1ull, a.limit);
      unsigned long long warm = 0;
      if (tid == 0) warm = plainLoad(tw + nb);  // now in this CU's L1
      arrive<true>(tbar, lb, G, 2ull, a.limit);
      if (tid == 0) *reinterpret_cast<volatile unsigned long long*>(tw + lb) = 2ull;
      arrive<true>(tbar, lb, G, 3ull, a.limit);
      if (tid == 0 && (warm != 1ull || ntLoad(tw + nb) != 2ull)) __hip_atomic_store(fail, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      arrive<true>(tbar, lb, G, 4ull, a.limit);
      if (tid == 0) placementOk = __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull ? 1 : 0;
      __syncthreads();
      if (!placementOk) {
        if (lb == 0 && tid == 0) a.st->commError = 2;
        return;
      }
    }
  }
  HierBar hb{};
  unsigned long long kbar = 0;
  if (MODE == 2) {
    // registration: how many workgroups sit on which XCD (the barrier words were zeroed by the host before this launch)
    hb.base = a.bar + ((2 * G + 16 + kXccStride - 1) / kXccStride) * kXccStride;
    hb.flag = a.bar + G;
    hb.limit = a.limit;
    hb.xcc = xccId();
    if (tid == 0) __hip_atomic_fetch_add(hb.reg(hb.xcc), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrive<false>(a.bar, lb, G, 1ull, a.limit);
    if (tid < kWave) {
      const unsigned long long c = tid < kXccSlots ? __hip_atomic_load(hb.reg(tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      const unsigned long long act = __ballot(c > 0);
      const unsigned long long mine = __shfl(c, hb.xcc);
      if (tid == 0) { hierN = (int)mine; hierActive = (uint32_t)act; }
    }
    __syncthreads();
    hb.nLocal = hierN;
    hb.active = hierActive;
  }
  auto meet = [&](unsigned long long epoch) {
    if (MODE == 2) arriveHier(hb, ++kbar);
    else arrive<LOCAL>(a.bar, lb, G, epoch, a.limit);
  };
  unsigned long long tPrev = a.prof ? wall_clock64() : 0ull;
  auto stamp = [&](int k) {
    if (a.prof && lb == 0 && tid == 0) { const unsigned long long t = wall_clock64(); a.prof[k] += t - tPrev; tPrev = t; }
  };
  // The primal step of this thread's FIRST column: what no decision changes (cost, bounds, diagonal of Q) stays in
  // registers for the whole launch, and what the decision selects between — both parities of x and A'y, the running
  // sum — is fetched BEFORE the decision is computed (the values exist once the third barrier of the previous trial
  // has passed), so that the step behind the decision is arithmetic and one store.  Same operations, same order.
  const int j0 = lb * kSpmvThreads + tid;
  const bool own0 = j0 < a.v.n;
  const int jc = own0 ? j0 : 0;
  const double c0 = a.v.cost[jc], u0 = a.v.upper[jc], l0 = a.v.lower[jc], q0 = a.v.qdiag ? a.v.qdiag[jc] : 0.0;
  double px[2], pa[2], pxs;
  auto prefetchPrimal = [&]() {
    px[0] = ldM<LOCAL>(a.v.x[0] + jc); px[1] = ldM<LOCAL>(a.v.x[1] + jc);
    pa[0] = ldM<LOCAL>(a.v.aty[0] + jc); pa[1] = ldM<LOCAL>(a.v.aty[1] + jc);
    pxs = ldM<LOCAL>(a.v.xSum + jc);
  };
  // PINA: (x, A'y) of the columns phase A gathers, both parities — in flight across the decision like the own column's
  constexpr int kPerA = CHUNK_A / kSpmvThreads;
  double gx[2][PINA ? kPerA : 1], ga[2][PINA ? kPerA : 1];
  auto prefetchGather = [&]() {
    if (PINA && bA.have) {
#pragma unroll
      for (int k = 0; k < kPerA; ++k) {
        const int j = bA.ci[k];
        gx[0][k] = ldM<LOCAL>(a.v.x[0] + j); gx[1][k] = ldM<LOCAL>(a.v.x[1] + j);
        ga[0][k] = ldM<LOCAL>(a.v.aty[0] + j); ga[1][k] = ldM<LOCAL>(a.v.aty[1] + j);
      }
    }
  };
  prefetchPrimal();
  prefetchGather();
  for (int trial = 0; trial < a.maxTrials; ++trial) {
    if (sh.halted) break;
    const int cur = sh.cur, nxt = cur ^ 1;
    const double tau = sh.tau, sigma = sh.sigma, avgW = sh.avgW, avgWx = sh.avgWx;
    const unsigned long long e0 = 4ull * (unsigned long long)sh.nTrials + 1ull;  // (+1: the placement check used 4 nTrials + 1 of the first trial)
    // ---- P: primal step on a share of the columns ----
    if (own0) {
      const double xv = cur ? px[1] : px[0];
      if (avgWx != 0.0) stM<LOCAL>(a.v.xSum + j0, pxs + avgWx * xv);  // deferred PDHG_Update_Average (step.c:437)
      double t = xv;
      t += (-tau) * c0;
      t += tau * (cur ? pa[1] : pa[0]);
      if (a.v.qdiag) t = t / (1.0 + tau * q0);
      t = t < u0 ? t : u0;
      t = t > l0 ? t : l0;
      stM<LOCAL>(a.v.x[nxt] + j0, t);
    }
    for (int j = j0 + G * kSpmvThreads; j < a.v.n; j += G * kSpmvThreads) {
      const double xv = ldM<LOCAL>(a.v.x[cur] + j);
      if (avgWx != 0.0) stM<LOCAL>(a.v.xSum + j, ldM<LOCAL>(a.v.xSum + j) + avgWx * xv);
      double t = xv;
      t += (-tau) * a.v.cost[j];
      t += tau * ldM<LOCAL>(a.v.aty[cur] + j);
      if (a.v.qdiag) t = t / (1.0 + tau * a.v.qdiag[j]);
      const double u = a.v.upper[j], l = a.v.lower[j];
      t = t < u ? t : u;
      t = t > l ? t : l;
      stM<LOCAL>(a.v.x[nxt] + j, t);
    }
    stamp(0);
    if (!PINA) meet(e0 + 1);
    stamp(1);
    // ---- A: A x+ and the dual step ----
    double* partDY = PINA && (sh.nTrials & 1) ? a.partDY + a.nPartA : a.partDY;
    if (bA.have) {
      double acc0 = 0.0, acc1 = 0.0;
      double xIn[kPerA];
      if (PINA) {
#pragma unroll
        for (int k = 0; k < kPerA; ++k)
          xIn[k] = primalFrom(cur ? gx[1][k] : gx[0][k], cur ? ga[1][k] : ga[0][k], tau, bA.pc[k], bA.pl[k], bA.pu[k], a.v.qdiag != nullptr, bA.pq[k]);
      }
      smallSpmvBlock<CHUNK_A, true, LOCAL, PINA>(a, a.A, bA, cur, sigma, avgW, prod, acc0, acc1, xIn);
      const double t = blockSum<kSpmvThreads>(acc0, scratch[0]);
      if (tid == 0) stM<LOCAL>(partDY + bA.slot_, t);
    }
    for (int tb = lb; tb < nTBA; tb += G) {  // long rows: segment tasks
      smallLongBlock<true, LOCAL>(a, a.LA, tb, cur, sigma, avgW, scratch[0], partDY, nullptr);
      __syncthreads();
    }
    stamp(2);
    meet(e0 + 2);
    stamp(3);
    // ---- T: A'y+ with the movement / interaction partials ----
    if (bAt.have) {
      double acc0 = 0.0, acc1 = 0.0;
      smallSpmvBlock<CHUNK_AT, false, LOCAL>(a, a.At, bAt, cur, sigma, avgW, prod, acc0, acc1);
      const double t0 = blockSum<kSpmvThreads>(acc0, scratch[0]);
      const double t1 = blockSum<kSpmvThreads>(acc1, scratch[1]);
      if (tid == 0) { stM<LOCAL>(a.partDX + bAt.slot_, t0); stM<LOCAL>(a.partInter + bAt.slot_, t1); }
    }
    for (int tb = lb; tb < nTBAt; tb += G) {  // long columns: segment tasks
      smallLongBlock<false, LOCAL>(a, a.LAt, tb, cur, sigma, avgW, scratch[0], a.partDX, a.partInter);
      __syncthreads();
    }
    stamp(4);
    meet(e0 + 3);
    stamp(5);
    prefetchPrimal();  // for the next trial's primal step: in flight while the decision is computed
    prefetchGather();
    // ---- D: the decision, identical in every workgroup ----
    // (the timeout flag of the barriers: fetched next to the partials, looked at behind the decision)
    const unsigned long long timedOut = tid == 0 ? __hip_atomic_load(a.bar + G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    double dY2, dX2, inter;
    trialSumsT<LOCAL ? 2 : 1, (CHUNK_A > kChunkSmall || CHUNK_AT > kChunkSmall)>(partDY, a.nPartA, a.partDX, a.partInter, a.nPartAt, tscr, dY2, dX2, inter);
    if (tid == 0) {
      decideUpdate<true>(&sh, dX2, dY2, inter);
      if (timedOut) { sh.commError = 1; sh.halted = 1; }
    }
    __syncthreads();
    stamp(6);
    if (a.prof && lb == 0 && tid == 0) a.prof[7] += 1;
  }
  if (lb == 0 && tid < (int)(sizeof(DevState) / 4)) reinterpret_cast<uint32_t*>(a.st)[tid] = reinterpret_cast<const uint32_t*>(&sh)[tid];
}

using SmallKernel = void (*)(const SmallArgs);
template <bool PINA>
SmallKernel pickT(int chunkA, int chunkAt, int mode) {
  if (chunkA == kChunkSmall && chunkAt == kChunkSmall)
    return mode == 1 ? k_trials_small<kChunkSmall, kChunkSmall, 1, PINA> : mode == 2 ? k_trials_small<kChunkSmall, kChunkSmall, 2, PINA>
                                                                                     : k_trials_small<kChunkSmall, kChunkSmall, 0, PINA>;
  // mid-size operands (2048-entry blocks): hundreds of workgroups, always the hierarchical barrier, and the P phase stays
  // (measured with the primal step inside phase A, 100k x 100k: 26.8 -> 46.9 us per trial — eight entries per lane, five
  // agent-scope or cached loads each, 204 bytes of scratch per lane)
  if (mode != 2 || PINA) return nullptr;
  if (chunkA == kChunk && chunkAt == kChunk) return k_trials_small<kChunk, kChunk, 2, false>;
  if (chunkA == kChunk && chunkAt == kChunkSmall) return k_trials_small<kChunk, kChunkSmall, 2, false>;
  if (chunkA == kChunkSmall && chunkAt == kChunk) return k_trials_small<kChunkSmall, kChunk, 2, false>;
  return nullptr;
}
SmallKernel pick(int chunkA, int chunkAt, int mode, bool pina = false) { return pina ? pickT<true>(chunkA, chunkAt, mode) : pickT<false>(chunkA, chunkAt, mode); }

}  // namespace

// Workgroups the persistent launch would use (0: this pair of operands does not qualify) and how many the device
// keeps resident at once.
int smallTrialsGrid(const MatView& A, const MatView& At, int32_t n, int device, int* residentOut, bool primalInA) {
  *residentOut = 0;
  // (long majors ride along as segment tasks; beyond kLongSlotCap of them their contributions need the k_long_groups launch)
  if (A.useSlab || At.useSlab || A.lng.contrib != nullptr || At.lng.contrib != nullptr) return 0;
  // (the long rows' segment tasks gather x+ from memory: with them the P phase stays — standmps 11.2 -> 11.8 us per trial without it)
  if (primalInA && A.lng.nTasks > 0) return 0;
  SmallKernel k = pick(A.csr.chunk, At.csr.chunk, 2, primalInA);  // (the variants of one chunk pair differ little; mode 2 exists for all)
  if (!k || A.csr.nBlocks <= 0 || At.csr.nBlocks <= 0) return 0;
  int perCu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k, kSpmvThreads, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  // (blocks of >= 82 SGPRs: the hardware admits fewer per CU than the occupancy query says — MI355X_MICROARCH.md; stay far below)
  *residentOut = (perCu < 4 ? perCu : 4) * cus;
  int g = A.csr.nBlocks > At.csr.nBlocks ? A.csr.nBlocks : At.csr.nBlocks;
  const int gv = (n + kSpmvThreads - 1) / kSpmvThreads;
  if (gv > g) g = gv < 64 ? gv : (g > 64 ? g : 64);  // the primal step alone never asks for more than 64 workgroups
  return g;
}

void launchSmallTrials(const MatView& A, const MatView& At, const IterVecs& v, DevState* st, double* partDY, double* partDX,
                       double* partInter, unsigned long long* bar, int32_t grid, int32_t maxTrials, int mode, hipStream_t s,
                       int32_t timeoutMs, bool failRollCall, bool selfTest, unsigned long long seq, bool primalInA) {
  const bool xcdLocal = mode == 1;
  static_assert(kHierBarWords == kSmallHierWords, "barrier buffer layout");
  // mode 2: the hierarchical barrier's counters (and the roll-call word) start from zero in every launch; the other modes
  // keep their words — arrival epochs grow with the trial counter, the roll-call count with the launches (seq = 1, 2, ...
  // since the caller zeroed the buffer), so no memset launch sits between two launches of the loop
  if (mode == 2) (void)hipMemsetAsync(bar, 0, smallBarWords(grid) * sizeof(unsigned long long), s);
  SmallArgs a{};
  a.expect = (mode == 2 ? (unsigned long long)grid : seq * (unsigned long long)grid) + (failRollCall ? 1ull : 0ull);
  a.selfTest = xcdLocal && selfTest ? 1 : 0;
  a.primalInA = primalInA ? 1 : 0;
  if (a.selfTest)  // its words: the tail of the buffer
    (void)hipMemsetAsync(bar + smallBarWords(grid) - (size_t)(2 * grid + 8), 0, (size_t)(2 * grid + 8) * sizeof(unsigned long long), s);
  a.limit = (unsigned long long)(timeoutMs > 0 ? timeoutMs : 1000) * 100000ull;
  a.LA = A.lng; a.LAt = At.lng; a.nPartA = A.nPartials; a.nPartAt = At.nPartials;
  a.A = A.csr; a.At = At.csr; a.v = v; a.st = st; a.partDY = partDY; a.partDX = partDX; a.partInter = partInter; a.bar = bar;
  a.xcdA = A.xcdMap; a.xcdAt = At.xcdMap; a.maxTrials = maxTrials;
  static unsigned long long* prof = [] {  // PDLP_MI355X_SMALL_PROF=1: per-phase ticks, printed at exit (development)
    unsigned long long* p = nullptr;
    if (devEnv("PDLP_MI355X_SMALL_PROF") && hipMalloc((void**)&p, 64) == hipSuccess) {
      (void)hipMemset(p, 0, 64);
      static unsigned long long* keep = p;
      atexit([] {
        unsigned long long h[8];
        if (hipMemcpy(h, keep, 64, hipMemcpyDeviceToHost) == hipSuccess && h[7])
          fprintf(stderr, "small-LP phases, us per trial over %llu trials: P %.2f | bar %.2f | A %.2f | bar %.2f | T %.2f | bar %.2f | D %.2f\n", h[7],
                  h[0] * 0.01 / h[7], h[1] * 0.01 / h[7], h[2] * 0.01 / h[7], h[3] * 0.01 / h[7], h[4] * 0.01 / h[7], h[5] * 0.01 / h[7], h[6] * 0.01 / h[7]);
      });
    }
    return p;
  }();
  a.prof = prof;
  hipLaunchKernelGGL(pick(A.csr.chunk, At.csr.chunk, mode, primalInA), dim3(xcdLocal ? 8 * grid : grid), dim3(kSpmvThreads), 0, s, a);
}

}  // namespace pdlp
