// pdlp_devfn.hpp — device functions shared by the kernel translation units
// (pdlp_kernels.hip, pdlp_mesh.hip): deterministic wave/block sums and the
// accept/reject + step-size update of the adaptive rule.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "pdlp_detmath.h"
#include "pdlp_kernels.hpp"

namespace pdlp {
namespace {

constexpr int kWave = 64;

// Is a check iteration due?  Asked by every kernel of a device-driven check (pdlp_kernels.hpp CheckCtl): the device
// has halted, at an iteration of the reference's schedule (nIter < 10, nIter % interval == 0, the last iteration:
// cupdlp_solver.c:953-962), and the solve has not ended.  The fixed-work loop stops AT its target without a check.
__device__ __forceinline__ bool checkDue(const DevState* st, const CheckCtl* cc) {
  if (!st->halted || st->commError || cc->terminated) return false;
  const int it = st->nIter;
  if (!cc->terminate && it >= cc->iterLimit) return false;
  return it < 10 || it % cc->interval == 0 || (cc->terminate && it == cc->optIterLimit - 1);
}
__device__ __forceinline__ bool gateOpen(const CheckGate& g) {
  if (g.flag && *g.flag == 0) return false;
  return g.st == nullptr || checkDue(g.st, g.cc);
}

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

// Sum of the products q = qb .. qe-1 of one major in the padded LDS strip, left to right (the reference's order).
// The adds are a dependent chain (25fv47 has a major of 340 entries), so nothing but the adds may sit in the loop.
// Eight products occupy nine consecutive slots, one of which is a pad slot (slot(q) = q + q / 8) that holds -0.0 —
// adding it changes no sum (x + -0.0 == x for every x) — so the nine reads use immediate offsets from ONE address that
// advances by 72 bytes: no per-element index arithmetic (which cost more than the adds: phase A of the small-LP
// loop 6.3 -> 3.9 us on 25fv47).  The reads of the next window are issued before the current one is added.  The
// caller has filled the pad slots (padSlots) before the products were written.
__device__ __forceinline__ void padSlots(double* prod, int nSlots, int tid, int nThreads) {
  for (int i = 8 + 9 * tid; i < nSlots; i += 9 * nThreads) prod[i] = -0.0;
}
__device__ __forceinline__ double majorSum(const double* prod, int qb, int qe) {
  double s = 0.0;
  int q = qb;
  if (q + 8 <= qe) {
    const double* w = prod + slot(q);
    double t[9], u[9];
    auto load = [&](double (&d)[9], const double* p) {
#pragma unroll
      for (int k = 0; k < 9; ++k) d[k] = p[k];
    };
    auto add = [&](const double (&d)[9]) {
#pragma unroll
      for (int k = 0; k < 9; ++k) s += d[k];
    };
    load(t, w);
    q += 8;
    w += 9;
    while (q + 16 <= qe) {  // two windows per turn: no register rotation; the scheduler must not sink the reads to their uses
      load(u, w);
      __builtin_amdgcn_sched_barrier(0);
      add(t);
      __builtin_amdgcn_sched_barrier(0);
      load(t, w + 9);
      __builtin_amdgcn_sched_barrier(0);
      add(u);
      q += 16;
      w += 18;
    }
    if (q + 8 <= qe) {
      load(u, w);
      __builtin_amdgcn_sched_barrier(0);
      add(t);
      add(u);
      q += 8;
    } else {
      add(t);
    }
  }
  for (; q < qe; ++q) s += prod[slot(q)];
  return s;
}

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
// must cover the trial counter — Solver::refreshPowTable keeps >= 4096 entries ahead at every stop; if they ever
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

// Block-uniform read through the scalar (constant) path: s_load counts on lgkmcnt,
// so it never forces a wait on the vector-memory prefetches in flight.
template <typename T>
__device__ __forceinline__ T ldUniform(const T* p) {
  return *(const __attribute__((address_space(4))) T*)(p);
}

// XCD-aware work assignment.  Workgroup b runs on XCD b % 8 (observed dispatch order; only a speed
// assumption), and each XCD has its own L2.  Consecutive work blocks own consecutive majors, which in a
// structured LP (network blocks, staircases) touch neighbouring minors: giving XCD x the CONTIGUOUS range of
// logical blocks [x*nB/8, (x+1)*nB/8) keeps the part of the gathered vector an XCD needs at 1/8 of it instead of
// all of it (measured on the block-angular LP of bench.py --config c: A x 73 -> 50 us, L2 misses 2.5 M -> 0.9 M;
// its transpose prefers round robin, 25 vs 39 us, and a random matrix does not care), so the mapping is chosen
// per operand by timing both at setup (tuneXcdMap).  Results do not depend on it: partials are indexed by the
// logical block.
__device__ __forceinline__ int xcdContiguousBlock(int b, int nB) {
  constexpr int kXcds = 8;
  const int xcd = b % kXcds, i = b / kXcds;
  const int qlo = nB / kXcds, r = nB % kXcds;
  return xcd < r ? xcd * (qlo + 1) + i : r * (qlo + 1) + (xcd - r) * qlo + i;
}

// Agent-scope relaxed accesses (global_load/store ... sc1): write-through stores, L1-bypassing loads — the only
// way data crosses workgroups INSIDE a launch on this part (eight XCDs with private L2s, per-CU L1s that are never
// refreshed by other CUs' stores).
__device__ __forceinline__ void stAgent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ldAgent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT));
}

// Grid barrier for a grid whose blocks are ALL resident.  Called by wave 0 of every block after lane 0's stores of the
// block's published words.  Every block owns one 8-byte arrival word; arriving = storing the epoch there (after the
// published words have landed); waiting = sweeping all arrival words, nBlocks / 64 per lane, until every one carries
// at least the epoch (a fast block may already stand at the launch's next barrier).  One store and one sweep: no
// atomic round trips, no counter to reset (measured the same as the XCD-hierarchical counter barrier it replaced in
// the fused trial: 64.0 us per launch either way).  Epochs grow with the trial counter (the words are zeroed when a
// solve starts).  LOCAL: every block sits on ONE XCD (the caller has checked it): the words go through that XCD's L2
// — ordinary stores, non-temporal (L1-bypassing) loads — instead of through memory.
//   Not every block may be resident: plain launches promise nothing about co-residency, and a second tenant of the
// device (another process, another stream) can hold the CUs a block of this grid needs.  A wait that has not ended
// after limitTicks (100 MHz wall clock; Solver: PDLP_MI355X_BARRIER_TIMEOUT_MS, 1 s) gives up: the block POISONS its
// own arrival word (bit 62: still >= every epoch, so nobody waits for it any more) and raises the flag word behind
// the arrival words; every sweep that meets a poisoned word fails too — the blocks that time out, the ones that sweep
// later and the stragglers that only start once the others have left all return false, so a barrier either holds for
// every block or for none (the poison is set before any straggler arrives, i.e. before any sweep can complete; a
// sweep pass that straddles the two events by a microsecond would be the exception — the poison is sticky, so the
// NEXT barrier of any block then meets a poisoned word of an older epoch and reports kBarBroken: an error, never a
// silently wrong iterate).  On kBarFailed the caller leaves the trial undecided and the host falls back to plain
// launches (Solver::syncState).
#ifndef PDLP_BAR_SLEEP
#define PDLP_BAR_SLEEP 1
#endif
constexpr unsigned long long kBarPoison = 1ull << 62;
enum : int { kBarOk = 0, kBarFailed = 1, kBarBroken = 2 };
// gridArrive + gridWait = gridBarrier.  Apart: a block that has loads in flight which nobody else needs (operands it
// prefetches for the phase behind the barrier) arrives FIRST and issues them afterwards — the arrival waits for vmcnt(0),
// and with the prefetches in front of it that wait was a whole HBM round trip under load (round 6: the fused trial released
// its barrier 2-5 us after the last block's epilogue).
template <bool LOCAL = false>
__device__ __forceinline__ void gridArrive(unsigned long long* bar, int blk, unsigned long long epoch, int lane) {
  if (lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's published words have landed
    if (LOCAL) *reinterpret_cast<volatile unsigned long long*>(bar + blk) = epoch;
    else __hip_atomic_store(bar + blk, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <bool LOCAL = false>
__device__ __forceinline__ int gridWait(unsigned long long* bar, int blk, int nBlocks, unsigned long long epoch, int lane,
                                         unsigned long long limitTicks) {
  auto ld = [&](const unsigned long long* p) -> unsigned long long {
    if (!LOCAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  };
  auto st = [&](unsigned long long* p, unsigned long long v) {
    if (LOCAL) *reinterpret_cast<volatile unsigned long long*>(p) = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const unsigned long long t0 = wall_clock64();
  // (Round 6 also tried loading all words of a sweep back to back — eight at 512 blocks — instead of this loop: the release
  // came no earlier, and the sixteen registers of the word array spilled in the fused kernel.  What the release waits for
  // is the arrival, see gridArrive.)
  for (uint32_t spins = 0;; ++spins) {
    bool ok = true, bad = false, stale = false;
    for (int i = lane; i < nBlocks; i += kWave) {
      const unsigned long long v = ld(bar + i);
      ok = ok && v >= epoch;
      bad = bad || (v & kBarPoison) != 0;
      stale = stale || ((v & kBarPoison) != 0 && (v & ~kBarPoison) < epoch);
    }
    if (__any(stale)) return kBarBroken;
    if (__any(bad)) return kBarFailed;
    if (__all(ok)) return kBarOk;
    __builtin_amdgcn_s_sleep(PDLP_BAR_SLEEP);
    if ((spins & 31u) == 31u && wall_clock64() - t0 > limitTicks) {
      if (lane == 0) {
        st(bar + blk, epoch | kBarPoison);
        __hip_atomic_store(bar + nBlocks, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return kBarFailed;
    }
  }
}

template <bool LOCAL = false>
__device__ __forceinline__ int gridBarrier(unsigned long long* bar, int blk, int nBlocks, unsigned long long epoch, int lane,
                                            unsigned long long limitTicks) {
  gridArrive<LOCAL>(bar, blk, epoch, lane);
  return gridWait<LOCAL>(bar, blk, nBlocks, epoch, lane, limitTicks);
}

// Roll call at the START of a launch whose workgroups will meet at grid barriers: before it has written anything,
// every workgroup adds itself to one counter word (zeroed by the host in front of the launch) and waits until all G
// have.  If they do not within limitTicks — the device is shared and not all of them fit next to the other tenant's
// work — the launch must change NOTHING and say so, and that verdict has to be the same in every workgroup, also in
// one that only starts after the others have given up.  Hence one word decides: a workgroup that gives up sets the
// poison bit with a compare-and-swap that fails once the count is complete; an arrival that finds the poison bit (its
// own fetch-add returns it) leaves at once; a waiter that reads the complete count without the bit has won for good —
// the count never drops and the bit can no longer be set.  Called by wave 0; the verdict is returned in every lane.
__device__ __forceinline__ bool rollCall(unsigned long long* word, unsigned long long G, unsigned long long limitTicks, int lane) {
  // (the count is cumulative over the launches of a solve: 62 bits of it, next to the poison bit)
  int verdict = 0;  // 1: everybody is here, 2: not this time
  if (lane == 0) {
    unsigned long long v = __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t spins = 0;; ++spins) {
      if (v & kBarPoison) { verdict = 2; break; }
      if ((v & (kBarPoison - 1ull)) >= G) { verdict = 1; break; }
      __builtin_amdgcn_s_sleep(2);
      if ((spins & 15u) == 15u && wall_clock64() - t0 > limitTicks) {
        unsigned long long cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
          if (cur & kBarPoison) { verdict = 2; break; }
          if ((cur & (kBarPoison - 1ull)) >= G) { verdict = 1; break; }
          if (__hip_atomic_compare_exchange_strong(word, &cur, cur | kBarPoison, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            verdict = 2;
            break;
          }
        }
        break;
      }
      v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return __shfl(verdict, 0) == 1;
}

// XCD-hierarchical grid barrier for HUNDREDS of resident blocks on all XCDs.  With the sweep above every block polls
// every arrival word through memory: at 490 blocks that is 490 x 31 line requests per round onto the few channels
// that hold the words, and a barrier costs 4.8 us (100k x 100k LP, measured back to back).  Here the blocks of one
// XCD meet at a counter of their own first (they are grouped by the XCC id they read from the hardware register, so
// no placement is assumed; one sweep barrier at the start of a launch makes the number of blocks per XCD known):
// the last one to arrive speaks for the XCD — it publishes the barrier index in the XCD's word (agent scope), waits
// for the words of all XCDs that have blocks, then releases its XCD through a word in that XCD's L2 (ordinary store:
// the L1 is write-through) which the others poll with L1-bypassing loads.  Memory sees 16 pollers instead of 490:
// 2.5-3.6 us per barrier including the drain of the phase's stores.  (Measured alternative: per-block arrival words
// in the L2 swept by a fixed leader instead of the counter — 3.7-4.2 us, one more polling stage.)  The words are
// zeroed by the host before every launch; k = 1, 2, ... counts the barriers of the launch.  Data published before
// the barrier must have been stored with agent scope and drained (s_waitcnt vmcnt(0)) by every wave before the
// block arrives — as for the sweep.
constexpr int kXccSlots = 16;   // XCC_ID has four bits
constexpr int kXccStride = 32;  // words between two XCDs' words: 256 bytes
struct HierBar {
  unsigned long long limit;  // 100 MHz ticks a wait may last
  unsigned long long* base;  // 4 arrays of kXccSlots * kXccStride words: registration counts, arrival counters, L2 release words, agent-scope XCD words
  unsigned long long* flag;  // timeout flag
  int xcc, nLocal;
  uint32_t active;  // bit x: XCD x has blocks of this launch
  __device__ __forceinline__ unsigned long long* reg(int x) const { return base + x * kXccStride; }
  __device__ __forceinline__ unsigned long long* cnt(int x) const { return base + (kXccSlots + x) * kXccStride; }
  __device__ __forceinline__ unsigned long long* rel(int x) const { return base + (2 * kXccSlots + x) * kXccStride; }
  __device__ __forceinline__ unsigned long long* glob(int x) const { return base + (3 * kXccSlots + x) * kXccStride; }
};
constexpr int kHierBarWords = 4 * kXccSlots * kXccStride;
__device__ __forceinline__ int xccId() {
  int id;
  asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(id));  // HW_REG_XCC_ID[3:0]
  return id;
}
// called by wave 0 of every block, after the block's stores have drained and its waves have met
__device__ __forceinline__ void hierBarrier(const HierBar& h, unsigned long long k, int lane) {
  auto ldL2 = [&](const unsigned long long* p) -> unsigned long long {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  };
  auto giveUp = [&]() {
    if (lane == 0) __hip_atomic_store(h.flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  unsigned long long old = 0;
  const unsigned long long t0 = wall_clock64();
  if (lane == 0) old = __hip_atomic_fetch_add(h.cnt(h.xcc), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __shfl(old, 0);
  if (old + 1 == k * (unsigned long long)h.nLocal) {  // the last block of this XCD
    if (lane == 0) __hip_atomic_store(h.glob(h.xcc), k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t spins = 0;; ++spins) {
      bool ok = true;
      if (lane < kXccSlots && ((h.active >> lane) & 1u)) ok = __hip_atomic_load(h.glob(lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= k;
      if (__all(ok)) break;
      if ((spins & 63u) == 63u && wall_clock64() - t0 > h.limit) { giveUp(); break; }
    }
    if (lane == 0) *reinterpret_cast<volatile unsigned long long*>(h.rel(h.xcc)) = k;
  } else {
    for (uint32_t spins = 0;; ++spins) {
      if (ldL2(h.rel(h.xcc)) >= k) break;
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 63u) == 63u && wall_clock64() - t0 > h.limit) { giveUp(); break; }
    }
  }
}

// Epilogue operands that do not depend on the SpMV result.
struct Pre { double a, b, c, d, e; };

// Fixed-order sums of the three per-block partial arrays by 256 threads: lane t sums elements t, t+256, ...
// (4 independent chains), then wave shuffle tree, then the 4 wave results in order.  Results valid in thread 0.
// Shared by k_decide, k_decide_primal and the fused trial kernel, so all take identical decisions.  AGENT = 1: the
// partials were written by other workgroups of the SAME launch (agent-scope loads); 2: the same inside ONE XCD (the
// persistent loop's XCD-local mode: non-temporal loads, served by the shared L2); threads >= 256 of a larger
// block only take part in the barrier.
// WIDE (the persistent loop of mid-size LPs, up to 1024 partials per array): up to four partials per lane and array,
// again fetched together; the additions are laneSum's for that count.
template <int AGENT, bool WIDE = false>
__device__ __forceinline__ void trialSumsT(const double* __restrict__ partDY, int nDY, const double* __restrict__ partDX,
                                           const double* __restrict__ partInter, int nDX, double (*scratch)[kVecThreads / kWave],
                                           double& dY2, double& dX2, double& inter, const double* __restrict__ partQ = nullptr,
                                           int nQ = 0, double* qint = nullptr) {
  const int tid = threadIdx.x;
  auto ld = [&](const double* q) { return AGENT == 2 ? ldStream(q) : AGENT == 1 ? ldAgent(q) : *q; };
  auto laneSum = [&](const double* __restrict__ p, int count) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = tid;
    for (; i + 3 * kVecThreads < count; i += 4 * kVecThreads) {
      const double a0 = ld(p + i), a1 = ld(p + i + kVecThreads), a2 = ld(p + i + 2 * kVecThreads), a3 = ld(p + i + 3 * kVecThreads);
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; i < count; i += kVecThreads) s0 += ld(p + i);
    return (s0 + s1) + (s2 + s3);
  };
  if (tid < kVecThreads) {
    double vY, vX, vI, vQ;
    if (nDY <= kVecThreads && nDX <= kVecThreads && nQ <= kVecThreads) {
      // at most one partial per lane (up to 256 work blocks per operand — the 1M x 1M slab launches and every Netlib-class
      // LP): the loads of all arrays leave together, ONE memory round trip instead of one per array; the additions are
      // those of laneSum for a single element
      const double aY = partDY && tid < nDY ? ld(partDY + tid) : 0.0;
      const double aX = tid < nDX ? ld(partDX + tid) : 0.0;
      const double aI = tid < nDX ? ld(partInter + tid) : 0.0;
      const double aQ = partQ && tid < nQ ? ld(partQ + tid) : 0.0;
      auto one = [](double a, bool have) { double s0 = 0.0; if (have) s0 += a; return (s0 + 0.0) + (0.0 + 0.0); };
      vY = one(aY, partDY && tid < nDY); vX = one(aX, tid < nDX); vI = one(aI, tid < nDX); vQ = one(aQ, partQ && tid < nQ);
    } else if (WIDE && !partQ && nDY <= 4 * kVecThreads && nDX <= 4 * kVecThreads) {
      double eY[4], eX[4], eI[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = tid + k * kVecThreads;
        eY[k] = partDY && i < nDY ? ld(partDY + i) : 0.0;
        eX[k] = i < nDX ? ld(partDX + i) : 0.0;
        eI[k] = i < nDX ? ld(partInter + i) : 0.0;
      }
      auto four = [&](const double* e, int count) {  // laneSum for count <= 1024: one full round of four chains, or a tail on chain 0
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (tid + 3 * kVecThreads < count) {
          s0 += e[0]; s1 += e[1]; s2 += e[2]; s3 += e[3];
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (tid + k * kVecThreads < count) s0 += e[k];
        }
        return (s0 + s1) + (s2 + s3);
      };
      vY = partDY ? four(eY, nDY) : 0.0;
      vX = four(eX, nDX);
      vI = four(eI, nDX);
      vQ = 0.0;
    } else {
      vY = partDY ? laneSum(partDY, nDY) : 0.0;
      vX = laneSum(partDX, nDX);
      vI = laneSum(partInter, nDX);
      vQ = partQ ? laneSum(partQ, nQ) : 0.0;  // (QP with off-diagonal Hessian entries: dx . N dx)
    }
    vY = waveSum(vY); vX = waveSum(vX); vI = waveSum(vI);
    if (partQ) vQ = waveSum(vQ);
    const int lane = tid & (kWave - 1), w = tid / kWave;
    if (lane == 0) { scratch[0][w] = vY; scratch[1][w] = vX; scratch[2][w] = vI; scratch[3][w] = vQ; }
  }
  __syncthreads();
  dY2 = dX2 = inter = 0.0;
  if (tid == 0) {
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < kVecThreads / kWave; ++i) { dY2 += scratch[0][i]; dX2 += scratch[1][i]; inter += scratch[2][i]; q += scratch[3][i]; }
    if (qint) *qint = q;
  }
}
__device__ __forceinline__ void trialSums(const double* __restrict__ partDY, int nDY, const double* __restrict__ partDX,
                                          const double* __restrict__ partInter, int nDX, double (*scratch)[kVecThreads / kWave],
                                          double& dY2, double& dX2, double& inter, const double* __restrict__ partQ = nullptr,
                                          int nQ = 0, double* qint = nullptr) {
  trialSumsT<0>(partDY, nDY, partDX, partInter, nDX, scratch, dY2, dX2, inter, partQ, nQ, qint);
}

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
