// pdlp_mesh.hip — see pdlp_mesh.hpp.  Kernels that talk to the peers' arenas over
// xGMI and the host-side IPC rendezvous.
//
// Memory-model notes (gfx950, HSA): the arenas are uncached (or fine-grained) device memory — the kinds
// that are coherent between agents inside a kernel — and every access to an arena is a SYSTEM-SCOPE relaxed atomic
// (`global_store/load ... sc0 sc1`: write-through / read-through, no reliance on any cache state).
// Publishing: each wave drains its stores (`s_waitcnt vmcnt(0)`: the writes have landed), the block
// barrier collects the waves, one lane takes a ticket, and the block that takes the last ticket stores the
// epoch flags — ordering by COMPLETION, so no L2 write-back fence (measured 5-8 us per kernel with
// release/acquire fences, MI355X_MICROARCH "publish-large": write-through wins).  Consuming: wave 0 polls
// the flags (one lane per peer), block barrier, then system-scope loads of the payload.  Buffers are
// never re-written before every reader has signalled a later phase (hazard analysis in DESIGN.md §6), so
// there is no double buffering.
#include "pdlp_mesh.hpp"

#include <cstddef>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

#include "pdlp_devfn.hpp"
#include "pdlp_device.hpp"

// The direct exchange maps the arenas of ranks in OTHER processes through HIP IPC, which on dmabuf-only hosts needs
// HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment when the HSA runtime starts.  That is the LAUNCHER's business (bench.py
// and the test launchers set it; the hipIpcGetMemHandle error message names it): a library must not change the
// environment of the process that loads it.  Ranks inside one process (pdlp_mi355x_solve with num_devices > 1, what
// Highs::run() reaches) map each other through peer access and need nothing.
namespace pdlp {

namespace {

constexpr int kFlagStride = 128;                    // bytes between two flags (own cache line each)

__device__ __forceinline__ long long* flagAt(const MeshArgs& ma, int rank, int kind, int src) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  return (long long*)(mv->arena[rank] + mv->offFlags + ((size_t)kind * kMeshMaxRanks + src) * kFlagStride);
}
__device__ __forceinline__ double* recvX(const MeshArgs& ma, int rank) {
  const MeshView* __restrict__ mv = ma.v; (void)mv; return (double*)(mv->arena[rank] + mv->offRecvX); }
__device__ __forceinline__ double* recvP(const MeshArgs& ma, int rank, int src) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  return (double*)(mv->arena[rank] + mv->offRecvP) + (size_t)src * mv->sliceMax;
}
__device__ __forceinline__ double* recvY(const MeshArgs& ma, int rank) {
  const MeshView* __restrict__ mv = ma.v; (void)mv; return (double*)(mv->arena[rank] + mv->offRecvY); }
__device__ __forceinline__ double* mailAt(const MeshArgs& ma, int rank, bool hot, int src) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  return (double*)(mv->arena[rank] + (hot ? mv->offMailHot : mv->offMailGen)) + (size_t)src * kMeshMailDoubles;
}

__device__ __forceinline__ void sysStore(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sysLoad(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void drainStores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One thread: publish epoch e of `kind` to every peer.  The caller guarantees that the payload stores
// (write-through, system scope) of every contributing wave have COMPLETED (drainStores + barrier + ticket).
__device__ void signalPeers(const MeshArgs& ma, int kind, long long e) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  const int G = ma.G, g = ma.g;
  // Optional belt and braces (PDLP_MI355X_MESH_FENCES=1, off by default: +5..8 us per exchange kernel): a
  // system-scope release fence in front of the flag stores, for a machine on which "the write-through payload
  // stores have been acknowledged" should turn out not to order them before the flag at the peer.
  if (ma.fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  for (int h = 0; h < G; ++h) {
    if (h == g) continue;
    __hip_atomic_store(flagAt(ma, h, kind, g), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Whole block: wait until every peer has published epoch >= e of `kind`.  Wave 0 spins (one lane
// per peer); the payload is then read with system-scope loads (sysLoad), which need no invalidate.
// Returns false on timeout (or if another kernel already failed); the caller bails out.
__device__ bool waitPeers(const MeshArgs& ma, int kind, long long e) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  const long long tEnter = (threadIdx.x == 0 && kind < 3) ? wall_clock64() : 0;
  __syncthreads();
  if (threadIdx.x < kWave) {
    const int h = threadIdx.x, G = ma.G, g = ma.g;
    if (h < G && h != g) {
      const long long* f = flagAt(ma, g, kind, h);
      const long long t0 = wall_clock64(), budget = ma.waitTicks;
      int polls = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
        if ((++polls & 255) == 0) {
          if (wall_clock64() - t0 > budget ||
              __hip_atomic_load(&ma.ms->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            ok = 0;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
  const bool good = ok != 0;
  if (ma.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // (payload reads are system-scope loads anyway)
  if (threadIdx.x == 0 && blockIdx.x == 0 && kind < 3) {  // hot-loop exchanges: how long (the first block of) this kernel waited
    atomicAdd(&ma.ms->waitTicks[kind], (unsigned long long)(wall_clock64() - tEnter));
    atomicAdd(&ma.ms->waitCount[kind], 1ull);
  }
  __syncthreads();
  return good;
}

__device__ void fail(const MeshArgs& ma, DevState* st) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&ma.ms->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st) { st->halted = 1; st->commError = 1; }
  }
}

// Multi-block producer: every wave drains its stores, the block barrier collects them, one lane takes a
// ticket; the block that takes the last ticket publishes the epoch.
__device__ void lastBlockSignal(const MeshArgs& ma, int kind, long long e, int ticket) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  drainStores();  // this wave's write-through stores have landed in the peers' arenas
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev =
        __hip_atomic_fetch_add(&ma.ms->counter[ticket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {  // every other block drained its stores before it took its ticket
      __hip_atomic_store(&ma.ms->counter[ticket], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      signalPeers(ma, kind, e);
    }
  }
}

__device__ __forceinline__ bool dead(const MeshArgs& ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  return __hip_atomic_load(&ma.ms->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// Peer base pointers in registers (statically indexed, fully unrolled: no scratch).
struct PeerPtrs { double* p[kMeshMaxRanks]; };
__device__ __forceinline__ void peerRecvX(const MeshArgs& ma, PeerPtrs& P) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  const int G = ma.G, g = ma.g;
#pragma unroll
  for (int h = 0; h < kMeshMaxRanks; ++h) P.p[h] = (h < G && h != g) ? recvX(ma, h) : nullptr;
}

// ONE block per rank ever spins on the peers' flags (a spinning grid could starve the very kernels it waits
// for when several ranks share a device, as in the single-GPU tests): the wait is its own tiny kernel,
// and the kernel boundary orders it before the consumers of the payload.
// st != nullptr: hot loop (epoch from seq); else a generic collective with the host-counted epoch eGen.
__global__ __launch_bounds__(kWave) void k_mesh_wait(DevState* st, const MeshArgs ma, int kind, long long eGen) {
  if ((st && st->halted) || dead(ma)) return;
  const long long e = (st || eGen < 0) ? ma.ms->seq + 1 : eGen;  // eGen < 0: hot loop without a DevState (HiPDLP)
  if (!waitPeers(ma, kind, e)) fail(ma, st);
}

// ---- hot loop ------------------------------------------------------------------------
// x+ = clamp(x - tau (c - A'y), l, u) on the own column slice (cupdlp_step.c:16-40), stored
// locally and pushed into every peer's recvX.
__device__ __forceinline__ void primalStepAndPush(const IterVecs& v, const DevState* st, const MeshArgs& ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  const int cur = st->cur, nxt = cur ^ 1;
  const double tau = st->tau, avgW = st->avgWx;
  const double* __restrict__ x = v.x[cur];
  const double* __restrict__ aty = v.aty[cur];
  double* __restrict__ xn = v.x[nxt];
  const int c0 = mv->colOff[ma.g];
  PeerPtrs peer;
  peerRecvX(ma, peer);
  const int stride = gridDim.x * blockDim.x;
  const int last = v.n - 1;
  // four independent elements per pass (clamped, unconditional loads: all 20 loads in flight)
  for (int j0 = blockIdx.x * blockDim.x + threadIdx.x; j0 < v.n; j0 += 4 * stride) {
    double xv[4], cv[4], av[4], uv[4], lv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = min(j0 + q * stride, last);
      xv[q] = x[j]; cv[q] = v.cost[j]; av[q] = aty[j]; uv[q] = v.upper[j]; lv[q] = v.lower[j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + q * stride;
      if (j > last) break;
      if (avgW != 0.0) v.xSum[j] += avgW * xv[q];  // deferred PDHG_Update_Average (step.c:437)
      double t = xv[q];
      t += (-tau) * cv[q];
      t += tau * av[q];
      if (v.qdiag) t = t / (1.0 + tau * v.qdiag[j]);  // QP prox step (diagonal Q)
      t = t < uv[q] ? t : uv[q];
      t = t > lv[q] ? t : lv[q];
      xn[j] = t;
#pragma unroll
      for (int h = 0; h < kMeshMaxRanks; ++h)
        if (peer.p[h]) sysStore(peer.p[h] + c0 + j, t);
    }
  }
}

// The other ranks' pieces of an all-gathered vector: receive area (uncached) -> dst (ordinary memory, so that the
// SpMV gathers hit L2).  [lo, hi) is the own piece, already in place.
__device__ __forceinline__ void copyOthers(const double* __restrict__ src, double* __restrict__ dst, int lo, int hi, int len) {
  const int stride = gridDim.x * blockDim.x;
  const int other = len - (hi - lo), lastQ = other - 1;
  for (int q0 = blockIdx.x * blockDim.x + threadIdx.x; q0 < other; q0 += 4 * stride) {
    double t[4];
    int jj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // four independent system-scope loads in flight
      const int q = min(q0 + u * stride, lastQ);
      jj[u] = q < lo ? q : q + (hi - lo);
      t[u] = sysLoad(src + jj[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q0 + u * stride <= lastQ) dst[jj[u]] = t[u];
  }
}

__global__ __launch_bounds__(kVecThreads) void k_mesh_primal_step(const IterVecs v, const DevState* st,
                                                                  const MeshArgs ma) {
  if (st->halted || dead(ma)) return;
  const long long e = ma.ms->seq + 1;
  primalStepAndPush(v, st, ma);
  lastBlockSignal(ma, kFlagX, e, 0);
}

// x+ of the other column slices: recvX -> x[nxt].
__global__ __launch_bounds__(kVecThreads) void k_mesh_wait_copy_x(const IterVecs v, DevState* st,
                                                                  const MeshArgs ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (st->halted || dead(ma)) return;  // (k_mesh_wait ran before: flag X has arrived, or the exchange is dead)
  if (ma.fusedWait && !waitPeers(ma, kFlagX, ma.ms->seq + 1)) { fail(ma, st); return; }  // ... or every block waits itself
  copyOthers(recvX(ma, ma.g), v.x[st->cur ^ 1], mv->colOff[ma.g], mv->colOff[ma.g + 1], v.n);
}

// Round 6, every rank on a GPU of its own (MeshArgs::fusedWait == 2): the whole X exchange in ONE launch — primal step
// on the own slice pushed to the peers, the block that drains last publishes the epoch, then every block waits for the
// peers' epochs and copies its share of their slices.  vc = the own column slice, xFull = the two full-length iterates.
// No block waits for anything before ITS pushes are out and a grid of at most 256 workgroups of 256 threads is resident
// at once on a device of its own, so the waits cannot keep a producer off the CUs.
__global__ __launch_bounds__(kVecThreads) void k_mesh_primal_x(const IterVecs vc, double* __restrict__ x0Full,
                                                               double* __restrict__ x1Full, int nFull, DevState* st,
                                                               const MeshArgs ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (st->halted || dead(ma)) return;
  const long long e = ma.ms->seq + 1;
  primalStepAndPush(vc, st, ma);
  lastBlockSignal(ma, kFlagX, e, 0);
  if (!waitPeers(ma, kFlagX, e)) { fail(ma, st); return; }
  copyOthers(recvX(ma, ma.g), (st->cur ^ 1) ? x1Full : x0Full, mv->colOff[ma.g], mv->colOff[ma.g + 1], nFull);
}

// ---- "two all-gathers" layout: all-gather of y+ (rows) --------------------------------------------------
// y+[r0:r1) (just written by the dual-step epilogue of A_g x+) -> every peer's recvY; flag P.
__device__ __forceinline__ void pushRows(const double* __restrict__ yn, const MeshArgs& ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  const int r0 = mv->rowOff[ma.g], r1 = mv->rowOff[ma.g + 1];
  const int G = ma.G, g = ma.g;
  PeerPtrs peer;
#pragma unroll
  for (int h = 0; h < kMeshMaxRanks; ++h) peer.p[h] = (h < G && h != g) ? recvY(ma, h) : nullptr;
  const int stride = gridDim.x * blockDim.x, len = r1 - r0;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < len; i0 += 4 * stride) {
    double t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = yn[r0 + min(i0 + u * stride, len - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * stride;
      if (i >= len) break;
#pragma unroll
      for (int h = 0; h < kMeshMaxRanks; ++h)
        if (peer.p[h]) sysStore(peer.p[h] + r0 + i, t[u]);
    }
  }
}
__global__ __launch_bounds__(kVecThreads) void k_mesh_push_y(const double* __restrict__ y0, const double* __restrict__ y1,
                                                             const DevState* st, const MeshArgs ma) {
  if (st->halted || dead(ma)) return;
  const long long e = ma.ms->seq + 1;
  pushRows((st->cur ^ 1) ? y1 : y0, ma);
  lastBlockSignal(ma, kFlagP, e, 1);
}
// y+ of the other row blocks: recvY -> y[nxt] (ordinary memory, so that the A'y gathers hit L2).
__global__ __launch_bounds__(kVecThreads) void k_mesh_wait_copy_y(double* __restrict__ y0, double* __restrict__ y1, int m,
                                                                  const DevState* st, const MeshArgs ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (st->halted || dead(ma)) return;  // (k_mesh_wait ran before: flag P has arrived, or the exchange is dead)
  if (ma.fusedWait && !waitPeers(ma, kFlagP, ma.ms->seq + 1)) { fail(ma, const_cast<DevState*>(st)); return; }
  copyOthers(recvY(ma, ma.g), (st->cur ^ 1) ? y1 : y0, mv->rowOff[ma.g], mv->rowOff[ma.g + 1], m);
}
// The whole Y exchange in one launch (fusedWait == 2, see k_mesh_primal_x).
__global__ __launch_bounds__(kVecThreads) void k_mesh_y(double* __restrict__ y0, double* __restrict__ y1, int m, DevState* st,
                                                        const MeshArgs ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (st->halted || dead(ma)) return;
  const long long e = ma.ms->seq + 1;
  double* __restrict__ yn = (st->cur ^ 1) ? y1 : y0;
  pushRows(yn, ma);
  lastBlockSignal(ma, kFlagP, e, 1);
  if (!waitPeers(ma, kFlagP, e)) { fail(ma, st); return; }
  copyOthers(recvY(ma, ma.g), yn, mv->rowOff[ma.g], mv->rowOff[ma.g + 1], m);
}

// partial[slice of owner h] -> h's recvP[g], one short coalesced loop per peer.  st != nullptr:
// hot loop (epoch from seq, flag P), else generic (epoch eGen, flag Gen).
__global__ __launch_bounds__(kVecThreads) void k_mesh_push_partial(const double* __restrict__ partial,
                                                                   const DevState* st, const MeshArgs ma,
                                                                   long long eGen) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if ((st && st->halted) || dead(ma)) return;
  const bool hot = st || eGen < 0;
  const long long e = hot ? ma.ms->seq + 1 : eGen;
  const int G = ma.G, g = ma.g;
  const int stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
  for (int h = 0; h < G; ++h) {
    if (h == g) continue;
    const int lo = mv->colOff[h], len = mv->colOff[h + 1] - lo;
    double* __restrict__ dst = recvP(ma, h, g);
    const double* __restrict__ src = partial + lo;
    for (int j0 = first; j0 < len; j0 += 4 * stride) {
      double t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = src[min(j0 + u * stride, len - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u * stride < len) sysStore(dst + j0 + u * stride, t[u]);
    }
  }
  lastBlockSignal(ma, hot ? kFlagP : kFlagGen, e, 1);
}

// The G contributions to the own slice, in rank order: src[h] = recvP[h] (or the own partial).
__device__ __forceinline__ void reduceSources(const MeshArgs& ma, const double* ownPartialSlice,
                                              PeerPtrs& S) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  const int G = ma.G, g = ma.g;
#pragma unroll
  for (int h = 0; h < kMeshMaxRanks; ++h)
    S.p[h] = h >= G ? nullptr : (h == g ? const_cast<double*>(ownPartialSlice) : recvP(ma, g, h));
}
__device__ __forceinline__ double orderedSum(const PeerPtrs& S, int j) {
  double s = 0.0;
#pragma unroll
  for (int h = 0; h < kMeshMaxRanks; ++h)
    if (S.p[h]) s += sysLoad(S.p[h] + j);
  return s;
}

// aty+[slice] = sum_h partial_h[slice]; movement / interaction partials of the slice
// (cupdlp_linalg.c:772-801).
__global__ __launch_bounds__(kVecThreads) void k_mesh_reduce_interact(const IterVecs v, DevState* st,
                                                                      const MeshArgs ma,
                                                                      const double* __restrict__ partial,
                                                                      double* partDX, double* partInter) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (st->halted || dead(ma)) return;  // (k_mesh_wait ran before: flag P has arrived)
  __shared__ double scratch[2][kVecThreads / kWave];
  const int cur = st->cur, nxt = cur ^ 1;
  PeerPtrs src;
  reduceSources(ma, partial + mv->colOff[ma.g], src);
  const double* __restrict__ xc = v.x[cur];
  const double* __restrict__ xn = v.x[nxt];
  const double* __restrict__ ac = v.aty[cur];
  double* __restrict__ an = v.aty[nxt];
  double a0 = 0.0, a1 = 0.0;
  const int stride = gridDim.x * blockDim.x;
  const int last = v.n - 1;
  for (int j0 = blockIdx.x * blockDim.x + threadIdx.x; j0 < v.n; j0 += 4 * stride) {
    double sv[4], dxv[4], acv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = min(j0 + q * stride, last);
      sv[q] = orderedSum(src, j);
      dxv[q] = xc[j] - xn[j];
      acv[q] = ac[j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // accumulation order: ascending j within the lane
      const int j = j0 + q * stride;
      if (j > last) break;
      const double da = acv[q] - sv[q];
      an[j] = sv[q];
      a0 += dxv[q] * dxv[q];
      a1 += dxv[q] * da;
    }
  }
  const double t0 = blockSum<kVecThreads>(a0, scratch[0]);
  const double t1 = blockSum<kVecThreads>(a1, scratch[1]);
  if (threadIdx.x == 0) { partDX[blockIdx.x] = t0; partInter[blockIdx.x] = t1; }
}

// One block: local sums of the three partial arrays -> every rank's mailbox -> rank-ordered
// totals -> the accept/reject decision (identical bits, hence identical decisions, everywhere).
__global__ __launch_bounds__(kVecThreads) void k_mesh_decide(DevState* st, const MeshArgs ma,
                                                             const double* __restrict__ partDY, int nDY,
                                                             const double* __restrict__ partDX,
                                                             const double* __restrict__ partInter, int nDX) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (st->halted) return;
  if (dead(ma)) { fail(ma, st); return; }  // lets the host loop stop
  const long long e = ma.ms->seq + 1;
  __shared__ double scratch[3][kVecThreads / kWave];
  const int tid = threadIdx.x;
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
  double vY = laneSum(partDY, nDY), vX = laneSum(partDX, nDX), vI = laneSum(partInter, nDX);
  vY = waveSum(vY); vX = waveSum(vX); vI = waveSum(vI);
  const int lane = tid & (kWave - 1), w = tid / kWave;
  if (lane == 0) { scratch[0][w] = vY; scratch[1][w] = vX; scratch[2][w] = vI; }
  __syncthreads();
  if (tid == 0) {
    double dY2 = 0.0, dX2 = 0.0, inter = 0.0;
#pragma unroll
    for (int i = 0; i < kVecThreads / kWave; ++i) { dY2 += scratch[0][i]; dX2 += scratch[1][i]; inter += scratch[2][i]; }
    for (int h = 0; h < ma.G; ++h) {
      double* box = mailAt(ma, h, true, ma.g);
      sysStore(box + 0, dX2); sysStore(box + 1, dY2); sysStore(box + 2, inter);
    }
    drainStores();
    signalPeers(ma, kFlagS, e);
  }
  if (!waitPeers(ma, kFlagS, e)) { fail(ma, st); return; }
  if (tid != 0) return;
  double dX2 = 0.0, dY2 = 0.0, inter = 0.0;
  for (int h = 0; h < ma.G; ++h) {
    const double* box = mailAt(ma, ma.g, true, h);
    dX2 += sysLoad(box + 0); dY2 += sysLoad(box + 1); inter += sysLoad(box + 2);
  }
  decideUpdate(st, dX2, dY2, inter);
  ma.ms->seq = e;
}

// ---- HiPDLP step, sharded (pdlp_halpern.cpp): P exchange -> primal side on the column slice -> X exchange
// of the REFLECTED x -> dual side on the local rows.  No scalar phase: step sizes only change at restarts.
// h holds column-sliced pointers (offset by c0) except rx, which is the full-length reflected vector.
__global__ __launch_bounds__(kVecThreads) void k_mesh_h_reduce_primal(const HalpernVecs h, int nLoc,
                                                                      const double* __restrict__ partial,
                                                                      const MeshArgs ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (dead(ma)) return;
  const long long e = ma.ms->seq + 1;
  const HalpernState hs = *h.hs;
  const int k = hs.hIter + h.kOff;
  const double w = (double)k / ((double)k + 1.0);
  const int c0 = mv->colOff[ma.g];
  PeerPtrs src, peer;
  reduceSources(ma, partial + c0, src);
  peerRecvX(ma, peer);
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nLoc; j += stride) {
    const double s = orderedSum(src, j);  // (A'y)_j, contributions in rank order
    const Pre p{h.xc[j], h.cost[j], h.xa[j], h.lower[j], h.upper[j]};
    halpernPrimal(h, j, s, p, hs.tau, hs.rho, w);  // writes xc, rx (slice positions), xn/slack on major steps
    const double rx = h.rx[j];
#pragma unroll
    for (int q = 0; q < kMeshMaxRanks; ++q)
      if (peer.p[q]) sysStore(peer.p[q] + c0 + j, rx);
  }
  lastBlockSignal(ma, kFlagX, e, 0);
}
// reflected x of the other column slices: recvX -> rx
__global__ __launch_bounds__(kVecThreads) void k_mesh_h_copy_x(double* __restrict__ rx, int n, const MeshArgs ma) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (dead(ma)) return;
  copyOthers(recvX(ma, ma.g), rx, mv->colOff[ma.g], mv->colOff[ma.g + 1], n);
}
__global__ void k_mesh_bump(const MeshArgs ma) {
  if (dead(ma)) return;
  ma.ms->seq = ma.ms->seq + 1;
}

// ---- generic collectives (host-counted epochs, off the hot path) ---------------------------
__device__ __forceinline__ void pushSlice(const double* __restrict__ vec, int lo, int hi, const MeshArgs& ma) {
  PeerPtrs peer;
  peerRecvX(ma, peer);
  const int stride = gridDim.x * blockDim.x;
  for (int j = lo + blockIdx.x * blockDim.x + threadIdx.x; j < hi; j += stride) {
    const double t = vec[j];
#pragma unroll
    for (int h = 0; h < kMeshMaxRanks; ++h)
      if (peer.p[h]) sysStore(peer.p[h] + j, t);
  }
}
__global__ __launch_bounds__(kVecThreads) void k_mesh_push_slice(const double* __restrict__ vec, int lo, int hi,
                                                                 const MeshArgs ma, long long e) {
  if (dead(ma)) return;
  pushSlice(vec, lo, hi, ma);
  lastBlockSignal(ma, kFlagGen, e, 2);
}

__global__ __launch_bounds__(kVecThreads) void k_mesh_wait_copy(double* __restrict__ vec, int lo, int hi, int len,
                                                                const MeshArgs ma, long long e) {
  if (dead(ma)) return;
  copyOthers(recvX(ma, ma.g), vec, lo, hi, len);
}

// Whole block: true in the block that arrives last at `ticket` (every block of the grid must call it once).
__device__ bool lastBlockHere(const MeshArgs& ma, int ticket) {
  __shared__ int lastOne;
  __syncthreads();  // every wave of the block has finished what came before
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(&ma.ms->counter[ticket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lastOne = prev == gridDim.x - 1;
    if (lastOne) __hip_atomic_store(&ma.ms->counter[ticket], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return lastOne != 0;
}

// Round 6 (fusedWait == 2): a generic all-gather in ONE launch instead of four — push, epoch by the block that drains
// last, every block waits and copies its share, and the block that finishes copying last holds the rendezvous that
// k_mesh_barrier held (the receive area may be overwritten by the next collective only after every rank has read it).
__global__ __launch_bounds__(kVecThreads) void k_mesh_allgather(double* __restrict__ vec, int lo, int hi, int len,
                                                                const MeshArgs ma, long long e) {
  if (dead(ma)) return;
  pushSlice(vec, lo, hi, ma);
  lastBlockSignal(ma, kFlagGen, e, 2);
  if (!waitPeers(ma, kFlagGen, e)) { fail(ma, nullptr); return; }
  copyOthers(recvX(ma, ma.g), vec, lo, hi, len);  // (a thread's loads have returned before it reaches the ticket: their values were stored)
  if (!lastBlockHere(ma, 3)) return;
  if (threadIdx.x == 0) signalPeers(ma, kFlagBar, e);
  if (!waitPeers(ma, kFlagBar, e)) fail(ma, nullptr);
}

__global__ __launch_bounds__(kVecThreads) void k_mesh_wait_reduce(const double* __restrict__ partial,
                                                                  double* __restrict__ dst, const MeshArgs ma,
                                                                  long long e) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;
  if (dead(ma)) return;
  const int c0 = mv->colOff[ma.g], len = mv->colOff[ma.g + 1] - c0;
  PeerPtrs src;
  reduceSources(ma, partial + c0, src);
  const int stride = gridDim.x * blockDim.x;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < len; j += stride) dst[c0 + j] = orderedSum(src, j);
}

// Rendezvous of all ranks: nobody passes before everybody has finished the work queued before it.
__global__ __launch_bounds__(kVecThreads) void k_mesh_barrier(const MeshArgs ma, long long e) {
  const MeshView* __restrict__ mv = ma.v; (void)mv;

  if (dead(ma)) return;
  if (threadIdx.x == 0) signalPeers(ma, kFlagBar, e);
  if (!waitPeers(ma, kFlagBar, e)) fail(ma, nullptr);
}

// Checksum of a vector's BIT PATTERNS, independent of the summation order: every element is rotated by its
// index and XORed (wave shuffle, then one integer atomic per wave).  *acc must be zero on entry.
__global__ __launch_bounds__(kVecThreads) void k_mesh_checksum(const double* __restrict__ v, long long len,
                                                               unsigned long long* acc) {
  unsigned long long x = 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
    const int r = (int)(i & 63);
    x ^= (b << r) | (r ? (b >> (64 - r)) : 0ull);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) x ^= __shfl_down(x, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0 && x) atomicXor(acc, x);
}
// buf[3*g .. 3*g+2] = the checksum in three 22-bit pieces (exact in a double), zero elsewhere
__global__ void k_mesh_checksum_pack(const unsigned long long* acc, double* buf, int G, int g) {
  const int t = threadIdx.x;
  if (t < 3 * G) {
    const unsigned long long c = *acc;
    const int slot = t / 3, piece = t % 3;
    buf[t] = slot == g ? (double)((c >> (22 * piece)) & (piece == 2 ? 0xfffffull : 0x3fffffull)) : 0.0;
  }
}

// buf[0:k) summed over the ranks in rank order by ONE block (the body of k_mesh_allreduce_scalars).
__device__ void allReduceBlock(double* buf, int k, const MeshArgs& ma, long long e, bool agentLoads) {
  const int tid = threadIdx.x;
  if (tid < k) {
    const double t = agentLoads ? __hip_atomic_load(buf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : buf[tid];
    for (int h = 0; h < ma.G; ++h) sysStore(mailAt(ma, h, false, ma.g) + tid, t);
  }
  drainStores();
  __syncthreads();
  if (tid == 0) signalPeers(ma, kFlagGen, e);
  if (!waitPeers(ma, kFlagGen, e)) { fail(ma, nullptr); return; }
  if (tid < k) {
    double s = 0.0;
    for (int h = 0; h < ma.G; ++h) s += sysLoad(mailAt(ma, ma.g, false, h) + tid);
    buf[tid] = s;
  }
  __syncthreads();
  // the mailboxes may be overwritten by the next all-reduce only after every rank has read them
  if (tid == 0) signalPeers(ma, kFlagBar, e);
  if (!waitPeers(ma, kFlagBar, e)) fail(ma, nullptr);
}

__global__ __launch_bounds__(kVecThreads) void k_mesh_allreduce_scalars(double* buf, int k, const MeshArgs ma,
                                                                        long long e) {
  if (dead(ma)) return;
  allReduceBlock(buf, k, ma, e, false);
}

// Round 6: the statistics of a sharded check — k_final_reduce2 (one block per quantity, fixed-order sum of its per-block
// partials, gated like every kernel of a check) and the all-reduce over the ranks in ONE launch: the block that finishes
// last runs the exchange.  The exchange is never gated (its epochs are counted by the host on every rank alike).
__global__ __launch_bounds__(kVecThreads) void k_mesh_reduce2_allreduce(const double* partials, int pstride, int nQ0, int nBlocks0,
                                                                        int nBlocks1, double* out, const CheckGate g,
                                                                        const MeshArgs ma, long long e) {
  if (dead(ma)) return;
  __shared__ double scratch[kVecThreads / kWave];
  if (gateOpen(g)) {
    const double s = reducePartials(partials + (size_t)blockIdx.x * pstride, (int)blockIdx.x < nQ0 ? nBlocks0 : nBlocks1, scratch);
    if (threadIdx.x == 0) __hip_atomic_store(out + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  drainStores();  // the sum is in memory before this block's ticket
  if (!lastBlockHere(ma, 3)) return;
  allReduceBlock(out, (int)gridDim.x, ma, e, true);
}

// The two norms of the primal-weight update (k_restart_vec's partials): each summed in fixed order by the one block,
// then added over the ranks — two k_final_reduce launches and the all-reduce in one.
__global__ __launch_bounds__(kVecThreads) void k_mesh_norms_allreduce(const double* partX, int nX, const double* partY, int nY,
                                                                      double* norms, const MeshArgs ma, long long e) {
  if (dead(ma)) return;
  __shared__ double scratch[2][kVecThreads / kWave];
  const double sx = reducePartials(partX, nX, scratch[0]);
  const double sy = reducePartials(partY, nY, scratch[1]);
  if (threadIdx.x == 0) { norms[0] = sx; norms[1] = sy; }
  __syncthreads();
  allReduceBlock(norms, 2, ma, e, false);
}

// Grids of the mesh kernels.  Producers end with one ticket atomic per block on a single counter
// (~12 ns each, serialised): few blocks.  Consumers only poll their flags: many blocks, a handful of
// elements per thread, so that the system-scope payload loads (not pipelined by the compiler) overlap.
// PDLP_MI355X_MESH_BLOCKS overrides the producer cap.
int32_t capped(int64_t len, int cap) {
  int64_t b = (len + kVecThreads - 1) / kVecThreads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int32_t)b;
}
int32_t meshBlocks(int64_t len) {  // producers
  static const int cap = [] {
    const char* e = devEnv("PDLP_MI355X_MESH_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 128;
  }();
  return capped(len, cap);
}
int32_t meshConsumerBlocks(int64_t len) { return capped((len + 3) / 4, 1024); }  // ~4 elements per thread

}  // namespace

int32_t meshGrid(int64_t len) { return meshConsumerBlocks(len); }

// ---- launchers (dmv = the view in device memory) --------------------------------------------
void launchMeshPrimalStep(const IterVecs& vc, const DevState* st, const MeshArgs& dmv, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_primal_step, dim3(meshBlocks(vc.n)), dim3(kVecThreads), 0, s, vc, st, dmv);
}
void launchMeshWaitCopyX(const IterVecs& vf, const DevState* st, const MeshArgs& dmv, hipStream_t s) {
  if (!dmv.fusedWait) hipLaunchKernelGGL(k_mesh_wait, dim3(1), dim3(kWave), 0, s, const_cast<DevState*>(st), dmv, (int)kFlagX, 0LL);
  // (every block polls when the wait is fused: one block per CU at most)
  hipLaunchKernelGGL(k_mesh_wait_copy_x, dim3(dmv.fusedWait ? capped((vf.n + 3) / 4, 256) : meshConsumerBlocks(vf.n)), dim3(kVecThreads), 0,
                     s, vf, const_cast<DevState*>(st), dmv);
}
void launchMeshPushPartial(const double* partial, int32_t n, const DevState* st, const MeshArgs& dmv, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_push_partial, dim3(meshBlocks(n)), dim3(kVecThreads), 0, s, partial, st, dmv, 0LL);
}
void launchMeshReduceInteract(const IterVecs& vc, const DevState* st, const MeshArgs& dmv, const double* partial,
                              double* partDX, double* partInter, int32_t nBlocks, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_wait, dim3(1), dim3(kWave), 0, s, const_cast<DevState*>(st), dmv, (int)kFlagP, 0LL);
  hipLaunchKernelGGL(k_mesh_reduce_interact, dim3(nBlocks), dim3(kVecThreads), 0, s, vc, const_cast<DevState*>(st),
                     dmv, partial, partDX, partInter);
}
void launchMeshDecide(DevState* st, const MeshArgs& dmv, const double* partDY, int32_t nDY, const double* partDX,
                      const double* partInter, int32_t nDX, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_decide, dim3(1), dim3(kVecThreads), 0, s, st, dmv, partDY, nDY, partDX, partInter, nDX);
}

void launchMeshPushY(const IterVecs& vf, const double* const yFull[2], const DevState* st, const MeshArgs& dmv, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_push_y, dim3(meshBlocks(std::max(vf.m, 1))), dim3(kVecThreads), 0, s, yFull[0], yFull[1], st, dmv);
}
void launchMeshWaitCopyY(double* const yFull[2], int32_t m, const DevState* st, const MeshArgs& dmv, hipStream_t s) {
  if (!dmv.fusedWait) hipLaunchKernelGGL(k_mesh_wait, dim3(1), dim3(kWave), 0, s, const_cast<DevState*>(st), dmv, (int)kFlagP, 0LL);
  hipLaunchKernelGGL(k_mesh_wait_copy_y, dim3(dmv.fusedWait ? capped((std::max(m, 1) + 3) / 4, 256) : meshConsumerBlocks(std::max(m, 1))),
                     dim3(kVecThreads), 0, s, yFull[0], yFull[1], m, st, dmv);
}

// fusedWait == 2: the X / Y exchange of a trial in one launch each (see k_mesh_primal_x)
void launchMeshPrimalX(const IterVecs& vc, double* const xFull[2], int32_t nFull, DevState* st, const MeshArgs& dmv, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_primal_x, dim3(capped((std::max(nFull, 1) + 3) / 4, 256)), dim3(kVecThreads), 0, s, vc, xFull[0], xFull[1], nFull,
                     st, dmv);
}
void launchMeshY(double* const yFull[2], int32_t m, DevState* st, const MeshArgs& dmv, hipStream_t s) {
  hipLaunchKernelGGL(k_mesh_y, dim3(capped((std::max(m, 1) + 3) / 4, 256)), dim3(kVecThreads), 0, s, yFull[0], yFull[1], m, st, dmv);
}

void launchMeshHalpernStep(const MatView& A, const MatView& At, const HalpernVecs& hFull, const HalpernVecs& hCol,
                           int32_t n, int32_t nLoc, double* partial, const MeshArgs& ma, hipStream_t s) {
  // 1. partial A_g' y_current, pushed to the slice owners (flag P)
  launchSpmvPlain(At, hFull.yc, partial, s);
  hipLaunchKernelGGL(k_mesh_push_partial, dim3(meshBlocks(n)), dim3(kVecThreads), 0, s, partial,
                     (const DevState*)nullptr, ma, -1LL);
  // 2. rank-ordered reduce + primal projection / reflection / blend on the slice; reflected x pushed (flag X)
  hipLaunchKernelGGL(k_mesh_wait, dim3(1), dim3(kWave), 0, s, (DevState*)nullptr, ma, (int)kFlagP, -1LL);
  hipLaunchKernelGGL(k_mesh_h_reduce_primal, dim3(meshBlocks(nLoc)), dim3(kVecThreads), 0, s, hCol, nLoc, partial, ma);
  // 3. the other slices of the reflected x, then the dual side on the local rows
  hipLaunchKernelGGL(k_mesh_wait, dim3(1), dim3(kWave), 0, s, (DevState*)nullptr, ma, (int)kFlagX, -1LL);
  hipLaunchKernelGGL(k_mesh_h_copy_x, dim3(meshConsumerBlocks(n)), dim3(kVecThreads), 0, s, hFull.rx, n, ma);
  launchHalpernDual(A, hFull, s);
  hipLaunchKernelGGL(k_mesh_bump, dim3(1), dim3(1), 0, s, ma);
}

// ---- host side ------------------------------------------------------------------------------
namespace {

// Rendezvous segment in POSIX shared memory (all ranks are processes of one node).
struct ShmSlot {
  std::atomic<uint32_t> ready;
  uint32_t pid;
  uint64_t arenaBytes;
  uint64_t rawPtr;   // the arena's device pointer: what a rank of the SAME process maps (peer access, no IPC)
  int32_t device;    // HIP device ordinal of the owner
  int32_t pad_;
  uint64_t busHash;  // hash of the owner's PCI bus id: two ranks on the same physical GPU have the same
  hipIpcMemHandle_t handle;
  std::atomic<uint32_t> agree[64];  // round -> 1 ok / 2 not ok
};
struct ShmSeg {
  std::atomic<uint32_t> barrier[16];
  ShmSlot slot[kMeshMaxRanks];
};

uint64_t fnv64(const void* p, size_t len) {
  const unsigned char* b = (const unsigned char*)p;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < len; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

// PDLP_MI355X_MESH_FENCES: 1 = release / acquire fences around the flags; 2 = the same AND every step of an exchange a
// kernel of its own (single-block wait kernels: the most conservative form, the last stop in front of RCCL)
int32_t meshFenceLevel() {
  const char* e = getenv("PDLP_MI355X_MESH_FENCES");
  return e ? std::min(std::max(atoi(e), 0), 2) : 0;
}
int32_t meshFences() { return meshFenceLevel() != 0 ? 1 : 0; }

}  // namespace

void Mesh::hostBarrier(int slot, double timeoutSec) {
  ShmSeg* seg = (ShmSeg*)shm_;
  seg->barrier[slot].fetch_add(1, std::memory_order_acq_rel);
  const auto t0 = std::chrono::steady_clock::now();
  while (seg->barrier[slot].load(std::memory_order_acquire) < (uint32_t)v_.G) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSec)
      throw std::runtime_error("pdlp_mi355x mesh: timed out waiting for the other ranks (host rendezvous)");
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

Mesh::Mesh(int32_t rank, int32_t world, const void* id128, int32_t n, int32_t m, const std::vector<int32_t>& rowOff,
           hipStream_t s)
    : n_(n) {
  try {
    construct(rank, world, id128, n, m, rowOff, s);
  } catch (...) {
    // a constructor that throws never runs its destructor: free what was allocated, but do not wait for the
    // peers (the exchange is unusable; they time out in their own rendezvous and fall back as well)
    if (state_) (void)hipMemset(state_, 0xff, sizeof(MeshState));  // marks the exchange broken
    release();
    throw;
  }
}

void Mesh::construct(int32_t rank, int32_t world, const void* id128, int32_t n, int32_t m,
                     const std::vector<int32_t>& rowOff, hipStream_t s) {
  if (world > kMeshMaxRanks) throw std::runtime_error("pdlp_mi355x mesh: more than 16 ranks");
  if (world > 1 && !id128) throw std::runtime_error("pdlp_mi355x mesh: a communicator id is needed");
  v_.G = world;
  v_.g = rank;
  for (int h = 0; h <= kMeshMaxRanks; ++h) {
    const int hh = h < world ? h : world;
    v_.colOff[h] = (int32_t)((int64_t)n * hh / world);
    v_.rowOff[h] = hh < (int)rowOff.size() ? rowOff[hh] : m;
  }
  int64_t sliceMax = 1;
  for (int h = 0; h < world; ++h) sliceMax = std::max<int64_t>(sliceMax, v_.colOff[h + 1] - v_.colOff[h]);
  v_.sliceMax = sliceMax;
  v_.waitTicks = 1000000000LL;  // 10 s of the 100 MHz wall clock; PDLP_MI355X_MESH_TIMEOUT_MS overrides
  if (const char* t = getenv("PDLP_MI355X_MESH_TIMEOUT_MS")) v_.waitTicks = std::max(1LL, atoll(t)) * 100000LL;
  size_t off = 0;
  v_.offFlags = (int64_t)off;   off += alignUp((size_t)kNumMeshFlags * kMeshMaxRanks * kFlagStride, 4096);
  v_.offMailHot = (int64_t)off; off += alignUp((size_t)kMeshMaxRanks * kMeshMailDoubles * 8, 4096);
  v_.offMailGen = (int64_t)off; off += alignUp((size_t)kMeshMaxRanks * kMeshMailDoubles * 8, 4096);
  v_.offRecvX = (int64_t)off;   off += alignUp((size_t)std::max(n, m) * 8 + 8, 4096);
  v_.offRecvP = (int64_t)off;   off += alignUp((size_t)world * sliceMax * 8, 4096);
  v_.offRecvY = (int64_t)off;   off += alignUp((size_t)std::max(m, 1) * 8 + 8, 4096);
  arenaBytes_ = off;

  // memory that is coherent between agents inside a kernel (PDLP_MI355X_MESH_MEM=finegrained|coarse are
  // diagnostic switches; coarse only for single-GPU protocol tests)
  const char* mm = devEnv("PDLP_MI355X_MESH_MEM");
  if (mm && !strcmp(mm, "coarse")) {
    PDLP_HIP(hipMalloc(&arena_, arenaBytes_));
  } else if (mm && !strcmp(mm, "finegrained")) {
    PDLP_HIP(hipExtMallocWithFlags(&arena_, arenaBytes_, hipDeviceMallocFinegrained));
  } else if (hipExtMallocWithFlags(&arena_, arenaBytes_, hipDeviceMallocUncached) != hipSuccess) {
    // default: uncached (MTYPE_UC) device memory — what RCCL uses for its peer-visible buffers; every arena
    // access is a system-scope write-/read-through anyway.  Fine-grained as the second choice.
    (void)hipGetLastError();
    PDLP_HIP(hipExtMallocWithFlags(&arena_, arenaBytes_, hipDeviceMallocFinegrained));
  }
  PDLP_HIP(hipMemsetAsync(arena_, 0, arenaBytes_, s));
  PDLP_HIP(hipMalloc((void**)&state_, sizeof(MeshState)));
  PDLP_HIP(hipMemsetAsync(state_, 0, sizeof(MeshState), s));
  PDLP_HIP(hipStreamSynchronize(s));
  v_.ms = state_;
  for (int h = 0; h < kMeshMaxRanks; ++h) v_.arena[h] = nullptr;
  v_.arena[rank] = (char*)arena_;
  PDLP_HIP(hipMalloc((void**)&dView_, sizeof(MeshView)));
  PDLP_HIP(hipMalloc((void**)&testV_, sizeof(double) * (size_t)std::max(n, 64)));  // selfTest() / verifyReplicated() scratch
  PDLP_HIP(hipMalloc((void**)&testP_, sizeof(double) * (size_t)std::max(n, 64)));
  if (world == 1) {
    PDLP_HIP(hipMemcpy(dView_, &v_, sizeof(MeshView), hipMemcpyHostToDevice));
    setupOk_ = true;
    args_ = MeshArgs{dView_, state_, v_.G, v_.g, v_.waitTicks, meshFences(), 0};
    return;
  }

  char name[64];
  snprintf(name, sizeof(name), "/pdlp_mesh_%016llx", (unsigned long long)fnv64(id128, 128));
  shmName_ = name;
  const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) throw std::runtime_error("pdlp_mi355x mesh: shm_open failed");
  shmBytes_ = sizeof(ShmSeg);
  if (ftruncate(fd, (off_t)shmBytes_) != 0) { close(fd); throw std::runtime_error("pdlp_mi355x mesh: ftruncate failed"); }
  shm_ = mmap(nullptr, shmBytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (shm_ == MAP_FAILED) { shm_ = nullptr; throw std::runtime_error("pdlp_mi355x mesh: mmap failed"); }
  ShmSeg* seg = (ShmSeg*)shm_;
  ShmSlot& mine = seg->slot[rank];
  // From here on a failure must not desert the other ranks: it is recorded, this rank keeps taking part
  // in the rendezvous, and selfTest()/allAgree() then turn it into "every rank falls back to RCCL".
  // One process, one host thread per device (pdlp_mi355x_solve with num_devices > 1): the peers' arenas
  // are ordinary pointers of this address space and only need peer access; across processes they are
  // mapped through HIP IPC.  A rank whose IPC export fails is still usable by same-process peers (ready = 3).
  bool ok = true;
  int myDevice = 0;
  (void)hipGetDevice(&myDevice);
  const char* forceIpc = devEnv("PDLP_MI355X_MESH_FORCE_IPC");  // diagnostic: IPC also inside one process
  const bool allLocal = !(forceIpc && atoi(forceIpc) != 0);
  const bool exported = hipIpcGetMemHandle(&mine.handle, arena_) == hipSuccess;
  if (!exported) {
    (void)hipGetLastError();
    // peers in OTHER processes cannot map this arena then (same-process peers use the raw pointer): say why
    const char* legacy = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
    fprintf(stderr, "pdlp_mi355x[rank %d]: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=%s). Hosts whose driver only "
                    "supports dmabuf IPC need HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the first HIP call of the "
                    "process; ranks in other processes will fall back to RCCL.\n", rank, legacy ? legacy : "(unset)");
  }
  mine.arenaBytes = arenaBytes_;
  mine.rawPtr = (uint64_t)(uintptr_t)arena_;
  mine.device = myDevice;
  {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, myDevice) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
    mine.busHash = bus[0] ? fnv64(bus, strlen(bus)) : 0ull;  // (0: unknown — treated as shared)
  }
  mine.pid = (uint32_t)getpid();
  mine.ready.store(exported ? 1u : 3u, std::memory_order_release);
  hostBarrier(0, 60.0);
  if (rank == 0) shm_unlink(name);  // every rank has it mapped; nothing is left behind on a crash
  for (int h = 0; h < world; ++h) {
    if (h == rank) continue;
    const uint32_t rdy = seg->slot[h].ready.load(std::memory_order_acquire);
    const bool samePid = allLocal && seg->slot[h].pid == mine.pid;
    if (!(rdy == 1u || (rdy == 3u && samePid)) || seg->slot[h].arenaBytes != arenaBytes_) {
      ok = false;
      continue;
    }
    void* p = nullptr;
    if (allLocal && seg->slot[h].pid == mine.pid) {
      p = (void*)(uintptr_t)seg->slot[h].rawPtr;
      if (seg->slot[h].device != myDevice) {
        const hipError_t e = hipDeviceEnablePeerAccess(seg->slot[h].device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
        (void)hipGetLastError();
      }
      if (!ok) continue;
      v_.arena[h] = (char*)p;
      continue;
    }
    if (hipIpcOpenMemHandle(&p, seg->slot[h].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      ok = false;
      continue;
    }
    v_.arena[h] = (char*)p;
    ipcMapped_[h] = true;
  }
  if (ok) ok = hipMemcpy(dView_, &v_, sizeof(MeshView), hipMemcpyHostToDevice) == hipSuccess;
  setupOk_ = ok;
  // does every rank have a physical GPU of its own?  (every rank reads the same table: the same answer everywhere)
  bool own = true;
  for (int a = 0; a < world; ++a)
    for (int b = a + 1; b < world; ++b)
      if (seg->slot[a].busHash == 0ull || seg->slot[a].busHash == seg->slot[b].busHash) own = false;
  int fusedWait = own ? 2 : 0;
  if (meshFenceLevel() == 2) fusedWait = 0;
  if (const char* e = devEnv("PDLP_MI355X_MESH_FUSED_WAIT")) fusedWait = std::min(std::max(atoi(e), 0), 2);
  args_ = MeshArgs{dView_, state_, v_.G, v_.g, v_.waitTicks, meshFences(), fusedWait};
  hostBarrier(1, 60.0);
}

void Mesh::release() noexcept {
  if (v_.G > 1 && shm_) {
    // nobody frees an arena that a peer's kernel may still write to — unless the exchange already broke
    // (a peer vanished or timed out): then waiting for it again would only delay the error
    (void)hipDeviceSynchronize();
    MeshState h{};
    const bool broken = !state_ || hipMemcpy(&h, state_, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || h.error != 0;
    if (!broken) { try { hostBarrier(2, 20.0); } catch (...) {} }
    for (int r = 0; r < v_.G; ++r)
      if (r != v_.g && v_.arena[r] && ipcMapped_[r]) (void)hipIpcCloseMemHandle(v_.arena[r]);
    if (!broken) { try { hostBarrier(3, 20.0); } catch (...) {} }
  }
  if (arena_) (void)hipFree(arena_);
  if (state_) (void)hipFree(state_);
  if (dView_) (void)hipFree(dView_);
  if (testV_) (void)hipFree(testV_);
  if (testP_) (void)hipFree(testP_);
  if (shm_) munmap(shm_, shmBytes_);
  arena_ = nullptr; state_ = nullptr; dView_ = nullptr; shm_ = nullptr; testV_ = nullptr; testP_ = nullptr;
}

Mesh::~Mesh() { release(); }

void Mesh::allGather(double* vec, bool byRows, hipStream_t s) {
  if (v_.G == 1) return;
  const int32_t* off = byRows ? v_.rowOff : v_.colOff;
  const int32_t lo = off[v_.g], hi = off[v_.g + 1], len = off[v_.G];
  const long long e = ++epoch_;
  if (args_.fusedWait == 2) {  // every rank on a GPU of its own: one launch
    hipLaunchKernelGGL(k_mesh_allgather, dim3(capped((len + 3) / 4, 256)), dim3(kVecThreads), 0, s, vec, lo, hi, len, args_, e);
    return;
  }
  hipLaunchKernelGGL(k_mesh_push_slice, dim3(meshBlocks(hi - lo)), dim3(kVecThreads), 0, s, vec, lo, hi, args_, e);
  hipLaunchKernelGGL(k_mesh_wait, dim3(1), dim3(kWave), 0, s, (DevState*)nullptr, args_, (int)kFlagGen, e);
  hipLaunchKernelGGL(k_mesh_wait_copy, dim3(meshConsumerBlocks(len)), dim3(kVecThreads), 0, s, vec, lo, hi, len, args_, e);
  hipLaunchKernelGGL(k_mesh_barrier, dim3(1), dim3(kVecThreads), 0, s, args_, e);
}

void Mesh::reduceScatterCols(const double* partial, double* dst, hipStream_t s) {
  const long long e = ++epoch_;
  const int32_t n = v_.colOff[v_.G];
  hipLaunchKernelGGL(k_mesh_push_partial, dim3(meshBlocks(n)), dim3(kVecThreads), 0, s, partial,
                     (const DevState*)nullptr, args_, e);
  hipLaunchKernelGGL(k_mesh_wait, dim3(1), dim3(kWave), 0, s, (DevState*)nullptr, args_, (int)kFlagGen, e);
  hipLaunchKernelGGL(k_mesh_wait_reduce, dim3(meshConsumerBlocks(c1() - c0())), dim3(kVecThreads), 0, s, partial, dst, args_, e);
  hipLaunchKernelGGL(k_mesh_barrier, dim3(1), dim3(kVecThreads), 0, s, args_, e);
}

void Mesh::allReduceScalars(double* buf, int32_t k, hipStream_t s) {
  if (k > kMeshMailDoubles) throw std::runtime_error("pdlp_mi355x mesh: too many scalars in one all-reduce");
  if (v_.G == 1) return;
  const long long e = ++epoch_;
  hipLaunchKernelGGL(k_mesh_allreduce_scalars, dim3(1), dim3(kVecThreads), 0, s, buf, k, args_, e);
}

void Mesh::reduce2AllReduce(const double* partials, int32_t stride, int32_t nQ0, int32_t nBlocks0, int32_t nQ1, int32_t nBlocks1,
                            double* out, CheckGate g, hipStream_t s) {
  if (nQ0 + nQ1 > kMeshMailDoubles) throw std::runtime_error("pdlp_mi355x mesh: too many scalars in one all-reduce");
  if (v_.G == 1 || args_.fusedWait != 2) {
    launchFinalReduce2(partials, stride, nQ0, nBlocks0, nQ1, nBlocks1, out, g, s);
    allReduceScalars(out, nQ0 + nQ1, s);
    return;
  }
  const long long e = ++epoch_;
  hipLaunchKernelGGL(k_mesh_reduce2_allreduce, dim3(nQ0 + nQ1), dim3(kVecThreads), 0, s, partials, stride, nQ0, nBlocks0, nBlocks1, out, g,
                     args_, e);
}

void Mesh::normsAllReduce(const double* partX, int32_t nX, const double* partY, int32_t nY, double* norms, hipStream_t s) {
  if (v_.G == 1 || args_.fusedWait != 2) {
    launchFinalReduce(partX, nX, nX, 1, norms, s);
    launchFinalReduce(partY, nY, nY, 1, norms + 1, s);
    allReduceScalars(norms, 2, s);
    return;
  }
  const long long e = ++epoch_;
  hipLaunchKernelGGL(k_mesh_norms_allreduce, dim3(1), dim3(kVecThreads), 0, s, partX, nX, partY, nY, norms, args_, e);
}

// Every rank must hold the same bits in a replicated vector (x of the cuPDLP path, the reflected x of the
// HiPDLP path): a stale or torn read in one of the exchanges would make the copies differ.  The checksums of
// all ranks are exchanged (exact small integers through the scalar all-reduce) and compared; a mismatch is
// an error, never a silently wrong iterate.  Called at check iterations (1 in 40).
void Mesh::verifyReplicated(const double* vec, int64_t len, hipStream_t s) {
  if (v_.G == 1) return;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(testP_);
  double* buf = testV_;  // >= 4 doubles... the pack needs 3*G <= 48: see construct()
  PDLP_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_mesh_checksum, dim3(capped(len, 256)), dim3(kVecThreads), 0, s, vec, (long long)len, acc);
  hipLaunchKernelGGL(k_mesh_checksum_pack, dim3(1), dim3(64), 0, s, acc, buf, v_.G, v_.g);
  allReduceScalars(buf, 3 * v_.G, s);
  double host[3 * kMeshMaxRanks];
  PDLP_HIP(hipMemcpyAsync(host, buf, sizeof(double) * 3 * v_.G, hipMemcpyDeviceToHost, s));
  PDLP_HIP(hipStreamSynchronize(s));
  checkError(s);
  for (int h = 1; h < v_.G; ++h)
    for (int p = 0; p < 3; ++p)
      if (host[3 * h + p] != host[p])
        throw std::runtime_error("pdlp_mi355x mesh: the ranks hold different copies of a replicated vector (exchange "
                                 "inconsistent); rerun with PDLP_MI355X_EXCHANGE=rccl");
}

void Mesh::phaseStats(double usPerWait[3], double count[3], hipStream_t s) {
  MeshState h{};
  PDLP_HIP(hipMemcpyAsync(&h, state_, sizeof(h), hipMemcpyDeviceToHost, s));
  PDLP_HIP(hipStreamSynchronize(s));
  for (int k = 0; k < 3; ++k) {
    count[k] = (double)h.waitCount[k];
    usPerWait[k] = h.waitCount[k] ? (double)h.waitTicks[k] / (double)h.waitCount[k] / 100.0 : 0.0;  // 100 MHz clock
  }
  PDLP_HIP(hipMemsetAsync(reinterpret_cast<char*>(state_) + offsetof(MeshState, waitTicks), 0,
                          sizeof(h.waitTicks) + sizeof(h.waitCount), s));
  PDLP_HIP(hipStreamSynchronize(s));
}

void Mesh::checkError(hipStream_t s) {
  MeshState h{};
  PDLP_HIP(hipMemcpyAsync(&h, state_, sizeof(h), hipMemcpyDeviceToHost, s));
  PDLP_HIP(hipStreamSynchronize(s));
  if (h.error)
    throw std::runtime_error("pdlp_mi355x mesh: a peer did not answer in time (exchange timed out)");
}

bool Mesh::allAgree(bool ok) {
  if (v_.G == 1) return ok;
  ShmSeg* seg = (ShmSeg*)shm_;
  const int r = agreeRound_++;
  if (r >= 64) throw std::runtime_error("pdlp_mi355x mesh: too many agreement rounds");
  seg->slot[v_.g].agree[r].store(ok ? 1u : 2u, std::memory_order_release);
  bool all = true;
  const auto t0 = std::chrono::steady_clock::now();
  for (int h = 0; h < v_.G; ++h) {
    uint32_t a;
    while ((a = seg->slot[h].agree[r].load(std::memory_order_acquire)) == 0) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0)
        throw std::runtime_error("pdlp_mi355x mesh: timed out waiting for the other ranks (agreement)");
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    all = all && a == 1u;
  }
  return all;
}

// Known-answer test of the three exchange patterns with rank-dependent data.
bool Mesh::selfTest(hipStream_t s) {
  if (!setupOk_) return false;
  const int32_t n = n_;
  const int G = v_.G, g = v_.g;
  auto val = [](int rank, int j, int round) { return (double)((rank + 1) * 1000003 + j * 7 + round); };
  std::vector<double> host(n), got(n);
  // buffers allocated in construct() and freed in release(): a hipFree here would synchronise the whole
  // device — with several ranks of one process folded onto one device it would wait for a peer's kernel
  double* dv = testV_;
  double* dp = testP_;  // also carries the 3 test scalars
  bool ok = true;
  try {
    for (int round = 0; round < 3 && ok; ++round) {
      // all-gather by columns
      for (int j = 0; j < n; ++j) host[j] = (j >= c0() && j < c1()) ? val(g, j, round) : -1.0;
      PDLP_HIP(hipMemcpyAsync(dv, host.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
      allGather(dv, false, s);
      PDLP_HIP(hipMemcpyAsync(got.data(), dv, sizeof(double) * n, hipMemcpyDeviceToHost, s));
      checkError(s);
      for (int h = 0; h < G && ok; ++h)
        for (int j = v_.colOff[h]; j < v_.colOff[h + 1]; ++j)
          if (got[j] != val(h, j, round)) { ok = false; break; }
      // reduce-scatter
      for (int j = 0; j < n; ++j) host[j] = val(g, j, round + 10);
      PDLP_HIP(hipMemcpyAsync(dp, host.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
      reduceScatterCols(dp, dv, s);
      PDLP_HIP(hipMemcpyAsync(got.data(), dv, sizeof(double) * n, hipMemcpyDeviceToHost, s));
      checkError(s);
      for (int j = c0(); j < c1() && ok; ++j) {
        double e = 0.0;
        for (int h = 0; h < G; ++h) e += val(h, j, round + 10);
        if (got[j] != e) ok = false;
      }
      // scalars
      double sc[3] = {val(g, 1, round), val(g, 2, round), val(g, 3, round)};
      PDLP_HIP(hipMemcpyAsync(dp, sc, sizeof(sc), hipMemcpyHostToDevice, s));
      allReduceScalars(dp, 3, s);
      PDLP_HIP(hipMemcpyAsync(sc, dp, sizeof(sc), hipMemcpyDeviceToHost, s));
      checkError(s);
      for (int q = 0; q < 3; ++q) {
        double e = 0.0;
        for (int h = 0; h < G; ++h) e += val(h, q + 1, round);
        if (sc[q] != e) ok = false;
      }
    }
  } catch (...) {
    ok = false;
  }

  return ok;
}

}  // namespace pdlp
