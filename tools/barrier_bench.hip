// Micro-benchmark (round 4): what does a PHASE CHANGE cost on an MI355X in the geometry of the slab SpMV launches —
// 256 workgroups of 1024 threads, one per CU (128 KB of LDS each) — as a kernel boundary and as an in-kernel grid barrier?
// The question behind DESIGN.md section 3, "Why no persistent slab loop": a persistent trial loop trades the two kernel
// boundaries of a trial for two more grid barriers (plus the L1 invalidate a phase needs when the gathered vector changed
// under the running kernel).  Phases: every workgroup streams its share of a 96 MB buffer (the matrix stream of one SpMV)
// and stores 32 KB (the epilogue's vectors); K phases run
//   launches      as K launches of one hipGraph                           (what the 2-launch trial does)
//   barriers      inside ONE launch, separated by a sweep barrier         (one arrival word per workgroup, agent scope)
//   barriers+inv  the same with `buffer_inv sc1` behind every barrier     (agent-scope acquire by one wave / by every wave)
// and the same three with EMPTY phases (the synchronisation alone).  Prints microseconds per phase.
// Build: hipcc --offload-arch=gfx950 -O3 tools/barrier_bench.hip -o tools/barrier_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e = (x);                                                  \
    if (e != hipSuccess) {                                               \
      printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

constexpr int kThreads = 1024, kBlocks = 256, kWave = 64;

__device__ __forceinline__ void phaseBody(const double* __restrict__ src, double* __restrict__ dst, long perBlock, int outPerBlock, int work) {
  if (!work) return;
  const double* p = src + (long)blockIdx.x * perBlock;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (long i = threadIdx.x; i + 3 * kThreads < perBlock; i += 4 * kThreads) {
    const double a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + kThreads);
    const double c = __builtin_nontemporal_load(p + i + 2 * kThreads), d = __builtin_nontemporal_load(p + i + 3 * kThreads);
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  const double s = (s0 + s1) + (s2 + s3);
  for (int i = threadIdx.x; i < outPerBlock; i += kThreads) dst[(long)blockIdx.x * outPerBlock + i] = s + i;
}

__global__ __launch_bounds__(kThreads) void k_phase(const double* src, double* dst, long perBlock, int outPerBlock, int work) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) lds[0] = 0.0;
  phaseBody(src, dst, perBlock, outPerBlock, work);
}

__global__ __launch_bounds__(kThreads) void k_persistent(const double* src, double* dst, long perBlock, int outPerBlock, int work, int K,
                                                         unsigned long long* bar, unsigned long long base, int inv) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) lds[0] = 0.0;
  const int lane = threadIdx.x & (kWave - 1);
  for (int k = 0; k < K; ++k) {
    phaseBody(src, dst, perBlock, outPerBlock, work);
    // every wave drains its stores, the block meets, wave 0 arrives and sweeps
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < kWave) {
      const unsigned long long epoch = base + (unsigned long long)k + 1ull;
      if (lane == 0) __hip_atomic_store(bar + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        bool ok = true;
        for (int i = lane; i < (int)gridDim.x; i += kWave)
          ok = ok && __hip_atomic_load(bar + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (inv == 1 && threadIdx.x < kWave) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1 by ONE wave: the L1 belongs to the CU
    __syncthreads();
    if (inv == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                        // ... by every wave
  }
}

int main() {
  const long bytes = 96l << 20, n = bytes / 8, perBlock = n / kBlocks;
  const int outPerBlock = 4096;  // 32 KB per workgroup
  double *src, *dst;
  unsigned long long* bar;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, (size_t)kBlocks * outPerBlock * 8));
  CK(hipMalloc(&bar, (kBlocks + 8) * 8));
  CK(hipMemset(src, 0, bytes));
  CK(hipMemset(bar, 0, (kBlocks + 8) * 8));
  const size_t lds = 128 << 10;
  CK(hipFuncSetAttribute((const void*)k_phase, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)k_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int K = 84, reps = 20;  // 84 = the launches of one 42-trial graph
  unsigned long long base = 0;
  for (int work = 1; work >= 0; --work) {
    // ---- K launches in one graph ----
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_phase, dim3(kBlocks), dim3(kThreads), lds, s, src, dst, perBlock, outPerBlock, work);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double usLaunch = ms * 1e3 / (reps * K);
    // ---- one launch, K phases, barriers (without / with the invalidate) ----
    double usBar[3];
    for (int inv = 0; inv < 3; ++inv) {
      hipLaunchKernelGGL(k_persistent, dim3(kBlocks), dim3(kThreads), lds, s, src, dst, perBlock, outPerBlock, work, K, bar, base, inv);
      base += K;
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_persistent, dim3(kBlocks), dim3(kThreads), lds, s, src, dst, perBlock, outPerBlock, work, K, bar, base, inv);
        base += K;
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      usBar[inv] = ms * 1e3 / (reps * K);
    }
    printf("%s phases (256 x 1024 threads, 128 KB LDS%s): per phase %.2f us as graph launches | %.2f us with grid barriers | %.2f us "
           "with grid barriers + buffer_inv sc1 by one wave per workgroup | %.2f us with the invalidate in every wave\n",
           work ? "streaming" : "empty", work ? ", 96 MB read + 8 MB written per phase" : "", usLaunch, usBar[0], usBar[1], usBar[2]);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
