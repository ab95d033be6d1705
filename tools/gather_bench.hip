// Micro-benchmark: throughput of fully divergent 8-byte gathers on gfx950 from
// tables of different sizes (L1-, L2-, MALL-resident) and from LDS.  Used to
// size the slab SpMV.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__);        \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <int PER>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ idx, const double* __restrict__ tab,
                                                double* out, long n) {
  long base = ((long)blockIdx.x * 256) * PER + threadIdx.x;
  int ci[PER];
  double v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    long q = base + (long)k * 256;
    ci[k] = idx[q < n ? q : n - 1];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) v[k] = tab[ci[k]];
  double s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) s += v[k];
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

// table staged in LDS (T doubles); each block gathers PER*256 entries from it
template <int PER, int T>
__global__ __launch_bounds__(256) void k_gather_lds(const int* __restrict__ idx, const double* __restrict__ tab,
                                                    double* out, long n) {
  __shared__ double t[T];
  for (int i = threadIdx.x; i < T; i += 256) t[i] = tab[i];
  __syncthreads();
  long base = ((long)blockIdx.x * 256) * PER + threadIdx.x;
  int ci[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    long q = base + (long)k * 256;
    ci[k] = idx[q < n ? q : n - 1] & (T - 1);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) s += t[ci[k]];
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

// only the coalesced index stream (what the gather kernels pay besides the gathers)
template <int PER>
__global__ __launch_bounds__(256) void k_stream(const int* __restrict__ idx, double* out, long n) {
  long base = ((long)blockIdx.x * 256) * PER + threadIdx.x;
  long s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    long q = base + (long)k * 256;
    s += idx[q < n ? q : n - 1];
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = (double)s;
}

int main() {
  const long n = 8 << 20;  // 8M gathers
  std::mt19937_64 rng(1);
  int* dIdx;
  double *dTab, *dOut;
  CK(hipMalloc(&dIdx, n * 4));
  CK(hipMalloc(&dTab, (8 << 20) * 8L));
  CK(hipMalloc(&dOut, n * 8));
  CK(hipMemset(dTab, 0, (8 << 20) * 8L));
  std::vector<int> h(n);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  constexpr int PER = 8;
  dim3 grid((n + 256 * PER - 1) / (256 * PER));
  float ms;
  auto timeit = [&](auto launch, const char* what) {
    for (int w = 0; w < 3; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    printf("%-58s %7.1f us  %7.1f G/s\n", what, ms * 1e3, n / ms / 1e6);
  };
  char name[128];
  for (long T : {4096L, 65536L, 262144L, 1048576L, 4194304L}) {
    for (auto& v : h) v = (int)(rng() % T);
    CK(hipMemcpy(dIdx, h.data(), n * 4, hipMemcpyHostToDevice));
    snprintf(name, sizeof name, "global gather, table %8ld doubles (%6ld KB)", T, T * 8 / 1024);
    timeit([&] { hipLaunchKernelGGL(k_gather<PER>, grid, dim3(256), 0, 0, dIdx, dTab, dOut, n); }, name);
    if (T == 4096)
      timeit([&] { hipLaunchKernelGGL((k_gather_lds<PER, 4096>), grid, dim3(256), 0, 0, dIdx, dTab, dOut, n); },
             "LDS gather, 32 KB table (+32 KB table load per 2048 gathers)");
  }
  timeit([&] { hipLaunchKernelGGL(k_stream<PER>, grid, dim3(256), 0, 0, dIdx, dOut, n); }, "index stream only (32 MB read + 8 MB write)");
  // sorted-by-line indices: same volume but adjacent lanes share 64-byte lines (8 doubles)
  for (long i = 0; i < n; ++i) h[i] = (int)((i / 8 * 8 + (rng() % 8)) % 1048576);
  CK(hipMemcpy(dIdx, h.data(), n * 4, hipMemcpyHostToDevice));
  timeit([&] { hipLaunchKernelGGL(k_gather<PER>, grid, dim3(256), 0, 0, dIdx, dTab, dOut, n); },
         "global gather, 8 lanes per 64B line (1M table)");
  return 0;
}
