// pdlp_setup.hip — GPU-side formulate / scale / transpose / slab layout (see pdlp_setup.hpp).
// One-off streaming passes and rocPRIM radix sorts; everything order-sensitive is
// done by one thread per major in the reference's traversal order so that the
// result is bit-identical to the host path and to the reference.
#include "pdlp_setup.hpp"

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <limits>

namespace pdlp {

namespace {

constexpr int kT = 256;
inline int gridFor(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + kT - 1) / kT, 65535 * 16)); }
#define GSTRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

constexpr double kBoundInf = 1e20;

// ---- formulate -----------------------------------------------------------------
// infThresh / freeKind: 1e20 and kRowBound on the cuPDLP-C path (CupdlpWrapper.cpp:316-343); +inf and
// kRowFree on the HiPDLP path (pdhg.cc:175-197)
__device__ __forceinline__ bool isEqKind(int k) { return k == kRowEq || k == kRowBound || k == kRowFree; }
__global__ void k_row_classify(const double* __restrict__ lo, const double* __restrict__ up, int m, double infThresh,
                               int freeKind, int32_t* kind, int32_t* eqFlag, int32_t* ineqFlag, int32_t* slackFlag) {
  GSTRIDE(i, m) {
    const bool hl = lo[i] > -infThresh, hu = up[i] < infThresh;
    int k;
    if (hl && hu && lo[i] == up[i]) k = kRowEq;
    else if (hl && !hu) k = kRowGeq;
    else if (!hl && hu) k = kRowLeq;
    else if (hl && hu) k = kRowBound;  // ranged: a'x - z = 0 with a bounded slack
    else k = freeKind;                 // free: the same, unbounded slack
    kind[i] = k;
    eqFlag[i] = isEqKind(k) ? 1 : 0;
    ineqFlag[i] = (k == kRowLeq || k == kRowGeq) ? 1 : 0;
    slackFlag[i] = (k == kRowBound || k == kRowFree) ? 1 : 0;
  }
}

__device__ __forceinline__ double clampInfLo(double v) { return v < -kBoundInf ? -INFINITY : v; }
__device__ __forceinline__ double clampInfUp(double v) { return v > kBoundInf ? INFINITY : v; }

__global__ void k_row_finish(const double* __restrict__ lo, const double* __restrict__ up, const int32_t* kind,
                             const int32_t* eqRank, const int32_t* ineqRank, const int32_t* slackRank, int m, int n0,
                             int nEq, int64_t nnz0, int32_t* rowNewIdx, double* rhs, double* cost, double* lower,
                             double* upper, int32_t* cscBeg, int32_t* cscIdx, int32_t* cscCol, double* cscVal,
                             double* rowUpper /* HiPDLP form only, else nullptr */) {
  GSTRIDE(i, m) {
    const int k = kind[i];
    const int ni = isEqKind(k) ? eqRank[i] : nEq + ineqRank[i];
    rowNewIdx[i] = ni;
    double r;
    if (k == kRowEq) r = lo[i];
    else if (k == kRowBound || k == kRowFree) r = 0.0;
    else if (k == kRowLeq) r = -up[i];
    else r = lo[i];
    rhs[ni] = r;
    if (rowUpper) rowUpper[ni] = (k == kRowEq) ? up[i] : ((k == kRowBound || k == kRowFree) ? 0.0 : INFINITY);
    if (k == kRowBound || k == kRowFree) {
      const int j = n0 + slackRank[i];
      cost[j] = 0.0;
      lower[j] = rowUpper ? lo[i] : clampInfLo(lo[i]);
      upper[j] = rowUpper ? up[i] : clampInfUp(up[i]);
      const int64_t p = nnz0 + slackRank[i];
      cscBeg[j] = (int32_t)p;
      cscIdx[p] = ni;
      cscCol[p] = j;
      cscVal[p] = -1.0;
    }
  }
}

__global__ void k_col_setup(const double* __restrict__ c, const double* __restrict__ lo, const double* __restrict__ up,
                            const int32_t* __restrict__ aStart, int n0, double sense, int clampInf, double* cost,
                            double* lower, double* upper, int32_t* cscBeg) {
  GSTRIDE(j, n0) {
    cost[j] = c[j] * sense;
    lower[j] = clampInf ? clampInfLo(lo[j]) : lo[j];
    upper[j] = clampInf ? clampInfUp(up[j]) : up[j];
    cscBeg[j] = aStart[j];
  }
}

// Reference entry order inside a column: equality-type rows first, then inequality rows (LEQ negated).
__global__ void k_col_entries(const int32_t* __restrict__ aStart, const int32_t* __restrict__ aIndex,
                              const double* __restrict__ aValue, const int32_t* __restrict__ kind,
                              const int32_t* __restrict__ rowNewIdx, int n0, int m, int32_t* cscIdx, int32_t* cscCol,
                              double* cscVal, int32_t* badFlag) {
  GSTRIDE(j, n0) {
    const int b = aStart[j], e = aStart[j + 1];
    int k = b;
    for (int p = b; p < e; ++p) {
      const int r = aIndex[p];
      if (r < 0 || r >= m) { *badFlag = 1; continue; }
      const int t = kind[r];
      if (isEqKind(t)) { cscIdx[k] = rowNewIdx[r]; cscCol[k] = (int)j; cscVal[k] = aValue[p]; ++k; }
    }
    for (int p = b; p < e; ++p) {
      const int r = aIndex[p];
      if (r < 0 || r >= m) continue;
      const int t = kind[r];
      if (t == kRowLeq) { cscIdx[k] = rowNewIdx[r]; cscCol[k] = (int)j; cscVal[k] = -aValue[p]; ++k; }
      else if (t == kRowGeq) { cscIdx[k] = rowNewIdx[r]; cscCol[k] = (int)j; cscVal[k] = aValue[p]; ++k; }
    }
  }
}

// ---- transposes ----------------------------------------------------------------
__global__ void k_iota(int32_t* a, int64_t n) { GSTRIDE(i, n) a[i] = (int32_t)i; }
__global__ void k_fill_d(double* a, double v, int64_t n) { GSTRIDE(i, n) a[i] = v; }

// out entries q: minor = minorOf[perm[q]], val = valIn[perm[q]]
__global__ void k_gather_entries(const int32_t* __restrict__ perm, const int32_t* __restrict__ minorIn,
                                 const double* __restrict__ valIn, int64_t nnz, int32_t* minorOut, double* valOut) {
  GSTRIDE(q, nnz) {
    const int p = perm[q];
    minorOut[q] = minorIn[p];
    valOut[q] = valIn[p];
  }
}
// beg[i] = first q with sortedMajor[q] >= i  (i = 0..nMajor)
__global__ void k_lower_bounds(const int32_t* __restrict__ sortedKeys, int64_t nnz, int64_t nQueries, int32_t* out) {
  GSTRIDE(i, nQueries) {
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)(uint32_t)sortedKeys[mid] < i) lo = mid + 1; else hi = mid;
    }
    out[i] = (int32_t)lo;
  }
}

// ---- scaling --------------------------------------------------------------------
// one thread per major: max |a| (Ruiz) or left-to-right sum |a| (Pock-Chambolle, alpha = 1)
template <bool SUM>
__global__ void k_major_reduce(const int32_t* __restrict__ beg, const double* __restrict__ val, int nMajor,
                               double* out) {
  GSTRIDE(r, nMajor) {
    double s = 0.0;
    for (int p = beg[r]; p < beg[r + 1]; ++p) {
      const double a = fabs(val[p]);
      if (SUM) s += a;
      else if (a > s) s = a;
    }
    out[r] = (s == 0.0) ? 1.0 : sqrt(s);  // sqrt(max) / sqrt(sum^(1/1)); empty or zero majors -> 1
  }
}
// qdiag (QP, may be null): x = x'/cs  =>  1/2 q x^2 = 1/2 (q / cs^2) x'^2, two divisions per pass as applyScaling does
__global__ void k_apply_cols(const double* __restrict__ cs, int n, double* cost, double* lower, double* upper,
                             double* colScale, double* qdiag) {
  GSTRIDE(j, n) {
    cost[j] /= cs[j];
    lower[j] *= cs[j];
    upper[j] *= cs[j];
    colScale[j] *= cs[j];
    if (qdiag) qdiag[j] = (qdiag[j] / cs[j]) / cs[j];
  }
}
__global__ void k_apply_rows(const double* __restrict__ rs, int m, double* rhs, double* rowScale) {
  GSTRIDE(i, m) {
    rhs[i] /= rs[i];
    rowScale[i] *= rs[i];
  }
}
// a = (a / rs[row]) / cs[col], the operation order of scale_problem (cupdlp_scaling.c:17-45)
__global__ void k_scale_vals(const int32_t* __restrict__ rowOf, const int32_t* __restrict__ colOf,
                             const double* __restrict__ rs, const double* __restrict__ cs, int64_t nnz, double* val) {
  GSTRIDE(p, nnz) val[p] = (val[p] / rs[rowOf[p]]) / cs[colOf[p]];
}
// HiPDLP (scaling.cc): MODE 0 sqrt(max|a|) (Ruiz), 1 sqrt(sum|a|) (Pock-Chambolle, alpha 1), 2 sqrt(sqrt(sum a^2)) (L2);
// zero or empty majors -> 1
template <int MODE>
__global__ void k_major_reduce_h(const int32_t* __restrict__ beg, const double* __restrict__ val, int nMajor,
                                 double* out) {
  GSTRIDE(r, nMajor) {
    double s = 0.0;
    for (int p = beg[r]; p < beg[r + 1]; ++p) {
      const double a = fabs(val[p]);
      if (MODE == 0) s = a > s ? a : s;
      else if (MODE == 1) s += a;
      else s += val[p] * val[p];
    }
    if (MODE == 0) out[r] = (s == 0.0) ? 1.0 : sqrt(s);
    else if (MODE == 1) out[r] = s > 0.0 ? sqrt(s) : 1.0;
    else out[r] = s > 0.0 ? sqrt(sqrt(s)) : 1.0;
  }
}
__global__ void k_apply_rows_h(const double* __restrict__ rs, int m, double* rowLower, double* rowUpper,
                               double* rowScale) {
  GSTRIDE(i, m) {
    if (rowLower[i] > -INFINITY) rowLower[i] /= rs[i];
    if (rowUpper[i] < INFINITY) rowUpper[i] /= rs[i];
    rowScale[i] *= rs[i];
  }
}
__global__ void k_apply_cols_h(const double* __restrict__ cs, int n, double* cost, double* lower, double* upper,
                               double* colScale) {
  GSTRIDE(j, n) {
    cost[j] /= cs[j];
    if (lower[j] > -INFINITY) lower[j] *= cs[j];
    if (upper[j] < INFINITY) upper[j] *= cs[j];
    colScale[j] *= cs[j];
  }
}
// a /= (rs[row] * cs[col]): Scaling::applyScaling, scaling.cc:251-259
__global__ void k_scale_vals_h(const int32_t* __restrict__ rowOf, const int32_t* __restrict__ colOf,
                               const double* __restrict__ rs, const double* __restrict__ cs, int64_t nnz, double* val) {
  GSTRIDE(p, nnz) val[p] /= (rs[rowOf[p]] * cs[colOf[p]]);
}
__global__ void k_is_eq(const int32_t* __restrict__ kind, const int32_t* __restrict__ rowNewIdx, int m, uint8_t* isEq) {
  GSTRIDE(i, m) isEq[rowNewIdx[i]] = isEqKind(kind[i]) ? 1 : 0;
}
__global__ void k_absmax_partial(const double* __restrict__ val, int64_t nnz, double* partial) {
  __shared__ double sm[kT];
  double mx = 0.0;
  GSTRIDE(p, nnz) { const double a = fabs(val[p]); if (a > mx) mx = a; }
  sm[threadIdx.x] = mx;
  __syncthreads();
  for (int s = kT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s && sm[threadIdx.x + s] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

// ---- slab layout ------------------------------------------------------------------
// waveOf[r] = the wave that owns major r (waveBeg: first major of every wave, ascending; empty waves repeat a value)
__global__ void k_wave_of(const int32_t* __restrict__ waveBeg, int nWaves, int nMajor, int32_t* waveOf) {
  GSTRIDE(r, nMajor) {
    int lo = 0, hi = nWaves;  // last w with waveBeg[w] <= r
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (waveBeg[mid] <= (int)r) lo = mid; else hi = mid;
    }
    waveOf[r] = lo;
  }
}
__global__ void k_slab_keys(const int32_t* __restrict__ beg, const int32_t* __restrict__ major,
                            const int32_t* __restrict__ idx, int64_t nnz, const int32_t* __restrict__ waveOf, int S, int W,
                            int longLimit, uint32_t keyMax, uint32_t* keys) {
  GSTRIDE(p, nnz) {
    const int r = major[p];
    const int len = beg[r + 1] - beg[r];
    keys[p] = len > longLimit ? keyMax : (uint32_t)waveOf[r] * (uint32_t)S + ((uint32_t)idx[p] >> W);
  }
}
__global__ void k_slab_entries(const int32_t* __restrict__ perm, const int32_t* __restrict__ major,
                               const int32_t* __restrict__ idx, const double* __restrict__ valIn, int64_t nShort,
                               const int32_t* __restrict__ waveOf, const int32_t* __restrict__ waveBeg, int minorBits,
                               uint32_t* ent, double* val) {
  GSTRIDE(q, nShort) {
    const int p = perm[q];
    const int r = major[p];
    ent[q] = ((uint32_t)(r - waveBeg[waveOf[r]]) << minorBits) | (uint32_t)idx[p];
    val[q] = valIn[p];
  }
}
// wavePtr[w] = first q with sortedKey[q] >= w*S  (w = 0..nWaves; keyMax sorts after every wave)
__global__ void k_wave_ptr(const uint32_t* __restrict__ sortedKeys, int64_t nnz, int nWaves, int S, int32_t* out) {
  GSTRIDE(w, nWaves + 1) {
    const uint64_t key = (uint64_t)w * (uint64_t)S;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((uint64_t)sortedKeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    out[w] = (int32_t)lo;
  }
}
__global__ void k_long_mask(const int32_t* __restrict__ beg, int nMajor, int longLimit, int nWords,
                            uint32_t* mask, int32_t* longFlag) {
  GSTRIDE(w, nWords) {  // bit r of the mask: major r is a long one
    uint32_t bits = 0;
    for (int k = 0; k < 32; ++k) {
      const int64_t r = w * 32 + k;
      if (r < nMajor && beg[r + 1] - beg[r] > longLimit) bits |= 1u << k;
    }
    mask[w] = bits;
  }
  GSTRIDE(r, nMajor) longFlag[r] = (beg[r + 1] - beg[r] > longLimit) ? 1 : 0;
}
// long-major compaction
__global__ void k_long_rows(const int32_t* __restrict__ beg, const int32_t* __restrict__ longFlag,
                            const int32_t* __restrict__ longRank, int nMajor, int32_t* longMap, int32_t* longLen) {
  GSTRIDE(r, nMajor) if (longFlag[r]) { longMap[longRank[r]] = (int)r; longLen[longRank[r]] = beg[r + 1] - beg[r]; }
}
__global__ void k_long_copy(const int32_t* __restrict__ beg, const int32_t* __restrict__ idx,
                            const double* __restrict__ val, const int32_t* __restrict__ longMap,
                            const int32_t* __restrict__ longBeg, int32_t* idxOut, double* valOut) {
  const int c = blockIdx.x;  // one block per long major
  const int r = longMap[c];
  const int src = beg[r], len = beg[r + 1] - beg[r], dst = longBeg[c];
  for (int k = threadIdx.x; k < len; k += blockDim.x) { idxOut[dst + k] = idx[src + k]; valOut[dst + k] = val[src + k]; }
}

// ---- rocPRIM helpers -----------------------------------------------------------------
void exclusiveSum(const int32_t* in, int32_t* out, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  size_t bytes = 0;
  PDLP_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), s));
  DeviceArray<char> tmp;
  tmp.alloc(bytes);
  PDLP_HIP(rocprim::exclusive_scan(tmp.get(), bytes, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), s));
  PDLP_HIP(hipStreamSynchronize(s));
}
int bitsFor(uint64_t maxKey) {
  int b = 1;
  while (b < 32 && (maxKey >> b) != 0) ++b;
  return b;
}
// stable sort of (key, position) pairs; returns sorted keys and the permutation
void sortByKey(const uint32_t* keysIn, int64_t n, uint64_t maxKey, DeviceArray<uint32_t>& keysOut,
               DeviceArray<int32_t>& perm, hipStream_t s) {
  keysOut.alloc((size_t)n);
  perm.alloc((size_t)n);
  if (n <= 0) return;
  DeviceArray<int32_t> iota;
  iota.alloc((size_t)n);
  hipLaunchKernelGGL(k_iota, dim3(gridFor(n)), dim3(kT), 0, s, iota.get(), n);
  size_t bytes = 0;
  const unsigned endBit = (unsigned)bitsFor(maxKey);
  PDLP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keysIn, keysOut.get(), iota.get(), perm.get(), (size_t)n, 0u,
                                     endBit, s));
  DeviceArray<char> tmp;
  tmp.alloc(bytes);
  PDLP_HIP(rocprim::radix_sort_pairs(tmp.get(), bytes, keysIn, keysOut.get(), iota.get(), perm.get(), (size_t)n, 0u,
                                     endBit, s));
  PDLP_HIP(hipStreamSynchronize(s));
}

template <typename T>
T fetchOne(const T* dev, hipStream_t s) {
  T v;
  PDLP_HIP(hipMemcpyAsync(&v, dev, sizeof(T), hipMemcpyDeviceToHost, s));
  PDLP_HIP(hipStreamSynchronize(s));
  return v;
}

// CSR <-> CSC by a stable sort on the minor index: the output majors keep their
// entries in ascending input-major order (what cupdlp_dcs_transpose produces).
void transposeOnDevice(const int32_t* majorIn, const int32_t* minorIn, const double* valIn, int64_t nnz,
                       int32_t nMajorOut, int32_t nMinorOut, hipStream_t s, DeviceCsrData& out) {
  out.nMajor = nMajorOut;
  out.nMinor = nMinorOut;
  out.nnz = nnz;
  DeviceArray<uint32_t> keys;
  DeviceArray<int32_t> perm;
  sortByKey(reinterpret_cast<const uint32_t*>(minorIn), nnz, (uint64_t)std::max(nMajorOut, 1), keys, perm, s);
  out.major.alloc((size_t)nnz);
  out.idx.alloc((size_t)nnz + 1);
  out.val.alloc((size_t)nnz + 1);
  out.idx.zero(s);
  out.val.zero(s);
  out.beg.alloc((size_t)nMajorOut + 1);
  if (nnz > 0) {
    PDLP_HIP(hipMemcpyAsync(out.major.get(), keys.get(), sizeof(int32_t) * nnz, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_gather_entries, dim3(gridFor(nnz)), dim3(kT), 0, s, perm.get(), majorIn, valIn, nnz,
                       out.idx.get(), out.val.get());
  }
  hipLaunchKernelGGL(k_lower_bounds, dim3(gridFor(nMajorOut + 1)), dim3(kT), 0, s,
                     reinterpret_cast<const int32_t*>(keys.get()), nnz, (int64_t)nMajorOut + 1, out.beg.get());
  PDLP_HIP(hipStreamSynchronize(s));
}

}  // namespace

void gpuPrepare(const pdlp_problem_t& P, bool doScale, hipStream_t s, DeviceProblem& D, const HipdlpSetup* hp) {
  const bool H = hp != nullptr;  // HiPDLP form (pdhg.cc:152-357 + scaling.cc) instead of the cuPDLP-C one
  validateProblem(P);
  const int32_t n0 = P.num_col, m = P.num_row;
  const int64_t nnz0 = n0 > 0 ? P.a_start[n0] : 0;
  if (nnz0 > 0 && (!P.a_index || !P.a_value)) throw std::runtime_error("null matrix arrays");
  for (int32_t j = 0; j < n0; ++j)
    if (P.a_start[j + 1] < P.a_start[j]) throw std::runtime_error("a_start not monotone");
  D.n0 = n0;
  D.m = m;
  D.offset = P.offset;
  D.sense = P.sense < 0 ? -1.0 : 1.0;
  const double costSense = H ? 1.0 : D.sense;  // HiPDLP does not apply the objective sense (pdhg.cc:171)

  // upload the caller's arrays
  DeviceArray<int32_t> aStart, aIndex;
  DeviceArray<double> aValue, cIn, clIn, cuIn, rlIn, ruIn;
  aStart.alloc((size_t)n0 + 1); aIndex.alloc((size_t)nnz0); aValue.alloc((size_t)nnz0);
  cIn.alloc(n0); clIn.alloc(n0); cuIn.alloc(n0); rlIn.alloc(m); ruIn.alloc(m);
  aStart.upload(P.a_start, (size_t)n0 + 1, s); aIndex.upload(P.a_index, (size_t)nnz0, s);
  aValue.upload(P.a_value, (size_t)nnz0, s);
  cIn.upload(P.col_cost, n0, s); clIn.upload(P.col_lower, n0, s); cuIn.upload(P.col_upper, n0, s);
  rlIn.upload(P.row_lower, m, s); ruIn.upload(P.row_upper, m, s);

  // rows: classify, rank, permute
  DeviceArray<int32_t> kind, eqF, inF, slF, eqR, inR, slR, rowNew;
  kind.alloc(m); eqF.alloc(m); inF.alloc(m); slF.alloc(m); eqR.alloc(m); inR.alloc(m); slR.alloc(m); rowNew.alloc(m);
  hipLaunchKernelGGL(k_row_classify, dim3(gridFor(m)), dim3(kT), 0, s, rlIn.get(), ruIn.get(), m,
                     H ? (double)INFINITY : kBoundInf, H ? (int)kRowFree : (int)kRowBound, kind.get(), eqF.get(),
                     inF.get(), slF.get());
  exclusiveSum(eqF.get(), eqR.get(), m, s);
  exclusiveSum(inF.get(), inR.get(), m, s);
  exclusiveSum(slF.get(), slR.get(), m, s);
  int32_t nEq = 0, nSlack = 0;
  if (m > 0) {
    nEq = fetchOne(eqR.get() + (m - 1), s) + fetchOne(eqF.get() + (m - 1), s);
    nSlack = fetchOne(slR.get() + (m - 1), s) + fetchOne(slF.get() + (m - 1), s);
  }
  if ((int64_t)n0 + nSlack > std::numeric_limits<int32_t>::max() ||
      nnz0 + nSlack > std::numeric_limits<int32_t>::max())
    throw std::runtime_error("problem exceeds 32-bit index range");
  const int32_t n = n0 + nSlack;
  const int64_t nnz = nnz0 + nSlack;
  D.n = n; D.nEqs = nEq; D.nnz = nnz;

  D.cost.alloc(n); D.lower.alloc(n); D.upper.alloc(n); D.rhs.alloc(m); D.colScale.alloc(n); D.rowScale.alloc(m);
  if (H) { D.rowUpper.alloc(m); D.rowIsEq.alloc((size_t)m); }
  DeviceArray<int32_t> cscBeg, cscIdx, cscCol, bad;
  DeviceArray<double> cscVal;
  cscBeg.alloc((size_t)n + 1); cscIdx.alloc((size_t)nnz); cscCol.alloc((size_t)nnz); cscVal.alloc((size_t)nnz);
  bad.alloc(1);
  bad.zero(s);
  hipLaunchKernelGGL(k_col_setup, dim3(gridFor(n0)), dim3(kT), 0, s, cIn.get(), clIn.get(), cuIn.get(), aStart.get(),
                     n0, costSense, H ? 0 : 1, D.cost.get(), D.lower.get(), D.upper.get(), cscBeg.get());
  hipLaunchKernelGGL(k_row_finish, dim3(gridFor(m)), dim3(kT), 0, s, rlIn.get(), ruIn.get(), kind.get(), eqR.get(),
                     inR.get(), slR.get(), m, n0, nEq, nnz0, rowNew.get(), D.rhs.get(), D.cost.get(), D.lower.get(),
                     D.upper.get(), cscBeg.get(), cscIdx.get(), cscCol.get(), cscVal.get(),
                     H ? D.rowUpper.get() : nullptr);
  if (H) hipLaunchKernelGGL(k_is_eq, dim3(gridFor(m)), dim3(kT), 0, s, kind.get(), rowNew.get(), m, D.rowIsEq.get());
  {
    const int32_t last = (int32_t)nnz;
    PDLP_HIP(hipMemcpyAsync(cscBeg.get() + n, &last, sizeof(int32_t), hipMemcpyHostToDevice, s));
    PDLP_HIP(hipStreamSynchronize(s));
  }
  hipLaunchKernelGGL(k_col_entries, dim3(gridFor(n0)), dim3(kT), 0, s, aStart.get(), aIndex.get(), aValue.get(),
                     kind.get(), rowNew.get(), n0, m, cscIdx.get(), cscCol.get(), cscVal.get(), bad.get());
  if (fetchOne(bad.get(), s) != 0) throw std::runtime_error("row index out of range");

  // host copies of the row bookkeeping; norms of the UNSCALED formulated data, summed
  // left to right on the host exactly as Init_Scaling does (cupdlp_scaling.c:395-425)
  D.rowKind.resize(m);
  D.rowNewIdx.resize(m);
  kind.download(D.rowKind.data(), m, s);
  rowNew.download(D.rowNewIdx.data(), m, s);
  PDLP_HIP(hipStreamSynchronize(s));
  {
    double sc = 0.0;
    for (int32_t j = 0; j < n0; ++j) { const double v = P.col_cost[j] * costSense; sc += v * v; }
    D.normCost = std::sqrt(sc);  // slack costs are 0
    double sr = 0.0;  // permuted order: equality-type rows first, then inequalities
    for (int32_t i = 0; i < m; ++i)
      if (D.rowKind[i] == kRowEq) sr += P.row_lower[i] * P.row_lower[i];
      else if (D.rowKind[i] == kRowBound || D.rowKind[i] == kRowFree) sr += 0.0;
    for (int32_t i = 0; i < m; ++i)
      if (D.rowKind[i] == kRowLeq) sr += (-P.row_upper[i]) * (-P.row_upper[i]);
      else if (D.rowKind[i] == kRowGeq) sr += P.row_lower[i] * P.row_lower[i];
    D.normRhs = std::sqrt(sr);
  }

  // A by rows (ascending column): stable sort of the column-major entries by row
  transposeOnDevice(cscCol.get(), cscIdx.get(), cscVal.get(), nnz, m, n, s, D.A);

  // scaling: Ruiz x10 in the infinity norm, then Pock-Chambolle alpha = 1 (cupdlp_scaling.c:47-231);
  // both copies of the matrix (reference-order CSC for the column passes, CSR for the row
  // passes) receive the same two divisions, so they stay bit-identical
  hipLaunchKernelGGL(k_fill_d, dim3(gridFor(n)), dim3(kT), 0, s, D.colScale.get(), 1.0, (int64_t)n);
  hipLaunchKernelGGL(k_fill_d, dim3(gridFor(m)), dim3(kT), 0, s, D.rowScale.get(), 1.0, (int64_t)m);
  if (H) {
    // HiPDLP: columns are kept with ascending row index (pdhg.cc:311), so the column passes run on A' and
    // both copies receive the identical division a /= (rs * cs) (scaling.cc:251-259)
    transposeOnDevice(D.A.major.get(), D.A.idx.get(), D.A.val.get(), nnz, n, m, s, D.At);
    if (doScale && (hp->ruiz || hp->pc || hp->l2)) {
      DeviceArray<double> cs, rs;
      cs.alloc(n);
      rs.alloc(m);
      auto pass = [&](int mode) {
        if (mode == 0) {
          hipLaunchKernelGGL(k_major_reduce_h<0>, dim3(gridFor(n)), dim3(kT), 0, s, D.At.beg.get(), D.At.val.get(), n, cs.get());
          hipLaunchKernelGGL(k_major_reduce_h<0>, dim3(gridFor(m)), dim3(kT), 0, s, D.A.beg.get(), D.A.val.get(), m, rs.get());
        } else if (mode == 1) {
          hipLaunchKernelGGL(k_major_reduce_h<1>, dim3(gridFor(n)), dim3(kT), 0, s, D.At.beg.get(), D.At.val.get(), n, cs.get());
          hipLaunchKernelGGL(k_major_reduce_h<1>, dim3(gridFor(m)), dim3(kT), 0, s, D.A.beg.get(), D.A.val.get(), m, rs.get());
        } else {
          hipLaunchKernelGGL(k_major_reduce_h<2>, dim3(gridFor(n)), dim3(kT), 0, s, D.At.beg.get(), D.At.val.get(), n, cs.get());
          hipLaunchKernelGGL(k_major_reduce_h<2>, dim3(gridFor(m)), dim3(kT), 0, s, D.A.beg.get(), D.A.val.get(), m, rs.get());
        }
        hipLaunchKernelGGL(k_apply_cols_h, dim3(gridFor(n)), dim3(kT), 0, s, cs.get(), n, D.cost.get(), D.lower.get(),
                           D.upper.get(), D.colScale.get());
        hipLaunchKernelGGL(k_apply_rows_h, dim3(gridFor(m)), dim3(kT), 0, s, rs.get(), m, D.rhs.get(),
                           D.rowUpper.get(), D.rowScale.get());
        // A: major = row, idx = column;  A': major = column, idx = row
        hipLaunchKernelGGL(k_scale_vals_h, dim3(gridFor(nnz)), dim3(kT), 0, s, D.A.major.get(), D.A.idx.get(), rs.get(),
                           cs.get(), nnz, D.A.val.get());
        hipLaunchKernelGGL(k_scale_vals_h, dim3(gridFor(nnz)), dim3(kT), 0, s, D.At.idx.get(), D.At.major.get(), rs.get(),
                           cs.get(), nnz, D.At.val.get());
      };
      if (hp->ruiz) for (int it = 0; it < hp->ruizIters; ++it) pass(0);
      if (hp->pc) pass(1);
      if (hp->l2) pass(2);
      D.scaled = true;
      PDLP_HIP(hipStreamSynchronize(s));
    }
    D.hColScale.resize(n);
    D.hRowScale.resize(m);
    D.colScale.download(D.hColScale.data(), n, s);
    D.rowScale.download(D.hRowScale.data(), m, s);
    PDLP_HIP(hipStreamSynchronize(s));
    return;
  }
  // QP (cuPDLP-C form): the diagonal of Q, with the objective sense, rides along with the column scaling
  {
    std::vector<double> q;
    extractDiagonalHessian(P, D.sense, n, q);
    if (!q.empty()) {
      D.qdiag.alloc((size_t)n);
      D.qdiag.upload(q.data(), (size_t)n, s);
      PDLP_HIP(hipStreamSynchronize(s));  // q goes out of scope
    }
  }
  if (doScale) {
    DeviceArray<double> cs, rs;
    cs.alloc(n);
    rs.alloc(m);
    auto pass = [&](bool sum) {
      if (sum) {
        hipLaunchKernelGGL(k_major_reduce<true>, dim3(gridFor(n)), dim3(kT), 0, s, cscBeg.get(), cscVal.get(), n, cs.get());
        hipLaunchKernelGGL(k_major_reduce<true>, dim3(gridFor(m)), dim3(kT), 0, s, D.A.beg.get(), D.A.val.get(), m, rs.get());
      } else {
        hipLaunchKernelGGL(k_major_reduce<false>, dim3(gridFor(n)), dim3(kT), 0, s, cscBeg.get(), cscVal.get(), n, cs.get());
        hipLaunchKernelGGL(k_major_reduce<false>, dim3(gridFor(m)), dim3(kT), 0, s, D.A.beg.get(), D.A.val.get(), m, rs.get());
      }
      hipLaunchKernelGGL(k_apply_cols, dim3(gridFor(n)), dim3(kT), 0, s, cs.get(), n, D.cost.get(), D.lower.get(),
                         D.upper.get(), D.colScale.get(), D.qdiag.size() ? D.qdiag.get() : (double*)nullptr);
      hipLaunchKernelGGL(k_apply_rows, dim3(gridFor(m)), dim3(kT), 0, s, rs.get(), m, D.rhs.get(), D.rowScale.get());
      hipLaunchKernelGGL(k_scale_vals, dim3(gridFor(nnz)), dim3(kT), 0, s, cscIdx.get(), cscCol.get(), rs.get(),
                         cs.get(), nnz, cscVal.get());
      hipLaunchKernelGGL(k_scale_vals, dim3(gridFor(nnz)), dim3(kT), 0, s, D.A.major.get(), D.A.idx.get(), rs.get(),
                         cs.get(), nnz, D.A.val.get());
    };
    for (int it = 0; it < 10; ++it) pass(false);
    pass(true);
    D.scaled = true;
    PDLP_HIP(hipStreamSynchronize(s));
  }

  // max |a_ij| of the scaled matrix (initial step size, cupdlp_step.c:360-365)
  {
    const int nb = 1024;
    DeviceArray<double> part;
    part.alloc(nb);
    hipLaunchKernelGGL(k_absmax_partial, dim3(nb), dim3(kT), 0, s, cscVal.get(), nnz, part.get());
    std::vector<double> h(nb);
    part.download(h.data(), nb, s);
    PDLP_HIP(hipStreamSynchronize(s));
    D.matNormInf = 0.0;
    for (double v : h) D.matNormInf = std::max(D.matNormInf, v);
  }

  // A' by columns with ascending row: stable sort of the row-major entries by column
  transposeOnDevice(D.A.major.get(), D.A.idx.get(), D.A.val.get(), nnz, n, m, s, D.At);

  // host copies: scale vectors (postsolve / hot start) and the left-to-right sums of
  // the scaled c and b that PDHG_Init_Step_Sizes needs (cupdlp_step.c:349-358)
  D.hColScale.resize(n);
  D.hRowScale.resize(m);
  std::vector<double> hc(n), hb(m);
  D.colScale.download(D.hColScale.data(), n, s);
  D.rowScale.download(D.hRowScale.data(), m, s);
  D.cost.download(hc.data(), n, s);
  D.rhs.download(hb.data(), m, s);
  PDLP_HIP(hipStreamSynchronize(s));
  D.sumCost2 = 0.0;
  for (double v : hc) D.sumCost2 += v * v;
  D.sumRhs2 = 0.0;
  for (double v : hb) D.sumRhs2 += v * v;
}

// pdlp_host.cpp slabColdCounts on the device: how many majors touch every minor, then the cold entries of every major
__global__ void k_minor_count(const int32_t* __restrict__ idx, int64_t nnz, int32_t* count) {
  GSTRIDE(p, nnz) atomicAdd(count + idx[p], 1);
}
__global__ void k_major_cold(const int32_t* __restrict__ beg, const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                             int nMajor, int longLimit, int far, int hot, int32_t* cold) {
  GSTRIDE(r, nMajor) {
    const int p0 = beg[r], len = beg[r + 1] - beg[r];
    int c = 0;
    if (len >= 2 && len <= longLimit) {
      const int mid = idx[p0 + len / 2];
      for (int p = p0; p < p0 + len; ++p) {
        const int j = idx[p];
        const int d = j > mid ? j - mid : mid - j;
        if (d >= far && count[j] <= hot) ++c;
      }
    }
    cold[r] = c;
  }
}

void gpuSlabPartition(const DeviceCsrData& M, int32_t longLimit, int32_t majorCost, hipStream_t s, DeviceSlabLayout& L) {
  // the partition by work is sequential and cheap: on the host, from the major starts (4 bytes per major over PCIe),
  // by the same function the host-side build uses
  std::vector<int32_t> hb((size_t)M.nMajor + 1), hc((size_t)std::max(M.nMajor, 1), 0);
  DeviceArray<int32_t> count, cold;
  count.alloc((size_t)std::max(M.nMinor, 1));
  cold.alloc(hc.size());
  count.zero(s);
  if (M.nnz > 0) hipLaunchKernelGGL(k_minor_count, dim3(gridFor(M.nnz)), dim3(kT), 0, s, M.idx.get(), M.nnz, count.get());
  if (M.nMajor > 0)
    hipLaunchKernelGGL(k_major_cold, dim3(gridFor(M.nMajor)), dim3(kT), 0, s, M.beg.get(), M.idx.get(), count.get(), M.nMajor, longLimit,
                       kSlabFar, kSlabHotCount, cold.get());
  M.beg.download(hb.data(), hb.size(), s);
  if (M.nMajor > 0) cold.download(hc.data(), (size_t)M.nMajor, s);
  PDLP_HIP(hipStreamSynchronize(s));
  SlabPartition P = slabPartition(hb.data(), hc.data(), M.nMajor, M.nMinor, longLimit, majorCost);
  L.rowsPerBlock = P.maxRowsPerBlock;
  L.nBlocks = P.nBlocks;
  L.minorBits = P.minorBits;
  L.hostWaveBeg = std::move(P.waveBeg);
  L.waveBeg.alloc(L.hostWaveBeg.size());
  L.waveBeg.upload(L.hostWaveBeg.data(), L.hostWaveBeg.size(), s);
  PDLP_HIP(hipStreamSynchronize(s));
}

void gpuBuildSlabLayout(const DeviceCsrData& M, int32_t longLimit, int32_t W, int32_t majorCost, hipStream_t s, DeviceSlabLayout& L) {
  const int32_t nMajor = M.nMajor, nMinor = M.nMinor;
  const int64_t nnz = M.nnz;
  if (L.hostWaveBeg.empty()) gpuSlabPartition(M, longLimit, majorCost, s, L);
  const int32_t nWaves = L.nBlocks * kSlabWavesPerBlock;
  DeviceArray<int32_t> waveOf;
  waveOf.alloc((size_t)std::max(nMajor, 1));
  if (nMajor > 0) hipLaunchKernelGGL(k_wave_of, dim3(gridFor(nMajor)), dim3(kT), 0, s, L.waveBeg.get(), nWaves, nMajor, waveOf.get());
  const int32_t S = std::max(1, (int32_t)(((int64_t)nMinor + ((int64_t)1 << W) - 1) >> W));
  const int64_t nSeg = (int64_t)nWaves * S;
  if (nSeg >= (int64_t)0x7fffffff) throw std::runtime_error("slab layout: too many segments");
  const uint32_t keyMax = (uint32_t)nSeg;  // sorts after every real segment

  // long majors: mask, map, compact CSR
  const int nWords = (nMajor + 31) / 32 + 1;
  L.longMask.alloc((size_t)nWords);
  DeviceArray<int32_t> longFlag, longRank;
  longFlag.alloc((size_t)std::max(nMajor, 1));
  longRank.alloc((size_t)std::max(nMajor, 1));
  hipLaunchKernelGGL(k_long_mask, dim3(gridFor(std::max(nWords, nMajor))), dim3(kT), 0, s, M.beg.get(), nMajor,
                     longLimit, nWords, L.longMask.get(), longFlag.get());
  exclusiveSum(longFlag.get(), longRank.get(), nMajor, s);
  L.nLong = nMajor > 0 ? fetchOne(longRank.get() + (nMajor - 1), s) + fetchOne(longFlag.get() + (nMajor - 1), s) : 0;
  L.longMap.alloc((size_t)std::max(L.nLong, 1));
  L.longCsr.nMajor = L.nLong;
  L.longCsr.nMinor = nMinor;
  L.hostLongBeg.assign(1, 0);
  if (L.nLong > 0) {
    DeviceArray<int32_t> longLen;
    longLen.alloc(L.nLong);
    hipLaunchKernelGGL(k_long_rows, dim3(gridFor(nMajor)), dim3(kT), 0, s, M.beg.get(), longFlag.get(), longRank.get(),
                       nMajor, L.longMap.get(), longLen.get());
    std::vector<int32_t> hl(L.nLong);
    longLen.download(hl.data(), L.nLong, s);
    PDLP_HIP(hipStreamSynchronize(s));
    L.hostLongBeg.resize((size_t)L.nLong + 1);
    for (int32_t c = 0; c < L.nLong; ++c) L.hostLongBeg[c + 1] = L.hostLongBeg[c] + hl[c];
    const int64_t nnzLong = L.hostLongBeg[L.nLong];
    L.longCsr.nnz = nnzLong;
    L.longCsr.beg.alloc((size_t)L.nLong + 1);
    L.longCsr.beg.upload(L.hostLongBeg.data(), (size_t)L.nLong + 1, s);
    L.longCsr.idx.alloc((size_t)nnzLong + 1);
    L.longCsr.val.alloc((size_t)nnzLong + 1);
    L.longCsr.idx.zero(s);
    L.longCsr.val.zero(s);
    hipLaunchKernelGGL(k_long_copy, dim3(L.nLong), dim3(kT), 0, s, M.beg.get(), M.idx.get(), M.val.get(),
                       L.longMap.get(), L.longCsr.beg.get(), L.longCsr.idx.get(), L.longCsr.val.get());
    PDLP_HIP(hipStreamSynchronize(s));
  } else {
    L.longCsr.nnz = 0;
    L.longCsr.beg.alloc(1);
    L.longCsr.beg.zero(s);
    L.longCsr.idx.alloc(1);
    L.longCsr.val.alloc(1);
  }

  // short entries sorted by (wave, slab); the stable sort keeps (major, minor) order inside
  DeviceArray<uint32_t> keys, sortedKeys;
  DeviceArray<int32_t> perm;
  keys.alloc((size_t)std::max<int64_t>(nnz, 1));
  if (nnz > 0)
    hipLaunchKernelGGL(k_slab_keys, dim3(gridFor(nnz)), dim3(kT), 0, s, M.beg.get(), M.major.get(), M.idx.get(), nnz,
                       waveOf.get(), S, W, longLimit, keyMax, keys.get());
  sortByKey(keys.get(), nnz, (uint64_t)keyMax, sortedKeys, perm, s);
  L.wavePtr.alloc((size_t)nWaves + 1);
  hipLaunchKernelGGL(k_wave_ptr, dim3(gridFor(nWaves + 1)), dim3(kT), 0, s, sortedKeys.get(), nnz, nWaves, S,
                     L.wavePtr.get());
  L.nnzShort = fetchOne(L.wavePtr.get() + nWaves, s);
  L.ent.alloc((size_t)L.nnzShort + 1);
  L.val.alloc((size_t)L.nnzShort + 1);
  L.ent.zero(s);
  L.val.zero(s);
  if (L.nnzShort > 0)
    hipLaunchKernelGGL(k_slab_entries, dim3(gridFor(L.nnzShort)), dim3(kT), 0, s, perm.get(), M.major.get(),
                       M.idx.get(), M.val.get(), L.nnzShort, waveOf.get(), L.waveBeg.get(), L.minorBits, L.ent.get(), L.val.get());
  PDLP_HIP(hipStreamSynchronize(s));
}

}  // namespace pdlp
