/*
 * gpu_order.h — TEST INFRASTRUCTURE ONLY (shared by pdlp_oracle.c and hipdlp_oracle.c).
 *
 * The summation orders of the HIP kernels' reductions, restated on the CPU so that the oracles
 * can follow a GPU solve bit for bit ("device reduction order" mode, opt->reserved[0] == 1):
 * per-lane strided accumulation, 64-lane shuffle tree (waveSum), fixed-order sum of the 4 wave
 * results (blockSum), 4-chain sum of the per-block partials (reducePartials) —
 * highs_amd/csrc/pdlp_devfn.hpp, pdlp_kernels.hip (vector kernels), pdlp_halpern.hip.
 * Constants mirror pdlp_kernels.hpp: 256 lanes per block, vector grids capped at 2048 blocks.
 */
#ifndef ORACLE_GPU_ORDER_H_
#define ORACLE_GPU_ORDER_H_
#include <string.h>

enum { G_T = 256, G_WAVE = 64, G_CHUNK = 2048, G_CHUNK_SMALL = 512, G_MAXMAJ = 2048, G_MAXGRID = 2048 };
/* spmvChunkFor (pdlp_kernels.hpp): work-plan block size of a CSR stream with this many nonzeros */
static inline int g_chunk_for(long nnz) { return nnz < (1L << 18) ? G_CHUNK_SMALL : G_CHUNK; }
/* slab SpMV (k_spmv_slab): 1024-thread blocks of 16 waves.  The majors are dealt to blocks (and, inside a block, to its
 * waves — which does not matter for any sum) by WORK: pdlp_host.cpp slabPartition, restated here.  Work of a major of len
 * entries = len + 3 x its cold entries (g_slab_cold: they count four times) + len * min(len, 64) / 32 + majorCost (integer division; majorCost alone for a long major, whose segment
 * tasks run elsewhere; majorCost = 2 for the operand by rows, 10 for the transposed one, whose launch also carries the
 * next primal step of every column).  nBlocks = ceil(nMajor / 256) capped at 256 (more only when 256 blocks of 16384 majors do not hold the
 * operand); block b takes majors while it is closer to ceil(work left / blocks left) with the next major than without,
 * at least one and at most 16384, and never so few / many that the blocks behind it could not hold / would not get the
 * rest. */
enum { G_SLAB_T = 1024, G_SLAB_WAVES = 16, G_SLAB_BLOCKS = 256, G_SLAB_BLOCK_CAP = 16384, G_SLAB_MIN_ROWS = 256, G_SLAB_MAJOR_COST_ROWS = 2, G_SLAB_MAJOR_COST_COLS = 10 };
static inline int g_slab_fits(int nMajor, int nMinor) { /* the minor index must fit 28 bits of an entry */
  (void)nMajor;
  return (long)nMinor <= (1L << 28);
}
/* blockBeg[0..nBlocks] (caller provides room for g_slab_blocks_room(nMajor) ints); returns nBlocks */
static inline long g_slab_blocks_room(long nMajor) { return G_SLAB_BLOCKS + nMajor / 16 + 3; } /* (a wave holds >= 16 majors: g_slab_fits) */
/* cold entries of every major (pdlp_host.cpp slabColdCounts): 2^17 or more minors away from the major's middle entry, in a
 * minor that at most 64 majors touch; they count four times.  cold: caller's scratch of nMajor ints, count: of nMinor ints */
static inline void g_slab_cold(const int* beg, const int* idx, int nMajor, int nMinor, int longLimit, int* cold, int* count) {
  for (int j = 0; j < nMinor; ++j) count[j] = 0;
  for (long p = 0; p < (nMajor > 0 ? beg[nMajor] : 0); ++p) count[idx[p]]++;
  for (int r = 0; r < nMajor; ++r) {
    const int p0 = beg[r], len = beg[r + 1] - beg[r];
    int c = 0;
    if (len >= 2 && len <= longLimit) {
      const int mid = idx[p0 + len / 2];
      for (int p = p0; p < p0 + len; ++p) {
        const int d = idx[p] > mid ? idx[p] - mid : mid - idx[p];
        if (d >= (1 << 17) && count[idx[p]] <= 64) ++c;
      }
    }
    cold[r] = c;
  }
}
static inline int g_slab_blocks(const int* beg, const int* cold, int nMajor, int nMinor, int longLimit, int majorCost, int* blockBeg) {
  int mb = 0;
  while ((1L << mb) < (long)nMinor) ++mb;
  if (mb < 4) mb = 4;
  long waveCap = 1L << (32 - mb);
  if (waveCap > G_SLAB_BLOCK_CAP) waveCap = G_SLAB_BLOCK_CAP;
  long cap = waveCap * G_SLAB_WAVES;
  if (cap > G_SLAB_BLOCK_CAP) cap = G_SLAB_BLOCK_CAP;
  long nB = ((long)nMajor + G_SLAB_MIN_ROWS - 1) / G_SLAB_MIN_ROWS;
  if (nB > G_SLAB_BLOCKS) nB = G_SLAB_BLOCKS;
  if (nB < ((long)nMajor + cap - 1) / cap) nB = ((long)nMajor + cap - 1) / cap;
#define G_WORK(r, len) ((len) > longLimit ? (long)majorCost : (long)(len) + 3L * cold[r] + ((long)(len) * ((len) < 64 ? (len) : 64)) / 32 + majorCost)
  long rem = 0;
  for (int r = 0; r < nMajor; ++r) { const int len = beg[r + 1] - beg[r]; rem += G_WORK(r, len); }
  int r = 0;
  blockBeg[0] = 0;
  for (long u = 0; u < nB; ++u) {
    const long left = nB - u, target = (rem + left - 1) / left, rows = (long)nMajor - r;
    long minRows = rows - (left - 1) * cap, maxRows = rows - (left - 1);
    if (minRows < 1) minRows = rows > 0 ? 1 : 0;
    if (maxRows < 1) maxRows = 1;
    if (maxRows > cap) maxRows = cap;
    long acc = 0, cnt = 0;
    while (cnt < rows && cnt < maxRows) {
      const int len = beg[r + 1] - beg[r];
      const long c = G_WORK(r, len);
      if (cnt >= minRows && 2 * acc + c > 2 * target) break;
      acc += c; ++r; ++cnt;
    }
    rem -= acc;
    blockBeg[u + 1] = r;
  }
#undef G_WORK
  return (int)nB;
}

static inline double g_wave_tree(const double* lane /* [64] */) {
  double v[G_WAVE], t[G_WAVE];
  memcpy(v, lane, sizeof(v));
  for (int off = G_WAVE / 2; off > 0; off >>= 1) {
    for (int i = 0; i < G_WAVE; ++i) t[i] = v[i] + (i + off < G_WAVE ? v[i + off] : v[i]); /* __shfl_down */
    memcpy(v, t, sizeof(v));
  }
  return v[0];
}
static inline double g_block_sum_n(const double* perThread, int nThreads) {
  double r = 0.0;
  for (int w = 0; w < nThreads / G_WAVE; ++w) r += g_wave_tree(perThread + w * G_WAVE);
  return r;
}
static inline double g_block_sum(const double* perThread /* [256] */) { return g_block_sum_n(perThread, G_T); }
/* A long major (pdlp_kernels.hip longTask): segments of 512 * 2^k entries (smallest k with at most 64 segments);
 * within a segment lane l adds the products of the entries l, l+64, ... in ascending order, then the 64-lane
 * shuffle tree; the segment sums are added left to right. */
enum { G_LONG_SEG = 512, G_LONG_MAXSEG = 64, G_LONG_SLOT_CAP = 2048, G_SLAB_LONG = 256 };
static inline double g_long_major_sum(const int* idx, const double* val, const double* in, int p0, int len) {
  long seg = G_LONG_SEG;
  while ((len + seg - 1) / seg > G_LONG_MAXSEG) seg *= 2;
  double total = 0.0;
  for (long sb = 0; sb < len; sb += seg) {
    const long se = sb + seg < len ? sb + seg : len;
    double lane[G_WAVE];
    for (int l = 0; l < G_WAVE; ++l) {
      double s = 0.0;
      for (long q = sb + l; q < se; q += G_WAVE) s += val[p0 + q] * in[idx[p0 + q]];
      lane[l] = s;
    }
    total += g_wave_tree(lane);
  }
  return total;
}
/* value of one major: left to right up to longLimit entries, segment tasks beyond */
static inline double g_major_sum(const int* beg, const int* idx, const double* val, const double* in, int r, int longLimit) {
  const int p0 = beg[r], p1 = beg[r + 1];
  if (p1 - p0 <= longLimit) {
    double s = 0.0;
    for (int p = p0; p < p1; ++p) s += val[p] * in[idx[p]];
    return s;
  }
  return g_long_major_sum(idx, val, in, p0, p1 - p0);
}
/* the longest major that is still summed left to right: the CSR stream's chunk (by its nnz), or 256 in the slab
 * layout.  layoutMode as in pdlp_oracle.c: 0 = the product's automatic rule (slab when the gathered vector has
 * >= 2^18 entries), 1 = CSR, 2 = slab. */
static inline int g_long_major_chunk(const int* beg, int nMajor, int nMinor, int layoutMode) {
  int slab = layoutMode == 2 || (layoutMode == 0 && nMinor >= (1 << 18));
  if (slab && (nMajor <= 0 || !g_slab_fits(nMajor, nMinor))) slab = 0;
  if (!slab) return g_chunk_for(nMajor > 0 ? beg[nMajor] : 0);
  return G_SLAB_LONG;
}
/* reducePartials: lane t sums p[t], p[t+256], ... in 4 independent chains */
static inline double g_reduce_partials(const double* p, int count) {
  double lane[G_T];
  for (int t = 0; t < G_T; ++t) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = t;
    for (; i + 3 * G_T < count; i += 4 * G_T) { s0 += p[i]; s1 += p[i + G_T]; s2 += p[i + 2 * G_T]; s3 += p[i + 3 * G_T]; }
    for (; i < count; i += G_T) s0 += p[i];
    lane[t] = (s0 + s1) + (s2 + s3);
  }
  return g_block_sum(lane);
}
static inline int g_vec_blocks(int len) {
  long b = ((long)len + G_T - 1) / G_T;
  if (b < 1) b = 1;
  if (b > G_MAXGRID) b = G_MAXGRID;
  return (int)b;
}
/* grid-stride reduction of f(i), i < len, as the vector kernels do it */
typedef double (*g_elem_fn)(const void* ctx, int i);
static inline double g_grid_sum(int len, g_elem_fn f, const void* ctx, double* partScratch) {
  const int nb = g_vec_blocks(len > 0 ? len : 1), stride = nb * G_T;
  for (int b = 0; b < nb; ++b) {
    double lane[G_T];
    for (int t = 0; t < G_T; ++t) {
      double a = 0.0;
      for (long i = (long)b * G_T + t; i < len; i += stride) a += f(ctx, (int)i);
      lane[t] = a;
    }
    partScratch[b] = g_block_sum(lane);
  }
  return g_reduce_partials(partScratch, nb);
}


#endif
