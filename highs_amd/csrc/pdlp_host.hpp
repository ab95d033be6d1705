// pdlp_host.hpp — host-side problem preparation for the MI355X PDLP path.
//
// Turns the caller's HighsLp-shaped arrays (pdlp_problem_t) into the standard
// form cuPDLP-C iterates on, with the same conventions as the reference so
// that results are comparable row for row:
//   formulate      <-> formulateLP_highs          highs/pdlp/CupdlpWrapper.cpp:280-448
//   scale          <-> Init_Scaling/PDHG_Scale_Data  cupdlp_scaling.c:395-425,233-393
//   build_csr/csc  <-> csc2csr / cupdlp_dcs_transpose cupdlp_utils.c:1222, cupdlp_cs.c:189
//   row partition  <-> (no reference: SURVEY §8e multi-GPU row blocks)
#pragma once
#include <cstdarg>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pdlp_mi355x.h"
#include "pdlp_env.hpp"

namespace pdlp {


// One formatted log line to the caller's sink (pdlp_params_t::log_callback, e.g. highsLogUser) or, without
// one, to stdout like the reference's cuPDLP-C.
void logLine(const pdlp_params_t& opt, int level, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
void logLineV(const pdlp_params_t& opt, int level, const char* fmt, va_list ap);

enum RowKind : int32_t { kRowEq = 0, kRowLeq = 1, kRowGeq = 2, kRowBound = 3 };  // cupdlp_defs.h types

// Sparse matrix in compressed form; "major" is rows for CSR, columns for CSC.
struct Compressed {
  std::vector<int32_t> beg;   // [nMajor+1]
  std::vector<int32_t> idx;   // [nnz] minor index
  std::vector<double> val;    // [nnz]
};

struct StandardForm {
  int32_t n = 0;      // columns incl. one slack per BOUND row
  int32_t m = 0;      // rows
  int32_t n0 = 0;     // original columns
  int32_t nEqs = 0;   // EQ + BOUND rows, permuted first
  int64_t nnz = 0;
  Compressed csc;     // reference column order (EQ/BOUND entries first)
  Compressed csr;     // rows with ascending column index
  Compressed cscSorted;  // columns with ascending row index (device copy for A'y)
  std::vector<double> cost, rhs, lower, upper;
  std::vector<double> qdiag;     // diagonal of Q (with the sense, scaled like cost twice); empty = LP
  Compressed qoff;               // off-diagonal part of Q: symmetric, both triangles, by rows with ascending column
                                 // (with the sense, q_ij / (cs_i cs_j)); beg empty = none
  std::vector<double> rowUpper;  // HiPDLP form only (rhs is then the row LOWER bound)
  std::vector<uint8_t> rowIsEq;  // HiPDLP form only: per PERMUTED row (is_equality_row_)
  std::vector<int32_t> rowKind;    // per ORIGINAL row
  std::vector<int32_t> rowNewIdx;  // original row -> permuted row
  std::vector<double> colScale, rowScale;
  bool scaled = false;
  double offset = 0.0, sense = 1.0;
  double normCost = 0.0, normRhs = 0.0;  // of the unscaled formulated data
  double matNormInf = 0.0;               // max |a_ij| of the (scaled) matrix
};

// Throws std::runtime_error on malformed input.
void validateProblem(const pdlp_problem_t& P);
void requireConstraints(const pdlp_problem_t& P);  // throws for LPs without rows / columns / nonzeros
void extractDiagonalHessian(const pdlp_problem_t& P, double sense, int32_t n, std::vector<double>& q);
// The Hessian of a QP, given as HiGHS gives it (model/HighsHessian.h:22-34: lower triangle, column-wise), split into
// its diagonal (prox step) and its off-diagonal part N (explicit N x term): qoff = N with both triangles, by rows with
// ascending column, repeated entries added up; empty when Q is diagonal.  Entries above the diagonal and a negative
// diagonal (for this objective sense) are errors.
void extractHessian(const pdlp_problem_t& P, double sense, int32_t n, std::vector<double>& qdiag, Compressed& qoff);
bool hessianHasOffDiagonal(const pdlp_problem_t& P);
void formulate(const pdlp_problem_t& P, StandardForm& F);
void scale(StandardForm& F, int ruizTimes = 10, double pcAlpha = 1.0);
void finalize(StandardForm& F);  // CSR + row-sorted CSC + matNormInf

// ---- HiPDLP path (solver="hipdlp") -------------------------------------------------------------
// preprocessLp, hipdlp/pdhg.cc:152-357: same row kinds as above but classified with +-inf (not 1e20),
// free rows get their own kind (4), rows keep BOTH bounds (rhs = row_lower, rowUpper), the costs do
// NOT take the objective sense, and column entries are sorted by permuted row index.
enum { kRowFree = 4 };
void formulateHipdlp(const pdlp_problem_t& P, StandardForm& F);
// Scaling::scaleProblem, hipdlp/scaling.cc:31-262: Ruiz (inf-norm) x ruizIters, Pock-Chambolle
// (alpha 1), L2 — each optional (pdlp_scaling_mode bits 1, 4, 2).
void scaleHipdlp(StandardForm& F, bool ruiz, bool pc, bool l2, int ruizIters);

// Contiguous row blocks balanced by nonzeros: returns world+1 row offsets.
std::vector<int32_t> rowPartition(const Compressed& csr, int32_t m, int32_t world);

// Row slab [r0,r1) of F as (csr slab, csc-of-slab with local row indices).
void extractSlab(const StandardForm& F, int32_t r0, int32_t r1, Compressed& csrSlab, Compressed& cscSlab);

// CSR-adaptive launch plan: consecutive majors are grouped into work blocks of at most `chunk` nonzeros (whole
// majors, summed left to right by one lane each); a major longer than `chunk` belongs to no block — it is cut
// into segment tasks (LongPlan).
struct StreamPlan {
  std::vector<int32_t> blockBeg;    // [4*nBlocks] first and end major, first and end entry of each block
  int32_t nBlocks = 0;
  std::vector<int32_t> longMajors;  // majors longer than chunk, ascending
};
StreamPlan planStream(const std::vector<int32_t>& beg, int32_t nMajor, int32_t chunk, int32_t maxMajorsPerBlock);

// Segment tasks of the long majors of one operand (the device view is pdlp_kernels.hpp LongMat / LongTask).  A long
// major is cut into segments of 512 * 2^k nonzeros (smallest k with at most 64 segments), one task each; tasks are
// handed to workgroups of wavesPerBlock waves, W consecutive tasks each.
//   Stream layout (homeOf == nullptr): tasks in (major, segment) order; a major with at most W segments never straddles
// two workgroups (idle tasks pad the list) and its segment sums meet in LDS.
//   Slab layout (homeOf given): the XCD-affine deal.  homeOf(pBeg, pEnd) = the XCD (0..7) whose streaming blocks gather
// from the stretch of the vector the segment's entries [pBeg, pEnd) lie in; task workgroup lb runs on XCD
// (firstXcd + lb) % 8 and takes tasks of that home (see planLong).  Segment sums of majors with more than one segment
// meet in HBM slots (LongTask::first + seg); nSegSlots of them.
struct LongTaskHost { int32_t pBeg, pEnd, c, first, nSeg, major, contained, seg; };  // = LongTask
struct LongPlan {
  std::vector<LongTaskHost> tasks;
  int32_t nLong = 0, nTasks = 0, nSegSlots = 0;
};
// longMajors: indices into beg; vecIndex: result-vector index of each of them (nullptr: the index itself)
LongPlan planLong(const std::vector<int32_t>& beg, const std::vector<int32_t>& longMajors, const int32_t* vecIndex,
                  int32_t wavesPerBlock, const std::function<int(int32_t, int32_t)>* homeOf = nullptr, int32_t firstXcd = 0);

// XCD of logical slab block b under the contiguous block -> XCD map (the inverse of pdlp_devfn.hpp xcdContiguousBlock:
// XCD x runs the logical blocks [x*nB/8, (x+1)*nB/8), the first nB % 8 XCDs one more)
inline int32_t xcdOfLogicalBlock(int32_t b, int32_t nB) {
  const int32_t qlo = nB / 8, r = nB % 8;
  if (b < r * (qlo + 1)) return b / (qlo + 1);
  return qlo > 0 ? r + (b - r * (qlo + 1)) / qlo : 0;
}
// Which XCD's streaming blocks gather from which stretch of the vector: tile t = minors [t << tileLog2, (t+1) << tileLog2);
// owner[t] = the XCD (contiguous map) whose blocks hold the most short-major entries in it, the nearest owned tile's
// owner where nobody gathers.  hist: [8 * nTiles] entry counts per (XCD, tile).
int32_t xcdTileLog2(int32_t nMinor);
inline int32_t xcdTileCount(int32_t nMinor, int32_t tileLog2) { return (int32_t)((((int64_t)(nMinor > 1 ? nMinor : 1) - 1) >> tileLog2) + 1); }
std::vector<int8_t> xcdTileOwners(const std::vector<int32_t>& hist, int32_t nTiles);
// home XCD of the entries idx[pBeg, pEnd) (ascending minors): the most frequent owner among 8 sample entries
int xcdHomeOf(const int32_t* idx, int32_t pBeg, int32_t pEnd, const std::vector<int8_t>& owner, int32_t tileLog2);

// Slab layout (the layout of k_spmv_slab).  The gathered vector of a random sparse LP (8 MB at
// n = 1M) does not fit one XCD's 4 MB L2, so a plain CSR stream pays one fabric request per 8-byte
// gather.  Here every wave sweeps the gathered vector slab by slab (slab = 2^slabWidthLog2 consecutive
// minor indices, 1 MB by default), in step with all the others, so the slab being gathered from stays
// in every XCD's L2.  The unit of ownership is the WAVE: a block of 16 waves (one block per CU) owns
// the consecutive majors [waveBeg[16 b], waveBeg[16 b + 16]), wave w of it [waveBeg[16 b + w],
// waveBeg[16 b + w + 1]).  Blocks and waves are cut by WORK, not by major count (slabPartition below):
// with skewed major lengths a block of equal major COUNT streams up to twice the mean number of entries
// and the launch is its slowest block (round 4, per-block phase profile).  A wave's nonzeros are ONE
// dense stream sorted by (minor >> slabWidthLog2, local major, minor): no windows, no per-slab padding.
// An entry packs (localMajor << minorBits | minor), local = major - the wave's first major, with the
// GLOBAL minor, so the kernel needs no slab table at all — a slab boundary inside a 64-entry group
// shows up as a descent of the local major.  Majors longer than `longLimit` are left out (marked in
// longMask, one bit per major) and handled as segment tasks.
constexpr int32_t kSlabWidthLog2 = 17;  // 1 MB slabs: 56.2 vs 57.0 us per A x at the bench size (15..18 within 1.5 %)
constexpr int32_t kSlabWavesPerBlock = 16;
constexpr int32_t kSlabTargetBlocks = 256;  // CUs of an MI355X
constexpr int32_t kSlabBlockRowCap = 16384; // majors per block: 128 KB of LDS accumulators (gfx950 has 160 KB per CU)
constexpr int32_t kSlabMinRowsPerBlock = 256;
// work of a major besides its entries, in entries: its epilogue.  The operand by rows (A x+: dual step, ~7 vector
// loads and stores per row) and the transposed one (A'y+ with the interaction sums AND the next primal step of the column:
// ~10, most of them behind the grid barrier) differ: with one value for both, config d lost 3 us on one launch or the
// other; 6 / 10 / 16 for the transposed operand: config d 72.3 / 70.6-71.0 / 71.4 us per iteration, config c unchanged
constexpr int32_t kSlabMajorCostRows = 2, kSlabMajorCostCols = 10;

// The partition of the majors over blocks and waves.  Work of a major of len entries =
//     len + 3 cold + len * min(len, 64) / 32 + majorCost          (majorCost alone for a long major)
//   * len * min(len, 64) / 32: the entries of a run of equal majors inside a 64-entry group are added by ONE lane, so a
//     group made of one run of 64 costs about three times a group of eight runs of eight (config d, blocks of equal ENTRY
//     counts: the block with the longest rows still streamed 1.66x the mean time);
//   * cold: its COLD entries count kSlabColdWeight = 4 times.  Cold = kSlabFar or more minors away from the major's middle
//     entry AND in a minor that at most kSlabHotCount majors touch: the gather leaves the part of the gathered vector the
//     block works in and meets nobody else's — a miss all the way to HBM, and a wave has ONE group of gathers in flight, so
//     a group of cold entries is a memory round trip of its own (config c: the block that owns 512 rows of 12 random
//     columns ended 7 us after the others in round 5; at weight 2, once the XCD-affine tasks had taken the wasted traffic
//     out of the launch, still 4 us — 34.1 against a mean of 29.8; weight 3 / 4 on one box: 32.3 / 32.6, launch 35.0 ->
//     33.5 / 33.6 us, round 6).
//     Far entries in minors that many majors touch — the dense columns of config d, the dense rows of config c seen from
//     its columns — are the hottest lines of the vector and cost nothing extra (a span-only rule measured worse there:
//     profiles/r05_development_measurements.md section 4);
//   * majorCost: the epilogue, kSlabMajorCostRows / Cols above.
// nBlocks = ceil(nMajor / 256) capped at 256 (more only when 256 blocks of 16384 majors do not hold the operand).
// Blocks are filled one after the other: block b takes majors while it is closer to ceil(work left / blocks left)
// with the next major than without, but at least one, at most kSlabBlockRowCap, and never so few / many that the
// blocks behind it could not hold / would not get the rest; the 16 waves of a block are filled the same way from the
// block's majors (cap: 2^(32 - minorBits) majors, the local-major field of an entry).  Sequential and exact in
// integers: the device-side set-up (pdlp_setup.hip) counts the cold entries with two small kernels, downloads them
// with the major starts (8 bytes per major) and calls this same function; oracle/gpu_order.h restates it.
int64_t slabMajorWork(int32_t len, int32_t nCold, int32_t longLimit, int32_t majorCost);
constexpr int32_t kSlabFar = 1 << 17, kSlabHotCount = 64, kSlabColdWeight = 4;
// cold[r] for every major r (0 for long and single-entry majors)
void slabColdCounts(const int32_t* beg, const int32_t* idx, int32_t nMajor, int32_t nMinor, int32_t longLimit, int32_t* cold);
struct SlabPartition {
  int32_t nBlocks = 0, minorBits = 0, maxRowsPerBlock = 0;
  std::vector<int32_t> waveBeg;  // [16*nBlocks+1] first major of every wave
  int32_t blockBeg(int32_t b) const { return waveBeg[(size_t)b * kSlabWavesPerBlock]; }
};
// false: the minor index does not fit the entry packing (nMinor > 2^28: the caller uses the CSR stream kernel)
bool slabFits(int32_t nMajor, int32_t nMinor);
// cold: slabColdCounts, or nullptr (none)
SlabPartition slabPartition(const int32_t* beg, const int32_t* cold, int32_t nMajor, int32_t nMinor, int32_t longLimit,
                            int32_t majorCost);

struct SlabLayout {
  int32_t rowsPerBlock = 0;  // most majors in one block (LDS accumulators)
  int32_t nBlocks = 0, minorBits = 0, slabWidthLog2 = 0;
  std::vector<int32_t> waveBeg;   // [16*nBlocks+1] first major of every wave
  std::vector<int32_t> wavePtr;   // [16*nBlocks+1] entry offsets
  std::vector<uint32_t> ent;      // [nnzShort]
  std::vector<double> val;        // [nnzShort]
  std::vector<uint32_t> longMask; // [ceil(nMajor/32)] bit r: major r is a long one
  Compressed longCsr;             // compacted long majors
  std::vector<int32_t> longMap;   // compact index -> major
};
void buildSlabLayout(const Compressed& csr, int32_t nMajor, int32_t nMinor, int32_t longLimit, int32_t slabWidthLog2,
                     int32_t majorCost, SlabLayout& out);
// Host build of what pdlp_kernels.hip k_block_span computes on the device: per block of the partition the span (lo, hi) and
// entry count of its short majors, and the histogram [8 * nTiles] of their entries per (XCD of the contiguous map, tile).
std::vector<int32_t> slabTileHistogram(const int32_t* beg, const int32_t* idx, const SlabPartition& part, int32_t longLimit,
                                       int32_t tileLog2, int32_t nTiles, std::vector<int32_t>& lo, std::vector<int32_t>& hi,
                                       std::vector<int32_t>& cnt);
// The segment tasks of a slab operand's long majors.  longBeg / longIdx: the compact CSR of the long majors (nLong of
// them), longMap: compact index -> major.  Tasks per workgroup (taskGroup, out): 16 — or, `balance`, ceil(segments / 256),
// at most 16: ONE task workgroup per CU.  They run NEXT to the streaming blocks (two blocks per CU), and 128 of them on 256
// CUs slow down half of the streaming blocks (bench.py --config c, A x+: blocks sharing their CU 37 us, the others 26.5);
// the fused A'y+ launch can only carry them as co-resident workgroups when there is at most one per CU (config d: 2 304
// segments -> 256 workgroups of 9; round 5 ran 144 of 16 there and the launch waited for the 144 blocks that shared a CU).
// tileOwner (nullptr: tasks in (major, segment) order): the XCD-affine deal of planLong, nSlabBlocks streaming blocks in
// front of the task workgroups.
LongPlan planSlabTasks(const std::vector<int32_t>& longBeg, const int32_t* longIdx, int32_t nLong, const int32_t* longMap, bool balance,
                       const std::vector<int8_t>* tileOwner, int32_t tileLog2, int32_t nSlabBlocks, int32_t& taskGroup);

}  // namespace pdlp
