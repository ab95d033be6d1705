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
#include <string>
#include <vector>

#include "../../include/pdlp_mi355x.h"

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
// handed to workgroups of wavesPerBlock waves, W consecutive tasks each, and a major with at most W segments never
// straddles two workgroups (idle tasks pad the list).
struct LongTaskHost { int32_t pBeg, pEnd, c, first, nSeg, major, contained, pad_; };  // = LongTask
struct LongPlan {
  std::vector<LongTaskHost> tasks;
  int32_t nLong = 0, nTasks = 0;
};
// longMajors: indices into beg; vecIndex: result-vector index of each of them (nullptr: the index itself)
LongPlan planLong(const std::vector<int32_t>& beg, const std::vector<int32_t>& longMajors, const int32_t* vecIndex,
                  int32_t wavesPerBlock);

// Slab layout (the layout of k_spmv_slab).  The gathered vector of a random sparse LP (8 MB at
// n = 1M) does not fit one XCD's 4 MB L2, so a plain CSR stream pays one fabric request per 8-byte
// gather.  Here every wave sweeps the gathered vector slab by slab (slab = 2^slabWidthLog2 consecutive
// minor indices, 1 MB by default), in step with all the others, so the slab being gathered from stays
// in every XCD's L2.  The unit of ownership is the WAVE: a block of 16 waves (one block per CU) owns
// the consecutive majors [waveBeg[16 b], waveBeg[16 b + 16]), wave w of it [waveBeg[16 b + w],
// waveBeg[16 b + w + 1]).  Blocks and waves are cut by WORK, not by major count (SlabPlan below):
// with skewed major lengths a block of equal major COUNT streams up to twice the mean number of entries
// and the launch is its slowest block (round 4, per-block phase profile).
//   Three classes of majors by length:
//   * regular (at most longLimit = 256 entries): in the owning wave's list, sorted by (minor >> slabWidthLog2,
//     local major, minor); the first lane of a run of equal majors adds the run left to right.  An entry packs
//     (localMajor << minorBits | minor), local = major - the wave's first major, with the GLOBAL minor — no slab table:
//     a slab boundary inside a 64-entry group shows up as a descent of the local major;
//   * medium (longLimit < entries <= medMax): cut into segments of 512 entries that are dealt to the 16 waves of the
//     block that OWNS the major and appended to those waves' lists behind their regular entries (each segment padded
//     to whole 64-entry groups with zero-valued entries).  A wave streams them through the same register pipeline:
//     lane l adds the products of the segment's entries l, l+64, ... in a register, 64-lane shuffle tree, the segment
//     sum goes to a slot in LDS; after the stream one lane adds a major's segment sums left to right.  That is the
//     summation order of the segment tasks below (oracle: g_long_major_sum) — no extra workgroups, no tickets, nothing
//     crosses the block — and the major then is an ordinary major of the block for the epilogue;
//   * long (more than medMax entries): left out (marked in longMask, one bit per major), segment tasks as before
//     (LongPlan): a major that would unbalance its block.
struct SlabSeg { int32_t src, len, dst, slot; };  // CSR position of the first entry, entries (<= 512), position in ent/val, slot in the block
struct SlabPlan {
  int32_t nBlocks = 0, minorBits = 0, longLimit = 0, medMax = 0;
  int32_t maxRowsPerBlock = 0, maxSlotsPerBlock = 0;
  int64_t listLen = 0;                // entries of all lists incl. padding
  std::vector<int32_t> waveBeg;       // [16*nBlocks+1] first major of every wave
  std::vector<int32_t> waveReg;       // [16*nBlocks] regular entries of the wave
  std::vector<int32_t> wavePtr;       // [16*nBlocks+1] list offsets: regular entries, (pad to 64 if segments follow), segments
  std::vector<int32_t> waveSegBeg;    // [16*nBlocks+1] the wave's segments in segs / segDesc
  std::vector<SlabSeg> segs;          // wave by wave, in list order
  std::vector<uint32_t> segDesc;      // [nSegs] slot << 16 | len (what the kernel reads)
  std::vector<int32_t> blockMedBeg;   // [nBlocks+1] the block's medium majors in medDesc
  std::vector<uint32_t> medDesc;      // [2*nMed] {major - first major of the block, firstSlot << 8 | nSeg}
  int32_t blockBeg(int32_t b) const { return waveBeg[(size_t)b * 16]; }
};
constexpr int32_t kSlabWidthLog2 = 17;  // 1 MB slabs: 56.2 vs 57.0 us per A x at the bench size (15..18 within 1.5 %)
constexpr int32_t kSlabWavesPerBlock = 16;
constexpr int32_t kSlabTargetBlocks = 256;  // CUs of an MI355X
constexpr int32_t kSlabBlockUnitCap = 16384;// LDS doubles per block for the majors' accumulators AND the segment slots: 128 KB of 160
constexpr int32_t kSlabMinRowsPerBlock = 256;
constexpr int32_t kSlabMajorCost = 2;       // work of a major besides its entries (epilogue), in entries
constexpr int32_t kSlabSegment = 512;       // entries per in-block segment (= pdlp_kernels.hpp kLongSegment: the same sums)
constexpr int32_t kSlabMedMaxCap = 16384;   // a medium major has at most 32 segments

// The plan.  Work of major r = (its entries — half of them for a medium major, none for a long one) + kSlabMajorCost; LDS units of major r = 1 + its
// segments (medium majors).  nBlocks = ceil(nMajor / 256) capped at 256 (more only when the units do not fit);
// medMax = half of the mean work per block, within [512, 16384].  Blocks are filled one after the other: block b
// takes majors while it is closer to ceil(work left / blocks left) with the next major than without, at least one,
// at most kSlabBlockUnitCap units, and never so few / many that the blocks behind it could not hold / would not get
// the rest.  Inside a block: segment i of the block (medium majors in order, their segments in order) goes to wave
// i mod 16; the majors are then dealt to the 16 waves the same way by work, each wave's target lowered by the segment
// entries it already has (cap: 2^(32 - minorBits) majors, the local-major field of an entry).  Sequential and exact
// in integers: the device-side set-up (pdlp_setup.hip) calls this same function on the downloaded major starts;
// oracle/gpu_order.h restates the BLOCK boundaries and medMax (all that sums depend on).
// false: the minor index does not fit the entry packing (nMinor > 2^28: the caller uses the CSR stream kernel)
bool slabFits(int32_t nMajor, int32_t nMinor);
SlabPlan slabPlan(const int32_t* beg, int32_t nMajor, int32_t nMinor, int32_t longLimit);

struct SlabLayout {
  SlabPlan plan;
  int32_t slabWidthLog2 = 0;
  std::vector<uint32_t> ent;      // [plan.listLen]
  std::vector<double> val;        // [plan.listLen]
  std::vector<uint32_t> longMask; // [ceil(nMajor/32)+1] bit r: major r is a long one
  Compressed longCsr;             // compacted long majors
  std::vector<int32_t> longMap;   // compact index -> major
};
// plan: a plan computed before (slabPlan of the same arguments), or nullptr
void buildSlabLayout(const Compressed& csr, int32_t nMajor, int32_t nMinor, int32_t longLimit, int32_t slabWidthLog2,
                     SlabLayout& out, const SlabPlan* plan = nullptr);

}  // namespace pdlp
