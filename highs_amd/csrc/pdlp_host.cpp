// pdlp_host.cpp — see pdlp_host.hpp.
#include "pdlp_host.hpp"

#include <algorithm>
#include <cstdio>
#include <functional>
#include <cmath>
#include <limits>
#include <mutex>
#include <set>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace pdlp {

const char* devEnv(const char* name) {
  const char* v = getenv(name);
  if (!v) return nullptr;
  const char* d = getenv("PDLP_MI355X_DEV");
  if (d && atoi(d) != 0) return v;
  static std::mutex mu;
  static std::set<std::string> told;
  std::lock_guard<std::mutex> g(mu);
  if (told.insert(name).second)
    fprintf(stderr, "pdlp_mi355x: %s is a development switch and is ignored (set PDLP_MI355X_DEV=1 to enable development switches)\n", name);
  return nullptr;
}

void logLineV(const pdlp_params_t& opt, int level, const char* fmt, va_list ap) {
  if (!opt.log_callback) {
    vprintf(fmt, ap);
    fflush(stdout);
    return;
  }
  char buf[1024];
  vsnprintf(buf, sizeof(buf), fmt, ap);
  opt.log_callback(opt.log_ctx, level, buf);
}
void logLine(const pdlp_params_t& opt, int level, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  logLineV(opt, level, fmt, ap);
  va_end(ap);
}

namespace {
constexpr double kBoundInf = 1e20;  // CupdlpWrapper.cpp:316-317,375-378
const double kInf = std::numeric_limits<double>::infinity();
}  // namespace

// Structural checks of the caller's CSC arrays (shared by both formulations and the device-side setup):
// the starts must begin at 0, never decrease and end at num_nz; row indices must be in range.
void validateProblem(const pdlp_problem_t& P) {
  if (P.num_col < 0 || P.num_row < 0) throw std::runtime_error("negative dimensions");
  if (P.num_col > 0 && (!P.a_start || !P.col_cost || !P.col_lower || !P.col_upper))
    throw std::runtime_error("null column arrays");
  if (P.num_row > 0 && (!P.row_lower || !P.row_upper)) throw std::runtime_error("null row arrays");
  if (P.num_col == 0) return;
  if (P.a_start[0] != 0) throw std::runtime_error("a_start[0] must be 0");
  for (int32_t j = 0; j < P.num_col; ++j)
    if (P.a_start[j + 1] < P.a_start[j]) throw std::runtime_error("a_start must not decrease");
  const int64_t nnz = P.a_start[P.num_col];
  if (P.num_nz > 0 && nnz > P.num_nz) throw std::runtime_error("a_start[num_col] exceeds num_nz");
  if (nnz > 0 && (!P.a_index || !P.a_value)) throw std::runtime_error("null matrix arrays");
  for (int64_t p = 0; p < nnz; ++p)
    if (P.a_index[p] < 0 || P.a_index[p] >= P.num_row) throw std::runtime_error("row index out of range");
}

// Diagonal of the (lower-triangular, column-wise) HighsHessian times the objective sense, padded with zeros
// for the slack columns; empty when there is no quadratic term.  Off-diagonal nonzeros are an error: the
// primal step of this library is the closed-form proximal step of a SEPARABLE quadratic.
void extractDiagonalHessian(const pdlp_problem_t& P, double sense, int32_t n, std::vector<double>& q) {
  Compressed off;
  extractHessian(P, sense, n, q, off);
  if (!off.beg.empty())
    throw std::runtime_error("pdlp_mi355x: this set-up path takes diagonal Hessians only (off-diagonal entries present)");
}

bool hessianHasOffDiagonal(const pdlp_problem_t& P) {
  if (P.q_dim <= 0 || !P.q_start || !P.q_index || !P.q_value) return false;
  for (int32_t j = 0; j < P.q_dim; ++j)
    for (int32_t p = P.q_start[j]; p < P.q_start[j + 1]; ++p)
      if (P.q_index[p] != j && P.q_value[p] != 0.0) return true;
  return false;
}

void extractHessian(const pdlp_problem_t& P, double sense, int32_t n, std::vector<double>& q, Compressed& off) {
  q.clear();
  off = Compressed();
  if (P.q_dim <= 0 || !P.q_start) return;
  if (P.q_dim > P.num_col) throw std::runtime_error("Hessian dimension exceeds the number of columns");
  const int32_t nq = P.q_start[P.q_dim];
  if (nq > 0 && (!P.q_index || !P.q_value)) throw std::runtime_error("null Hessian arrays");
  bool any = false;
  std::vector<double> d((size_t)n, 0.0);
  std::vector<int32_t> cnt((size_t)n + 1, 0);
  int64_t nOff = 0;
  for (int32_t j = 0; j < P.q_dim; ++j)
    for (int32_t p = P.q_start[j]; p < P.q_start[j + 1]; ++p) {
      const int32_t i = P.q_index[p];
      if (i < 0 || i >= P.q_dim) throw std::runtime_error("Hessian index out of range");
      if (P.q_value[p] == 0.0) continue;
      any = true;
      if (i == j) { d[j] += P.q_value[p] * sense; continue; }
      if (i < j)
        throw std::runtime_error("pdlp_mi355x: the Hessian must be given by its lower triangle (entry (" + std::to_string(i) + "," +
                                 std::to_string(j) + ") lies above the diagonal)");
      ++cnt[i + 1]; ++cnt[j + 1];
      nOff += 2;
    }
  for (double v : d)
    if (v < 0.0) throw std::runtime_error("pdlp_mi355x: the Hessian is not positive semidefinite for this objective sense");
  if (any) q = std::move(d);
  if (nOff == 0) return;
  if (nOff > 0x7fffffff) throw std::runtime_error("Hessian too large");
  // both triangles by rows; columns visited in ascending order, so row i receives its entries (i, j < i) in ascending
  // j from the first pass and (i, j > i) in ascending j from the second: sorted without a sort
  for (int32_t r = 0; r < n; ++r) cnt[r + 1] += cnt[r];
  std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
  std::vector<int32_t> idx((size_t)nOff);
  std::vector<double> val((size_t)nOff);
  for (int32_t j = 0; j < P.q_dim; ++j)  // lower entries (i > j) into row i: column j ascending over the outer loop
    for (int32_t p = P.q_start[j]; p < P.q_start[j + 1]; ++p) {
      const int32_t i = P.q_index[p];
      if (i == j || P.q_value[p] == 0.0) continue;
      idx[pos[i]] = j; val[pos[i]++] = P.q_value[p] * sense;
    }
  // mirrored entries (j, i) into row j: within column j the caller's order of i; sort each row's tail by column
  std::vector<int32_t> tailBeg(pos);
  for (int32_t j = 0; j < P.q_dim; ++j)
    for (int32_t p = P.q_start[j]; p < P.q_start[j + 1]; ++p) {
      const int32_t i = P.q_index[p];
      if (i == j || P.q_value[p] == 0.0) continue;
      idx[pos[j]] = i; val[pos[j]++] = P.q_value[p] * sense;
    }
  for (int32_t r = 0; r < n; ++r) {  // (stable insertion sort: the rows of a column usually come ascending already)
    for (int32_t a = tailBeg[r] + 1; a < cnt[r + 1]; ++a) {
      const int32_t ci = idx[a];
      const double cv = val[a];
      int32_t b = a - 1;
      while (b >= tailBeg[r] && idx[b] > ci) { idx[b + 1] = idx[b]; val[b + 1] = val[b]; --b; }
      idx[b + 1] = ci; val[b + 1] = cv;
    }
  }
  // repeated (row, column) pairs are added up, left to right
  off.beg.assign((size_t)n + 1, 0);
  for (int32_t r = 0; r < n; ++r) {
    off.beg[r] = (int32_t)off.idx.size();
    for (int32_t a = cnt[r]; a < cnt[r + 1]; ++a) {
      if (a > cnt[r] && idx[a] == off.idx.back()) off.val.back() += val[a];
      else { off.idx.push_back(idx[a]); off.val.push_back(val[a]); }
    }
  }
  off.beg[n] = (int32_t)off.idx.size();
}

// HiGHS never hands an LP without rows or without matrix nonzeros to a solver: solveLp() answers those itself
// (solveUnconstrainedLp, lp_data/HighsSolve.cpp:61-66).  PDHG has nothing to iterate on there (the initial step
// size is 1/max|a_ij|), so a direct caller of the C ABI gets an error instead of a loop that never ends.
void requireConstraints(const pdlp_problem_t& P) {
  const int64_t nnz = P.num_col > 0 && P.a_start ? (int64_t)P.a_start[P.num_col] : 0;
  bool anyNonzero = false;
  for (int64_t p = 0; p < nnz && !anyNonzero; ++p) anyNonzero = P.a_value[p] != 0.0;
  if (P.num_row == 0 || P.num_col == 0 || !anyNonzero)
    throw std::runtime_error("pdlp_mi355x: the LP has no rows, no columns or no matrix nonzeros — HiGHS solves such "
                             "LPs itself (solveUnconstrainedLp) before the PDLP path");
}

void formulate(const pdlp_problem_t& P, StandardForm& F) {
  validateProblem(P);
  const int32_t n0 = P.num_col, m = P.num_row;
  const int64_t nnz0 = n0 > 0 ? P.a_start[n0] : 0;
  if (nnz0 > 0 && (!P.a_index || !P.a_value)) throw std::runtime_error("null matrix arrays");
  F = StandardForm();
  F.n0 = n0;
  F.m = m;
  F.offset = P.offset;
  F.sense = P.sense < 0 ? -1.0 : 1.0;

  // Classify rows.  Ranged AND free rows become  a'x - z = 0  with a bounded
  // slack z (CupdlpWrapper.cpp:320-343).
  F.rowKind.resize(m);
  F.rowNewIdx.resize(m);
  int32_t nSlack = 0, nEq = 0;
  for (int32_t i = 0; i < m; ++i) {
    const bool lo = P.row_lower[i] > -kBoundInf, up = P.row_upper[i] < kBoundInf;
    RowKind k;
    if (lo && up && P.row_lower[i] == P.row_upper[i]) k = kRowEq;
    else if (lo && !up) k = kRowGeq;
    else if (!lo && up) k = kRowLeq;
    else k = kRowBound;
    F.rowKind[i] = k;
    if (k == kRowEq || k == kRowBound) ++nEq;
    if (k == kRowBound) ++nSlack;
  }
  if ((int64_t)n0 + nSlack > std::numeric_limits<int32_t>::max() ||
      nnz0 + nSlack > std::numeric_limits<int32_t>::max())
    throw std::runtime_error("problem exceeds 32-bit index range");
  F.n = n0 + nSlack;
  F.nEqs = nEq;
  F.nnz = nnz0 + nSlack;

  // Row permutation: equalities (EQ, BOUND) first in original order, then
  // inequalities; LEQ rows are negated into GEQ form (:382-404).
  F.rhs.assign(m, 0.0);
  int32_t eqPos = 0, inPos = nEq;
  for (int32_t i = 0; i < m; ++i) {
    switch (F.rowKind[i]) {
      case kRowEq: F.rowNewIdx[i] = eqPos; F.rhs[eqPos++] = P.row_lower[i]; break;
      case kRowBound: F.rowNewIdx[i] = eqPos; F.rhs[eqPos++] = 0.0; break;
      case kRowLeq: F.rowNewIdx[i] = inPos; F.rhs[inPos++] = -P.row_upper[i]; break;
      default: F.rowNewIdx[i] = inPos; F.rhs[inPos++] = P.row_lower[i]; break;
    }
  }

  // Columns: cost takes the sense, bounds beyond +-1e20 become infinite.
  F.cost.assign(F.n, 0.0);
  F.lower.resize(F.n);
  F.upper.resize(F.n);
  for (int32_t j = 0; j < n0; ++j) {
    F.cost[j] = P.col_cost[j] * F.sense;
    F.lower[j] = P.col_lower[j];
    F.upper[j] = P.col_upper[j];
  }
  for (int32_t i = 0, j = n0; i < m; ++i)
    if (F.rowKind[i] == kRowBound) { F.lower[j] = P.row_lower[i]; F.upper[j] = P.row_upper[i]; ++j; }
  for (int32_t j = 0; j < F.n; ++j) {
    if (F.lower[j] < -kBoundInf) F.lower[j] = -kInf;
    if (F.upper[j] > kBoundInf) F.upper[j] = kInf;
  }
  extractHessian(P, F.sense, F.n, F.qdiag, F.qoff);

  // Matrix in the reference's entry order: per column, equality-type entries
  // first, then inequality entries (LEQ negated) (:413-436); one -1 per slack.
  Compressed& A = F.csc;
  A.beg.resize((size_t)F.n + 1);
  A.idx.resize((size_t)F.nnz);
  A.val.resize((size_t)F.nnz);
  int64_t k = 0;
  for (int32_t j = 0; j < n0; ++j) {
    A.beg[j] = (int32_t)k;
    const int32_t b = P.a_start[j], e = P.a_start[j + 1];
    if (e < b) throw std::runtime_error("a_start not monotone");
    for (int32_t p = b; p < e; ++p) {
      const int32_t r = P.a_index[p];
      if (r < 0 || r >= m) throw std::runtime_error("row index out of range");
      const int32_t kind = F.rowKind[r];
      if (kind == kRowEq || kind == kRowBound) { A.idx[k] = F.rowNewIdx[r]; A.val[k] = P.a_value[p]; ++k; }
    }
    for (int32_t p = b; p < e; ++p) {
      const int32_t r = P.a_index[p];
      const int32_t kind = F.rowKind[r];
      if (kind == kRowLeq) { A.idx[k] = F.rowNewIdx[r]; A.val[k] = -P.a_value[p]; ++k; }
      else if (kind == kRowGeq) { A.idx[k] = F.rowNewIdx[r]; A.val[k] = P.a_value[p]; ++k; }
    }
  }
  for (int32_t i = 0, j = n0; i < m; ++i)
    if (F.rowKind[i] == kRowBound) { A.beg[j] = (int32_t)k; A.idx[k] = F.rowNewIdx[i]; A.val[k] = -1.0; ++k; ++j; }
  A.beg[F.n] = (int32_t)k;

  // Termination norms are those of the UNSCALED formulated data (Init_Scaling
  // runs before PDHG_Scale_Data, CupdlpWrapper.cpp:110 vs :153).
  double s = 0.0;
  for (double v : F.cost) s += v * v;
  F.normCost = std::sqrt(s);
  s = 0.0;
  for (double v : F.rhs) s += v * v;
  F.normRhs = std::sqrt(s);
  F.colScale.assign(F.n, 1.0);
  F.rowScale.assign(m, 1.0);
}

namespace {
// One diagonal rescaling D_r^-1 A D_c^-1 with the reference's operation order
// (scale_problem, cupdlp_scaling.c:17-45): rows first, then columns.
void applyScaling(StandardForm& F, const std::vector<double>& cs, const std::vector<double>& rs) {
  Compressed& A = F.csc;
  for (int32_t j = 0; j < F.n; ++j) {
    F.cost[j] /= cs[j];
    F.lower[j] *= cs[j];
    F.upper[j] *= cs[j];
    F.colScale[j] *= cs[j];
  }
  if (!F.qdiag.empty())  // x = x'/cs  =>  1/2 q x^2 = 1/2 (q / cs^2) x'^2
    for (int32_t j = 0; j < F.n; ++j) F.qdiag[j] = (F.qdiag[j] / cs[j]) / cs[j];
  if (!F.qoff.beg.empty())
    for (int32_t r = 0; r < F.n; ++r)
      for (int32_t p = F.qoff.beg[r]; p < F.qoff.beg[r + 1]; ++p) F.qoff.val[p] = (F.qoff.val[p] / cs[r]) / cs[F.qoff.idx[p]];
  for (int32_t i = 0; i < F.m; ++i) {
    F.rhs[i] /= rs[i];
    F.rowScale[i] *= rs[i];
  }
  for (int32_t j = 0; j < F.n; ++j) {
    const double c = cs[j];
    for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) A.val[p] = (A.val[p] / rs[A.idx[p]]) / c;
  }
}
}  // namespace

void scale(StandardForm& F, int ruizTimes, double pcAlpha) {
  Compressed& A = F.csc;
  std::vector<double> cs(F.n), rs(F.m);
  // Ruiz equilibration in the infinity norm (cupdlp_ruiz_scaling :47-120)
  for (int it = 0; it < ruizTimes; ++it) {
    std::fill(rs.begin(), rs.end(), 0.0);
    for (int32_t j = 0; j < F.n; ++j) {
      double mx = 0.0;
      for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) {
        const double a = std::fabs(A.val[p]);
        if (a > mx) mx = a;
        if (rs[A.idx[p]] < a) rs[A.idx[p]] = a;
      }
      cs[j] = mx == 0.0 ? 1.0 : std::sqrt(mx);
    }
    for (int32_t i = 0; i < F.m; ++i) rs[i] = rs[i] == 0.0 ? 1.0 : std::sqrt(rs[i]);
    applyScaling(F, cs, rs);
  }
  // Pock-Chambolle (cupdlp_pc_scaling :174-231).  The reference fixes alpha=1
  // (Init_Scaling :409), for which pow(|a|,alpha) and pow(s,1/alpha) are exact
  // identities; the general form is kept for other alpha.
  if (pcAlpha < 0.0 || pcAlpha > 2.0) throw std::runtime_error("PC alpha must be in [0,2]");
  if (F.m > 0) {
    std::fill(rs.begin(), rs.end(), 0.0);
    for (int32_t j = 0; j < F.n; ++j) {
      double sc = 0.0;
      for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) {
        const double a = std::fabs(A.val[p]);
        sc += std::pow(a, pcAlpha);
        rs[A.idx[p]] += std::pow(a, 2.0 - pcAlpha);
      }
      sc = std::sqrt(std::pow(sc, 1.0 / pcAlpha));
      cs[j] = sc == 0.0 ? 1.0 : sc;
    }
    for (int32_t i = 0; i < F.m; ++i) {
      const double r = std::sqrt(std::pow(rs[i], 1.0 / (2.0 - pcAlpha)));
      rs[i] = r == 0.0 ? 1.0 : r;
    }
  } else {
    std::fill(cs.begin(), cs.end(), 1.0);
  }
  applyScaling(F, cs, rs);
  F.scaled = true;
}

namespace {
// Counting transpose: output majors hold their entries in ascending input-major order.
void transpose(const Compressed& in, int32_t nMajorIn, int32_t nMajorOut, Compressed& out) {
  const int64_t nnz = in.beg[nMajorIn];
  out.beg.assign((size_t)nMajorOut + 1, 0);
  out.idx.resize((size_t)nnz);
  out.val.resize((size_t)nnz);
  for (int64_t p = 0; p < nnz; ++p) ++out.beg[in.idx[p] + 1];
  for (int32_t i = 0; i < nMajorOut; ++i) out.beg[i + 1] += out.beg[i];
  std::vector<int32_t> pos(out.beg.begin(), out.beg.end() - 1);
  for (int32_t j = 0; j < nMajorIn; ++j)
    for (int32_t p = in.beg[j]; p < in.beg[j + 1]; ++p) {
      const int32_t q = pos[in.idx[p]]++;
      out.idx[q] = j;
      out.val[q] = in.val[p];
    }
}
}  // namespace

void finalize(StandardForm& F) {
  transpose(F.csc, F.n, F.m, F.csr);        // rows, ascending column
  transpose(F.csr, F.m, F.n, F.cscSorted);  // columns, ascending row
  double mx = 0.0;
  for (double v : F.csc.val) mx = std::max(mx, std::fabs(v));
  F.matNormInf = mx;
}

std::vector<int32_t> rowPartition(const Compressed& csr, int32_t m, int32_t world) {
  std::vector<int32_t> off((size_t)world + 1, m);
  off[0] = 0;
  const int64_t nnz = csr.beg[m];
  // weight = nnz + rows so that empty rows still spread evenly
  const double total = (double)nnz + (double)m;
  int32_t r = 0;
  for (int32_t g = 1; g < world; ++g) {
    const double target = total * (double)g / (double)world;
    while (r < m && (double)csr.beg[r] + (double)r < target) ++r;
    off[g] = r;
  }
  off[world] = m;
  return off;
}

void extractSlab(const StandardForm& F, int32_t r0, int32_t r1, Compressed& csrSlab, Compressed& cscSlab) {
  const int32_t mLoc = r1 - r0;
  const int32_t b = F.csr.beg[r0], e = F.csr.beg[r1];
  csrSlab.beg.resize((size_t)mLoc + 1);
  for (int32_t i = 0; i <= mLoc; ++i) csrSlab.beg[i] = F.csr.beg[r0 + i] - b;
  csrSlab.idx.assign(F.csr.idx.begin() + b, F.csr.idx.begin() + e);
  csrSlab.val.assign(F.csr.val.begin() + b, F.csr.val.begin() + e);
  transpose(csrSlab, mLoc, F.n, cscSlab);
}

StreamPlan planStream(const std::vector<int32_t>& beg, int32_t nMajor, int32_t chunk, int32_t maxMajorsPerBlock) {
  StreamPlan plan;
  int32_t start = 0;
  while (start < nMajor) {
    if (beg[start + 1] - beg[start] > chunk) {  // a long major: segment tasks, no block
      plan.longMajors.push_back(start);
      ++start;
      continue;
    }
    const int32_t base = beg[start];
    int32_t end = start;
    // extend while the block stays within `chunk` nonzeros and the major cap
    while (end < nMajor && end - start < maxMajorsPerBlock && beg[end + 1] - base <= chunk) ++end;
    plan.blockBeg.push_back(start);
    plan.blockBeg.push_back(end);
    plan.blockBeg.push_back(beg[start]);  // (first and end entry ride along: one dependent load less in the kernels)
    plan.blockBeg.push_back(beg[end]);
    start = end;
  }
  plan.nBlocks = (int32_t)plan.blockBeg.size() / 4;
  return plan;
}

LongPlan planLong(const std::vector<int32_t>& beg, const std::vector<int32_t>& longMajors, const int32_t* vecIndex,
                  int32_t W, const std::function<int(int32_t, int32_t)>* homeOf, int32_t firstXcd) {
  constexpr int32_t kSeg = 512, kMaxSeg = 64;  // pdlp_kernels.hpp kLongSegment, kLongMaxSegments
  LongPlan L;
  L.nLong = (int32_t)longMajors.size();
  std::vector<LongTaskHost> real;  // XCD-affine order: the tasks in (major, segment) order first
  int32_t slot = 0;                // first segment-sum slot of the major
  for (int32_t c = 0; c < L.nLong; ++c) {
    const int32_t r = longMajors[c];
    const int32_t p0 = beg[r], len = beg[r + 1] - beg[r];
    int64_t seg = kSeg;
    while ((len + seg - 1) / seg > kMaxSeg) seg *= 2;
    const int32_t nSeg = (int32_t)((len + seg - 1) / seg);
    const bool contained = homeOf ? nSeg == 1 : nSeg <= W;
    if (contained && !homeOf)
      while ((int32_t)(L.tasks.size() % W) + nSeg > W) L.tasks.push_back(LongTaskHost{0, 0, -1, (int32_t)L.tasks.size(), 1, 0, 1, 0});
    const int32_t first = homeOf ? slot : (int32_t)L.tasks.size();
    for (int32_t k = 0; k < nSeg; ++k) {
      const int64_t a = p0 + (int64_t)k * seg, b = std::min<int64_t>(p0 + len, a + seg);
      (homeOf ? real : L.tasks).push_back(LongTaskHost{(int32_t)a, (int32_t)b, c, first, nSeg, vecIndex ? vecIndex[c] : r, contained ? 1 : 0, k});
    }
    slot += nSeg;
  }
  if (homeOf) {
    // XCD-affine deal (slab layout).  Task workgroup lb is workgroup nStreamingBlocks + lb of its launch and runs on XCD
    // (firstXcd + lb) % 8 (dispatch order; a speed assumption only).  A task goes to a workgroup of the XCD whose streaming
    // blocks gather from the same stretch of the vector (homeOf), so that its gathers hit lines that L2 holds anyway instead
    // of pulling a second copy of them into another XCD's L2 (bench.py --config c, A x+: 2.18x the needed HBM bytes before).
    // No XCD takes more than its share of the workgroups: what does not fit goes to the XCD with the most room.  Which
    // workgroup runs a task changes no sum: a major's segment sums meet in HBM slots (first + seg) and are added left to right.
    constexpr int kXcds = 8;
    const int32_t nReal = (int32_t)real.size();
    const int32_t G = (nReal + W - 1) / W;
    int32_t cap[kXcds] = {0};
    for (int32_t lb = 0; lb < G; ++lb) cap[(firstXcd + lb) % kXcds] += W;
    std::vector<int32_t> list[kXcds], spill;
    for (int32_t i = 0; i < nReal; ++i) {
      int h = (*homeOf)(real[i].pBeg, real[i].pEnd);
      h = ((h % kXcds) + kXcds) % kXcds;
      if ((int32_t)list[h].size() < cap[h]) list[h].push_back(i); else spill.push_back(i);
    }
    for (int32_t i : spill) {
      int best = 0;
      for (int x = 1; x < kXcds; ++x)
        if (cap[x] - (int32_t)list[x].size() > cap[best] - (int32_t)list[best].size()) best = x;
      list[best].push_back(i);
    }
    size_t next[kXcds] = {0};
    L.tasks.reserve((size_t)G * W);
    for (int32_t lb = 0; lb < G; ++lb) {
      const int x = (firstXcd + lb) % kXcds;
      for (int32_t w = 0; w < W; ++w) {
        if (next[x] < list[x].size()) L.tasks.push_back(real[(size_t)list[x][next[x]++]]);
        else L.tasks.push_back(LongTaskHost{0, 0, -1, 0, 1, 0, 1, 0});  // idle
      }
    }
    L.nSegSlots = slot;
  }
  L.nTasks = (int32_t)L.tasks.size();
  if (!homeOf) L.nSegSlots = L.nTasks;
  return L;
}

int32_t xcdTileLog2(int32_t nMinor) {
  int b = 0;
  while (((int64_t)1 << b) < (int64_t)nMinor) ++b;
  return std::max(14, b - 12);  // at most 4096 tiles (an LDS histogram of 16 KB in the device build)
}
std::vector<int8_t> xcdTileOwners(const std::vector<int32_t>& hist, int32_t nTiles) {
  std::vector<int8_t> owner((size_t)std::max(nTiles, 1), -1);
  for (int32_t t = 0; t < nTiles; ++t) {
    int32_t best = 0;
    for (int x = 0; x < 8; ++x)
      if (hist[(size_t)x * nTiles + t] > best) { best = hist[(size_t)x * nTiles + t]; owner[t] = (int8_t)x; }
  }
  // tiles nobody gathers from: the nearest owned tile to the left, else to the right, else proportional
  int8_t last = -1;
  for (int32_t t = 0; t < nTiles; ++t) { if (owner[t] >= 0) last = owner[t]; else owner[t] = last; }
  last = -1;
  for (int32_t t = nTiles - 1; t >= 0; --t) { if (owner[t] >= 0) last = owner[t]; else owner[t] = last; }
  for (int32_t t = 0; t < nTiles; ++t) if (owner[t] < 0) owner[t] = (int8_t)(((int64_t)t * 8) / nTiles);
  return owner;
}
int xcdHomeOf(const int32_t* idx, int32_t pBeg, int32_t pEnd, const std::vector<int8_t>& owner, int32_t tileLog2) {
  const int64_t len = (int64_t)pEnd - pBeg;
  if (len <= 0 || owner.empty()) return 0;
  int votes[8] = {0};
  for (int k = 0; k < 8; ++k) {
    const int64_t p = pBeg + ((2 * k + 1) * len) / 16;
    size_t t = (size_t)(idx[p] >> tileLog2);
    if (t >= owner.size()) t = owner.size() - 1;
    ++votes[owner[t] & 7];
  }
  int best = 0;
  for (int x = 1; x < 8; ++x) if (votes[x] > votes[best]) best = x;
  return best;
}

namespace {
int32_t bitsFor(int64_t count) {  // smallest b with 2^b >= count
  int b = 0;
  while (((int64_t)1 << b) < count) ++b;
  return b;
}
// Fills `units` consecutive ranges of [r0, r1) by work (see SlabPartition): out[0..units] boundaries.
void fillByWork(const int32_t* beg, const int32_t* cold, int32_t longLimit, int32_t majorCost, int32_t r0, int32_t r1, int32_t units, int64_t cap, bool nonEmpty,
                int32_t* out) {
  auto cost = [&](int32_t r) -> int64_t { return slabMajorWork(beg[r + 1] - beg[r], cold ? cold[r] : 0, longLimit, majorCost); };
  int64_t rem = 0;
  for (int32_t r = r0; r < r1; ++r) rem += cost(r);
  int32_t r = r0;
  out[0] = r0;
  for (int32_t u = 0; u < units; ++u) {
    const int64_t left = units - u;
    const int64_t target = (rem + left - 1) / left;
    const int64_t rows = (int64_t)r1 - r;
    const int64_t minRows = std::max<int64_t>(nonEmpty && rows > 0 ? 1 : 0, rows - (left - 1) * cap);
    const int64_t maxRows = std::min<int64_t>(cap, nonEmpty ? std::max<int64_t>(rows - (left - 1), 1) : rows);
    int64_t acc = 0, cnt = 0;
    while (cnt < rows && cnt < maxRows) {  // a major is taken while the range is closer to its target with it than without
      const int64_t c = cost(r);
      if (cnt >= minRows && 2 * acc + c > 2 * target) break;
      acc += c; ++r; ++cnt;
    }
    rem -= acc;
    out[u + 1] = r;
  }
}
}  // namespace

int64_t slabMajorWork(int32_t len, int32_t nCold, int32_t longLimit, int32_t majorCost) {
  if (len > longLimit) return majorCost;  // (its segment tasks run elsewhere)
  return (int64_t)len + (int64_t)nCold * (kSlabColdWeight - 1) + ((int64_t)len * std::min(len, 64)) / 32 + majorCost;
}

void slabColdCounts(const int32_t* beg, const int32_t* idx, int32_t nMajor, int32_t nMinor, int32_t longLimit, int32_t* cold) {
  std::vector<int32_t> count((size_t)std::max(nMinor, 1), 0);
  const int64_t nnz = nMajor > 0 ? beg[nMajor] : 0;
  for (int64_t p = 0; p < nnz; ++p) ++count[idx[p]];
  for (int32_t r = 0; r < nMajor; ++r) {
    const int32_t p0 = beg[r], len = beg[r + 1] - beg[r];
    int32_t c = 0;
    if (len >= 2 && len <= longLimit) {
      const int32_t mid = idx[p0 + len / 2];
      for (int32_t p = p0; p < p0 + len; ++p) {
        const int32_t d = idx[p] > mid ? idx[p] - mid : mid - idx[p];
        if (d >= kSlabFar && count[idx[p]] <= kSlabHotCount) ++c;
      }
    }
    cold[r] = c;
  }
}

bool slabFits(int32_t nMajor, int32_t nMinor) {
  (void)nMajor;
  return bitsFor(nMinor) <= 28;  // at least 16 majors per wave
}

SlabPartition slabPartition(const int32_t* beg, const int32_t* cold, int32_t nMajor, int32_t nMinor, int32_t longLimit,
                            int32_t majorCost) {
  SlabPartition P;
  if (!slabFits(nMajor, nMinor)) throw std::runtime_error("slab layout: minor index does not fit the entry packing");
  P.minorBits = std::max(bitsFor(nMinor), 4);
  const int64_t waveCap = std::min<int64_t>((int64_t)1 << (32 - P.minorBits), kSlabBlockRowCap);
  const int64_t blockCap = std::min<int64_t>(kSlabBlockRowCap, waveCap * kSlabWavesPerBlock);
  int64_t nB = ((int64_t)nMajor + kSlabMinRowsPerBlock - 1) / kSlabMinRowsPerBlock;
  nB = std::min<int64_t>(nB, kSlabTargetBlocks);
  nB = std::max<int64_t>(nB, ((int64_t)nMajor + blockCap - 1) / blockCap);
  P.nBlocks = (int32_t)nB;
  P.waveBeg.assign((size_t)nB * kSlabWavesPerBlock + 1, 0);
  if (nB == 0) return P;
  std::vector<int32_t> blockBeg((size_t)nB + 1);
  fillByWork(beg, cold, longLimit, majorCost, 0, nMajor, (int32_t)nB, blockCap, true, blockBeg.data());
  for (int32_t b = 0; b < (int32_t)nB; ++b) {
    fillByWork(beg, cold, longLimit, majorCost, blockBeg[b], blockBeg[b + 1], kSlabWavesPerBlock, waveCap, false,
               P.waveBeg.data() + (size_t)b * kSlabWavesPerBlock);
    P.maxRowsPerBlock = std::max(P.maxRowsPerBlock, blockBeg[b + 1] - blockBeg[b]);
  }
  return P;
}

void buildSlabLayout(const Compressed& csr, int32_t nMajor, int32_t nMinor, int32_t longLimit, int32_t slabWidthLog2,
                     int32_t majorCost, SlabLayout& out) {
  out = SlabLayout();
  std::vector<int32_t> cold((size_t)std::max(nMajor, 1));
  slabColdCounts(csr.beg.data(), csr.idx.data(), nMajor, nMinor, longLimit, cold.data());
  SlabPartition P = slabPartition(csr.beg.data(), cold.data(), nMajor, nMinor, longLimit, majorCost);
  out.rowsPerBlock = P.maxRowsPerBlock;
  out.nBlocks = P.nBlocks;
  out.minorBits = P.minorBits;
  out.slabWidthLog2 = slabWidthLog2;
  out.waveBeg = std::move(P.waveBeg);
  const int32_t nWaves = out.nBlocks * kSlabWavesPerBlock;
  const int32_t S = std::max(1, (int32_t)(((int64_t)nMinor + ((int64_t)1 << slabWidthLog2) - 1) >> slabWidthLog2));
  if ((int64_t)nWaves * S >= (int64_t)0x7fffffff) throw std::runtime_error("slab layout: too many segments");
  out.longMask.assign(((size_t)nMajor + 31) / 32 + 1, 0u);
  out.longCsr.beg.push_back(0);
  // pass 1: per (wave, slab) counts; long majors go to the side CSR
  std::vector<int32_t> count((size_t)nWaves * S, 0);
  for (int32_t w = 0; w < nWaves; ++w)
    for (int32_t r = out.waveBeg[w]; r < out.waveBeg[w + 1]; ++r) {
      const int32_t len = csr.beg[r + 1] - csr.beg[r];
      if (len > longLimit) {
        out.longMask[(size_t)r >> 5] |= 1u << (r & 31);
        out.longMap.push_back(r);
        out.longCsr.idx.insert(out.longCsr.idx.end(), csr.idx.begin() + csr.beg[r], csr.idx.begin() + csr.beg[r + 1]);
        out.longCsr.val.insert(out.longCsr.val.end(), csr.val.begin() + csr.beg[r], csr.val.begin() + csr.beg[r + 1]);
        out.longCsr.beg.push_back((int32_t)out.longCsr.idx.size());
        continue;
      }
      for (int32_t p = csr.beg[r]; p < csr.beg[r + 1]; ++p) ++count[(size_t)w * S + (csr.idx[p] >> slabWidthLog2)];
    }
  // exclusive scan in (wave, slab) order; wavePtr = the wave boundaries of it
  out.wavePtr.assign((size_t)nWaves + 1, 0);
  std::vector<int32_t> pos((size_t)nWaves * S);
  int64_t acc = 0;
  for (int32_t w = 0; w < nWaves; ++w) {
    out.wavePtr[w] = (int32_t)acc;
    for (int32_t k = 0; k < S; ++k) {
      pos[(size_t)w * S + k] = (int32_t)acc;
      acc += count[(size_t)w * S + k];
    }
  }
  out.wavePtr[nWaves] = (int32_t)acc;
  out.ent.resize((size_t)acc);
  out.val.resize((size_t)acc);
  // pass 2: majors in order, minors ascending within a major => each (wave, slab) segment comes out
  // sorted by (local major, minor)
  for (int32_t w = 0; w < nWaves; ++w)
    for (int32_t r = out.waveBeg[w]; r < out.waveBeg[w + 1]; ++r) {
      const int32_t len = csr.beg[r + 1] - csr.beg[r];
      if (len > longLimit) continue;
      const uint32_t lr = (uint32_t)(r - out.waveBeg[w]);
      for (int32_t p = csr.beg[r]; p < csr.beg[r + 1]; ++p) {
        const int32_t c = csr.idx[p];
        const int32_t q = pos[(size_t)w * S + (c >> slabWidthLog2)]++;
        out.ent[q] = (lr << out.minorBits) | (uint32_t)c;
        out.val[q] = csr.val[p];
      }
    }
}


std::vector<int32_t> slabTileHistogram(const int32_t* beg, const int32_t* idx, const SlabPartition& part, int32_t longLimit,
                                       int32_t tileLog2, int32_t nTiles, std::vector<int32_t>& lo, std::vector<int32_t>& hi,
                                       std::vector<int32_t>& cnt) {
  const int32_t nB = part.nBlocks;
  lo.assign((size_t)nB, std::numeric_limits<int32_t>::max());
  hi.assign((size_t)nB, -1);
  cnt.assign((size_t)nB, 0);
  std::vector<int32_t> hist((size_t)8 * nTiles, 0);
  for (int32_t b = 0; b < nB; ++b) {
    int32_t* hx = hist.data() + (size_t)xcdOfLogicalBlock(b, nB) * nTiles;
    for (int32_t r = part.blockBeg(b); r < part.blockBeg(b + 1); ++r) {
      const int32_t p0 = beg[r], p1 = beg[r + 1];
      if (p1 <= p0 || p1 - p0 > longLimit) continue;
      lo[b] = std::min(lo[b], idx[p0]); hi[b] = std::max(hi[b], idx[p1 - 1]); cnt[b] += p1 - p0;
      for (int32_t p = p0; p < p1; ++p) ++hx[idx[p] >> tileLog2];
    }
  }
  return hist;
}

LongPlan planSlabTasks(const std::vector<int32_t>& longBeg, const int32_t* longIdx, int32_t nLong, const int32_t* longMap, bool balance,
                       const std::vector<int8_t>* tileOwner, int32_t tileLog2, int32_t nSlabBlocks, int32_t& taskGroup) {
  constexpr int32_t kSeg = 512, kMaxSeg = 64;  // pdlp_kernels.hpp kLongSegment, kLongMaxSegments
  taskGroup = kSlabWavesPerBlock;
  if (balance) {
    int64_t segs = 0;
    for (int32_t c = 0; c < nLong; ++c) {
      const int64_t len = longBeg[c + 1] - longBeg[c];
      int64_t seg = kSeg;
      while ((len + seg - 1) / seg > kMaxSeg) seg *= 2;
      segs += (len + seg - 1) / seg;
    }
    // one task workgroup per CU where there are that many tasks: ceil(segments / 256) tasks each, 1 .. 16
    taskGroup = (int32_t)std::min<int64_t>(kSlabWavesPerBlock, std::max<int64_t>(1, (segs + kSlabTargetBlocks - 1) / kSlabTargetBlocks));
  }
  std::vector<int32_t> all((size_t)nLong);
  for (int32_t c = 0; c < nLong; ++c) all[c] = c;
  std::function<int(int32_t, int32_t)> homeOf;
  if (tileOwner && longIdx)
    homeOf = [&](int32_t pBeg, int32_t pEnd) { return xcdHomeOf(longIdx, pBeg, pEnd, *tileOwner, tileLog2); };
  return planLong(longBeg, all, longMap, taskGroup, homeOf ? &homeOf : nullptr, nSlabBlocks % 8);
}

}  // namespace pdlp
