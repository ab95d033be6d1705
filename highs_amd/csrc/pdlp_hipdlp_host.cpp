// pdlp_hipdlp_host.cpp — host-side preparation of the HiPDLP path (see pdlp_host.hpp).
#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <utility>

#include "pdlp_host.hpp"

namespace pdlp {

namespace {
const double kInf = std::numeric_limits<double>::infinity();
}

void formulateHipdlp(const pdlp_problem_t& P, StandardForm& F) {
  validateProblem(P);
  {
    std::vector<double> q;
    extractDiagonalHessian(P, 1.0, P.num_col, q);
    if (!q.empty()) throw std::runtime_error("pdlp_mi355x: quadratic objectives are solved by the pdlp path only (algorithm = 0)");
  }
  const int32_t n0 = P.num_col, m = P.num_row;
  const int64_t nnz0 = n0 > 0 ? P.a_start[n0] : 0;
  if (nnz0 > 0 && (!P.a_index || !P.a_value)) throw std::runtime_error("null matrix arrays");
  F = StandardForm();
  F.n0 = n0;
  F.m = m;
  F.offset = P.offset;
  F.sense = P.sense < 0 ? -1.0 : 1.0;  // only used for col_dual (pdhg.cc:481)

  F.rowKind.resize(m);
  F.rowNewIdx.resize(m);
  int32_t nSlack = 0, nEq = 0;
  for (int32_t i = 0; i < m; ++i) {  // pdhg.cc:175-197
    const bool lo = P.row_lower[i] > -kInf, up = P.row_upper[i] < kInf;
    int32_t k;
    if (lo && up) k = (P.row_lower[i] == P.row_upper[i]) ? (int32_t)kRowEq : (int32_t)kRowBound;
    else if (lo) k = kRowGeq;
    else if (up) k = kRowLeq;
    else k = kRowFree;
    F.rowKind[i] = k;
    if (k == kRowEq || k == kRowBound || k == kRowFree) ++nEq;
    if (k == kRowBound || k == kRowFree) ++nSlack;
  }
  if ((int64_t)n0 + nSlack > std::numeric_limits<int32_t>::max() ||
      nnz0 + nSlack > std::numeric_limits<int32_t>::max())
    throw std::runtime_error("problem exceeds 32-bit index range");
  F.n = n0 + nSlack;
  F.nEqs = nEq;
  F.nnz = nnz0 + nSlack;
  auto isEqKind = [](int32_t k) { return k == kRowEq || k == kRowBound || k == kRowFree; };
  int32_t eqPos = 0, inPos = nEq;
  for (int32_t i = 0; i < m; ++i) F.rowNewIdx[i] = isEqKind(F.rowKind[i]) ? eqPos++ : inPos++;
  F.rowIsEq.assign((size_t)m, 0);
  for (int32_t i = 0; i < m; ++i) F.rowIsEq[F.rowNewIdx[i]] = isEqKind(F.rowKind[i]) ? 1 : 0;

  F.cost.assign(F.n, 0.0);
  F.lower.resize(F.n);
  F.upper.resize(F.n);
  for (int32_t j = 0; j < n0; ++j) { F.cost[j] = P.col_cost[j]; F.lower[j] = P.col_lower[j]; F.upper[j] = P.col_upper[j]; }
  for (int32_t i = 0, j = n0; i < m; ++i)
    if (F.rowKind[i] == kRowBound || F.rowKind[i] == kRowFree) { F.lower[j] = P.row_lower[i]; F.upper[j] = P.row_upper[i]; ++j; }
  F.rhs.assign(m, 0.0);
  F.rowUpper.assign(m, 0.0);
  for (int32_t i = 0; i < m; ++i) {  // pdhg.cc:253-277
    const int32_t r = F.rowNewIdx[i];
    switch (F.rowKind[i]) {
      case kRowEq: F.rhs[r] = P.row_lower[i]; F.rowUpper[r] = P.row_upper[i]; break;
      case kRowGeq: F.rhs[r] = P.row_lower[i]; F.rowUpper[r] = kInf; break;
      case kRowLeq: F.rhs[r] = -P.row_upper[i]; F.rowUpper[r] = kInf; break;
      default: F.rhs[r] = 0.0; F.rowUpper[r] = 0.0; break;
    }
  }
  Compressed& A = F.csc;
  A.beg.resize((size_t)F.n + 1);
  A.idx.resize((size_t)F.nnz);
  A.val.resize((size_t)F.nnz);
  int64_t k = 0;
  std::vector<std::pair<int32_t, double>> col;
  for (int32_t j = 0; j < n0; ++j) {  // pdhg.cc:296-320: entries sorted by (new row, value)
    A.beg[j] = (int32_t)k;
    const int32_t b = P.a_start[j], e = P.a_start[j + 1];
    if (e < b) throw std::runtime_error("a_start not monotone");
    col.clear();
    for (int32_t p = b; p < e; ++p) {
      const int32_t r = P.a_index[p];
      if (r < 0 || r >= m) throw std::runtime_error("row index out of range");
      double v = P.a_value[p];
      if (F.rowKind[r] == kRowLeq) v = -v;
      col.push_back({F.rowNewIdx[r], v});
    }
    std::sort(col.begin(), col.end());
    for (const auto& en : col) { A.idx[k] = en.first; A.val[k] = en.second; ++k; }
  }
  for (int32_t i = 0, j = n0; i < m; ++i)
    if (F.rowKind[i] == kRowBound || F.rowKind[i] == kRowFree) {
      A.beg[j] = (int32_t)k; A.idx[k] = F.rowNewIdx[i]; A.val[k] = -1.0; ++k; ++j;
    }
  A.beg[F.n] = (int32_t)k;
  // unscaled_c_norm_ / unscaled_rhs_norm_ (pdhg.cc:343-344): norm2 = sqrt(dot), left to right
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
// Scaling::applyScaling, scaling.cc:222-262 (+ cumulative factors)
void applyScalingHipdlp(StandardForm& F, const std::vector<double>& cs, const std::vector<double>& rs) {
  Compressed& A = F.csc;
  for (int32_t j = 0; j < F.n; ++j) F.cost[j] /= cs[j];
  for (int32_t j = 0; j < F.n; ++j) {
    if (F.lower[j] > -kInf) F.lower[j] *= cs[j];
    if (F.upper[j] < kInf) F.upper[j] *= cs[j];
  }
  for (int32_t i = 0; i < F.m; ++i) {
    if (F.rhs[i] > -kInf) F.rhs[i] /= rs[i];
    if (F.rowUpper[i] < kInf) F.rowUpper[i] /= rs[i];
  }
  for (int32_t j = 0; j < F.n; ++j)
    for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) A.val[p] /= (rs[A.idx[p]] * cs[j]);
  for (int32_t j = 0; j < F.n; ++j) F.colScale[j] *= cs[j];
  for (int32_t i = 0; i < F.m; ++i) F.rowScale[i] *= rs[i];
}
}  // namespace

void scaleHipdlp(StandardForm& F, bool ruiz, bool pc, bool l2, int ruizIters) {
  Compressed& A = F.csc;
  std::vector<double> cs(F.n), rs(F.m);
  F.scaled = false;
  if (ruiz) {  // applyRuizScaling, scaling.cc:59-125
    for (int it = 0; it < ruizIters; ++it) {
      std::fill(rs.begin(), rs.end(), 0.0);
      for (int32_t j = 0; j < F.n; ++j) {
        double mx = 0.0;
        for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) mx = std::max(mx, std::fabs(A.val[p]));
        cs[j] = (A.beg[j + 1] > A.beg[j]) ? std::sqrt(mx) : 0.0;
        if (cs[j] == 0.0) cs[j] = 1.0;
      }
      for (int32_t j = 0; j < F.n; ++j)
        for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) rs[A.idx[p]] = std::max(rs[A.idx[p]], std::fabs(A.val[p]));
      for (int32_t i = 0; i < F.m; ++i) rs[i] = rs[i] == 0.0 ? 1.0 : std::sqrt(rs[i]);
      applyScalingHipdlp(F, cs, rs);
    }
    F.scaled = true;
  }
  if (pc) {  // applyPockChambolleScaling, :127-178, alpha = 1: pow(v, 1) is the identity
    std::fill(rs.begin(), rs.end(), 0.0);
    for (int32_t j = 0; j < F.n; ++j) {
      double s = 0.0;
      for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) s += std::fabs(A.val[p]);
      cs[j] = s > 0.0 ? std::sqrt(s) : 1.0;
    }
    for (int32_t j = 0; j < F.n; ++j)
      for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) rs[A.idx[p]] += std::fabs(A.val[p]);
    for (int32_t i = 0; i < F.m; ++i) rs[i] = rs[i] > 0.0 ? std::sqrt(rs[i]) : 1.0;
    applyScalingHipdlp(F, cs, rs);
    F.scaled = true;
  }
  if (l2) {  // applyL2Scaling, :180-220
    std::fill(rs.begin(), rs.end(), 0.0);
    for (int32_t j = 0; j < F.n; ++j) {
      double s = 0.0;
      for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) s += A.val[p] * A.val[p];
      cs[j] = s > 0.0 ? std::sqrt(std::sqrt(s)) : 1.0;
    }
    for (int32_t j = 0; j < F.n; ++j)
      for (int32_t p = A.beg[j]; p < A.beg[j + 1]; ++p) rs[A.idx[p]] += A.val[p] * A.val[p];
    for (int32_t i = 0; i < F.m; ++i) rs[i] = rs[i] > 0.0 ? std::sqrt(std::sqrt(rs[i])) : 1.0;
    applyScalingHipdlp(F, cs, rs);
    F.scaled = true;
  }
}

}  // namespace pdlp
