// pdlp_synth.cpp — synthetic random sparse box-constrained LP generator of
// SURVEY.md §8(d) / BASELINE.md §3 (the workload the headline metric is quoted
// on).  Draw order matters and is kept exactly: seed std::mt19937_64(seed);
// x*_j ~ U(0,1) for all j; c_j ~ N(0,1) for all j; then per row: k = nnz/m
// column draws rng()%n, sort+unique, one N(0,1) value per kept column
// (accumulating a_i.x*), and for odd rows one U(0,1) slack.  Even rows are
// equalities at x*, odd rows are  a_i.x <= a_i.x* + U.  Bounds 0 <= x <= 1.
// The LP is returned column-wise, rows ascending within a column (what HiGHS'
// passModel produces from the row-wise input).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "../../include/pdlp_mi355x.h"

namespace {
int genSynthetic(int32_t m, int32_t n, int64_t nnz_target, uint64_t seed, pdlp_problem_t* P) {
  if (!P || m <= 0 || n <= 0 || nnz_target < m) return 1;
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> N(0.0, 1.0);
  const int perRow = (int)(nnz_target / m);
  std::vector<double> xs(n);
  for (auto& v : xs) v = U(rng);
  double* cost = (double*)malloc(sizeof(double) * n);
  for (int j = 0; j < n; ++j) cost[j] = N(rng);
  std::vector<int32_t> rStart((size_t)m + 1), rIdx;
  std::vector<double> rVal;
  rIdx.reserve((size_t)nnz_target);
  rVal.reserve((size_t)nnz_target);
  double* rl = (double*)malloc(sizeof(double) * m);
  double* ru = (double*)malloc(sizeof(double) * m);
  std::vector<int> cols(perRow);
  const double inf = std::numeric_limits<double>::infinity();
  for (int i = 0; i < m; ++i) {
    rStart[i] = (int32_t)rIdx.size();
    for (int k = 0; k < perRow; ++k) cols[k] = (int)(rng() % (uint64_t)n);
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    double ax = 0;
    for (int c : cols) {
      const double v = N(rng);
      rIdx.push_back(c);
      rVal.push_back(v);
      ax += v * xs[c];
    }
    cols.resize(perRow);
    if (i % 2 == 0) { rl[i] = ax; ru[i] = ax; }
    else { rl[i] = -inf; ru[i] = ax + U(rng); }
  }
  rStart[m] = (int32_t)rIdx.size();
  const int64_t nnz = (int64_t)rIdx.size();
  // row-wise -> column-wise (counting transpose, rows ascending per column)
  int32_t* aStart = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  int32_t* aIndex = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
  double* aValue = (double*)malloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
  for (int64_t p = 0; p < nnz; ++p) ++aStart[rIdx[p] + 1];
  for (int j = 0; j < n; ++j) aStart[j + 1] += aStart[j];
  std::vector<int32_t> pos(aStart, aStart + n);
  for (int i = 0; i < m; ++i)
    for (int32_t p = rStart[i]; p < rStart[i + 1]; ++p) {
      const int32_t q = pos[rIdx[p]]++;
      aIndex[q] = i;
      aValue[q] = rVal[p];
    }
  double* cl = (double*)malloc(sizeof(double) * n);
  double* cu = (double*)malloc(sizeof(double) * n);
  for (int j = 0; j < n; ++j) { cl[j] = 0.0; cu[j] = 1.0; }
  memset(P, 0, sizeof(*P));
  P->num_col = n; P->num_row = m; P->num_nz = nnz;
  P->a_start = aStart; P->a_index = aIndex; P->a_value = aValue;
  P->col_cost = cost; P->col_lower = cl; P->col_upper = cu; P->row_lower = rl; P->row_upper = ru;
  P->offset = 0.0; P->sense = 1;
  return 0;
}
}  // namespace

// Nothing throws across the boundary: an allocation failure of the work vectors becomes return code 1.
extern "C" int pdlp_mi355x_gen_synthetic(int32_t m, int32_t n, int64_t nnz_target, uint64_t seed,
                                         pdlp_problem_t* P) {
  try {
    return genSynthetic(m, n, nnz_target, seed, P);
  } catch (...) {
    return 1;
  }
}

extern "C" void pdlp_mi355x_free_problem(pdlp_problem_t* P) {
  if (!P) return;
  free((void*)P->a_start); free((void*)P->a_index); free((void*)P->a_value);
  free((void*)P->col_cost); free((void*)P->col_lower); free((void*)P->col_upper);
  free((void*)P->row_lower); free((void*)P->row_upper);
  memset(P, 0, sizeof(*P));
}
