// pdlp_api.cpp — the extern "C" boundary (include/pdlp_mi355x.h).  Exceptions
// never cross it: every entry point catches, records the message for
// pdlp_mi355x_last_error() and returns non-zero (-> HighsStatus::kError).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pdlp_halpern.hpp"
#include "pdlp_mps.hpp"
#include "pdlp_solver.hpp"
#include "pdlp_detmath.h"

struct pdlp_mi355x_solver {
  pdlp::SolverBase* impl;
};

namespace {
thread_local std::string g_lastError;

template <typename Fn>
int guarded(Fn&& fn) {
  try {
    fn();
    return 0;
  } catch (const std::exception& e) {
    g_lastError = e.what();
  } catch (...) {
    g_lastError = "unknown exception";
  }
  return 1;
}
}  // namespace

namespace {
template <typename T>
T* dupVec(const std::vector<T>& v) {
  T* p = (T*)malloc(sizeof(T) * (v.size() ? v.size() : 1));
  if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}
}  // namespace

extern "C" {

const char* pdlp_mi355x_last_error(void) { return g_lastError.c_str(); }
int pdlp_mi355x_abi_version(void) { return PDLP_MI355X_ABI_VERSION; }

void pdlp_mi355x_default_params(pdlp_params_t* opt) {
  if (!opt) return;
  memset(opt, 0, sizeof(*opt));
  opt->primal_tol = 1e-7;  // kDefaultKktTolerance, HConst.h:345
  opt->dual_tol = 1e-7;
  opt->gap_tol = 1e-7;
  opt->time_limit = std::numeric_limits<double>::infinity();
  opt->iter_limit = std::numeric_limits<int32_t>::max();
  opt->features_off = 0;
  opt->restart_method = 1;
  opt->log_level = 0;
  opt->device = 0;
  opt->check_interval = 0;
  opt->algorithm = 0;
  opt->scaling_mode = 5;        // kPdlpScalingRuiz + kPdlpScalingPC (HighsOptions.h:1345-1349)
  opt->ruiz_iterations = 10;    // HighsOptions.h:1353-1355
  opt->step_size_strategy = 1;  // kPdlpStepSizeStrategyAdaptive (HighsOptions.h:1374-1378)
}

int pdlp_mi355x_create(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_mi355x_solver_t** out) {
  return pdlp_mi355x_create_sharded(P, opt, 0, 1, nullptr, out);
}

int pdlp_mi355x_create_sharded(const pdlp_problem_t* P, const pdlp_params_t* opt, int32_t rank, int32_t world,
                               const void* id128, pdlp_mi355x_solver_t** out) {
  return guarded([&] {
    if (!P || !opt || !out) throw std::runtime_error("null argument");
    *out = nullptr;
    pdlp::SolverBase* s = nullptr;
    if (opt->algorithm == 1) {
      s = new pdlp::HalpernSolver(*P, *opt, rank, world, id128);
    } else if (opt->algorithm == 0) {
      s = new pdlp::Solver(*P, *opt, rank, world, id128);
    } else {
      throw std::runtime_error("unknown algorithm (0 = cuPDLP-C path, 1 = HiPDLP path)");
    }
    *out = new pdlp_mi355x_solver{s};
  });
}

int pdlp_mi355x_run(pdlp_mi355x_solver_t* s, pdlp_result_t* R) {
  return guarded([&] {
    if (!s || !s->impl) throw std::runtime_error("null solver");
    s->impl->run(R);
  });
}

void pdlp_mi355x_destroy(pdlp_mi355x_solver_t* s) {
  if (!s) return;
  try {
    delete s->impl;
  } catch (...) {
  }
  delete s;
}

namespace {
// The multi-GPU form of the one-call boundary: the LP is row-block sharded over G devices of THIS process,
// one host thread per device (each thread owns its device's solver from construction to destruction; the
// ranks meet inside Mesh / Comm exactly as G processes would).  Rank 0 fills the caller's result.
void solveSharded(const pdlp_problem_t& P, const pdlp_params_t& opt, int G, pdlp_result_t* R) {
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0)
    throw std::runtime_error("pdlp_mi355x: no HIP device available (this library has no CPU fallback)");
  // PDLP_MI355X_FOLD_DEVICES=1 (tests on a 1-GPU box): several ranks share a physical device
  const char* f = pdlp::devEnv("PDLP_MI355X_FOLD_DEVICES");
  const bool fold = f && atoi(f) != 0;
  if (!fold && opt.device + G > nDev)
    throw std::runtime_error("pdlp_mi355x: num_devices = " + std::to_string(G) + " starting at device " +
                             std::to_string(opt.device) + ", but only " + std::to_string(nDev) + " HIP devices are visible");
  if (G > 16) throw std::runtime_error("pdlp_mi355x: at most 16 devices");
  unsigned char id[128];
  {
    std::random_device rd;
    for (unsigned char& b : id) b = (unsigned char)rd();
  }
  std::vector<std::string> errors((size_t)G);
  std::vector<pdlp_result_t> scratch((size_t)G);
  // PDLP_MI355X_VERIFY_RANKS=1 (tests): every rank returns its full solution, compared bit for bit below
  const char* vr = pdlp::devEnv("PDLP_MI355X_VERIFY_RANKS");
  const bool verify = vr && atoi(vr) != 0;
  std::vector<std::vector<double>> keep;
  if (verify) keep.resize((size_t)G * 4);
  std::vector<std::thread> workers;
  for (int g = 0; g < G; ++g) {
    workers.emplace_back([&, g] {
      try {
        pdlp_params_t o = opt;
        o.device = fold ? (opt.device + g) % nDev : opt.device + g;
        o.num_devices = 1;
        if (hipSetDevice(o.device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
        std::unique_ptr<pdlp::SolverBase> s;
        if (o.algorithm == 1) s.reset(new pdlp::HalpernSolver(P, o, g, G, id));
        else if (o.algorithm == 0) s.reset(new pdlp::Solver(P, o, g, G, id));
        else throw std::runtime_error("unknown algorithm (0 = cuPDLP-C path, 1 = HiPDLP path)");
        memset(&scratch[g], 0, sizeof(pdlp_result_t));
        if (verify && g > 0) {
          keep[4 * g + 0].assign((size_t)P.num_col, 0.0); keep[4 * g + 1].assign((size_t)P.num_col, 0.0);
          keep[4 * g + 2].assign((size_t)P.num_row + 1, 0.0); keep[4 * g + 3].assign((size_t)P.num_row + 1, 0.0);
          scratch[g].col_value = keep[4 * g + 0].data(); scratch[g].col_dual = keep[4 * g + 1].data();
          scratch[g].row_value = keep[4 * g + 2].data(); scratch[g].row_dual = keep[4 * g + 3].data();
        }
        s->run(g == 0 ? R : &scratch[g]);
        if (g == 0) scratch[0] = *R;
      } catch (const std::exception& e) {
        errors[g] = e.what()[0] ? e.what() : "error";
      } catch (...) {
        errors[g] = "unknown exception";
      }
    });
  }
  for (std::thread& t : workers) t.join();
  for (int g = 0; g < G; ++g)
    if (!errors[g].empty()) throw std::runtime_error("device rank " + std::to_string(g) + ": " + errors[g]);
  // every rank takes every decision from rank-ordered sums, so they must all report the same bits
  for (int g = 1; g < G; ++g) {
    const pdlp_result_t& a = scratch[0];
    const pdlp_result_t& b = scratch[g];
    bool same = a.term_code == b.term_code && a.num_iter == b.num_iter && a.num_trials == b.num_trials &&
                memcmp(&a.primal_obj, &b.primal_obj, sizeof(double)) == 0 && memcmp(&a.dual_obj, &b.dual_obj, sizeof(double)) == 0;
    if (same && verify && R->col_value && R->row_dual)
      same = memcmp(R->col_value, b.col_value, sizeof(double) * (size_t)P.num_col) == 0 &&
             memcmp(R->row_dual, b.row_dual, sizeof(double) * (size_t)P.num_row) == 0;
    if (!same) throw std::runtime_error("device ranks 0 and " + std::to_string(g) + " disagree (exchange inconsistent)");
  }
}
}  // namespace

int pdlp_mi355x_solve(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_result_t* R) {
  int G = opt ? opt->num_devices : 0;
  if (G <= 0) {
    const char* e = getenv("PDLP_MI355X_DEVICES");
    G = e ? atoi(e) : 1;
  }
  if (G > 1) {
    return guarded([&] {
      if (!P || !opt || !R) throw std::runtime_error("null argument");
      solveSharded(*P, *opt, G, R);
    });
  }
  pdlp_mi355x_solver_t* s = nullptr;
  int rc = pdlp_mi355x_create(P, opt, &s);
  if (rc == 0) rc = pdlp_mi355x_run(s, R);
  pdlp_mi355x_destroy(s);
  return rc;
}

int pdlp_mi355x_dims(const pdlp_mi355x_solver_t* s, int32_t* n, int32_t* m, int64_t* nnz, int32_t* nEqs) {
  return guarded([&] {
    if (!s || !s->impl) throw std::runtime_error("null solver");
    s->impl->dims(n, m, nnz, nEqs);
  });
}

int pdlp_mi355x_reset(pdlp_mi355x_solver_t* s) {
  return guarded([&] {
    if (!s || !s->impl) throw std::runtime_error("null solver");
    s->impl->reset();
  });
}

int pdlp_mi355x_iterate(pdlp_mi355x_solver_t* s, int32_t n_iters, pdlp_iter_stats_t* st) {
  return guarded([&] {
    if (!s || !s->impl) throw std::runtime_error("null solver");
    s->impl->iterate(n_iters, st);
  });
}

int pdlp_mi355x_get_vector(pdlp_mi355x_solver_t* s, const char* name, double* host, int64_t len) {
  return guarded([&] {
    if (!s || !s->impl || !name || !host) throw std::runtime_error("null argument");
    s->impl->getVector(name, host, len);
  });
}

int pdlp_mi355x_set_vector(pdlp_mi355x_solver_t* s, const char* name, const double* host, int64_t len) {
  return guarded([&] {
    if (!s || !s->impl || !name || !host) throw std::runtime_error("null argument");
    s->impl->setVector(name, host, len);
  });
}

int pdlp_mi355x_stage(pdlp_mi355x_solver_t* s, const char* stage, double* scalars_out, int32_t n_scalars) {
  return guarded([&] {
    if (!s || !s->impl || !stage) throw std::runtime_error("null argument");
    s->impl->stage(stage, scalars_out, n_scalars);
  });
}

int pdlp_mi355x_time_kernel(pdlp_mi355x_solver_t* s, const char* kernel, int32_t reps, double* avg_ms) {
  return guarded([&] {
    if (!s || !s->impl || !kernel || !avg_ms) throw std::runtime_error("null argument");
    *avg_ms = s->impl->timeKernel(kernel, reps);
  });
}

int pdlp_mi355x_host_prepare(const pdlp_problem_t* P, const pdlp_params_t* opt, pdlp_prepared_t* out) {
  return guarded([&] {
    if (!P || !opt || !out) throw std::runtime_error("null argument");
    memset(out, 0, sizeof(*out));
    pdlp::StandardForm F;
    if (opt->algorithm == 1) {  // HiPDLP form: rhs = row lower bounds (row upper bounds are not exported)
      pdlp::formulateHipdlp(*P, F);
      if (!(opt->features_off & PDLP_FEATURE_SCALING_OFF))
        pdlp::scaleHipdlp(F, opt->scaling_mode & 1, opt->scaling_mode & 4, opt->scaling_mode & 2, opt->ruiz_iterations);
    } else {
      pdlp::formulate(*P, F);
      if (!(opt->features_off & PDLP_FEATURE_SCALING_OFF)) pdlp::scale(F);
    }
    pdlp::finalize(F);
    out->n = F.n; out->m = F.m; out->n_eqs = F.nEqs; out->n_orig = F.n0; out->nnz = F.nnz;
    out->csr_beg = dupVec(F.csr.beg); out->csr_idx = dupVec(F.csr.idx); out->csr_val = dupVec(F.csr.val);
    out->csc_beg = dupVec(F.cscSorted.beg); out->csc_idx = dupVec(F.cscSorted.idx); out->csc_val = dupVec(F.cscSorted.val);
    out->cost = dupVec(F.cost); out->rhs = dupVec(F.rhs); out->lower = dupVec(F.lower); out->upper = dupVec(F.upper);
    out->col_scale = dupVec(F.colScale); out->row_scale = dupVec(F.rowScale);
    out->row_kind = dupVec(F.rowKind); out->row_new_idx = dupVec(F.rowNewIdx);
    out->norm_cost = F.normCost; out->norm_rhs = F.normRhs; out->mat_norm_inf = F.matNormInf;
    out->spmv_blocks_ax = pdlp::planStream(F.csr.beg, F.m, pdlp::spmvChunkFor(F.nnz), pdlp::kMaxMajorsPerBlock).nBlocks;
    out->spmv_blocks_aty = pdlp::planStream(F.cscSorted.beg, F.n, pdlp::spmvChunkFor(F.nnz), pdlp::kMaxMajorsPerBlock).nBlocks;
  });
}

void pdlp_mi355x_free_prepared(pdlp_prepared_t* o) {
  if (!o) return;
  free(o->csr_beg); free(o->csr_idx); free(o->csr_val); free(o->csc_beg); free(o->csc_idx); free(o->csc_val);
  free(o->cost); free(o->rhs); free(o->lower); free(o->upper); free(o->col_scale); free(o->row_scale);
  free(o->row_kind); free(o->row_new_idx);
  memset(o, 0, sizeof(*o));
}

int pdlp_mi355x_row_partition(const pdlp_prepared_t* prep, int32_t world, int32_t* offsets) {
  return guarded([&] {
    if (!prep || !offsets || world < 1) throw std::runtime_error("bad argument");
    pdlp::Compressed csr;
    csr.beg.assign(prep->csr_beg, prep->csr_beg + prep->m + 1);
    std::vector<int32_t> off = pdlp::rowPartition(csr, prep->m, world);
    for (int32_t g = 0; g <= world; ++g) offsets[g] = off[g];
  });
}

int pdlp_mi355x_host_slab_layout(const pdlp_prepared_t* prep, int32_t which, int32_t long_limit,
                                 pdlp_slab_layout_t* out) {
  return guarded([&] {
    if (!prep || !out) throw std::runtime_error("null argument");
    memset(out, 0, sizeof(*out));
    const int32_t nMajor = which ? prep->n : prep->m, nMinor = which ? prep->m : prep->n;
    pdlp::Compressed c;
    const int32_t* beg = which ? prep->csc_beg : prep->csr_beg;
    c.beg.assign(beg, beg + nMajor + 1);
    c.idx.assign(which ? prep->csc_idx : prep->csr_idx, (which ? prep->csc_idx : prep->csr_idx) + prep->nnz);
    c.val.assign(which ? prep->csc_val : prep->csr_val, (which ? prep->csc_val : prep->csr_val) + prep->nnz);
    pdlp::SlabLayout L;
    pdlp::buildSlabLayout(c, nMajor, nMinor, long_limit, pdlp::kSlabWidthLog2, which ? pdlp::kSlabMajorCostCols : pdlp::kSlabMajorCostRows, L);
    out->rows_per_block = L.rowsPerBlock; out->rows_per_wave = 0; out->n_blocks = L.nBlocks;
    out->minor_bits = L.minorBits; out->slab_width_log2 = L.slabWidthLog2;
    out->n_long = (int32_t)L.longMap.size(); out->nnz_short = (int64_t)L.ent.size();
    out->wave_ptr = dupVec(L.wavePtr); out->ent = dupVec(L.ent); out->val = dupVec(L.val);
    out->long_mask = dupVec(L.longMask); out->long_map = dupVec(L.longMap); out->wave_beg = dupVec(L.waveBeg);
  });
}

void pdlp_mi355x_free_slab_layout(pdlp_slab_layout_t* o) {
  if (!o) return;
  free(o->wave_ptr); free(o->ent); free(o->val); free(o->long_mask); free(o->long_map); free(o->wave_beg);
  memset(o, 0, sizeof(*o));
}

int pdlp_mi355x_host_task_plan(const pdlp_prepared_t* prep, int32_t which, int32_t long_limit, int32_t balance,
                               pdlp_task_plan_t* out) {
  return guarded([&] {
    if (!prep || !out) throw std::runtime_error("null argument");
    memset(out, 0, sizeof(*out));
    const int32_t nMajor = which ? prep->n : prep->m, nMinor = which ? prep->m : prep->n;
    pdlp::Compressed c;
    const int32_t* beg = which ? prep->csc_beg : prep->csr_beg;
    c.beg.assign(beg, beg + nMajor + 1);
    c.idx.assign(which ? prep->csc_idx : prep->csr_idx, (which ? prep->csc_idx : prep->csr_idx) + prep->nnz);
    c.val.assign(which ? prep->csc_val : prep->csr_val, (which ? prep->csc_val : prep->csr_val) + prep->nnz);
    const int32_t majorCost = which ? pdlp::kSlabMajorCostCols : pdlp::kSlabMajorCostRows;
    // the same calls, in the same order, as DeviceMatrix::upload (pdlp_solver.cpp)
    std::vector<int32_t> cold((size_t)std::max(nMajor, 1));
    pdlp::slabColdCounts(c.beg.data(), c.idx.data(), nMajor, nMinor, long_limit, cold.data());
    const pdlp::SlabPartition part = pdlp::slabPartition(c.beg.data(), cold.data(), nMajor, nMinor, long_limit, majorCost);
    std::vector<int32_t> lo, hi, cnt;
    const int32_t tileLog2 = pdlp::xcdTileLog2(nMinor), nTiles = pdlp::xcdTileCount(nMinor, tileLog2);
    const std::vector<int32_t> hist = pdlp::slabTileHistogram(c.beg.data(), c.idx.data(), part, long_limit, tileLog2, nTiles, lo, hi, cnt);
    const std::vector<int8_t> owner = pdlp::xcdTileOwners(hist, nTiles);
    pdlp::SlabLayout L;
    pdlp::buildSlabLayout(c, nMajor, nMinor, long_limit, pdlp::kSlabWidthLog2, majorCost, L);
    int32_t taskGroup = 0;
    const int32_t nLong = (int32_t)L.longMap.size();
    const pdlp::LongPlan P = pdlp::planSlabTasks(L.longCsr.beg, L.longCsr.idx.data(), nLong, L.longMap.data(), balance != 0, &owner, tileLog2,
                                                 L.nBlocks, taskGroup);
    out->n_tasks = P.nTasks; out->task_group = taskGroup; out->n_seg_slots = P.nSegSlots; out->n_long = nLong;
    out->n_blocks = L.nBlocks; out->tile_log2 = tileLog2; out->n_tiles = nTiles;
    std::vector<int32_t> flat((size_t)8 * P.nTasks);
    if (P.nTasks > 0) memcpy(flat.data(), P.tasks.data(), flat.size() * sizeof(int32_t));
    out->tasks = dupVec(flat); out->tile_owner = dupVec(owner); out->long_beg = dupVec(L.longCsr.beg); out->long_idx = dupVec(L.longCsr.idx);
  });
}

void pdlp_mi355x_free_task_plan(pdlp_task_plan_t* o) {
  if (!o) return;
  free(o->tasks); free(o->tile_owner); free(o->long_beg); free(o->long_idx);
  memset(o, 0, sizeof(*o));
}

void pdlp_mi355x_det_exp_log(int32_t n, const double* x, double* exp_out, double* log_out) {
  for (int32_t i = 0; i < n; ++i) { exp_out[i] = pdlp_det_exp(x[i]); log_out[i] = pdlp_det_log(x[i]); }
}

int64_t pdlp_mi355x_sizeof(int32_t which) {
  switch (which) {
    case 0: return sizeof(pdlp_problem_t);
    case 1: return sizeof(pdlp_params_t);
    case 2: return sizeof(pdlp_result_t);
    case 3: return sizeof(pdlp_iter_stats_t);
    case 4: return sizeof(pdlp_prepared_t);
    case 5: return sizeof(pdlp_slab_layout_t);
    case 6: return sizeof(pdlp_mps_model_t);
    case 7: return sizeof(pdlp_task_plan_t);
    default: return -1;
  }
}

// MPS ingest (pdlp_mps.cpp).  The arrays of *out are malloc'ed copies owned by the caller's struct.
int pdlp_mi355x_read_mps(const char* path, int32_t num_threads, pdlp_mps_model_t* out) {
  return pdlp_mi355x_read_mps_timed(path, num_threads, 0.0, out);
}

int pdlp_mi355x_read_mps_timed(const char* path, int32_t num_threads, double time_limit, pdlp_mps_model_t* out) {
  int status = 1;
  const int rc = guarded([&] {
    if (!path || !out) throw std::runtime_error("read_mps: NULL argument");
    memset(out, 0, sizeof(*out));
    pdlp::mps::Model M;
    status = pdlp::mps::readMps(path, num_threads, M, time_limit);
    if (status != pdlp::mps::kReadOk) {
      g_lastError = M.error;
      return;
    }
    auto dupStr = [](const std::string& s) {
      char* p = (char*)malloc(s.size() + 1);
      memcpy(p, s.data(), s.size());
      p[s.size()] = 0;
      return p;
    };
    pdlp_problem_t& P = out->lp;
    P.num_col = M.numCol;
    P.num_row = M.numRow;
    P.num_nz = (int64_t)M.aIndex.size();
    P.a_start = dupVec(M.aStart);
    P.a_index = dupVec(M.aIndex);
    P.a_value = dupVec(M.aValue);
    P.col_cost = dupVec(M.colCost);
    P.col_lower = dupVec(M.colLower);
    P.col_upper = dupVec(M.colUpper);
    P.row_lower = dupVec(M.rowLower);
    P.row_upper = dupVec(M.rowUpper);
    P.offset = M.offset;
    P.sense = M.sense;
    if (M.qDim > 0) {
      std::vector<int32_t> qs, qi;
      std::vector<double> qv;
      pdlp::mps::lowerTriangle(M, qs, qi, qv);
      P.q_dim = M.qDim;
      P.q_start = dupVec(qs);
      P.q_index = dupVec(qi);
      P.q_value = dupVec(qv);
      out->hessian_dim = M.qDim;
      out->hessian_start = dupVec(M.qStart);
      out->hessian_index = dupVec(M.qIndex);
      out->hessian_value = dupVec(M.qValue);
    }
    out->cost_row_location = M.costRowLocation;
    if (!M.integrality.empty()) {
      out->num_integrality = M.numCol;
      out->integrality = dupVec(M.integrality);
    }
    out->model_name = dupStr(M.modelName);
    out->objective_name = dupStr(M.objectiveName);
    if (!M.colNameStart.empty()) {
      char* pool = (char*)malloc(M.colNamePool.size() + 1);
      memcpy(pool, M.colNamePool.data(), M.colNamePool.size());
      out->col_name_pool = pool;
      out->col_name_start = dupVec(M.colNameStart);
    }
    if (!M.rowNameStart.empty()) {
      char* pool = (char*)malloc(M.rowNamePool.size() + 1);
      memcpy(pool, M.rowNamePool.data(), M.rowNamePool.size());
      out->row_name_pool = pool;
      out->row_name_start = dupVec(M.rowNameStart);
    }
    out->num_warnings = M.numWarnings;
    out->warning_issued = M.warningIssued ? 1 : 0;
    out->warnings = dupStr(M.warnings);
    out->threads = M.threads;
    out->file_bytes = M.fileBytes;
    out->seconds = M.seconds;
  });
  return rc ? 1 : status;
}

void pdlp_mi355x_free_mps_model(pdlp_mps_model_t* out) {
  if (!out) return;
  pdlp_problem_t& P = out->lp;
  const void* owned[] = {P.a_start, P.a_index, P.a_value, P.col_cost, P.col_lower, P.col_upper, P.row_lower, P.row_upper,
                         P.q_start, P.q_index, P.q_value, out->integrality, out->model_name, out->objective_name,
                         out->col_name_pool, out->col_name_start, out->row_name_pool, out->row_name_start,
                         out->hessian_start, out->hessian_index, out->hessian_value, out->warnings};
  for (const void* p : owned) free((void*)p);
  memset(out, 0, sizeof(*out));
}

int pdlp_mi355x_comm_unique_id(void* id128) {
  return guarded([&] {
    if (!id128) throw std::runtime_error("null argument");
    pdlp::Comm::uniqueId(id128);
  });
}

}  // extern "C"
