// pdlp_halpern.hpp — device-resident driver of the reference's second PDLP path, solver="hipdlp"
// (highs/pdlp/HiPdlpWrapper.cpp, hipdlp/pdhg.cc): restarted Halpern PDHG with reflection, fixed
// step sizes from a power method, PID-controlled primal weight.  One Halpern step is TWO launches —
// the A'y SpMV with the primal projection/reflection/blend as its epilogue, and the A x SpMV with
// the dual ones — and a block of 40 steps replays from a hipGraph; the host only acts at the
// reference's check iterations (every PDHG_CHECK_INTERVAL = 40 steps, pdhg.cc:32).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <string>
#include <vector>

#include "pdlp_mesh.hpp"
#include "pdlp_solver.hpp"

namespace pdlp {

// vector kernels of the check iterations (pdlp_halpern.hip); partial layouts as in pdlp_kernels.hpp
// gate (every launcher below): the kernel runs only while that device word is non-zero (nullptr: always) — the
// device-driven loop queues whole blocks ahead and lets the decision kernel switch them off
void launchHalpernFpeRows(const double* yn, const double* ry, double* dy, int32_t m, double* part, int32_t nBlocks,
                          hipStream_t s, const int32_t* gate = nullptr);
void launchHalpernFpeCols(const double* xn, const double* rx, const double* atd, int32_t n, double* partDx2,
                          double* partCross, int32_t nBlocks, hipStream_t s, const int32_t* gate = nullptr);
constexpr int kHRowStats = 2;  // 0: sum (((ax - rl) [min 0 on inequality rows]) * rowScale)^2   1: sum rl*y
void launchHalpernRowStats(const double* ax, const double* y, const double* rl, const double* rowScale,
                           const uint8_t* isEq, int32_t m, int scaled, double* part, int32_t stride, int32_t nBlocks,
                           hipStream_t s, const int32_t* gate = nullptr);
constexpr int kHColStats = 4;  // 0: sum ((c - A'y - s+ + s-) * colScale)^2  1: sum c*x  2: sum l*s+  3: sum u*s-
void launchHalpernColStats(const double* aty, const double* x, const double* cost, const double* lower,
                           const double* upper, const double* colScale, const double* cachedSlack, int32_t n,
                           int scaled, double* sp, double* sn, double* part, int32_t stride, int32_t nBlocks,
                           hipStream_t s, const int32_t* gate = nullptr);
// the device-driven loop (pdlp_halpernfn.hpp halpernDecide on the device, and what it gates)
void launchHalpernDecide(HalpernState* st, const double* stat, HalpernRecord* ring, hipStream_t s);
void launchHalpernRestartCopy(const HalpernState* st, double* xa, double* xc, const double* xn, int32_t n, double* ya, double* yc,
                              const double* yn, int32_t m, hipStream_t s);
void launchHalpernKeepOutput(const HalpernState* st, double* outX, const double* xn, int32_t n, double* outY, const double* yn, int32_t m,
                             hipStream_t s);
void launchDivScalar(double* v, double denom, int32_t len, hipStream_t s);  // v[i] /= denom

class HalpernSolver : public SolverBase {
 public:
  // world > 1: row-block shards over the direct xGMI mesh exchange (pdlp_mesh.hpp); id128 as for Solver
  HalpernSolver(const pdlp_problem_t& P, const pdlp_params_t& opt, int32_t rank = 0, int32_t world = 1,
                const void* id128 = nullptr);
  ~HalpernSolver() override;
  void run(pdlp_result_t* R) override;
  void iterate(int32_t nIters, pdlp_iter_stats_t* st) override;
  void reset() override;
  void dims(int32_t* n, int32_t* m, int64_t* nnz, int32_t* nEqs) const override;
  void getVector(const std::string& name, double* host, int64_t len) override;
  void setVector(const std::string& name, const double* host, int64_t len) override;
  void stage(const std::string& name, double* out, int32_t cap) override;
  double timeKernel(const std::string& name, int32_t reps) override;

 private:
  void construct(const pdlp_problem_t& P, const void* id128);
  HalpernVecs stepVecs(bool major, int32_t kOff) const;
  void sumOverRanks(double* devBuf, int32_t count);
  void gatherToHost(const double* devLocal, int32_t lo, int32_t hi, bool byRows, std::vector<double>& full);
  void spmvAt(const double* yLocal, double* atySliceInFull, const int32_t* gate = nullptr);  // A'y: full vector, or own column slice when sharded
  void release() noexcept;
  struct Res { double pObj = 0, dObj = 0, gap = 0, relGap = 0, pFeas = 0, dFeas = 0; };
  double powerMethod();
  void initStepSizes();
  void pushState();                            // the host's copy of the state record -> device
  void pullState();                            // ... and back (device-driven loop)
  void armState();                             // host copy: the loop may run (gate words follow the state)
  void enqueueStep(bool major, int32_t kOff);
  void enqueueMinorSteps();                    // steps 2..40
  void runBlock(bool fpeAfterFirst);           // steps 1..40 of one block (host-driven loop)
  void enqueueUnit();                          // block + check + decision + gated copies (device-driven loop)
  // gate (device word, nullptr = always): see pdlp_halpern.hpp launchHalpern*
  void enqueueFpe(int slot, const int32_t* gate);   // computeFixedPointError's three sums -> statOut_[slot..]
  void enqueueCheck(double* x, const double* y, bool cachedSlack, const int32_t* gate);  // A x, A'y + the sums of checkConvergence
  void enqueueDiff(const int32_t* gate);       // the two restart distances -> statOut_[kHSlotDiff..]
  void fetchStats(int count);                  // all queued statistics: one all-reduce, one download, one sync
  void restartCopies();
  void noteRecord(const HalpernRecord& r);     // a check's line: residuals for the result, log
  void doSolve(bool terminate, int64_t iterTarget);
  void postsolve(pdlp_result_t* R);
  double sum(const double* partials, int32_t nBlocks);
  double elapsed() const;
  void log(int level, const char* fmt, ...) const;
  std::pair<double*, int64_t> lookup(const std::string& name);

  pdlp_params_t opt_;
  StandardForm F_;
  // the caller's LP (postprocess computes row activities and the objective from it, pdhg.cc:409-468)
  std::vector<int32_t> origBeg_, origIdx_;
  std::vector<double> origVal_, origCost_;
  // sharding (rows [r0_, r1_) and column slice [c0_, c1_) are owned; everything when world == 1)
  int32_t rank_ = 0, world_ = 1, r0_ = 0, r1_ = 0, mLoc_ = 0, c0_ = 0, c1_ = 0, nLoc_ = 0;
  bool sharded_ = false;
  Mesh* mesh_ = nullptr;
  DeviceArray<double> commBuf_;  // partial A_g' y (n)
  hipStream_t stream_ = nullptr;
  DeviceMatrix dA_, dAt_;
  DeviceArray<double> xc_, yc_, xn_, yn_, rx_, ry_, xa_, ya_, slack_, sp_, sn_, outX_, outY_;
  DeviceArray<double> cost_, lower_, upper_, rl_, ru_, colScale_, rowScale_, tmpN_, tmpM_, tmpM2_, gatherBuf_;
  DeviceArray<uint8_t> isEq_;
  DeviceArray<double> part_, statOut_;
  DeviceArray<HalpernState> dState_;
  HalpernState* hostState_ = nullptr;  // pinned
  double* hostStats_ = nullptr;        // pinned
  int32_t stride_ = 0;
  // the scalars of PDLPSolver (pdhg.hpp) live in the state record (pdlp_kernels.hpp HalpernState: hostState_ is the host's
  // copy, dState_ the device's); mirrors of its counters for the result
  double lambda_ = 0;
  bool pid_ = true;
  bool devLoop_ = false;  // check iterations, restarts and the PID weight on the device (one GPU)
  int32_t nRestarts_ = 0, nChecks_ = 0;
  int64_t iters_ = 0;
  HalpernRecord* hostRing_ = nullptr;  // pinned: the checks' lines, written by the decision kernel
  hipGraphExec_t unitGraph_ = nullptr; // one unit of the device-driven loop
  int termStatus_ = -1;  // -1 not set, 0 optimal, 1 iteration limit, 2 time limit
  bool haveOutput_ = false;
  Res res_;
  hipGraphExec_t graphExec_ = nullptr;
  bool useGraph_ = true;
  // in-loop kernel timing (stage "profile_on"): HIP events around the two launches of every step
  bool profile_ = false;
  std::vector<hipEvent_t> profEvents_;
  int32_t profQueued_ = 0;
  double profAxMs_ = 0, profAtyMs_ = 0;
  int64_t profLaunches_ = 0;
  void profCollect();
  std::chrono::steady_clock::time_point solveBeg_;
  double setupSeconds_ = 0, solveSeconds_ = 0;
};

}  // namespace pdlp
