// pdlp_solver.hpp — device-resident PDHG driver (the MI355X counterpart of
// cuPDLP-C's LP_SolvePDHG / PDHG_Solve, cupdlp_solver.c:899-1498).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "pdlp_device.hpp"
#include "pdlp_host.hpp"
#include "pdlp_kernels.hpp"
#include "pdlp_mesh.hpp"
#include "pdlp_setup.hpp"

namespace pdlp {

// Environment switches of a solver, read once at creation (pdlp_solver.cpp DevSwitches::fromEnv); -1 = not set.
struct DevSwitches {
  int graph = -1, forceComm = 0, gpuSetup = -1;
  int slab = -1;          // PDLP_MI355X_SLAB: 0 CSR stream only, 1 slab layout, -1 automatic by the gathered vector's size
  int slabW = 0;          // PDLP_MI355X_SLAB_W: log2 of the slab width (development)
  int xcdMap = -1, slabPace = -1;
  int slabTune = 1;       // PDLP_MI355X_SLAB_TUNE=0 (development): the slab width by rule only, no timing of narrower slabs
  int affineTasks = 1;  // XCD-affine deal of the slab layout's segment tasks (0: (major, segment) order; A/B measurements)
  int fusedCoTasks = -1;  // PDLP_MI355X_FUSED_COTASKS: 0 = the fused trial's streaming blocks run the long columns' task passes themselves
  int touchTail = 1;      // PDLP_MI355X_TOUCH_TAIL=0 (development): no touching of the tail columns' operands in front of the fused trial's barrier
  int uniformBounds = 1;  // PDLP_MI355X_UNIFORM_BOUNDS=0 (development): the fused trial loads l and u of every column even where a block's columns share them
  int constCached = -1;   // PDLP_MI355X_CONST_CACHED=0|1 (development): c, l, u of the primal step non-temporal / ordinary loads (default: by size)
  int fused = -1, fusedStream = 0, persistent = -1, xcdLocal = -1, hierBarrier = -1, deviceCheck = -1, checkSmall = -1;
  int primalInA = -1;     // PDLP_MI355X_PRIMAL_IN_A: the persistent loop without its P phase (pdlp_small.hip PINA); -1 = where measured faster
  int barrierTimeoutMs = 1000;  // PDLP_MI355X_BARRIER_TIMEOUT_MS: how long a grid barrier / roll call waits for missing workgroups
  int fault = 0;          // PDLP_MI355X_FAULT (tests): 1 = the first persistent launch expects one workgroup too many,
                          // 2 = the 12th fused trial's barrier expects one block too many (both then time out and fall back)
  std::string exchange, meshLayout;
  static DevSwitches fromEnv();
};

// One operand matrix in HBM: CSR stream plan or slab layout for the majors that are summed left to right,
// segment tasks for the long ones (pdlp_host.hpp LongPlan).
struct DeviceMatrix {
  DeviceArray<int32_t> beg, idx, blockBeg, wavePtr, waveBeg;
  DeviceArray<uint32_t> ent, longMask;
  DeviceArray<double> val, slabVal;
  // long majors
  DeviceArray<LongTask> lTasks;
  DeviceArray<double> lSegSum, lContrib;
  DeviceArray<uint32_t> lTicket;
  int32_t nLong = 0, nTasks = 0, longSlots = 0, longGroup = 1, taskGroup = 4;
  // slab layout: size the task workgroups so that every CU gets one (uploadPlans).  Off for the operand whose tasks the
  // fused trial runs inside its streaming blocks (already spread evenly; full groups of 16 keep long columns in LDS)
  bool balanceTaskBlocks = true;
  bool affineTasks = true;  // slab layout: XCD-affine deal of the segment tasks (PDLP_MI355X_DEV_AFFINE_TASKS=0: (major, segment) order)
  int32_t majorCost = kSlabMajorCostRows;  // slab partition: work of a major besides its entries (the owner sets kSlabMajorCostCols on its transposed operand)
  int32_t fusedCoTasks = 0;  // MatView::coTaskBlocks (the fused trial's task workgroups), decided by the solver at set-up
  int32_t touchTail = 1;     // MatView::touchTail
  int32_t nMajor = 0, nBlocks = 0;  // nBlocks = CSR stream blocks
  int32_t chunk = kChunk;           // work-plan block size of the CSR stream (spmvChunkFor)
  int64_t nnz = 0;
  bool useSlab = false;
  int32_t noPace = 0;  // slab kernel without the per-group block barrier (SlabMat::noPace), see tuneXcdMap
  int32_t slabWidthLog2 = 0;  // log2 of the slab width the layout was built with (device build; see buildSlabTuned)
  double estRunLen = 1.0;     // estimated run length of equal majors at the widest slabs (device build)
  bool mapTuned = false;      // XCD map / pacing already chosen by timing (with the slab width)
  int32_t xcdMap = 1;  // block -> XCD assignment of the SpMV kernels (pdlp_kernels.hip xcdContiguousBlock), see tuneXcdMap
  SlabMat slab{};
  // sw.slab: 0 = CSR stream only, 1 = slab layout, -1 = auto by nMinor
  void upload(const Compressed& c, int32_t nMajor_, int32_t nMinor_, const DevSwitches& sw, hipStream_t s);
  // same, from a matrix that is already in HBM (GPU-side setup); takes M's arrays
  void buildFromDevice(DeviceCsrData& M, const DevSwitches& sw, hipStream_t s);
  MatView view() const;
  int32_t nPartials() const { return (useSlab ? slab.nBlocks : 0) + nBlocks + longSlots; }

 private:
  // hostLongIdx / tileOwner / tileLog2 (slab layout): the minors of the long majors on the host and the XCD that owns each
  // tile of the gathered vector — the segment tasks are dealt to workgroups of the XCD their entries live in (planLong)
  void uploadPlans(const std::vector<int32_t>& hostBeg, int32_t nCsrMajor, const int32_t* longVecIndex, hipStream_t s,
                   const int32_t* hostLongIdx = nullptr, const std::vector<int8_t>* tileOwner = nullptr, int32_t tileLog2 = 0);
  // per-block column span / entry count of the short majors -> do the blocks touch few stretches of the gathered vector densely?
  static bool touchesFewTiles(const std::vector<int32_t>& lo, const std::vector<int32_t>& hi, const std::vector<int32_t>& cnt);
};

// Picks M.xcdMap by timing the plain SpMV out = M * in with both block -> XCD assignments (a few launches; the
// result vector is scratch).  PDLP_MI355X_XCD_MAP=0|1 forces one.
// Returns the time of one plain SpMV with the choice made (ms; 0 for operands too small to time).
float tuneXcdMap(DeviceMatrix& M, const DevSwitches& sw, const double* in, double* out, hipStream_t s);
// M -> `out` like DeviceMatrix::buildFromDevice, plus (slab layout, >= 2^20 nonzeros) the slab width chosen by timing
// the plain SpMV at the rule's width, 2^13 and 2^11 (pdlp_solver.cpp buildSlabTuned holds the measurements behind it).
void buildSlabTuned(DeviceMatrix& out, DeviceCsrData& M, const DevSwitches& sw, hipStream_t s);

class Comm;  // RCCL wrapper (pdlp_comm.cpp)

// Host copy of cuPDLP's CUPDLPresobj numbers for one iterate.
struct Residuals {
  double pObj = 0, dObj = 0, gap = 0, relGap = 0, pFeas = 0, dFeas = 0;
  double pInfObj = 0, pInfRes = 1, dInfObj = 0, dInfRes = 1;
};

// What the C ABI drives: one implementation per reference path (cuPDLP-C: Solver below;
// HiPDLP: HalpernSolver, pdlp_halpern.hpp).
class SolverBase {
 public:
  virtual ~SolverBase() = default;
  virtual void run(pdlp_result_t* R) = 0;
  virtual void iterate(int32_t nIters, pdlp_iter_stats_t* st) = 0;
  virtual void reset() = 0;
  virtual void dims(int32_t* n, int32_t* m, int64_t* nnz, int32_t* nEqs) const = 0;
  virtual void getVector(const std::string& name, double* host, int64_t len) = 0;
  virtual void setVector(const std::string& name, const double* host, int64_t len) = 0;
  virtual void stage(const std::string& name, double* out, int32_t cap) = 0;
  virtual double timeKernel(const std::string& name, int32_t reps) = 0;
};

class Solver : public SolverBase {
 public:
  Solver(const pdlp_problem_t& P, const pdlp_params_t& opt, int32_t rank, int32_t world, const void* id128);
  ~Solver() override;

  void run(pdlp_result_t* R) override;                          // LP_SolvePDHG
  void iterate(int32_t nIters, pdlp_iter_stats_t* st) override;  // fixed-work loop for timing
  void reset() override;                                        // PDHG_Init_Step_Sizes + PDHG_Init_Variables

  void dims(int32_t* n, int32_t* m, int64_t* nnz, int32_t* nEqs) const override;
  void getVector(const std::string& name, double* host, int64_t len) override;
  void setVector(const std::string& name, const double* host, int64_t len) override;
  void stage(const std::string& name, double* out, int32_t cap) override;
  double timeKernel(const std::string& name, int32_t reps) override;

 private:
  // setup
  void construct(const pdlp_problem_t& P, const void* id128);
  void release() noexcept;
  void uploadProblem();
  void uploadProblemFromDevice(DeviceProblem& D);
  void uploadShardFromDevice(DeviceProblem& D);  // sharded, two-all-gathers layout: the rank's shard cut on the device
  static void downloadForm(DeviceProblem& D, StandardForm& F, hipStream_t s);
  void allocIterates();
  void initStepSizes();
  void initVariables();
  void applyHotStart();
  // hot loop
  void enqueueTrial();
  void enqueueBatch(int32_t todo);  // trials up to the next halt: one persistent launch / the captured graph / single trials
  void captureGraph();
  void runUntilHalt();
  void syncState();    // device -> host_
  void pushState(bool wait = true);    // host_ -> device
  // check iteration
  void computeAverage();
  void computeResiduals();
  bool checkTermination(const Residuals& r) const;
  bool checkInfeasibility();
  void restartIterate();
  int32_t nextCheckIter(int32_t it) const;
  void doSolve(bool terminate, int32_t iterBudget);        // check iterations driven by the host (sharded paths, profile mode)
  void doSolveDevice(bool terminate, int32_t iterBudget);  // check iterations on the device, several periods queued ahead
  void enqueueCheckDevice();
  void uploadCtl(bool terminate, int64_t iterLim);
  void downloadCtl();
  void processRecords(bool terminate, int64_t iterLim, int& logSinceHeader);
  void logCheckLine(int32_t it, const Residuals& cur, const Residuals& avg, double t, int& logSinceHeader) const;
  void postsolve(pdlp_result_t* R);
  // linear algebra on device (sharding-aware)
  void deviceAx(const double* x, double* axLocal);
  void deviceATy(const double* yLocal, double* aty);
  double reduceScalar(const double* partials, int32_t nBlocks, bool overRanks);
  void sumOverRanks(double* devBuf, int32_t count);  // RCCL all-reduce or mesh rank-ordered sum
  void gatherToHost(const double* devLocal, int32_t lo, int32_t hi, bool byRows, std::vector<double>& full);
  std::pair<double*, int64_t> lookup(const std::string& name);
  void log(int level, const char* fmt, ...) const;
  double elapsed() const;
  bool timeIsUp();  // elapsed() > time_limit, agreed across ranks when sharded (identical control flow)

  pdlp_params_t opt_;
  DevSwitches sw_;
  StandardForm F_;
  bool hasStart_ = false;
  std::vector<double> startX_, startY_;
  // sharding
  int32_t rank_ = 0, world_ = 1, r0_ = 0, r1_ = 0, mLoc_ = 0;
  Comm* comm_ = nullptr;
  bool gpuSetup_ = true;   // formulate/scale/transpose/slab layout on the device (pdlp_setup.hip)
  double sumCost2_ = 0, sumRhs2_ = 0;  // left-to-right sums of the scaled c, b
  bool sharded_ = false;  // row-block sharded kernel sequence (world > 1, or forced for testing)
  bool shardOnDevice_ = false;  // sharded + device-side set-up: the rank's shard is cut on the device (uploadShardFromDevice)
  // exchange of the sharded path: direct xGMI mesh (pdlp_mesh.hpp; columns are then sliced as
  // well, [c0_, c1_)) or, as the fallback, RCCL all-reduce with replicated column work
  Mesh* mesh_ = nullptr;
  bool meshMode_ = false;
  // Mesh layouts (PDLP_MI355X_MESH_LAYOUT): "colblock" (default) = row block for A x, column block A[:, c0:c1) for
  // A'y, two all-gathers per trial (x+ slices, y+ row blocks), no n-length partial; "partial" = the round-1 layout
  // (transpose of the row block, all-gather of x+ and reduce-scatter of the A_g'y partials).  With colblock every
  // rank keeps y, yAvg (and the power method's work vector) at FULL length m; yOff_ = r0_ is where its rows sit.
  bool colblock_ = false;
  int32_t yOff_ = 0, yLen_ = 0;
  double* yl(int k) const { return y_[k].get() + yOff_; }
  double* yAvgl() const { return yAvg_.get() + yOff_; }
  double* tmpMl() const { return tmpM_.get() + yOff_; }
  int32_t c0_ = 0, c1_ = 0, nLoc_ = 0;
  IterVecs vecsCol_{};  // vecs_ restricted to the own column slice
  IterVecs vecsAty_{};  // colblock layout: vecsCol_ with the FULL-length y (what the column-block A'y kernel gathers from)
  // device
  hipStream_t stream_ = nullptr;
  DeviceMatrix dA_, dAt_;
  // QP whose Hessian has off-diagonal entries: N = the off-diagonal part (symmetric, scaled with the
  // columns), N x by parity like A'y, N xAvg at checks, the partials of dx . N dx
  DeviceMatrix dQ_;
  bool hasQoff_ = false;
  DeviceArray<double> nx_[2], nxAvg_, partQ_;
  DeviceArray<double> x_[2], y_[2], ax_[2], aty_[2];
  DeviceArray<double> xAvg_, yAvg_, axAvg_, atyAvg_, xSum_, ySum_, xLast_, yLast_;
  DeviceArray<double> cost_, rhs_, lower_, upper_, colScale_, rowScale_, qdiag_;  // qdiag_: QP only
  DeviceArray<double> slackPos_, slackNeg_, slackPosAvg_, slackNegAvg_;
  DeviceArray<double> partDY_, partDX_, partInter_, statPart_, statOut_, commBuf_, tmpM_, gatherBuf_;
  // Two slots: the single-GPU loop alternates between them from trial to trial (k_decide_primal reads
  // one, writes the other); stPar_ = the slot that holds the state after everything enqueued so far.
  DeviceArray<DevState> dState_;
  int32_t stPar_ = 0, graphPar_ = 0;
  // Fused trial (2 launches, pdlp_kernels.hpp launchSpmvAtyFusedPrimal): the A'y kernel also takes the decision and
  // does the next primal step.  needPrimal_: the host has pushed a state since the last trial, so the next trial
  // starts with a stand-alone primal step.
  bool fused_ = false, needPrimal_ = true;
  // Small LPs: a batch of trials is ONE persistent launch (pdlp_small.hip); smallGrid_ = its workgroups
  bool persistent_ = false, xcdLocal_ = false, hierBar_ = false, primalInA_ = false;
  int smallMode() const { return xcdLocal_ ? 1 : hierBar_ ? 2 : 0; }
  int32_t smallGrid_ = 0;
  DeviceArray<unsigned long long> gridBar_;
  DeviceArray<int32_t> colBlockUni_;     // IterVecs::colBlockUni / colBlockBounds (fused slab launch)
  DeviceArray<double> colBlockBounds_;
  int64_t uniLowerCols_ = 0, uniUpperCols_ = 0;  // columns covered by a block-wide lower / upper bound (stage "uniform_bound_columns")
  int32_t barrierFallbacks_ = 0, smallLaunches_ = 0;
  unsigned long long smallSeq_ = 0;  // persistent launches since gridBar_ was zeroed (their roll call counts cumulatively)
  // (barrier rounds of the contexts of one device: ordered on the DEVICE by an event chain, see pdlp_solver.cpp)
  struct DeviceGate { std::mutex mu; hipEvent_t ev[2] = {nullptr, nullptr}; int cur = 0; bool recorded = false; };
  static DeviceGate& deviceGate(int device);
  std::unique_lock<std::mutex> beginBarrierRound();          // locked (and the stream ordered behind the last round) iff this solver launches grid barriers
  void endBarrierRound(std::unique_lock<std::mutex>& gate);  // marks the end of this round on the stream, releases the gate
  struct BarrierRound {  // a round in scope: ended (event recorded, gate released) on every way out, also by an exception
    Solver& self;
    std::unique_lock<std::mutex> gate;
    ~BarrierRound();
  };
  int32_t stalledRounds_ = 0, stalledSince_ = 0;  // consecutive device stops without an accepted trial
  // Device-driven check iterations (pdlp_kernels.hpp CheckCtl; PDLP_MI355X_DEVICE_CHECK=0 gives the host-driven loop back)
  bool devCheck_ = true;
  DeviceArray<CheckCtl> dCtl_;
  CheckCtl* hostCtl_ = nullptr;      // pinned staging copy
  CheckRecord* hostRing_ = nullptr;  // pinned: one record per queued check, written by the device
  static constexpr int32_t kRingSlots = 64;
  int64_t checkSeq_ = 0, checkSeen_ = 0;  // checks enqueued / records looked at
  DeviceArray<double> partRestartY_;
  // Small LPs: the check as ONE launch (pdlp_check.hip k_check_small); its barrier words and launch counter
  bool checkSmall_ = false;
  DeviceArray<unsigned long long> checkBar_;
  unsigned long long checkSmallSeq_ = 0;
  DevState* dst() const { return dState_.get() + stPar_; }
  DeviceArray<double> powRed_, powGrow_;  // host-tabulated powers of the trial counter (see DevState)
  void refreshPowTable();
  DevState* hostState_ = nullptr;  // pinned mirror
  double* hostStats_ = nullptr;    // pinned
  int32_t statStride_ = 0;
  IterVecs vecs_{};
  // scalar solver state (host side of CUPDLPresobj / CUPDLPiterates)
  Residuals cur_, avg_;
  double pFeasLR_ = 0, dFeasLR_ = 0, gapLR_ = 0, pFeasLC_ = 0, dFeasLC_ = 0, gapLC_ = 0;
  int32_t iLastRestartIter_ = 0, nRestarts_ = 0, nChecks_ = 0;
  int32_t termCode_ = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT, termIterate_ = 0;
  bool adaptive_ = true, restartOn_ = true;
  double feasTol_ = 1e-8;
  std::chrono::steady_clock::time_point solveBeg_;
  double setupSeconds_ = 0, solveSeconds_ = 0;
  // optional hipGraph of a batch of trials
  hipGraphExec_t graphExec_ = nullptr;
  int32_t graphTrials_ = 0;
  bool useGraph_ = true;
  // in-loop kernel timing (stage "profile_on"): HIP events around the two SpMV launches of every trial
  bool profile_ = false;
  std::vector<hipEvent_t> profEvents_;
  int32_t profTrialsQueued_ = 0;
  double profAxMs_ = 0, profAtyMs_ = 0;
  int64_t profLaunches_ = 0;
  void profCollect(int32_t realTrials);
};

// RCCL communicator, loaded lazily with dlopen so that the library itself has
// no link-time dependency on librccl.
class Comm {
 public:
  static void uniqueId(void* id128);
  Comm(int32_t rank, int32_t world, const void* id128);
  ~Comm();
  void allReduceSum(double* buf, size_t count, hipStream_t s);
  int32_t rank() const { return rank_; }
  int32_t world() const { return world_; }

 private:
  void* comm_ = nullptr;
  int32_t rank_, world_;
};

}  // namespace pdlp
