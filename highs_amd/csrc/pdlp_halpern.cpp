// pdlp_halpern.cpp — see pdlp_halpern.hpp.  Host control flow of PDLPSolver::solve
// (hipdlp/pdhg.cc:494-707); every vector lives in HBM.
#include "pdlp_halpern.hpp"
#include "pdlp_halpernfn.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace pdlp {

namespace {
constexpr int kCheckInterval = 40;  // PDHG_CHECK_INTERVAL, pdhg.cc:32
constexpr int kStatSlots = 8;       // rows of the partial-sum table
// statOut_/hostStats_ slots of one block's ONE download (doSolve): [fpe 3 | check 6 | fpe after the block's first step 3]
constexpr int kSlotFpe = kHSlotFpe, kSlotCheck = kHSlotCheck, kSlotFpe0 = kHSlotFpe0, kSlotDiff = kHSlotDiff, kStatOut = kHStatOut;
static_assert(kSlotFpe0 == kSlotCheck + kHRowStats + kHColStats && kSlotDiff == kSlotFpe0 + 3 && kSlotDiff + 2 <= kStatOut, "stat slots");
}  // namespace

double HalpernSolver::elapsed() const {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - solveBeg_).count();
}

void HalpernSolver::log(int level, const char* fmt, ...) const {
  if (opt_.log_level < level) return;
  va_list ap;
  va_start(ap, fmt);
  logLineV(opt_, level, fmt, ap);
  va_end(ap);
}

HalpernSolver::HalpernSolver(const pdlp_problem_t& P, const pdlp_params_t& opt, int32_t rank, int32_t world,
                             const void* id128)
    : opt_(opt), rank_(rank), world_(world) {
  try {
    construct(P, id128);
  } catch (...) {
    release();  // a throwing constructor never runs the destructor
    throw;
  }
}

void HalpernSolver::construct(const pdlp_problem_t& P, const void* id128) {
  const auto t0 = std::chrono::steady_clock::now();
  validateProblem(P);
  requireConstraints(P);
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0)
    throw std::runtime_error("pdlp_mi355x: no HIP device available (this library has no CPU fallback)");
  PDLP_HIP(hipSetDevice(opt_.device));
  PDLP_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  const DevSwitches sw = DevSwitches::fromEnv();  // the environment, once per solver
  if (sw.graph >= 0) useGraph_ = sw.graph != 0;
  if (world_ < 1 || rank_ < 0 || rank_ >= world_) throw std::runtime_error("bad rank/world");
  sharded_ = world_ > 1 || sw.forceComm != 0;
  // check iterations, restarts and the PID weight on the device (one GPU): whole blocks are queued ahead, the host reads
  // the checks' lines from a pinned ring.  PDLP_MI355X_DEVICE_CHECK=0 (development) / a sharded solve: the host decides.
  devLoop_ = !sharded_ && sw.deviceCheck != 0;
  if (rank_ == 0) log(1, "Solving with HiPDLP (restarted Halpern PDHG) on MI355X (gfx950, HIP)\n");
  if ((opt_.features_off & PDLP_FEATURE_RESTART_OFF) != 0)
    log(1, "HiPDLP uses Halpern restart only; ignoring the restart-off feature flag.\n");  // pdhg.cc:1846-1852
  pid_ = opt_.step_size_strategy != 0;  // 0 fixed; everything else runs as PID (pdhg.cc:1856-1864)

  if (P.num_col > 0 && P.a_start) {
    origBeg_.assign(P.a_start, P.a_start + P.num_col + 1);
    origIdx_.assign(P.a_index, P.a_index + origBeg_[P.num_col]);
    origVal_.assign(P.a_value, P.a_value + origBeg_[P.num_col]);
    origCost_.assign(P.col_cost, P.col_cost + P.num_col);
  }
  // quadratic objectives belong to the pdlp path (algorithm 0): refused here, before either set-up path is chosen
  // (the device-side set-up never looks at q_*)
  if (P.q_dim > 0 && P.q_start && P.q_value)
    for (int32_t p = 0; p < P.q_start[P.q_dim]; ++p)
      if (P.q_value[p] != 0.0)
        throw std::runtime_error("pdlp_mi355x: quadratic objectives are solved by the pdlp path only (algorithm = 0)");
  const bool doScale = !(opt_.features_off & PDLP_FEATURE_SCALING_OFF);
  // preprocessing + scaling + both orientations (+ slab layouts) on the device for big LPs, on the host
  // for small ones: same bits either way (tests)
  const int64_t nnzIn = P.num_col > 0 && P.a_start ? (int64_t)P.a_start[P.num_col] : 0;
  bool gpuSetup = nnzIn >= 200000;
  if (sw.gpuSetup >= 0) gpuSetup = sw.gpuSetup != 0;
  if (sharded_) gpuSetup = false;  // the row-block shards are cut on the host
  if (gpuSetup) {
    HipdlpSetup hs;
    hs.ruiz = opt_.scaling_mode & 1; hs.pc = opt_.scaling_mode & 4; hs.l2 = opt_.scaling_mode & 2;
    hs.ruizIters = opt_.ruiz_iterations;
    DeviceProblem D;
    gpuPrepare(P, doScale, stream_, D, &hs);
    F_ = StandardForm();
    F_.n = D.n; F_.m = D.m; F_.n0 = D.n0; F_.nEqs = D.nEqs; F_.nnz = D.nnz;
    F_.scaled = D.scaled; F_.offset = D.offset; F_.sense = D.sense;
    F_.normCost = D.normCost; F_.normRhs = D.normRhs;
    F_.rowKind = std::move(D.rowKind); F_.rowNewIdx = std::move(D.rowNewIdx);
    F_.colScale = std::move(D.hColScale); F_.rowScale = std::move(D.hRowScale);
    dAt_.majorCost = kSlabMajorCostCols;
    buildSlabTuned(dA_, D.A, sw, stream_);
    buildSlabTuned(dAt_, D.At, sw, stream_);
    cost_ = std::move(D.cost); lower_ = std::move(D.lower); upper_ = std::move(D.upper); rl_ = std::move(D.rhs);
    ru_ = std::move(D.rowUpper); colScale_ = std::move(D.colScale); rowScale_ = std::move(D.rowScale);
    isEq_ = std::move(D.rowIsEq);
  } else {
    formulateHipdlp(P, F_);
    if (doScale) scaleHipdlp(F_, opt_.scaling_mode & 1, opt_.scaling_mode & 4, opt_.scaling_mode & 2, opt_.ruiz_iterations);
    finalize(F_);  // rows ascending column; columns are already ascending row
    r0_ = 0; r1_ = F_.m;
    if (sharded_) {
      std::vector<int32_t> off = rowPartition(F_.csr, F_.m, world_);
      r0_ = off[rank_]; r1_ = off[rank_ + 1];
      mesh_ = new Mesh(rank_, world_, id128, F_.n, F_.m, off, stream_);
      const bool ok = mesh_->selfTest(stream_);
      if (!mesh_->allAgree(ok))
        throw std::runtime_error("pdlp_mi355x: the sharded HiPDLP path needs the direct xGMI exchange, whose self-test failed");
      c0_ = mesh_->c0(); c1_ = mesh_->c1();
      Compressed csrSlab, cscSlab;
      extractSlab(F_, r0_, r1_, csrSlab, cscSlab);
      dAt_.majorCost = kSlabMajorCostCols;
      dA_.upload(csrSlab, r1_ - r0_, F_.n, sw, stream_);
      dAt_.upload(cscSlab, F_.n, r1_ - r0_, sw, stream_);
      commBuf_.alloc((size_t)F_.n + 8);
      commBuf_.zero(stream_);
    } else {
      dAt_.majorCost = kSlabMajorCostCols;
      dA_.upload(F_.csr, F_.m, F_.n, sw, stream_);
      dAt_.upload(F_.cscSorted, F_.n, F_.m, sw, stream_);
    }
    auto up = [&](DeviceArray<double>& d, const double* h, size_t count) {
      d.alloc(count);
      d.upload(h, count, stream_);
    };
    const size_t mL = (size_t)(r1_ - r0_);
    up(cost_, F_.cost.data(), F_.n); up(lower_, F_.lower.data(), F_.n); up(upper_, F_.upper.data(), F_.n);
    up(colScale_, F_.colScale.data(), F_.n);
    up(rl_, F_.rhs.data() + r0_, mL); up(ru_, F_.rowUpper.data() + r0_, mL); up(rowScale_, F_.rowScale.data() + r0_, mL);
    isEq_.alloc(mL);
    isEq_.upload(F_.rowIsEq.data() + r0_, mL, stream_);
  }
  if (!sharded_) { r0_ = 0; r1_ = F_.m; }
  mLoc_ = r1_ - r0_;
  if (!mesh_) { c0_ = 0; c1_ = F_.n; }
  nLoc_ = c1_ - c0_;
  const int32_t n = F_.n, m = mLoc_;
  for (DeviceArray<double>* d : {&xc_, &xn_, &rx_, &xa_, &slack_, &sp_, &sn_, &outX_, &tmpN_}) { d->alloc(n); d->zero(stream_); }
  for (DeviceArray<double>* d : {&yc_, &yn_, &ry_, &ya_, &outY_, &tmpM_, &tmpM2_}) { d->alloc(m); d->zero(stream_); }
  stride_ = std::max(vecBlocks(std::max(n, 1)), vecBlocks(std::max(m, 1)));
  (void)m;
  part_.alloc((size_t)kStatSlots * stride_);
  statOut_.alloc(kStatOut);
  statOut_.zero(stream_);
  dState_.alloc(1);
  PDLP_HIP(hipHostMalloc((void**)&hostState_, sizeof(HalpernState), hipHostMallocDefault));
  PDLP_HIP(hipHostMalloc((void**)&hostStats_, sizeof(double) * kStatOut, hipHostMallocDefault));
  PDLP_HIP(hipHostMalloc((void**)&hostRing_, sizeof(HalpernRecord) * kHalpernRing, hipHostMallocDefault));
  memset(hostState_, 0, sizeof(HalpernState));
  memset(hostRing_, 0, sizeof(HalpernRecord) * kHalpernRing);
  PDLP_HIP(hipStreamSynchronize(stream_));
  F_.csc = Compressed(); F_.csr = Compressed(); F_.cscSorted = Compressed();
  // block -> XCD assignment of the two operands (scratch vectors: any input will do)
  tuneXcdMap(dA_, sw, tmpN_.get(), tmpM_.get(), stream_);
  tuneXcdMap(dAt_, sw, tmpM_.get(), tmpN_.get(), stream_);
  reset();
  setupSeconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HalpernSolver::release() noexcept {
  if (graphExec_) (void)hipGraphExecDestroy(graphExec_);
  if (unitGraph_) (void)hipGraphExecDestroy(unitGraph_);
  if (hostRing_) (void)hipHostFree(hostRing_);
  unitGraph_ = nullptr; hostRing_ = nullptr;
  for (hipEvent_t e : profEvents_) (void)hipEventDestroy(e);
  profEvents_.clear();
  if (hostState_) (void)hipHostFree(hostState_);
  if (hostStats_) (void)hipHostFree(hostStats_);
  delete mesh_;
  if (stream_) (void)hipStreamDestroy(stream_);
  graphExec_ = nullptr; hostState_ = nullptr; hostStats_ = nullptr; stream_ = nullptr; mesh_ = nullptr;
}

HalpernSolver::~HalpernSolver() { release(); }

void HalpernSolver::dims(int32_t* n, int32_t* m, int64_t* nnz, int32_t* nEqs) const {
  if (n) *n = F_.n;
  if (m) *m = F_.m;
  if (nnz) *nnz = F_.nnz;
  if (nEqs) *nEqs = F_.nEqs;
}

// Deterministic sum of per-block partials, brought to the host (and over the ranks when sharded).
double HalpernSolver::sum(const double* partials, int32_t nBlocks) {
  launchFinalReduce(partials, nBlocks, nBlocks, 1, statOut_.get(), stream_);
  if (sharded_) sumOverRanks(statOut_.get(), 1);
  PDLP_HIP(hipMemcpyAsync(hostStats_, statOut_.get(), sizeof(double), hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  return hostStats_[0];
}

void HalpernSolver::sumOverRanks(double* devBuf, int32_t count) {
  if (mesh_) mesh_->allReduceScalars(devBuf, count, stream_);
}

// A' y for row-local y: the full vector on one GPU; when sharded, the rank-ordered sum of the ranks'
// partials on the OWN column slice of `aty` (a full-length buffer)
void HalpernSolver::spmvAt(const double* yLocal, double* aty, const int32_t* gate) {
  if (!sharded_) {
    CheckGate g;
    g.flag = gate;
    launchSpmvPlain(dAt_.view(), yLocal, aty, stream_, g);
  } else {
    launchSpmvPlain(dAt_.view(), yLocal, commBuf_.get(), stream_);
    mesh_->reduceScatterCols(commBuf_.get(), aty, stream_);
  }
}

void HalpernSolver::gatherToHost(const double* devLocal, int32_t lo, int32_t hi, bool byRows, std::vector<double>& full) {
  const int32_t len = byRows ? F_.m : F_.n;
  full.assign((size_t)len, 0.0);
  // persistent scratch: a hipFree here would synchronise the whole device, and with several ranks of one
  // process on one device (the folded test mode) it would wait for a peer's kernel that waits for us
  DeviceArray<double>& g = gatherBuf_;
  if (g.size() < (size_t)std::max(len, 1)) g.alloc((size_t)std::max(std::max(F_.n, F_.m), 1));
  g.zero(stream_);
  PDLP_HIP(hipMemcpyAsync(g.get() + lo, devLocal, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToDevice, stream_));
  if (mesh_) mesh_->allGather(g.get(), byRows, stream_);
  g.download(full.data(), (size_t)len, stream_);
  PDLP_HIP(hipStreamSynchronize(stream_));
  if (mesh_) mesh_->checkError(stream_);  // a timed-out exchange must not return a half-gathered vector
}

// powerMethod, pdhg.cc:1529-1670 (kCuPdlpAATPowerMethod): 20 iterations on A A' from the ones vector
double HalpernSolver::powerMethod() {
  const int32_t n = F_.n, m = mLoc_;
  if (n == 0 || F_.m == 0) return 1.0;
  const int32_t nbM = vecBlocks(std::max(m, 1)), nbN = vecBlocks(std::max(nLoc_, 1));
  double* x = tmpM_.get();   // x_vec / z_vec (rows, local)
  double* z = tmpM2_.get();
  double* y = tmpN_.get();   // y_vec (columns)
  launchFill(x, 1.0, m, stream_);
  double lambda = 0.0;
  for (int it = 0; it < 20; ++it) {
    spmvAt(x, y);                                                 // y = A' x
    if (sharded_) mesh_->allGather(y, false, stream_);            // A y needs all of y
    launchSpmvPlain(dA_.view(), y, z, stream_);                   // z = A y
    launchDot(z, z, m, part_.get(), nbM, stream_);
    const double zn = std::sqrt(sum(part_.get(), nbM));
    launchDivScalar(z, zn, m, stream_);
    spmvAt(z, y);                                                 // w = A' q  (own slice when sharded)
    launchDot(y + c0_, y + c0_, nLoc_, part_.get(), nbN, stream_);
    lambda = sum(part_.get(), nbN);
    std::swap(x, z);  // x_vec = z_vec
  }
  return lambda;
}

// initializeStepSizes, pdhg.cc:1944-1977
void HalpernSolver::initStepSizes() {
  HalpernState& H = *hostState_;
  H.omega = (F_.normCost + 1.0) / (F_.normRhs + 1.0);
  H.primalWeight = H.omega;
  H.bestPrimalWeight = H.primalWeight;
  lambda_ = powerMethod();
  const double base = 0.998 / std::sqrt(lambda_);
  H.eta = base;
  H.tau = base / H.omega;
  H.sigma = base * H.omega;
  H.rho = 1.0;  // halpern_gamma, pdhg.cc:1913
  log(2, "Initial step sizes from power method lambda = %g: primal step = %g; dual step = %g, eta = %g, omega = %g\n",
      lambda_, H.tau, H.sigma, H.eta, H.omega);
}

void HalpernSolver::pushState() {
  PDLP_HIP(hipMemcpyAsync(dState_.get(), hostState_, sizeof(HalpernState), hipMemcpyHostToDevice, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
}
void HalpernSolver::pullState() {
  PDLP_HIP(hipMemcpyAsync(hostState_, dState_.get(), sizeof(HalpernState), hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  PDLP_HIP(hipGetLastError());  // a failed kernel launch since the last stop surfaces here
}
// the loop may run: the gate words of the block's kernels follow the state (stage / timing entry points, start of a solve)
void HalpernSolver::armState() {
  HalpernState& H = *hostState_;
  H.halted = 0; H.run = 1; H.runFpe0 = H.fpe0Pending; H.doRestart = 0;
}

// initializeStepSizes + initialize + the start of solve() (pdhg.cc:499-553)
void HalpernSolver::reset() {
  const int32_t n = F_.n, m = mLoc_;
  HalpernState& H = *hostState_;
  memset(&H, 0, sizeof(H));
  initStepSizes();
  H.bestGap = std::numeric_limits<double>::infinity();
  H.lastTrialFpe = std::numeric_limits<double>::infinity();
  H.normRhs = F_.normRhs; H.normCost = F_.normCost; H.offset = F_.offset; H.tol = opt_.gap_tol;  // params_.tolerance
  H.pid = pid_ ? 1 : 0;
  H.termStatus = -1;
  H.iterLimit = 0;
  armState();
  for (DeviceArray<double>* d : {&xc_, &xn_, &rx_, &xa_, &slack_, &sp_, &sn_, &outX_}) d->zero(stream_);
  for (DeviceArray<double>* d : {&yc_, &yn_, &ry_, &ya_, &outY_}) d->zero(stream_);
  launchProjectBounds(xc_.get(), lower_.get(), upper_.get(), n, stream_);  // linalg::projectBounds of x = 0
  PDLP_HIP(hipMemcpyAsync(xa_.get(), xc_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
  PDLP_HIP(hipMemcpyAsync(ya_.get(), yc_.get(), sizeof(double) * m, hipMemcpyDeviceToDevice, stream_));
  iters_ = 0;
  nRestarts_ = nChecks_ = 0;
  termStatus_ = -1;
  haveOutput_ = false;
  res_ = Res();
  pushState();
}

HalpernVecs HalpernSolver::stepVecs(bool major, int32_t kOff) const {
  HalpernVecs h{};
  h.xc = xc_.get(); h.yc = yc_.get(); h.xn = xn_.get(); h.yn = yn_.get(); h.rx = rx_.get(); h.ry = ry_.get();
  h.xa = xa_.get(); h.ya = ya_.get(); h.slack = slack_.get();
  h.cost = cost_.get(); h.lower = lower_.get(); h.upper = upper_.get(); h.rowLower = rl_.get(); h.rowUpper = ru_.get();
  h.hs = dState_.get();
  h.kOff = kOff;
  h.major = major ? 1 : 0;
  return h;
}

// performHalpernPdhgStep, pdhg.cc:961-1018, as two fused SpMV launches (one GPU), or the
// partial-A'y / exchange / slice-primal / exchange / local-dual sequence of pdlp_mesh.hpp (sharded)
void HalpernSolver::enqueueStep(bool major, int32_t kOff) {
  const HalpernVecs h = stepVecs(major, kOff);
  if (!sharded_ && profile_) {  // eager launches bracketed by events (4th event: cost of an empty event pair)
    while ((int32_t)profEvents_.size() < 4 * (profQueued_ + 1)) {
      hipEvent_t e;
      PDLP_HIP(hipEventCreate(&e));
      profEvents_.push_back(e);
    }
    hipEvent_t* ev = &profEvents_[4 * profQueued_++];
    PDLP_HIP(hipEventRecord(ev[0], stream_));
    launchHalpernPrimal(dAt_.view(), h, stream_);
    PDLP_HIP(hipEventRecord(ev[1], stream_));
    launchHalpernDual(dA_.view(), h, stream_);
    PDLP_HIP(hipEventRecord(ev[2], stream_));
    PDLP_HIP(hipEventRecord(ev[3], stream_));
    return;
  }
  if (!sharded_) {
    launchHalpernPrimal(dAt_.view(), h, stream_);
    launchHalpernDual(dA_.view(), h, stream_);
    return;
  }
  HalpernVecs hc = h;  // column vectors offset to the own slice
  hc.xc += c0_; hc.xn += c0_; hc.rx += c0_; hc.xa += c0_; hc.slack += c0_;
  hc.cost += c0_; hc.lower += c0_; hc.upper += c0_;
  launchMeshHalpernStep(dA_.view(), dAt_.view(), h, hc, F_.n, nLoc_, commBuf_.get(), mesh_->args(), stream_);
}

void HalpernSolver::profCollect() {
  PDLP_HIP(hipStreamSynchronize(stream_));
  for (int32_t t = 0; t < profQueued_; ++t) {
    float a = 0.f, b = 0.f, z = 0.f;
    PDLP_HIP(hipEventElapsedTime(&a, profEvents_[4 * t], profEvents_[4 * t + 1]));
    PDLP_HIP(hipEventElapsedTime(&b, profEvents_[4 * t + 1], profEvents_[4 * t + 2]));
    PDLP_HIP(hipEventElapsedTime(&z, profEvents_[4 * t + 2], profEvents_[4 * t + 3]));
    profAtyMs_ += a - z;
    profAxMs_ += b - z;
    ++profLaunches_;
  }
  profQueued_ = 0;
}

// The steps 2..40 of a block (the hipGraph of the host-driven loop; part of the unit graph of the device-driven one)
void HalpernSolver::enqueueMinorSteps() {
  for (int i = 2; i <= kCheckInterval - 1; ++i) enqueueStep(false, i);
  enqueueStep(true, kCheckInterval);
}

// One block of the main loop (pdhg.cc:578-641) in the HOST-driven loop: major step 1, [fixed-point error if a restart
// just happened], minor steps 2..39, major step 40.  The state on the device is the host's (pushed by the caller).
void HalpernSolver::runBlock(bool fpeAfterFirst) {
  enqueueStep(true, 1);
  if (fpeAfterFirst) enqueueFpe(kSlotFpe0, nullptr);  // read with the block's other statistics (fetchStats)
  if (profile_ && !sharded_) {
    enqueueMinorSteps();
    profCollect();
  } else if (useGraph_) {
    if (!graphExec_) {
      hipGraph_t graph = nullptr;
      PDLP_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      enqueueMinorSteps();
      PDLP_HIP(hipStreamEndCapture(stream_, &graph));
      PDLP_HIP(hipGraphInstantiate(&graphExec_, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
    }
    PDLP_HIP(hipGraphLaunch(graphExec_, stream_));
  } else {
    enqueueMinorSteps();
  }
}

// computeFixedPointError, pdhg.cc:709-739: the three sums into statOut_[slot..slot+2]
void HalpernSolver::enqueueFpe(int slot, const int32_t* gate) {
  const int32_t m = mLoc_;
  const int32_t nbM = vecBlocks(std::max(m, 1)), nbN = vecBlocks(std::max(nLoc_, 1));
  double* part = part_.get();
  launchHalpernFpeRows(yn_.get(), ry_.get(), tmpM_.get(), m, part, nbM, stream_, gate);
  spmvAt(tmpM_.get(), tmpN_.get(), gate);
  launchHalpernFpeCols(xn_.get() + c0_, rx_.get() + c0_, tmpN_.get() + c0_, nLoc_, part + stride_,
                       part + 2 * (size_t)stride_, nbN, stream_, gate);
  launchFinalReduce(part, stride_, nbM, 1, statOut_.get() + slot, stream_, gate);
  launchFinalReduce(part + stride_, stride_, nbN, 2, statOut_.get() + slot + 1, stream_, gate);
}

// The first `count` statistics slots: summed over the ranks, brought to the host, ONE stream synchronisation.
void HalpernSolver::fetchStats(int count) {
  if (sharded_) sumOverRanks(statOut_.get(), count);
  PDLP_HIP(hipMemcpyAsync(hostStats_, statOut_.get(), sizeof(double) * count, hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  PDLP_HIP(hipGetLastError());  // a failed kernel launch since the last check surfaces here
  if (mesh_) mesh_->checkError(stream_);
}

// runConvergenceCheck's "current" leg (pdhg.cc:820-833): A x, A'y and the six sums into statOut_[kSlotCheck..].
// x: full-length buffer whose own column slice is valid (all of it on one GPU); y: local rows.
void HalpernSolver::enqueueCheck(double* x, const double* y, bool cachedSlack, const int32_t* gate) {
  const int32_t m = mLoc_;
  const int32_t nbM = vecBlocks(std::max(m, 1)), nbN = vecBlocks(std::max(nLoc_, 1));
  const int sc = F_.scaled ? 1 : 0;
  const size_t co = (size_t)c0_;
  double* part = part_.get();
  double* out = statOut_.get() + kSlotCheck;
  CheckGate g;
  g.flag = gate;
  if (sharded_) mesh_->allGather(x, false, stream_);  // A x needs every column slice
  launchSpmvPlain(dA_.view(), x, tmpM_.get(), stream_, g);
  spmvAt(y, tmpN_.get(), gate);
  launchHalpernRowStats(tmpM_.get(), y, rl_.get(), rowScale_.get(), isEq_.get(), m, sc, part, stride_, nbM, stream_, gate);
  launchHalpernColStats(tmpN_.get() + co, x + co, cost_.get() + co, lower_.get() + co, upper_.get() + co,
                        colScale_.get() + co, cachedSlack ? slack_.get() + co : nullptr, nLoc_, sc, sp_.get() + co,
                        sn_.get() + co, part + (size_t)kHRowStats * stride_, stride_, nbN, stream_, gate);
  launchFinalReduce(part, stride_, nbM, kHRowStats, out, stream_, gate);
  launchFinalReduce(part + (size_t)kHRowStats * stride_, stride_, nbN, kHColStats, out + kHRowStats, stream_, gate);
}

// The two distances of updatePrimalWeightAtRestart (pdhg.cc:1979-2049), |x_next - x_anchor|^2 and |y_next - y_anchor|^2,
// into statOut_[kSlotDiff..]: part of every check (two vector passes), so that the decision needs no second round trip.
void HalpernSolver::enqueueDiff(const int32_t* gate) {
  const int32_t m = mLoc_;
  const int32_t nbM = vecBlocks(std::max(m, 1)), nbN = vecBlocks(std::max(nLoc_, 1));
  launchDiffNorm2(xn_.get() + c0_, xa_.get() + c0_, nLoc_, part_.get(), nbN, stream_, gate);
  launchDiffNorm2(yn_.get(), ya_.get(), m, part_.get() + stride_, nbM, stream_, gate);
  launchFinalReduce(part_.get(), stride_, nbN, 1, statOut_.get() + kSlotDiff, stream_, gate);
  launchFinalReduce(part_.get() + stride_, stride_, nbM, 1, statOut_.get() + kSlotDiff + 1, stream_, gate);
}

// pdhg.cc:663-692: anchor and current iterate <- pdhg iterate of the last major step (the host-driven loop; the primal
// weight has been updated by halpernDecide)
void HalpernSolver::restartCopies() {
  const int32_t n = F_.n, m = mLoc_;
  PDLP_HIP(hipMemcpyAsync(xa_.get(), xn_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
  PDLP_HIP(hipMemcpyAsync(ya_.get(), yn_.get(), sizeof(double) * m, hipMemcpyDeviceToDevice, stream_));
  PDLP_HIP(hipMemcpyAsync(xc_.get(), xn_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
  PDLP_HIP(hipMemcpyAsync(yc_.get(), yn_.get(), sizeof(double) * m, hipMemcpyDeviceToDevice, stream_));
}

void HalpernSolver::noteRecord(const HalpernRecord& r) {
  res_.pObj = r.pObj; res_.dObj = r.dObj; res_.gap = r.gap; res_.relGap = r.relGap; res_.pFeas = r.pFeas; res_.dFeas = r.dFeas;
  if (opt_.log_level > 1 && rank_ == 0)
    logLine(opt_, 2, "%9lld  %+15.8e  %+15.8e  %8.2e  %10.2e  %8.2e  fpe %8.2e  w %8.2e\n", (long long)r.iters, r.pObj, r.dObj,
            r.relGap, r.pFeas / (1.0 + F_.normRhs), r.dFeas / (1.0 + F_.normCost), r.fpe, r.primalWeight);
}

// One unit of the DEVICE-driven loop: a block of 40 steps, the initial fixed-point error behind its first step when a
// restart came before (gated), the check's statistics, the decision kernel and what it switches on (restart copies,
// output copy).  Every kernel is a no-op once the loop has halted; the whole unit replays from ONE hipGraph.
void HalpernSolver::enqueueUnit() {
  HalpernState* ds = dState_.get();
  const int32_t n = F_.n, m = mLoc_;
  enqueueStep(true, 1);
  enqueueFpe(kSlotFpe0, &ds->runFpe0);
  enqueueMinorSteps();
  enqueueFpe(kSlotFpe, &ds->run);
  enqueueCheck(xn_.get(), yn_.get(), true, &ds->run);
  enqueueDiff(&ds->run);
  launchHalpernDecide(ds, statOut_.get(), hostRing_, stream_);
  launchHalpernRestartCopy(ds, xa_.get(), xc_.get(), xn_.get(), n, ya_.get(), yc_.get(), yn_.get(), m, stream_);
  launchHalpernKeepOutput(ds, outX_.get(), xn_.get(), n, outY_.get(), yn_.get(), m, stream_);
}

// PDLPSolver::solve, pdhg.cc:494-707.  terminate = false: fixed-work loop for timing (same
// kernels, checks and restarts; convergence is ignored).
void HalpernSolver::doSolve(bool terminate, int64_t iterTarget) {
  const int32_t n = F_.n, m = mLoc_;
  HalpernState& H = *hostState_;
  auto keepOutput = [&](const double* x, const double* y) {
    PDLP_HIP(hipMemcpyAsync(outX_.get(), x, sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
    PDLP_HIP(hipMemcpyAsync(outY_.get(), y, sizeof(double) * m, hipMemcpyDeviceToDevice, stream_));
    haveOutput_ = true;
  };
  auto mirror = [&]() { iters_ = H.iters; nRestarts_ = H.nRestarts; nChecks_ = H.nChecks; };
  auto timeUp = [&]() {  // every rank must take the same branch: the ranks' clocks are OR-ed
    bool up = elapsed() > opt_.time_limit;
    if (sharded_ && std::isfinite(opt_.time_limit)) {
      hostStats_[0] = up ? 1.0 : 0.0;
      PDLP_HIP(hipMemcpyAsync(statOut_.get(), hostStats_, sizeof(double), hipMemcpyHostToDevice, stream_));
      sumOverRanks(statOut_.get(), 1);
      PDLP_HIP(hipMemcpyAsync(hostStats_, statOut_.get(), sizeof(double), hipMemcpyDeviceToHost, stream_));
      PDLP_HIP(hipStreamSynchronize(stream_));
      up = hostStats_[0] > 0.0;
    }
    return up;
  };
  const int64_t limit = terminate ? (int64_t)opt_.iter_limit : iterTarget;
  H.terminate = terminate ? 1 : 0;
  H.iterLimit = limit;
  H.converged = 0;
  armState();
  if (H.iters == 0 && terminate) {  // initial convergence check, pdhg.cc:563-570
    pushState();
    enqueueCheck(xc_.get(), yc_.get(), false, nullptr);
    fetchStats(kSlotFpe0);
    if (mesh_) mesh_->verifyReplicated(rx_.get(), F_.n, stream_);
    HalpernRecord r{};
    const bool conv = halpernResiduals(H, hostStats_, r);
    H.nChecks += 1;
    res_.pObj = r.pObj; res_.dObj = r.dObj; res_.gap = r.gap; res_.relGap = r.relGap; res_.pFeas = r.pFeas; res_.dFeas = r.dFeas;
    mirror();
    if (conv) {
      keepOutput(xc_.get(), yc_.get());
      termStatus_ = 0;
      return;
    }
  }
  if (H.iters >= limit) { if (terminate) termStatus_ = 1; return; }
  const bool device = devLoop_ && !profile_;
  if (device) {
    // ---- the device decides: whole units are queued ahead, the host reads the checks' lines ----
    pushState();
    if (!unitGraph_ && useGraph_) {
      hipGraph_t graph = nullptr;
      PDLP_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      enqueueUnit();
      PDLP_HIP(hipStreamEndCapture(stream_, &graph));
      PDLP_HIP(hipGraphInstantiate(&unitGraph_, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
    }
    int ahead = 1;
    for (;;) {
      if (terminate && timeUp()) { termStatus_ = 2; break; }
      const int32_t checksBefore = H.nChecks;
      const auto t0 = std::chrono::steady_clock::now();
      const int64_t unitsLeft = (limit - H.iters + kCheckInterval - 1) / kCheckInterval;
      const int units = (int)std::max<int64_t>(1, std::min<int64_t>(ahead, unitsLeft));
      for (int u = 0; u < units; ++u) {
        if (unitGraph_) PDLP_HIP(hipGraphLaunch(unitGraph_, stream_));
        else enqueueUnit();
      }
      pullState();
      for (int32_t c = checksBefore; c < H.nChecks; ++c) noteRecord(hostRing_[c % kHalpernRing]);
      mirror();
      if (H.halted) {
        if (H.converged) { haveOutput_ = true; termStatus_ = 0; }
        else if (terminate) termStatus_ = 1;
        break;
      }
      // queue depth: ~25 ms of work, at most half the ring
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      const int cap = ms > 0.0 ? std::max(1, std::min(kHalpernRing / 2, (int)(25.0 * units / ms))) : 1;
      ahead = std::min(ahead * 2, cap);
    }
    return;
  }
  // ---- the host decides (sharded solves, profile mode, PDLP_MI355X_DEVICE_CHECK=0): same kernels, same decision function ----
  while (H.iters < limit) {
    if (terminate && timeUp()) { termStatus_ = 2; return; }
    // One block, its fixed-point error(s), its convergence statistics and the restart distances are queued back to back
    // and read with ONE download: the host only decides (converged / restart) between blocks.
    const bool fpeAfterFirst = H.fpe0Pending != 0;
    pushState();
    runBlock(fpeAfterFirst);
    enqueueFpe(kSlotFpe, nullptr);
    enqueueCheck(xn_.get(), yn_.get(), true, nullptr);
    enqueueDiff(nullptr);
    fetchStats(kStatOut);
    if (mesh_) mesh_->verifyReplicated(rx_.get(), F_.n, stream_);  // the reflected x is the vector every rank holds in full
    const HalpernRecord r = halpernDecide(H, hostStats_);
    noteRecord(r);
    mirror();
    if (H.converged) {
      keepOutput(xn_.get(), yn_.get());
      termStatus_ = 0;
      return;
    }
    if (H.doRestart) restartCopies();
  }
  if (terminate) termStatus_ = 1;
}

void HalpernSolver::run(pdlp_result_t* R) {
  reset();
  solveBeg_ = std::chrono::steady_clock::now();
  doSolve(true, 0);
  solveSeconds_ = elapsed();
  if (opt_.log_level > 0 && rank_ == 0)
    logLine(opt_, 1, "\nHiPDLP: %s after %lld iterations (%d restarts): primal obj %+.10e, dual obj %+.10e, rel gap %.2e\n",
           termStatus_ == 0 ? "converged" : termStatus_ == 2 ? "time limit" : "iteration limit", (long long)iters_,
           nRestarts_, res_.pObj, res_.dObj, res_.relGap);
  if (R) postsolve(R);
}

void HalpernSolver::iterate(int32_t nIters, pdlp_iter_stats_t* st) {
  const int64_t it0 = iters_;
  const int32_t ck0 = nChecks_, rs0 = nRestarts_;
  profAxMs_ = profAtyMs_ = 0.0;
  profLaunches_ = 0;
  solveBeg_ = std::chrono::steady_clock::now();
  hipEvent_t e0, e1;
  PDLP_HIP(hipEventCreate(&e0));
  PDLP_HIP(hipEventCreate(&e1));
  PDLP_HIP(hipEventRecord(e0, stream_));
  const int64_t blocks = ((int64_t)nIters + kCheckInterval - 1) / kCheckInterval;
  doSolve(false, it0 + blocks * kCheckInterval);
  PDLP_HIP(hipEventRecord(e1, stream_));
  PDLP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PDLP_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (st) {
    memset(st, 0, sizeof(*st));
    st->iters = (int32_t)(iters_ - it0);
    st->trials = st->iters;
    st->checks = nChecks_ - ck0;
    st->restarts = nRestarts_ - rs0;
    st->gpu_ms = ms;
    st->wall_ms = elapsed() * 1e3;
    if (profLaunches_ > 0) {  // in-loop averages per launch (profile mode)
      st->spmv_ax_ms = profAxMs_ / (double)profLaunches_;
      st->spmv_aty_ms = profAtyMs_ / (double)profLaunches_;
      st->reserved[0] = (double)profLaunches_;
    }
  }
}

// unscaleSolution (pdhg.cc:1883-1897, scaling.cc:264-278) + postprocess (pdhg.cc:359-492)
void HalpernSolver::postsolve(pdlp_result_t* R) {
  const int32_t n = F_.n, m = F_.m, n0 = F_.n0;
  std::vector<double> x(n, 0.0), y(m, 0.0), sp(n), sn(n);
  // only a converged check writes the output vectors (pdhg.cc:866-877): otherwise they are the zero start
  if (!sharded_) {
    if (haveOutput_) {
      outX_.download(x.data(), n, stream_);
      outY_.download(y.data(), m, stream_);
    }
    sp_.download(sp.data(), n, stream_);
    sn_.download(sn.data(), n, stream_);
    PDLP_HIP(hipStreamSynchronize(stream_));
  } else {  // x was all-gathered by the converged check; y is row-local, the slacks column-sliced
    if (haveOutput_) {
      outX_.download(x.data(), n, stream_);
      PDLP_HIP(hipStreamSynchronize(stream_));
      gatherToHost(outY_.get(), r0_, r1_, true, y);
    }
    gatherToHost(sp_.get() + c0_, c0_, c1_, false, sp);
    gatherToHost(sn_.get() + c0_, c0_, c1_, false, sn);
  }
  if (F_.scaled) {
    for (int32_t j = 0; j < n; ++j) x[j] /= F_.colScale[j];
    for (int32_t i = 0; i < m; ++i) y[i] /= F_.rowScale[i];
  }
  for (int32_t j = 0; j < n; ++j) { sp[j] *= F_.colScale[j]; sn[j] *= F_.colScale[j]; }
  if (R->col_value) std::copy(x.begin(), x.begin() + n0, R->col_value);
  if (R->row_dual)
    for (int32_t i = 0; i < m; ++i) {
      const double v = y[F_.rowNewIdx[i]];
      R->row_dual[i] = F_.rowKind[i] == kRowLeq ? -v : v;
    }
  if (R->col_dual)
    for (int32_t j = 0; j < n0; ++j) R->col_dual[j] = (sp[j] - sn[j]) * F_.sense;
  if (R->row_value) {  // A_original x over the original columns, column by column (:447-468)
    std::fill(R->row_value, R->row_value + m, 0.0);
    for (int32_t c = 0; c < n0; ++c)
      for (int32_t p = origBeg_[c]; p < origBeg_[c + 1]; ++p) R->row_value[origIdx_[p]] += origVal_[p] * x[c];
  }
  double finalObj = F_.offset;  // results_.primal_obj, :409-414
  for (int32_t c = 0; c < n0; ++c) finalObj += origCost_[c] * x[c];
  res_.pObj = finalObj;
  R->value_valid = 1;
  R->dual_valid = 1;
  R->term_code = termStatus_ == 0 ? PDLP_TERM_OPTIMAL : PDLP_TERM_TIMELIMIT_OR_ITERLIMIT;
  R->reserved_i = termStatus_ == 2 ? 1 : 0;
  R->term_iterate = 0;
  R->num_iter = (int32_t)std::min<int64_t>(iters_, INT_MAX);
  R->num_trials = 0;
  R->num_restarts = nRestarts_;
  R->primal_obj = res_.pObj; R->dual_obj = res_.dObj; R->primal_feas = res_.pFeas; R->dual_feas = res_.dFeas;
  R->rel_gap = res_.relGap; R->norm_rhs = F_.normRhs; R->norm_cost = F_.normCost;
  R->setup_seconds = setupSeconds_;
  R->solve_seconds = solveSeconds_;
}

std::pair<double*, int64_t> HalpernSolver::lookup(const std::string& name) {
  const int64_t n = F_.n, m = mLoc_;  // row vectors are local when sharded
  if (name == "x") return {xc_.get(), n};
  if (name == "y") return {yc_.get(), m};
  if (name == "x_next") return {xn_.get(), n};
  if (name == "y_next") return {yn_.get(), m};
  if (name == "x_reflected") return {rx_.get(), n};
  if (name == "y_reflected") return {ry_.get(), m};
  if (name == "x_anchor") return {xa_.get(), n};
  if (name == "y_anchor") return {ya_.get(), m};
  if (name == "dual_slack") return {slack_.get(), n};
  if (name == "cost") return {cost_.get(), n};
  if (name == "lower") return {lower_.get(), n};
  if (name == "upper") return {upper_.get(), n};
  if (name == "rhs" || name == "row_lower") return {rl_.get(), m};
  if (name == "row_upper") return {ru_.get(), m};
  if (name == "col_scale") return {colScale_.get(), n};
  if (name == "row_scale") return {rowScale_.get(), m};
  throw std::runtime_error("unknown vector name: " + name);
}

void HalpernSolver::getVector(const std::string& name, double* host, int64_t len) {
  if (name == "steps") {  // {tau, sigma, omega, eta, lambda, primal weight, fpe, initial fpe}
    const HalpernState& H = *hostState_;
    const double v[8] = {H.tau, H.sigma, H.omega, H.eta, lambda_, H.primalWeight, H.fpe, H.initialFpe};
    for (int64_t i = 0; i < len && i < 8; ++i) host[i] = v[i];
    return;
  }
  auto [p, l] = lookup(name);
  if (l != len) throw std::runtime_error("length mismatch for vector " + name);
  PDLP_HIP(hipMemcpyAsync(host, p, sizeof(double) * len, hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
}

void HalpernSolver::setVector(const std::string& name, const double* host, int64_t len) {
  if (name == "steps") {  // {tau, sigma}: impose step sizes (parity tests)
    if (len < 2) throw std::runtime_error("steps needs tau, sigma");
    hostState_->tau = host[0];
    hostState_->sigma = host[1];
    pushState();
    return;
  }
  auto [p, l] = lookup(name);
  if (l != len) throw std::runtime_error("length mismatch for vector " + name);
  PDLP_HIP(hipMemcpyAsync(p, host, sizeof(double) * len, hipMemcpyHostToDevice, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
}

void HalpernSolver::stage(const std::string& name, double* out, int32_t cap) {
  auto put = [&](int i, double v) { if (out && i < cap) out[i] = v; };
  if (name == "block") {  // one block of 40 steps from the current state, then the fixed-point error
    armState();
    pushState();
    runBlock(false);
    enqueueFpe(kSlotFpe, nullptr);
    fetchStats(3);
    hostState_->fpe = halpernFpe(*hostState_, hostStats_ + kSlotFpe);
    put(0, hostState_->fpe);
  } else if (name == "steps") {  // out[0] = number of steps (1..40), the last one major: returns nothing
    const int k = out && cap > 0 ? (int)out[0] : kCheckInterval;
    if (k < 1 || k > kCheckInterval) throw std::runtime_error("steps: 1..40");
    armState();
    pushState();
    for (int i = 1; i <= k; ++i) enqueueStep(i == 1 || i == k, i);
  } else if (name == "mesh_phases") {  // {X, P, -} average wait in us, then the wait counts (since the last call)
    double us[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
    if (mesh_) mesh_->phaseStats(us, cnt, stream_);
    for (int k = 0; k < 3; ++k) { put(k, us[k]); put(3 + k, cnt[k]); }
  } else if (name == "exchange") {
    put(0, sharded_ ? 2.0 : 0.0);
  } else if (name == "profile_on" || name == "profile_off") {
    profile_ = name == "profile_on";
  } else {
    throw std::runtime_error("unknown stage: " + name);
  }
  PDLP_HIP(hipStreamSynchronize(stream_));
}

double HalpernSolver::timeKernel(const std::string& name, int32_t reps) {
  if (reps < 1) reps = 1;
  armState();
  pushState();
  const HalpernVecs h = stepVecs(false, 2);
  auto once = [&]() {
    if (name == "spmv_aty" || name == "halpern_primal") launchHalpernPrimal(dAt_.view(), h, stream_);
    else if (name == "spmv_ax" || name == "halpern_dual") launchHalpernDual(dA_.view(), h, stream_);
    else if (name == "trial") enqueueStep(false, 2);
    else throw std::runtime_error("unknown kernel: " + name);
  };
  for (int i = 0; i < 3; ++i) once();
  hipEvent_t e0, e1;
  PDLP_HIP(hipEventCreate(&e0));
  PDLP_HIP(hipEventCreate(&e1));
  PDLP_HIP(hipEventRecord(e0, stream_));
  for (int i = 0; i < reps; ++i) once();
  PDLP_HIP(hipEventRecord(e1, stream_));
  PDLP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PDLP_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (double)ms / reps;
}

}  // namespace pdlp
