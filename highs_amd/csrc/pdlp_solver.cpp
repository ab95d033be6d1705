// pdlp_solver.cpp — see pdlp_solver.hpp.  Host control flow of the restarted,
// adaptive-step PDHG loop; every vector lives in HBM and every per-trial
// decision is taken on the device, so the host only synchronises at the
// reference's check iterations (nIter < 10, nIter % 40 == 0, last iteration).
#include "pdlp_solver.hpp"
#include "pdlp_detmath.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pdlp {

namespace {
constexpr int kStatRowCur = 0, kStatRowAvg = kRowStats, kStatColCur = 2 * kRowStats,
              kStatColAvg = 2 * kRowStats + kColStats, kStatTotal = 2 * kRowStats + 2 * kColStats;
// One hipGraph holds a whole check period of trials plus two spare ones for rejected trials (a trial queued after
// the device has halted is three early-exit kernels): one graph launch per 40 iterations instead of four — the
// gap between two graph launches is ~40 us on this runtime, which was 4 us per iteration on small LPs.
constexpr int kGraphTrials = 42;
constexpr int kGraphMinTodo = 30;  // fewer trials than this to the next halt: launched one by one
constexpr int kCheckInterval = 40;  // CUPDLP_RELEASE_INTERVAL, cupdlp_defs.h:39
}  // namespace

namespace {
// Majors longer than this stay out of the slab layout (their runs would be added
// by a single lane); the CSR kernel's stream / block-per-major paths take them.
constexpr int32_t kSlabLongLimit = 256;
// The slab layout pays off once the gathered vector no longer fits an XCD's L2
// next to the streams: 2^18 doubles = 2 MB.
constexpr int32_t kSlabAutoMinor = 1 << 18;
}  // namespace

namespace {
// slab width (log2 of the minors per slab) of an operand: PDLP_MI355X_SLAB_W if given, 2^14 for operands whose blocks
// touch few stretches of the gathered vector densely, else 2^17 (1 MB of the gathered vector)
int32_t slabWidthFor(const DevSwitches& sw, bool fewTiles) {
  if (sw.slabW > 0) return sw.slabW;
  return fewTiles ? kSlabTileLog2 : kSlabWidthLog2;
}
// mode (PDLP_MI355X_SLAB or auto) -> is the slab layout used for an operand of this shape
bool chooseSlab(int mode, int32_t nMajor, int32_t nMinor) {
  const bool want = mode == 1 || (mode < 0 && nMinor >= kSlabAutoMinor);
  return want && nMajor > 0 && slabFits(nMajor, nMinor);  // minors that do not fit the packing: CSR stream
}
}  // namespace

// Stream plan (blocks of whole short majors) and segment tasks (long majors) of the CSR arrays in beg/idx/val.
// Slab layout: those arrays hold only the majors longer than kSlabLongLimit, so chunk = that limit and every one
// of them becomes segment tasks (longVecIndex = compact index -> major).
void DeviceMatrix::uploadPlans(const std::vector<int32_t>& hostBeg, int32_t nCsrMajor, const int32_t* longVecIndex,
                               hipStream_t s, const int32_t* hostLongIdx, const std::vector<int8_t>* tileOwner, int32_t tileLog2) {
  const int64_t nnzCsr = nCsrMajor > 0 ? (int64_t)hostBeg[nCsrMajor] : 0;
  chunk = spmvChunkFor(nnzCsr);
  StreamPlan plan = planStream(hostBeg, nCsrMajor, useSlab ? kSlabLongLimit : chunk, kMaxMajorsPerBlock);
  if (useSlab && plan.nBlocks != 0) throw std::runtime_error("slab layout: a short major in the side matrix");
  nBlocks = plan.nBlocks;
  blockBeg.alloc(plan.blockBeg.size());
  blockBeg.upload(plan.blockBeg.data(), plan.blockBeg.size(), s);
  std::vector<int32_t> vecIdx;
  if (longVecIndex)
    for (int32_t c : plan.longMajors) vecIdx.push_back(longVecIndex[c]);
  static_assert(sizeof(LongTask) == sizeof(LongTaskHost) && sizeof(LongTask) == 32, "task record layout");
  // tasks per workgroup.  Stream layout: the 4 waves of a workgroup, tasks in (major, segment) order.  Slab layout:
  // pdlp_host.hpp planSlabTasks (group size by the number of CUs, XCD-affine deal)
  LongPlan L;
  if (useSlab) {
    L = planSlabTasks(hostBeg, hostLongIdx, (int32_t)plan.longMajors.size(), longVecIndex, balanceTaskBlocks,
                      affineTasks && hostLongIdx && tileOwner ? tileOwner : nullptr, tileLog2, slab.nBlocks, taskGroup);
  } else {
    taskGroup = kSpmvThreads / 64;
    L = planLong(hostBeg, plan.longMajors, longVecIndex ? vecIdx.data() : nullptr, taskGroup);
  }
  nLong = L.nLong;
  nTasks = L.nTasks;
  longGroup = nLong > kLongSlotCap ? (nLong + kLongSlotCap - 1) / kLongSlotCap : 1;
  longSlots = (nLong + longGroup - 1) / longGroup;
  lTasks.alloc((size_t)nTasks);
  lTasks.upload(reinterpret_cast<const LongTask*>(L.tasks.data()), (size_t)nTasks, s);
  lSegSum.alloc((size_t)std::max(L.nSegSlots, 1));
  lSegSum.zero(s);
  lTicket.alloc((size_t)nLong);
  lTicket.zero(s);
  if (longGroup > 1) {
    lContrib.alloc((size_t)2 * nLong);
    lContrib.zero(s);
  }
  PDLP_HIP(hipStreamSynchronize(s));  // host vectors go out of scope
}

// Do the row blocks of this operand touch few 2^14-entry stretches of the gathered vector, densely (at most 8 per block,
// at least one gather per 8 of their elements, for blocks holding most of the entries)?  Then the slab layout is built
// with slabs of that width (pdlp_kernels.hpp kSlabTileLog2).
namespace { constexpr int32_t kLocalMaxTiles = 8; }
bool DeviceMatrix::touchesFewTiles(const std::vector<int32_t>& lo, const std::vector<int32_t>& hi, const std::vector<int32_t>& cnt) {
  int64_t good = 0, all = 0;
  for (size_t b = 0; b < lo.size(); ++b) {
    if (cnt[b] <= 0) continue;
    const int64_t t = (hi[b] >> kSlabTileLog2) - (lo[b] >> kSlabTileLog2) + 1;  // (an upper bound: tiles in between may be untouched)
    all += cnt[b];
    if (t <= kLocalMaxTiles && (int64_t)cnt[b] * 8 >= t * ((int64_t)1 << kSlabTileLog2)) good += cnt[b];
  }
  return all > 0 && good * 5 >= all * 4;
}

void DeviceMatrix::upload(const Compressed& cIn, int32_t nMajor_, int32_t nMinor_, const DevSwitches& sw, hipStream_t s) {
  nMajor = nMajor_;
  nnz = cIn.beg.empty() ? 0 : cIn.beg[nMajor_];
  useSlab = chooseSlab(sw.slab, nMajor_, nMinor_);
  affineTasks = sw.affineTasks != 0;
  const Compressed* c = &cIn;
  SlabLayout L;
  std::vector<int8_t> tileOwner;
  int32_t tileLog2 = 0;
  if (useSlab) {
    std::vector<int32_t> cold((size_t)std::max(nMajor_, 1));
    slabColdCounts(cIn.beg.data(), cIn.idx.data(), nMajor_, nMinor_, kSlabLongLimit, cold.data());
    const SlabPartition part = slabPartition(cIn.beg.data(), cold.data(), nMajor_, nMinor_, kSlabLongLimit, majorCost);
    std::vector<int32_t> lo, hi, cnt;
    tileLog2 = xcdTileLog2(nMinor_);
    const int32_t nTiles = xcdTileCount(nMinor_, tileLog2);
    const std::vector<int32_t> hist = slabTileHistogram(cIn.beg.data(), cIn.idx.data(), part, kSlabLongLimit, tileLog2, nTiles, lo, hi, cnt);
    tileOwner = xcdTileOwners(hist, nTiles);
    buildSlabLayout(cIn, nMajor_, nMinor_, kSlabLongLimit, slabWidthFor(sw, touchesFewTiles(lo, hi, cnt)), majorCost, L);
    if (L.rowsPerBlock > kSlabMaxRows) throw std::runtime_error("slab layout: too many majors per block");
    wavePtr.alloc(L.wavePtr.size());
    wavePtr.upload(L.wavePtr.data(), L.wavePtr.size(), s);
    waveBeg.alloc(L.waveBeg.size());
    waveBeg.upload(L.waveBeg.data(), L.waveBeg.size(), s);
    ent.alloc(L.ent.size() + 1);  // one pad element: an empty wave still reads its first entry
    slabVal.alloc(L.val.size() + 1);
    longMask.alloc(L.longMask.size());
    ent.zero(s);
    slabVal.zero(s);
    ent.upload(L.ent.data(), L.ent.size(), s);
    slabVal.upload(L.val.data(), L.val.size(), s);
    longMask.upload(L.longMask.data(), L.longMask.size(), s);
    slab = SlabMat{wavePtr.get(), ent.get(), slabVal.get(), longMask.get(), waveBeg.get(), nMajor_, L.nBlocks, L.rowsPerBlock, L.minorBits};
    c = &L.longCsr;
  }
  const int32_t nCsrMajor = useSlab ? (int32_t)L.longMap.size() : nMajor_;
  const int64_t nnzCsr = c->beg[nCsrMajor];
  beg.alloc(c->beg.size());
  idx.alloc((size_t)nnzCsr + 1);  // one pad element: the kernels clamp, never predicate, their loads
  val.alloc((size_t)nnzCsr + 1);
  idx.zero(s);
  val.zero(s);
  beg.upload(c->beg.data(), c->beg.size(), s);
  idx.upload(c->idx.data(), (size_t)nnzCsr, s);
  val.upload(c->val.data(), (size_t)nnzCsr, s);
  uploadPlans(c->beg, nCsrMajor, useSlab ? L.longMap.data() : nullptr, s, c->idx.data(), useSlab ? &tileOwner : nullptr, tileLog2);
  PDLP_HIP(hipStreamSynchronize(s));  // host vectors may go out of scope
}

void DeviceMatrix::buildFromDevice(DeviceCsrData& M, const DevSwitches& sw, hipStream_t s) {
  nMajor = M.nMajor;
  nnz = M.nnz;
  useSlab = chooseSlab(sw.slab, M.nMajor, M.nMinor);
  affineTasks = sw.affineTasks != 0;
  std::vector<int32_t> hostBeg, hostLongMap, hostLongIdx;
  std::vector<int8_t> tileOwner;
  int32_t tileLog2 = 0;
  int32_t nCsrMajor = nMajor;
  bool localM = false;
  if (useSlab) {
    DeviceSlabLayout L;
    gpuSlabPartition(M, kSlabLongLimit, majorCost, s, L);
    const int32_t nB = L.nBlocks;
    {  // per-block span of the short majors, from the CSR that is already in HBM
      // ... and which XCD's blocks gather from which stretch of the vector (the home of the long majors' segment tasks)
      std::vector<int32_t> lo((size_t)nB, INT_MAX), hi((size_t)nB, -1), cnt((size_t)nB, 0);
      tileLog2 = xcdTileLog2(M.nMinor);
      const int32_t nTiles = xcdTileCount(M.nMinor, tileLog2);
      std::vector<int32_t> hist((size_t)8 * nTiles, 0);
      DeviceArray<int32_t> dLo, dHi, dCn, dHist;
      dLo.alloc(nB); dHi.alloc(nB); dCn.alloc(nB); dHist.alloc(hist.size());
      dHist.zero(s);
      launchBlockSpan(M.beg.get(), M.idx.get(), L.waveBeg.get(), nB, kSlabLongLimit, dLo.get(), dHi.get(), dCn.get(), tileLog2, nTiles,
                      dHist.get(), s);
      dLo.download(lo.data(), nB, s); dHi.download(hi.data(), nB, s); dCn.download(cnt.data(), nB, s);
      dHist.download(hist.data(), hist.size(), s);
      PDLP_HIP(hipStreamSynchronize(s));
      localM = touchesFewTiles(lo, hi, cnt);
      tileOwner = xcdTileOwners(hist, nTiles);
      // how long the runs of equal majors are at the rule's widest slabs (2^17): entries per short major of a block, over
      // the number of slabs the block's span crosses — 1 on a random operand (every entry of a row in another slab), the
      // major's whole length where a block is local.  Below 2 no narrower slab can help (buildSlabTuned skips its candidates).
      double num = 0.0, den = 0.0;
      for (int32_t b = 0; b < nB; ++b) {
        if (cnt[b] <= 0) continue;
        const double majors = std::max(1, L.hostWaveBeg[(size_t)(b + 1) * kSlabWavesPerBlock] - L.hostWaveBeg[(size_t)b * kSlabWavesPerBlock]);
        const double meanLen = (double)cnt[b] / majors;
        const double slabs = (double)((hi[b] >> kSlabWidthLog2) - (lo[b] >> kSlabWidthLog2) + 1);
        num += (double)cnt[b] * (meanLen / std::min(std::max(meanLen, 1.0), slabs));
        den += (double)cnt[b];
      }
      estRunLen = den > 0.0 ? num / den : 1.0;
    }
    // (an operand whose blocks touch few 16384-entry tiles of the gathered vector densely gets slabs of that width: its
    // runs of equal majors are shorter, more lanes add in parallel — bench.py --config c, A x: 44.0 -> 41.1 us)
    slabWidthLog2 = slabWidthFor(sw, localM);
    gpuBuildSlabLayout(M, kSlabLongLimit, slabWidthLog2, majorCost, s, L);
    if (L.rowsPerBlock > kSlabMaxRows) throw std::runtime_error("slab layout: too many majors per block");
    wavePtr = std::move(L.wavePtr);
    waveBeg = std::move(L.waveBeg);
    ent = std::move(L.ent);
    slabVal = std::move(L.val);
    longMask = std::move(L.longMask);
    slab = SlabMat{wavePtr.get(), ent.get(), slabVal.get(), longMask.get(), waveBeg.get(), nMajor, L.nBlocks, L.rowsPerBlock, L.minorBits};
    beg = std::move(L.longCsr.beg);
    idx = std::move(L.longCsr.idx);
    val = std::move(L.longCsr.val);
    hostBeg = L.hostLongBeg;
    nCsrMajor = L.nLong;
    hostLongMap.resize((size_t)L.nLong);
    L.longMap.download(hostLongMap.data(), (size_t)L.nLong, s);
    hostLongIdx.resize((size_t)hostBeg[L.nLong] + 1);  // (the minors of the long majors: where their segments gather from)
    if (L.nLong > 0) idx.download(hostLongIdx.data(), (size_t)hostBeg[L.nLong], s);
    PDLP_HIP(hipStreamSynchronize(s));
  } else {
    beg = std::move(M.beg);
    idx = std::move(M.idx);  // allocated with one pad element
    val = std::move(M.val);
    hostBeg.resize((size_t)nMajor + 1);
    beg.download(hostBeg.data(), hostBeg.size(), s);
    PDLP_HIP(hipStreamSynchronize(s));
  }
  uploadPlans(hostBeg, nCsrMajor, useSlab ? hostLongMap.data() : nullptr, s, useSlab ? hostLongIdx.data() : nullptr,
              useSlab ? &tileOwner : nullptr, tileLog2);
}

MatView DeviceMatrix::view() const {
  MatView v{};
  const int32_t slabBlocks = useSlab ? slab.nBlocks : 0;
  v.csr = SpmvMat{beg.get(), idx.get(), val.get(), blockBeg.get(), nMajor, nBlocks, slabBlocks, chunk};
  v.slab = slab;
  v.slab.noPace = noPace;
  v.lng = LongMat{idx.get(), val.get(), lTasks.get(), lSegSum.get(), lTicket.get(), longGroup > 1 ? lContrib.get() : nullptr,
                  nLong, nTasks, slabBlocks + nBlocks, longSlots, longGroup, taskGroup};
  v.useSlab = useSlab ? 1 : 0;
  v.xcdMap = xcdMap;
  v.coTaskBlocks = fusedCoTasks;
  v.touchTail = touchTail;
  v.nPartials = nPartials();
  return v;
}

float tuneXcdMap(DeviceMatrix& M, const DevSwitches& sw, const double* in, double* out, hipStream_t s) {
  const bool em = sw.xcdMap >= 0;
  const bool ep = sw.slabPace >= 0;  // 1 = barrier per group (random operands), 0 = free-running waves
  if (em) M.xcdMap = sw.xcdMap != 0;
  if (ep) M.noPace = sw.slabPace == 0;
  if (M.nnz < 200000) {  // small operands live in every L2 anyway
    if (!em) M.xcdMap = 1;
    return 0.f;
  }
  if (M.mapTuned) return 0.f;  // (chosen together with the slab width: buildSlabTuned)
  hipEvent_t e0, e1;
  PDLP_HIP(hipEventCreate(&e0));
  PDLP_HIP(hipEventCreate(&e1));
  float best = 0.f;
  int bestMap = M.xcdMap, bestFree = M.noPace;
  bool first = true;
  for (int fr = 0; fr < (M.useSlab && !ep ? 2 : 1); ++fr) {
    for (int map = 0; map < (em ? 1 : 2); ++map) {
      if (!em) M.xcdMap = map;
      if (!ep) M.noPace = fr;
      launchSpmvPlain(M.view(), in, out, s);  // warm-up
      PDLP_HIP(hipEventRecord(e0, s));
      for (int r = 0; r < 3; ++r) launchSpmvPlain(M.view(), in, out, s);
      PDLP_HIP(hipEventRecord(e1, s));
      PDLP_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      PDLP_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (first || ms < 0.97f * best) { best = ms; bestMap = M.xcdMap; bestFree = M.noPace; first = false; }  // a change only when clearly faster
    }
  }
  M.xcdMap = bestMap;
  M.noPace = bestFree;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return best / 3.f;
}

// The slab WIDTH of an operand, by timing (round 6).  The width decides how long the runs of equal majors inside a
// 64-entry group are — a run is added by ONE lane — and which stretch of the gathered vector a CU's waves sweep together;
// it changes no sum (a major's runs are added in ascending minor order whatever the width).  The rule of round 3 (2^14 for
// operands whose blocks touch few stretches densely, else 2^17) is right for the random LP and misses structured operands
// whose majors are long AND local: the tall held-out LP's columns (35 entries within a few thousand rows) stream 20 %
// faster at 2^11, config c's rows (16 entries) 6 % faster at 2^13 — while config d's and the power-law LP's operands lose
// up to 35 % there.  So: the rule's layout first, then 2^13 and 2^11, each with its own XCD map / pacing (tuneXcdMap), the
// fastest plain SpMV stays.  Only for operands of at least 2^20 nonzeros built on the device (a candidate costs one more
// slab build, ~10 ms at 5 M nonzeros); PDLP_MI355X_SLAB_W (development) forces a width.
void buildSlabTuned(DeviceMatrix& out, DeviceCsrData& M, const DevSwitches& sw, hipStream_t s) {
  out.buildFromDevice(M, sw, s);
  if (!out.useSlab || sw.slabW > 0 || sw.slabTune == 0 || M.nnz < ((int64_t)1 << 20)) return;  // (the stream layout has consumed M's arrays)
  if (out.estRunLen < 2.0) return;  // runs of one entry (random operands): nothing a narrower slab could shorten
  DeviceArray<double> in, res;
  in.alloc((size_t)std::max(M.nMinor, 1));
  res.alloc((size_t)std::max(M.nMajor, 1));
  in.zero(s);
  float best = tuneXcdMap(out, sw, in.get(), res.get(), s);
  out.mapTuned = true;
  const int32_t autoW = out.slabWidthLog2;
  for (int32_t W : {13, 11}) {
    if (W == autoW || ((int64_t)1 << W) >= (int64_t)M.nMinor) continue;
    DevSwitches sw2 = sw;
    sw2.slabW = W;
    DeviceMatrix cand;
    cand.majorCost = out.majorCost;
    cand.balanceTaskBlocks = out.balanceTaskBlocks;
    cand.buildFromDevice(M, sw2, s);
    const float t = tuneXcdMap(cand, sw, in.get(), res.get(), s);
    cand.mapTuned = true;
    if (t < 0.97f * best) {  // a change only when clearly faster
      best = t;
      out = std::move(cand);
    }
    PDLP_HIP(hipStreamSynchronize(s));  // (the loser's arrays are released with no launch on them in flight)
  }
}

// The environment, read ONCE per solver (never per launch).  What a user may need is listed in INTEGRATION.md section 4;
// the rest are development and test switches.
DevSwitches DevSwitches::fromEnv() {
  DevSwitches w;
  // user switches: always read
  auto num = [](const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
  };
  // development / test switches: only with PDLP_MI355X_DEV=1 (pdlp_host.hpp devEnv)
  auto dev = [](const char* name, int dflt) {
    const char* e = devEnv(name);
    return e ? atoi(e) : dflt;
  };
  auto devStr = [](const char* name) {
    const char* e = devEnv(name);
    return std::string(e ? e : "");
  };
  w.gpuSetup = num("PDLP_MI355X_GPU_SETUP", -1);
  w.fused = num("PDLP_MI355X_FUSED", -1);
  w.persistent = num("PDLP_MI355X_PERSISTENT", -1);
  w.barrierTimeoutMs = num("PDLP_MI355X_BARRIER_TIMEOUT_MS", 1000);
  {
    const char* e = getenv("PDLP_MI355X_EXCHANGE");
    w.exchange = e ? e : "";
  }
  w.graph = dev("PDLP_MI355X_GRAPH", -1);
  w.forceComm = dev("PDLP_MI355X_FORCE_COMM", 0);
  w.slab = dev("PDLP_MI355X_SLAB", -1);
  w.slabW = dev("PDLP_MI355X_SLAB_W", 0);
  w.xcdMap = dev("PDLP_MI355X_XCD_MAP", -1);
  w.slabPace = dev("PDLP_MI355X_SLAB_PACE", -1);
  w.slabTune = dev("PDLP_MI355X_SLAB_TUNE", 1);
  w.affineTasks = dev("PDLP_MI355X_AFFINE_TASKS", 1);
  w.fusedStream = dev("PDLP_MI355X_FUSED_STREAM", 0);
  w.fusedCoTasks = dev("PDLP_MI355X_FUSED_COTASKS", -1);
  w.uniformBounds = dev("PDLP_MI355X_UNIFORM_BOUNDS", 1);
  w.constCached = dev("PDLP_MI355X_CONST_CACHED", -1);
  w.touchTail = dev("PDLP_MI355X_TOUCH_TAIL", 1);
  w.xcdLocal = dev("PDLP_MI355X_XCD_LOCAL", -1);
  w.hierBarrier = dev("PDLP_MI355X_HIER_BARRIER", -1);
  w.deviceCheck = dev("PDLP_MI355X_DEVICE_CHECK", -1);
  w.checkSmall = dev("PDLP_MI355X_CHECK_SMALL", -1);
  w.primalInA = dev("PDLP_MI355X_PRIMAL_IN_A", -1);
  w.fault = dev("PDLP_MI355X_FAULT", 0);
  if (w.fault) fprintf(stderr, "pdlp_mi355x: PDLP_MI355X_FAULT=%d is set — a TEST hook that makes a barrier launch time out on purpose; "
                               "this solve will stall for the barrier timeout and continue on the slower plain-launch path\n", w.fault);
  w.meshLayout = devStr("PDLP_MI355X_MESH_LAYOUT");
  return w;
}

// One launch sequence with in-kernel grid barriers at a time per DEVICE and process: two solvers (two Highs instances on
// two threads: SURVEY section 8(b), lp_data/HighsSolve.cpp:97-104) whose barrier launches interleave could each hold
// CUs the other one's last workgroups need.  The order is imposed on the device, not on the hosts: a solver takes the
// gate only while it ENQUEUES a round — its stream first waits for the event that ends the previous barrier round of
// any context on this device, and records the next event behind its own launches — and synchronises with the gate
// released, so the next context's round is queued (and starts on the device) while this one is still waiting for, and
// then looking at, its results.  (Round 4 held the gate across the synchronisation: two contexts never overlapped at
// all, not even host work with device work.)  Solvers without barrier launches (sharded, HiPDLP, after a fall-back)
// never take it.  The two events per device are created by the first solver that needs them and live as long as the
// process: no solver owns what another one's stream may still be waiting on.
Solver::DeviceGate& Solver::deviceGate(int device) {
  static DeviceGate gates[64];
  return gates[device >= 0 && device < 64 ? device : 0];
}
std::unique_lock<std::mutex> Solver::beginBarrierRound() {
  std::unique_lock<std::mutex> gate;
  if (!(persistent_ || fused_)) return gate;
  DeviceGate& G = deviceGate(opt_.device);
  gate = std::unique_lock<std::mutex>(G.mu);
  if (G.recorded) PDLP_HIP(hipStreamWaitEvent(stream_, G.ev[G.cur], 0));
  return gate;
}
void Solver::endBarrierRound(std::unique_lock<std::mutex>& gate) {
  if (!gate.owns_lock()) return;
  DeviceGate& G = deviceGate(opt_.device);
  const int nxt = G.cur ^ 1;
  if (!G.ev[nxt]) {
    PDLP_HIP(hipSetDevice(opt_.device));  // (the event belongs to the device of the gate, whichever thread gets here first)
    PDLP_HIP(hipEventCreateWithFlags(&G.ev[nxt], hipEventDisableTiming));
  }
  PDLP_HIP(hipEventRecord(G.ev[nxt], stream_));
  G.cur = nxt;
  G.recorded = true;
  gate.unlock();
}
// The end of a round is recorded on EVERY way out of it: if enqueueing throws (a failed launch or graph capture), the
// barrier kernels that are already queued on this stream must still be ordered in front of the next context's round.
Solver::BarrierRound::~BarrierRound() {
  if (!gate.owns_lock()) return;
  try {
    self.endBarrierRound(gate);
  } catch (...) {  // (recording failed too: the gate is released by the lock's destructor; the next round may time out and fall back)
  }
}

double Solver::elapsed() const {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - solveBeg_).count();
}

// Every rank must take the same branch (they issue the same collectives): when sharded, the
// time-limit test is the OR over the ranks' clocks.
bool Solver::timeIsUp() {
  bool up = elapsed() > opt_.time_limit;
  if (!sharded_ || !std::isfinite(opt_.time_limit)) return up;
  hostStats_[kStatTotal + 1] = up ? 1.0 : 0.0;
  PDLP_HIP(hipMemcpyAsync(statOut_.get() + kStatTotal + 1, hostStats_ + kStatTotal + 1, sizeof(double),
                          hipMemcpyHostToDevice, stream_));
  sumOverRanks(statOut_.get() + kStatTotal + 1, 1);
  PDLP_HIP(hipMemcpyAsync(hostStats_ + kStatTotal + 1, statOut_.get() + kStatTotal + 1, sizeof(double),
                          hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  return hostStats_[kStatTotal + 1] > 0.0;
}

void Solver::log(int level, const char* fmt, ...) const {
  if (opt_.log_level < level || rank_ != 0) return;
  va_list ap;
  va_start(ap, fmt);
  logLineV(opt_, level, fmt, ap);
  va_end(ap);
}

Solver::Solver(const pdlp_problem_t& P, const pdlp_params_t& opt, int32_t rank, int32_t world, const void* id128)
    : opt_(opt), rank_(rank), world_(world) {
  try {
    construct(P, id128);
  } catch (...) {
    release();  // a throwing constructor never runs the destructor
    throw;
  }
}

void Solver::construct(const pdlp_problem_t& P, const void* id128) {
  const auto t0 = std::chrono::steady_clock::now();
  if (world_ < 1 || rank_ < 0 || rank_ >= world_) throw std::runtime_error("bad rank/world");
  validateProblem(P);
  requireConstraints(P);
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0)
    throw std::runtime_error("pdlp_mi355x: no HIP device available (this library has no CPU fallback)");
  PDLP_HIP(hipSetDevice(opt_.device));
  PDLP_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  sw_ = DevSwitches::fromEnv();
  if (sw_.graph >= 0) useGraph_ = sw_.graph != 0;

  adaptive_ = !(opt_.features_off & PDLP_FEATURE_ADAPTIVE_STEP_OFF);
  restartOn_ = !(opt_.features_off & PDLP_FEATURE_RESTART_OFF) && opt_.restart_method != 0;

  log(1, "Solving with PDLP on MI355X (gfx950, HIP)\n");
  sharded_ = world_ > 1 || sw_.forceComm != 0;
  // GPU-side setup pays off once the matrix is big enough to amortise its ~40 launches / syncs;
  // small LPs are prepared on the host (same bits either way)
  const int64_t nnzIn = P.num_col > 0 && P.a_start ? (int64_t)P.a_start[P.num_col] : 0;
  gpuSetup_ = nnzIn >= 200000;
  if (sw_.gpuSetup >= 0) gpuSetup_ = sw_.gpuSetup != 0;
  hasQoff_ = hessianHasOffDiagonal(P);
  if (hasQoff_) {
    gpuSetup_ = false;  // the off-diagonal part of Q is scaled with the columns on the host (pdlp_host.cpp applyScaling)
    if (sharded_) throw std::runtime_error("pdlp_mi355x: a Hessian with off-diagonal entries is solved on one GPU (num_devices = 1)");
  }
  // sharded: every rank still prepares the WHOLE problem (Ruiz scaling couples all rows and columns) but does it on
  // its device and copies the result back once; only the row-block cut and the upload of its shard stay on the host
  const bool shardedGpuSetup = sharded_ && gpuSetup_;
  if (sharded_) gpuSetup_ = false;
  const bool doScale = !(opt_.features_off & PDLP_FEATURE_SCALING_OFF);
  DeviceProblem devProb;
  if (gpuSetup_) {
    // formulate + scale + both orientations on the device; F_ keeps only the host-side bookkeeping
    gpuPrepare(P, doScale, stream_, devProb);
    F_ = StandardForm();
    F_.n = devProb.n; F_.m = devProb.m; F_.n0 = devProb.n0; F_.nEqs = devProb.nEqs; F_.nnz = devProb.nnz;
    F_.scaled = devProb.scaled; F_.offset = devProb.offset; F_.sense = devProb.sense;
    F_.normCost = devProb.normCost; F_.normRhs = devProb.normRhs; F_.matNormInf = devProb.matNormInf;
    F_.rowKind = std::move(devProb.rowKind);
    F_.rowNewIdx = std::move(devProb.rowNewIdx);
    F_.colScale = std::move(devProb.hColScale);
    F_.rowScale = std::move(devProb.hRowScale);
    sumCost2_ = devProb.sumCost2;
    sumRhs2_ = devProb.sumRhs2;
  } else if (shardedGpuSetup) {
    gpuPrepare(P, doScale, stream_, devProb);
    sumCost2_ = devProb.sumCost2;
    sumRhs2_ = devProb.sumRhs2;
    // A rank keeps only its shard (round 6): in the two-all-gathers layout of the mesh exchange both operands of a rank
    // are CONTIGUOUS pieces of the two orientations that the device-side set-up has just built (rows [r0, r1) of A by rows,
    // columns [c0, c1) of A by columns), so they are cut on the device (uploadShardFromDevice below) and nothing but the
    // row starts (4 bytes per row, for the row partition) and the scale vectors crosses PCIe.  The RCCL exchange and the
    // round-1 mesh layout need the transpose of the row slab: the whole form comes to the host as before.
    shardOnDevice_ = sw_.exchange != "rccl" && sw_.meshLayout != "partial";
    if (shardOnDevice_) {
      F_ = StandardForm();
      F_.n = devProb.n; F_.m = devProb.m; F_.n0 = devProb.n0; F_.nEqs = devProb.nEqs; F_.nnz = devProb.nnz;
      F_.scaled = devProb.scaled; F_.offset = devProb.offset; F_.sense = devProb.sense;
      F_.normCost = devProb.normCost; F_.normRhs = devProb.normRhs; F_.matNormInf = devProb.matNormInf;
      F_.rowKind = std::move(devProb.rowKind);
      F_.rowNewIdx = std::move(devProb.rowNewIdx);
      F_.colScale = std::move(devProb.hColScale);
      F_.rowScale = std::move(devProb.hRowScale);
      F_.csr.beg.resize((size_t)F_.m + 1);
      devProb.A.beg.download(F_.csr.beg.data(), F_.csr.beg.size(), stream_);
      PDLP_HIP(hipStreamSynchronize(stream_));
    } else {
      downloadForm(devProb, F_, stream_);
      devProb = DeviceProblem();  // released before any exchange kernel of another rank can run on this device
    }
  } else {
    formulate(P, F_);
    if (doScale) scale(F_);
    finalize(F_);
    sumCost2_ = 0.0;
    for (double v : F_.cost) sumCost2_ += v * v;
    sumRhs2_ = 0.0;
    for (double v : F_.rhs) sumRhs2_ += v * v;
  }
  log(1, "Using cost norm = %9.3g and RHS norm = %9.3g\n", F_.normCost, F_.normRhs);
  if (F_.nnz < 50000)  // DESIGN.md section 3, "Small LPs": three dependent launches of ~5 us per iteration
    log(1, "Note: %lld nonzeros - at this size the GPU iteration is launch-latency-bound (~20-25 us) and not "
           "faster than the reference's CPU pdlp\n", (long long)F_.nnz);

  // hot start in formulated+scaled space (PDHG_PreSolve, cupdlp_solver.c:1217-1279)
  if (P.start_value_valid && P.start_dual_valid && P.start_col_value && P.start_row_value && P.start_row_dual) {
    startX_.assign(F_.n, 0.0);
    startY_.assign(F_.m, 0.0);
    int32_t j = 0;
    for (; j < F_.n0; ++j) startX_[j] = P.start_col_value[j];
    for (int32_t i = 0; i < F_.m; ++i) {
      const double mu = F_.rowKind[i] == kRowLeq ? -1.0 : 1.0;
      startY_[F_.rowNewIdx[i]] = F_.sense * mu * P.start_row_dual[i];
      if (F_.rowKind[i] == kRowBound) startX_[j++] = P.start_row_value[i];
    }
    if (F_.scaled) {
      for (int32_t k = 0; k < F_.n; ++k) startX_[k] *= F_.colScale[k];
      for (int32_t k = 0; k < F_.m; ++k) startY_[k] *= F_.rowScale[k];
    }
    hasStart_ = true;
  }
  // cuPDLP treats "either flag set" as has_variables (cupdlp_solver.c:1465) but
  // only fills x,y when both are; with one flag the start is the zero vector.

  // PDLP_MI355X_FORCE_COMM=1 runs the sharded kernel sequence and the RCCL
  // all-reduce with a single rank (lets a 1-GPU box exercise the multi-GPU path)
  c0_ = 0;
  c1_ = nLoc_ = F_.n;
  if (sharded_) {
    std::vector<int32_t> off = rowPartition(F_.csr, F_.m, world_);
    r0_ = off[rank_];
    r1_ = off[rank_ + 1];
    if (shardOnDevice_) {
      // (the mesh's column slices, pdlp_mesh.hip: n * h / world) — the shard is cut, and every temporary of the cut is
      // released, BEFORE the exchange exists: a hipFree synchronises the device, and with several ranks folded onto one
      // device it would wait for a peer's kernel that is waiting for this rank
      c0_ = (int32_t)((int64_t)F_.n * rank_ / world_);
      c1_ = (int32_t)((int64_t)F_.n * (rank_ + 1) / world_);
      nLoc_ = c1_ - c0_;
      mLoc_ = r1_ - r0_;
      uploadShardFromDevice(devProb);
      devProb = DeviceProblem();
      PDLP_HIP(hipStreamSynchronize(stream_));
    }
    // Exchange: the direct xGMI mesh unless PDLP_MI355X_EXCHANGE=rccl, or the mesh cannot be
    // set up / fails its known-answer test on some rank (then EVERY rank uses RCCL).
    bool wantMesh = sw_.exchange != "rccl";
    if (wantMesh) {
      try {
        mesh_ = new Mesh(rank_, world_, id128, F_.n, F_.m, off, stream_);
        const bool ok = mesh_->selfTest(stream_);
        if (!mesh_->allAgree(ok)) throw std::runtime_error("the exchange self-test failed on some rank");
      } catch (const std::exception& e) {
        // every failure path ends here on EVERY rank (a rank that fails keeps taking part in the
        // rendezvous and votes "not ok"; a rank that vanished makes the others time out)
        fprintf(stderr, "pdlp_mi355x[rank %d]: direct xGMI exchange unavailable (%s); using RCCL\n", rank_, e.what());
        delete mesh_;
        mesh_ = nullptr;
      }
    }
    meshMode_ = mesh_ != nullptr;
    if (shardOnDevice_ && !meshMode_) {
      // the direct exchange is not available after all: the RCCL path needs the transpose of the row slab — prepare
      // again and bring the whole form to the host (every rank takes this branch together: allAgree above)
      DeviceProblem again;
      gpuPrepare(P, doScale, stream_, again);
      downloadForm(again, F_, stream_);
      shardOnDevice_ = false;
    }
    if (meshMode_) {
      colblock_ = sw_.meshLayout != "partial";
      if (shardOnDevice_ && (c0_ != mesh_->c0() || c1_ != mesh_->c1())) throw std::runtime_error("pdlp_mi355x: column slices of the shard cut and of the exchange differ");
      c0_ = mesh_->c0();
      c1_ = mesh_->c1();
      nLoc_ = c1_ - c0_;
    } else {
      unsigned char localId[128];
      if (world_ == 1 && !id128) {
        Comm::uniqueId(localId);
        id128 = localId;
      }
      comm_ = new Comm(rank_, world_, id128);
    }
    log(1, "Row-block sharded over %d GPUs, exchange: %s\n", world_,
        !meshMode_ ? "RCCL all-reduce" : colblock_ ? "direct xGMI mesh, two all-gathers (x+ slices, y+ row blocks)"
                                                   : "direct xGMI mesh, all-gather of x+ and reduce-scatter of the A'y partials");
  } else {
    r0_ = 0;
    r1_ = F_.m;
  }
  mLoc_ = r1_ - r0_;
  if (gpuSetup_) uploadProblemFromDevice(devProb);
  else if (shardOnDevice_) allocIterates();  // (the operands and vectors of the shard are in place: uploadShardFromDevice)
  else uploadProblem();
  // block -> XCD assignment of the two operands (x_ / y_ are zero here: any input will do)
  tuneXcdMap(dA_, sw_, x_[0].get(), ax_[0].get(), stream_);
  tuneXcdMap(dAt_, sw_, y_[0].get(), sharded_ ? commBuf_.get() : aty_[0].get(), stream_);
  if (hasQoff_) tuneXcdMap(dQ_, sw_, x_[0].get(), nx_[0].get(), stream_);
  if (devEnv("PDLP_MI355X_SLAB_PROF") && rank_ == 0)  // (development: what the set-up chose, next to the per-block phase profile)
    for (const DeviceMatrix* M : {&dA_, &dAt_})
      fprintf(stderr, "slab operand %s: slab %d (width 2^%d, estimated run length %.1f), blocks %d, XCD map %s, %s, long majors %d, tasks %d in workgroups of %d\n", M == &dA_ ? "A" : "A'",
              (int)M->useSlab, M->slabWidthLog2, M->estRunLen, M->useSlab ? M->slab.nBlocks : M->nBlocks, M->xcdMap ? "contiguous" : "round robin", M->noPace ? "free-running waves" : "paced",
              M->nLong, M->nTasks, M->taskGroup);
  // 2-launch trial where the A' y grid is resident all at once (grid barrier inside the kernel); PDLP_MI355X_FUSED=0 forces
  // 3 launches.  Slab layout (one block per CU): on by default, 140.0 -> 135.0 us per iteration at 1M x 1M.  Stream
  // layout: off by default — measured in round 3 the barrier + decision tail costs what the separate launch did
  // (100k x 100k: 33.1 us fused vs 32.1; 25fv47: 19.5 vs 20.2); PDLP_MI355X_FUSED_STREAM=1 turns it on.
  if (!sharded_) {
    // (long columns: their segment tasks as workgroups of the fused launch, resident next to its streaming blocks, where
    // two 1024-thread blocks per CU fit — PDLP_MI355X_FUSED_COTASKS=0 keeps the task passes inside the streaming blocks)
    dAt_.fusedCoTasks = sw_.fusedCoTasks != 0 ? fusedCoTaskBlocks(dAt_.view(), opt_.device) : 0;
    dAt_.touchTail = sw_.touchTail != 0 ? 1 : 0;
    const MatView at = dAt_.view();
    const bool allowed = at.useSlab ? sw_.fused != 0 : sw_.fusedStream != 0;
    fused_ = allowed && !hasQoff_ && fusedAtyBlocks(at) > 0 && fusedAtyBlocksResident(at, opt_.device) >= fusedAtyBlocks(at);
    // Netlib-class LPs (both operands below 2^18 nonzeros, stream layout, no long majors): the whole trial batch is one
    // persistent launch with grid barriers between the phases (pdlp_small.hip); PDLP_MI355X_PERSISTENT=0 turns it off
    int resident = 0;
    int g = 0;
    if (sw_.persistent != 0 && !hasQoff_) {
      // two barriers per trial instead of three: phase A recomputes x+ of the columns it gathers (PDLP_MI355X_PRIMAL_IN_A=0/1)
      // — where that variant's registers still leave the whole grid resident
      primalInA_ = sw_.primalInA != 0;
      g = smallTrialsGrid(dA_.view(), at, F_.n, opt_.device, &resident, primalInA_);
      if (primalInA_ && (g == 0 || g > resident)) {
        primalInA_ = false;
        g = smallTrialsGrid(dA_.view(), at, F_.n, opt_.device, &resident, false);
      }
    }
    if (g > 0 && g <= resident) {
      persistent_ = true;
      smallGrid_ = g;
      fused_ = false;
      // one XCD has 32 CUs: up to one workgroup per CU the XCD-local mode wins (25fv47, 21 workgroups: 17.6 us per
      // iteration against 19.4 with agent-scope accesses on all XCDs and 20.3 with launches), beyond it the single L2 and
      // the shared CUs cost more than the memory round trips they save (80bau3b, 48 workgroups: 19.7 / 16.8 / 17.4)
      xcdLocal_ = g <= 32 && sw_.xcdLocal != 0;
      // beyond a few dozen workgroups the all-poll-all barrier is what a trial waits for (490 workgroups: 4.8 us per
      // barrier): meet per XCD in its L2 first (PDLP_MI355X_HIER_BARRIER=0/1 forces either)
      const bool smallChunks = dA_.view().csr.chunk == kChunkSmall && at.csr.chunk == kChunkSmall;
      hierBar_ = !smallChunks || (sw_.hierBarrier >= 0 ? sw_.hierBarrier != 0 : g > 64);
    }
    if (fused_) gridBar_.alloc(gridBarWords(fusedAtyBlocks(at)));
    if (fused_ && at.useSlab && sw_.uniformBounds != 0) {
      // bounds that all columns of a block of the fused launch share (x >= 0 without an upper bound is the rule): two scalars
      // per block instead of 8 / 16 bytes per column in the launch's bandwidth-bound tail (IterVecs::colBlockUni)
      colBlockUni_.alloc((size_t)at.slab.nBlocks);
      colBlockBounds_.alloc((size_t)at.slab.nBlocks * 2);
      launchBlockBounds(lower_.get(), upper_.get(), at.slab.waveBeg, at.slab.nBlocks, colBlockUni_.get(), colBlockBounds_.get(), stream_);
      std::vector<int32_t> uni((size_t)at.slab.nBlocks);
      std::vector<double> bnd((size_t)at.slab.nBlocks * 2);
      colBlockUni_.download(uni.data(), uni.size(), stream_);
      colBlockBounds_.download(bnd.data(), bnd.size(), stream_);
      PDLP_HIP(hipStreamSynchronize(stream_));
      bool allLower = true;  // (bit patterns: the kernel's ULO instantiation takes ONE scalar for all columns)
      for (size_t b = 0; b < uni.size(); ++b) allLower = allLower && (uni[b] & 1) && memcmp(&bnd[2 * b], &bnd[0], sizeof(double)) == 0;
      {  // (for bench.py's needed bytes: how many columns' lower / upper bounds the fused launch does not load)
        std::vector<int32_t> wb((size_t)at.slab.nBlocks * kSlabWavesPerBlock + 1);
        PDLP_HIP(hipMemcpyAsync(wb.data(), at.slab.waveBeg, wb.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        PDLP_HIP(hipStreamSynchronize(stream_));
        uniLowerCols_ = uniUpperCols_ = 0;
        for (size_t b = 0; b < uni.size(); ++b) {
          const int64_t cols = wb[(b + 1) * kSlabWavesPerBlock] - wb[b * kSlabWavesPerBlock];
          if (uni[b] & 1) uniLowerCols_ += cols;
          if (uni[b] & 2) uniUpperCols_ += cols;
        }
      }
      for (IterVecs* v : {&vecs_, &vecsCol_, &vecsAty_}) {
        v->colBlockUni = colBlockUni_.get(); v->colBlockBounds = colBlockBounds_.get(); v->lowerUniform = allLower ? 1 : 0;
      }
    }
    if (persistent_) gridBar_.alloc(smallBarWords(smallGrid_));
    // Netlib-class LPs (at most 64 workgroups): the check iteration as one launch too (PDLP_MI355X_CHECK_SMALL=0: ten launches)
    checkSmall_ = persistent_ && smallGrid_ <= 64 && sw_.checkSmall != 0 && checkSmallResident(dA_.view(), at, opt_.device) >= smallGrid_;
    if (checkSmall_) checkBar_.alloc((size_t)smallGrid_ + 8);
  }
  // check iterations on the device (single GPU; the sharded paths issue their check collectives from the host)
  // check iterations on the device: single GPU, and (round 5) the row-block sharded solve in the two-all-gathers layout of the
  // mesh exchange — its check collectives are enqueued with the check's kernels instead of being driven from the host
  devCheck_ = (!sharded_ || (meshMode_ && colblock_)) && sw_.deviceCheck != 0;
  reset();
  // the trial-batch graph is part of the setup, not of the first iterations
  if (useGraph_ && !persistent_ && (!sharded_ || meshMode_)) captureGraph();
  PDLP_HIP(hipStreamSynchronize(stream_));
  setupSeconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void Solver::release() noexcept {
  if (graphExec_) (void)hipGraphExecDestroy(graphExec_);
  for (hipEvent_t e : profEvents_) (void)hipEventDestroy(e);
  profEvents_.clear();
  if (hostState_) (void)hipHostFree(hostState_);
  if (hostStats_) (void)hipHostFree(hostStats_);
  if (hostCtl_) (void)hipHostFree(hostCtl_);
  if (hostRing_) (void)hipHostFree(hostRing_);
  hostCtl_ = nullptr; hostRing_ = nullptr;
  delete comm_;
  delete mesh_;
  if (stream_) (void)hipStreamDestroy(stream_);
  graphExec_ = nullptr; hostState_ = nullptr; hostStats_ = nullptr; comm_ = nullptr; mesh_ = nullptr; stream_ = nullptr;
}

Solver::~Solver() { release(); }

void Solver::uploadProblem() {
  const int32_t n = F_.n;
  dAt_.majorCost = kSlabMajorCostCols;
  if (!sharded_) {
    dA_.upload(F_.csr, F_.m, n, sw_, stream_);
    dAt_.upload(F_.cscSorted, n, F_.m, sw_, stream_);
  } else {
    Compressed csrSlab, cscSlab;
    extractSlab(F_, r0_, r1_, csrSlab, cscSlab);
    dA_.upload(csrSlab, mLoc_, n, sw_, stream_);
    if (colblock_) {  // A'y operand: the columns this rank owns, over ALL rows (rows ascending, as on one GPU)
      Compressed cb;
      const int32_t b = F_.cscSorted.beg[c0_], e = F_.cscSorted.beg[c1_];
      cb.beg.resize((size_t)nLoc_ + 1);
      for (int32_t j = 0; j <= nLoc_; ++j) cb.beg[j] = F_.cscSorted.beg[c0_ + j] - b;
      cb.idx.assign(F_.cscSorted.idx.begin() + b, F_.cscSorted.idx.begin() + e);
      cb.val.assign(F_.cscSorted.val.begin() + b, F_.cscSorted.val.begin() + e);
      dAt_.upload(cb, nLoc_, F_.m, sw_, stream_);
    } else {
      dAt_.upload(cscSlab, n, mLoc_, sw_, stream_);
    }
  }
  cost_.alloc(n); rhs_.alloc(mLoc_); lower_.alloc(n); upper_.alloc(n); colScale_.alloc(n); rowScale_.alloc(mLoc_);
  cost_.upload(F_.cost.data(), n, stream_);
  lower_.upload(F_.lower.data(), n, stream_);
  upper_.upload(F_.upper.data(), n, stream_);
  colScale_.upload(F_.colScale.data(), n, stream_);
  rhs_.upload(F_.rhs.data() + r0_, mLoc_, stream_);
  rowScale_.upload(F_.rowScale.data() + r0_, mLoc_, stream_);
  if (!F_.qdiag.empty()) {
    qdiag_.alloc(n);
    qdiag_.upload(F_.qdiag.data(), n, stream_);
    if (F_.qoff.beg.empty()) log(1, "Quadratic objective (diagonal Hessian): proximal primal step\n");
  }
  if (!F_.qoff.beg.empty()) {
    dQ_.upload(F_.qoff, n, n, sw_, stream_);
    log(1, "Quadratic objective (%lld off-diagonal Hessian entries): proximal step on the diagonal, explicit Q x term for the rest\n",
        (long long)F_.qoff.beg[n]);
    F_.qoff = Compressed();
  }
  allocIterates();
  // the big host copies are not needed any more (postsolve uses only the scale vectors and row maps)
  F_.csc = Compressed(); F_.csr = Compressed(); F_.cscSorted = Compressed();
}

// The device-prepared problem as the host form the shard cut works on (bit-identical to formulate + scale +
// finalize: tests test_gpu_setup_*).
void Solver::downloadForm(DeviceProblem& D, StandardForm& F, hipStream_t s) {
  F = StandardForm();
  F.n = D.n; F.m = D.m; F.n0 = D.n0; F.nEqs = D.nEqs; F.nnz = D.nnz;
  F.scaled = D.scaled; F.offset = D.offset; F.sense = D.sense;
  F.normCost = D.normCost; F.normRhs = D.normRhs; F.matNormInf = D.matNormInf;
  F.rowKind = std::move(D.rowKind);
  F.rowNewIdx = std::move(D.rowNewIdx);
  F.colScale = std::move(D.hColScale);
  F.rowScale = std::move(D.hRowScale);
  auto pull = [&](const DeviceCsrData& M, Compressed& C) {
    C.beg.resize((size_t)M.nMajor + 1);
    C.idx.resize((size_t)M.nnz);
    C.val.resize((size_t)M.nnz);
    M.beg.download(C.beg.data(), C.beg.size(), s);
    M.idx.download(C.idx.data(), C.idx.size(), s);
    M.val.download(C.val.data(), C.val.size(), s);
  };
  pull(D.A, F.csr);
  pull(D.At, F.cscSorted);
  F.cost.resize((size_t)D.n); F.lower.resize((size_t)D.n); F.upper.resize((size_t)D.n); F.rhs.resize((size_t)D.m);
  D.cost.download(F.cost.data(), F.cost.size(), s);
  D.lower.download(F.lower.data(), F.lower.size(), s);
  D.upper.download(F.upper.data(), F.upper.size(), s);
  D.rhs.download(F.rhs.data(), F.rhs.size(), s);
  if (D.qdiag.size()) {
    F.qdiag.resize((size_t)D.n);
    D.qdiag.download(F.qdiag.data(), F.qdiag.size(), s);
  }
  PDLP_HIP(hipStreamSynchronize(s));
}

void Solver::uploadProblemFromDevice(DeviceProblem& D) {
  dAt_.majorCost = kSlabMajorCostCols;
  buildSlabTuned(dA_, D.A, sw_, stream_);
  buildSlabTuned(dAt_, D.At, sw_, stream_);
  cost_ = std::move(D.cost); rhs_ = std::move(D.rhs); lower_ = std::move(D.lower); upper_ = std::move(D.upper);
  colScale_ = std::move(D.colScale); rowScale_ = std::move(D.rowScale);
  if (D.qdiag.size()) {
    qdiag_ = std::move(D.qdiag);
    log(1, "Quadratic objective (diagonal Hessian): proximal primal step\n");
  }
  allocIterates();
}

// rows / columns [lo, hi) of a device CSR as a matrix of its own (contiguous piece: two device-to-device copies, the
// major starts rebased on the host — 4 bytes per major of the piece)
static void sliceDeviceCsr(const DeviceCsrData& M, int32_t lo, int32_t hi, hipStream_t s, DeviceCsrData& out) {
  const int32_t nMaj = hi - lo;
  std::vector<int32_t> hb((size_t)nMaj + 1);
  PDLP_HIP(hipMemcpyAsync(hb.data(), M.beg.get() + lo, sizeof(int32_t) * hb.size(), hipMemcpyDeviceToHost, s));
  PDLP_HIP(hipStreamSynchronize(s));
  const int32_t p0 = hb[0];
  const int64_t nnz = (int64_t)hb[nMaj] - p0;
  for (int32_t& v : hb) v -= p0;
  out.nMajor = nMaj; out.nMinor = M.nMinor; out.nnz = nnz;
  out.beg.alloc(hb.size());
  out.beg.upload(hb.data(), hb.size(), s);
  out.idx.alloc((size_t)nnz + 1);  // one pad element: the kernels clamp, never predicate, their loads
  out.val.alloc((size_t)nnz + 1);
  out.major.alloc((size_t)std::max<int64_t>(nnz, 1));
  out.idx.zero(s);
  out.val.zero(s);
  if (nnz > 0) {
    PDLP_HIP(hipMemcpyAsync(out.idx.get(), M.idx.get() + p0, sizeof(int32_t) * nnz, hipMemcpyDeviceToDevice, s));
    PDLP_HIP(hipMemcpyAsync(out.val.get(), M.val.get() + p0, sizeof(double) * nnz, hipMemcpyDeviceToDevice, s));
    PDLP_HIP(hipMemcpyAsync(out.major.get(), M.major.get() + p0, sizeof(int32_t) * nnz, hipMemcpyDeviceToDevice, s));
    launchAddInt(out.major.get(), -lo, nnz, s);  // entry -> LOCAL major
  }
  PDLP_HIP(hipStreamSynchronize(s));  // hb goes out of scope
}

// The shard of this rank from the device-prepared problem (two-all-gathers layout): A x operand = rows [r0, r1) of A by
// rows, A'y operand = columns [c0, c1) of A by columns (over ALL rows: every column is summed in the single-GPU order),
// column vectors whole (x is replicated), row vectors cut to the own rows.  Same bits as the host cut (uploadProblem).
void Solver::uploadShardFromDevice(DeviceProblem& D) {
  dAt_.majorCost = kSlabMajorCostCols;
  {
    DeviceCsrData As, Ats;
    sliceDeviceCsr(D.A, r0_, r1_, stream_, As);
    sliceDeviceCsr(D.At, c0_, c1_, stream_, Ats);
    D.A = DeviceCsrData();
    D.At = DeviceCsrData();
    dA_.buildFromDevice(As, sw_, stream_);
    dAt_.buildFromDevice(Ats, sw_, stream_);
  }
  cost_ = std::move(D.cost); lower_ = std::move(D.lower); upper_ = std::move(D.upper); colScale_ = std::move(D.colScale);
  rhs_.alloc((size_t)std::max(mLoc_, 1));
  rowScale_.alloc((size_t)std::max(mLoc_, 1));
  if (mLoc_ > 0) {
    PDLP_HIP(hipMemcpyAsync(rhs_.get(), D.rhs.get() + r0_, sizeof(double) * mLoc_, hipMemcpyDeviceToDevice, stream_));
    PDLP_HIP(hipMemcpyAsync(rowScale_.get(), D.rowScale.get() + r0_, sizeof(double) * mLoc_, hipMemcpyDeviceToDevice, stream_));
  }
  if (D.qdiag.size()) {
    qdiag_ = std::move(D.qdiag);
    log(1, "Quadratic objective (diagonal Hessian): proximal primal step\n");
  }
  PDLP_HIP(hipStreamSynchronize(stream_));
}

void Solver::allocIterates() {
  const int32_t n = F_.n;
  yLen_ = colblock_ ? F_.m : mLoc_;
  yOff_ = colblock_ ? r0_ : 0;
  for (int k = 0; k < 2; ++k) {
    x_[k].alloc(n); y_[k].alloc(yLen_); ax_[k].alloc(mLoc_); aty_[k].alloc(n);
    x_[k].zero(stream_); y_[k].zero(stream_); ax_[k].zero(stream_); aty_[k].zero(stream_);
  }
  if (hasQoff_) {
    for (int k = 0; k < 2; ++k) { nx_[k].alloc(n); nx_[k].zero(stream_); }
    nxAvg_.alloc(n);
    nxAvg_.zero(stream_);
    partQ_.alloc(std::max(dQ_.nPartials(), 1));
  }
  xAvg_.alloc(n); yAvg_.alloc(yLen_); axAvg_.alloc(mLoc_); atyAvg_.alloc(n);
  xSum_.alloc(n); ySum_.alloc(mLoc_); xLast_.alloc(n); yLast_.alloc(mLoc_);
  slackPos_.alloc(n); slackNeg_.alloc(n); slackPosAvg_.alloc(n); slackNegAvg_.alloc(n);
  slackPos_.zero(stream_); slackNeg_.zero(stream_); slackPosAvg_.zero(stream_); slackNegAvg_.zero(stream_);
  tmpM_.alloc(yLen_);
  tmpM_.zero(stream_);

  const int32_t nbV = std::max(vecBlocks(n), vecBlocks(std::max(mLoc_, 1)));
  partDY_.alloc(2 * (size_t)std::max(dA_.nPartials(), 1));  // (two halves: the persistent loop without a P phase alternates)
  partDX_.alloc(std::max(std::max(dAt_.nPartials(), nbV), 1));
  partInter_.alloc(std::max(std::max(dAt_.nPartials(), nbV), 1));
  statStride_ = nbV;
  statPart_.alloc((size_t)kStatTotal * statStride_);
  statOut_.alloc(kStatTotal + 8);
  commBuf_.alloc((size_t)n + 8);
  commBuf_.zero(stream_);
  dState_.alloc(2);
  dState_.zero(stream_);
  PDLP_HIP(hipHostMalloc((void**)&hostState_, sizeof(DevState), hipHostMallocDefault));
  PDLP_HIP(hipHostMalloc((void**)&hostStats_, sizeof(double) * (kStatTotal + 8), hipHostMallocDefault));
  memset(hostState_, 0, sizeof(DevState));
  dCtl_.alloc(1);
  dCtl_.zero(stream_);
  PDLP_HIP(hipHostMalloc((void**)&hostCtl_, sizeof(CheckCtl), hipHostMallocDefault));
  PDLP_HIP(hipHostMalloc((void**)&hostRing_, sizeof(CheckRecord) * kRingSlots, hipHostMallocDefault));
  memset(hostCtl_, 0, sizeof(CheckCtl));
  memset(hostRing_, 0, sizeof(CheckRecord) * kRingSlots);
  partRestartY_.alloc((size_t)std::max(vecBlocks(std::max(mLoc_, 1)), 1));

  vecs_ = IterVecs{};
  for (int k = 0; k < 2; ++k) {
    vecs_.x[k] = x_[k].get(); vecs_.y[k] = yl(k); vecs_.ax[k] = ax_[k].get(); vecs_.aty[k] = aty_[k].get();
  }
  vecs_.xSum = xSum_.get(); vecs_.ySum = ySum_.get();
  vecs_.cost = cost_.get(); vecs_.rhs = rhs_.get(); vecs_.lower = lower_.get(); vecs_.upper = upper_.get();
  vecs_.qdiag = qdiag_.size() ? qdiag_.get() : nullptr;
  if (hasQoff_) { vecs_.nx[0] = nx_[0].get(); vecs_.nx[1] = nx_[1].get(); }
  vecs_.n = n; vecs_.m = mLoc_; vecs_.nEqs = F_.nEqs; vecs_.rowOffset = r0_;
  vecs_.constCached = sw_.constCached >= 0 ? (sw_.constCached != 0) : constCached(dA_.nnz, n);
  vecsCol_ = vecs_;
  for (int k = 0; k < 2; ++k) { vecsCol_.x[k] += c0_; vecsCol_.aty[k] += c0_; }
  vecsCol_.xSum += c0_; vecsCol_.cost += c0_; vecsCol_.lower += c0_; vecsCol_.upper += c0_;
  if (vecsCol_.qdiag) vecsCol_.qdiag += c0_;
  vecsCol_.n = nLoc_;
  vecsAty_ = vecsCol_;
  if (colblock_)  // the column-block A'y kernel gathers from the FULL y of the next parity
    for (int k = 0; k < 2; ++k) vecsAty_.y[k] = y_[k].get();
  PDLP_HIP(hipStreamSynchronize(stream_));
}

void Solver::dims(int32_t* n, int32_t* m, int64_t* nnz, int32_t* nEqs) const {
  if (n) *n = F_.n;
  if (m) *m = F_.m;
  if (nnz) *nnz = F_.nnz;
  if (nEqs) *nEqs = F_.nEqs;
}

void Solver::syncState() {
  // the decision of the last enqueued trial is still pending in the single-GPU loop: take it now
  if (!sharded_)
    launchDecide(dst(), partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(), dAt_.nPartials(), nullptr,
                 stream_, true, hasQoff_ ? partQ_.get() : nullptr, hasQoff_ ? dQ_.nPartials() : 0);
  PDLP_HIP(hipMemcpyAsync(hostState_, dst(), sizeof(DevState), hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  PDLP_HIP(hipGetLastError());  // a kernel launch that failed since the last stop (bad grid, LDS request, ...) surfaces here
  if (hostState_->commError == 2 && persistent_ && xcdLocal_) {
    // the persistent launch found its workers on more than one XCD and touched nothing: agent-scope accesses from now on
    xcdLocal_ = false;
    hostState_->commError = 0;
    PDLP_HIP(hipMemcpyAsync(dst(), hostState_, sizeof(DevState), hipMemcpyHostToDevice, stream_));
    PDLP_HIP(hipStreamSynchronize(stream_));
    log(1, "Note: the XCD-local trial loop is not placed on one XCD on this device; continuing with agent-scope accesses\n");
  }
  if (hostState_->commError == 3 && (persistent_ || fused_)) {
    // A launch with in-kernel grid barriers did not get all its workgroups resident in time (the device is shared): the
    // persistent launch has changed nothing (roll call), the fused trial is undecided (pdlp_kernels.hip fusedBarrierFailed:
    // only the pending average weight of y has been consumed).  From here on plain launches, which need no co-residency.
    const bool wasFused = fused_;
    log(1, "Note: the workgroups of a launch with grid barriers were not resident together within %d ms (shared device?); "
           "continuing with %s\n", sw_.barrierTimeoutMs, "3 launches per trial step");
    persistent_ = false;
    fused_ = false;
    xcdLocal_ = false;
    checkSmall_ = false;
    if (graphExec_) { (void)hipGraphExecDestroy(graphExec_); graphExec_ = nullptr; }
    hostState_->commError = 0;
    hostState_->halted = hostState_->nIter >= hostState_->haltIter ? 1 : 0;
    if (wasFused) hostState_->avgW = 0.0;
    gridBar_.zero(stream_);
    ++barrierFallbacks_;
    pushState();
  }
  if (hostState_->commError)
    throw std::runtime_error("pdlp_mi355x: a grid barrier or a peer did not answer in time (exchange timed out)");
}
// (k+1)^-0.3 and (k+1)^-0.6 of the adaptive step rule (cupdlp_step.c:279-284) for the next
// kPowWindow trial counters, computed with the host's pow so that the device takes exactly the
// CPU's values (the oracle can then follow a GPU solve bit for bit).
void Solver::refreshPowTable() {
  constexpr int32_t kPowWindow = 16384, kPowMargin = 4096;
  DevState& s = *hostState_;
  if (s.powRed && s.nTrials >= s.powBase && s.nTrials + kPowMargin < s.powBase + s.powCount) return;
  if (powRed_.size() == 0) { powRed_.alloc(kPowWindow); powGrow_.alloc(kPowWindow); }
  std::vector<double> a(kPowWindow), b(kPowWindow);
  // entry i serves the trial that raises nTrials to powBase + i
  for (int32_t i = 0; i < kPowWindow; ++i) {
    const double k1 = (double)(s.nTrials + i) + 1.0;
    a[i] = std::pow(k1, -0.3);
    b[i] = std::pow(k1, -0.6);
  }
  powRed_.upload(a.data(), kPowWindow, stream_);
  powGrow_.upload(b.data(), kPowWindow, stream_);
  PDLP_HIP(hipStreamSynchronize(stream_));
  s.powBase = s.nTrials;
  s.powCount = kPowWindow;
  s.powRed = powRed_.get();
  s.powGrow = powGrow_.get();
}

// wait = false: the copy stays queued in front of whatever is enqueued next.  Only for callers that do not touch
// hostState_ again before their next stream synchronisation (the pinned buffer is the source of the queued copy).
void Solver::pushState(bool wait) {
  refreshPowTable();
  hostState_->pending = 0;
  needPrimal_ = true;
  PDLP_HIP(hipMemcpyAsync(dst(), hostState_, sizeof(DevState), hipMemcpyHostToDevice, stream_));
  if (wait) PDLP_HIP(hipStreamSynchronize(stream_));
}

// ---- sharding-aware device linear algebra ----------------------------------
void Solver::deviceAx(const double* x, double* axLocal) { launchSpmvPlain(dA_.view(), x, axLocal, stream_); }

void Solver::deviceATy(const double* yLocal, double* aty) {
  if (!sharded_) {
    launchSpmvPlain(dAt_.view(), yLocal, aty, stream_);
  } else if (meshMode_ && colblock_) {
    // yLocal is this rank's rows inside a full-length vector: all-gather the rows, own columns of A'y from the
    // column block, all-gather of the slices (full vector everywhere, as the other layouts leave it)
    double* full = const_cast<double*>(yLocal) - yOff_;
    mesh_->allGather(full, true, stream_);
    launchSpmvPlain(dAt_.view(), full, aty + c0_, stream_);
    mesh_->allGather(aty, false, stream_);
  } else if (meshMode_) {
    // reduce-scatter of the partials to the column owners, then all-gather: full vector everywhere
    launchSpmvPlain(dAt_.view(), yLocal, commBuf_.get(), stream_);
    mesh_->reduceScatterCols(commBuf_.get(), aty, stream_);
    mesh_->allGather(aty, false, stream_);
  } else {
    launchSpmvPlain(dAt_.view(), yLocal, commBuf_.get(), stream_);
    comm_->allReduceSum(commBuf_.get(), (size_t)F_.n, stream_);
    PDLP_HIP(hipMemcpyAsync(aty, commBuf_.get(), sizeof(double) * F_.n, hipMemcpyDeviceToDevice, stream_));
  }
}

void Solver::sumOverRanks(double* devBuf, int32_t count) {
  if (meshMode_) mesh_->allReduceScalars(devBuf, count, stream_);
  else if (comm_) comm_->allReduceSum(devBuf, (size_t)count, stream_);
}

// Assemble a vector that is distributed by rows or columns on every rank's host
// (devLocal holds [lo,hi) of it).
void Solver::gatherToHost(const double* devLocal, int32_t lo, int32_t hi, bool byRows, std::vector<double>& full) {
  const int32_t len = byRows ? F_.m : F_.n;
  full.assign((size_t)len, 0.0);
  // persistent scratch: a hipFree here would synchronise the whole device, and with several ranks of one
  // process on one device (the folded test mode) it would wait for a peer's kernel that waits for us
  DeviceArray<double>& g = gatherBuf_;
  if (g.size() < (size_t)std::max(len, 1)) g.alloc((size_t)std::max(std::max(F_.n, F_.m), 1));
  g.zero(stream_);
  PDLP_HIP(hipMemcpyAsync(g.get() + lo, devLocal, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToDevice, stream_));
  if (meshMode_) mesh_->allGather(g.get(), byRows, stream_);
  else if (comm_) comm_->allReduceSum(g.get(), (size_t)len, stream_);
  g.download(full.data(), (size_t)len, stream_);
  PDLP_HIP(hipStreamSynchronize(stream_));
  if (meshMode_) mesh_->checkError(stream_);  // a timed-out exchange must not return a half-gathered vector
}

// Sum per-block partials on the device, bring the scalar to the host; row
// quantities are additionally summed over the row-block owners.
double Solver::reduceScalar(const double* partials, int32_t nBlocks, bool overRanks) {
  launchFinalReduce(partials, nBlocks, nBlocks, 1, statOut_.get() + kStatTotal, stream_);
  if (overRanks && sharded_) sumOverRanks(statOut_.get() + kStatTotal, 1);
  PDLP_HIP(hipMemcpyAsync(hostStats_ + kStatTotal, statOut_.get() + kStatTotal, sizeof(double),
                          hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  return hostStats_[kStatTotal];
}

// ---- initialisation ----------------------------------------------------------
// PDHG_Init_Step_Sizes, cupdlp_step.c:312-375.  The norms of the scaled c and b
// are taken on the host with the reference's left-to-right sums.
void Solver::initStepSizes() {
  DevState& s = *hostState_;
  const double a = sumCost2_, b = sumRhs2_;
  s.beta = (std::fmin(a, b) > 1e-6) ? a / b : 1.0;
  if (adaptive_) {
    s.primalStep = (1.0 / F_.matNormInf) / std::sqrt(s.beta);
    s.dualStep = s.primalStep * s.beta;
  } else {
    // PDHG_Power_Method, cupdlp_step.c:71-145: 20 iterations on A A'
    const int32_t n = F_.n;
    double lambda = 0.0;
    launchFill(tmpMl(), 1.0, mLoc_, stream_);
    const int32_t nbM = vecBlocks(std::max(mLoc_, 1)), nbN = vecBlocks(n);
    for (int it = 0; it < 20; ++it) {
      deviceATy(tmpMl(), aty_[0].get());
      deviceAx(aty_[0].get(), ax_[0].get());
      launchDot(ax_[0].get(), ax_[0].get(), mLoc_, partDX_.get(), nbM, stream_);
      const double qn = std::sqrt(reduceScalar(partDX_.get(), nbM, true));
      launchScaleCopy(tmpMl(), ax_[0].get(), 1.0 / qn, mLoc_, stream_);
      deviceATy(tmpMl(), aty_[0].get());
      launchDot(aty_[0].get(), aty_[0].get(), n, partDX_.get(), nbN, stream_);
      lambda = reduceScalar(partDX_.get(), nbN, false);
    }
    s.primalStep = 0.8 / std::sqrt(lambda);
    s.dualStep = s.primalStep;
    s.primalStep /= std::sqrt(s.beta);
    s.dualStep *= std::sqrt(s.beta);
    log(2, "Initial step sizes from power method lambda = %g: primal = %g; dual = %g\n", lambda, s.primalStep,
        s.dualStep);
  }
  iLastRestartIter_ = 0;
  s.sumPrimalStep = 0.0;
  s.sumDualStep = 0.0;
}

// PDHG_Init_Variables, cupdlp_solver.c:531-591
void Solver::initVariables() {
  const int32_t n = F_.n;
  if (hasStart_) {
    x_[0].upload(startX_.data(), n, stream_);
    y_[0].upload(startY_.data() + (r0_ - yOff_), yLen_, stream_);
  } else {
    x_[0].zero(stream_);
    y_[0].zero(stream_);
  }
  launchProjectBounds(x_[0].get(), lower_.get(), upper_.get(), n, stream_);
  deviceAx(x_[0].get(), ax_[0].get());
  deviceATy(yl(0), aty_[0].get());
  if (hasQoff_) launchSpmvPlain(dQ_.view(), x_[0].get(), nx_[0].get(), stream_);
  xSum_.zero(stream_); ySum_.zero(stream_); xAvg_.zero(stream_); yAvg_.zero(stream_);
  launchProjectBounds(xSum_.get(), lower_.get(), upper_.get(), n, stream_);  // :583-584
  launchProjectBounds(xAvg_.get(), lower_.get(), upper_.get(), n, stream_);
  xLast_.zero(stream_); yLast_.zero(stream_);
}

void Solver::reset() {
  DevState& s = *hostState_;
  memset(&s, 0, sizeof(s));
  if (fused_ || persistent_) { gridBar_.zero(stream_); smallSeq_ = 0; }  // arrival epochs follow the trial counter, which starts again
  if (checkSmall_) { checkBar_.zero(stream_); checkSmallSeq_ = 0; }
  s.adaptive = adaptive_ ? 1 : 0;
  initStepSizes();
  initVariables();
  s.eta = std::sqrt(s.primalStep * s.dualStep);
  if (adaptive_) {
    s.tau = s.eta / std::sqrt(s.beta);
    s.sigma = s.eta * std::sqrt(s.beta);
  } else {
    s.tau = s.primalStep;
    s.sigma = s.dualStep;
  }
  s.haltIter = INT_MAX;
  pushState();
  cur_ = Residuals();
  avg_ = Residuals();
  pFeasLR_ = dFeasLR_ = gapLR_ = pFeasLC_ = dFeasLC_ = gapLC_ = 0.0;
  nRestarts_ = 0;
  nChecks_ = 0;
  termCode_ = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT;
  termIterate_ = 0;
}

// ---- the hot loop --------------------------------------------------------------
// One trial step of cupdlp_step.c:241-257 (+ the decision, on the device).
void Solver::profCollect(int32_t realTrials) {
  // events of trials queued after the device halted time no-op kernels: only the first realTrials count
  for (int32_t t = 0; t < profTrialsQueued_ && t < realTrials; ++t) {
    float a = 0.f, b = 0.f, z = 0.f;
    PDLP_HIP(hipEventElapsedTime(&a, profEvents_[4 * t], profEvents_[4 * t + 1]));
    PDLP_HIP(hipEventElapsedTime(&b, profEvents_[4 * t + 1], profEvents_[4 * t + 2]));
    PDLP_HIP(hipEventElapsedTime(&z, profEvents_[4 * t + 2], profEvents_[4 * t + 3]));  // empty interval
    profAxMs_ += a - z;  // event-to-event time minus the cost of the event pair itself
    profAtyMs_ += b - z;
    ++profLaunches_;
  }
  profTrialsQueued_ = 0;
}

void Solver::enqueueTrial() {
  if (meshMode_) {
    // direct-exchange sequence (pdlp_mesh.hpp): X all-gather, P reduce-scatter, S scalars
    const MeshArgs& mv = mesh_->args();
    double* buf = commBuf_.get();
    const int32_t nb = meshGrid(std::max(nLoc_, 1));  // consumer grid: ~4 slice elements per thread
    const bool oneLaunch = mv.fusedWait == 2;  // every rank on a GPU of its own: an exchange is one kernel (five launches per trial)
    if (oneLaunch) {
      double* xFull[2] = {x_[0].get(), x_[1].get()};
      launchMeshPrimalX(vecsCol_, xFull, F_.n, dst(), mv, stream_);
    } else {
      launchMeshPrimalStep(vecsCol_, dst(), mv, stream_);
      launchMeshWaitCopyX(vecs_, dst(), mv, stream_);
    }
    launchSpmvAxDual(dA_.view(), vecs_, dst(), partDY_.get(), stream_);
    if (colblock_) {
      // Y all-gather of the dual step's rows, then A'y+ on the own COLUMNS from the column block: every column is
      // summed over all rows in the single-GPU order; no n-length partial is written, pushed and re-reduced
      double* yFull[2] = {y_[0].get(), y_[1].get()};
      if (oneLaunch) {
        launchMeshY(yFull, F_.m, dst(), mv, stream_);
      } else {
        launchMeshPushY(vecs_, yFull, dst(), mv, stream_);
        launchMeshWaitCopyY(yFull, F_.m, dst(), mv, stream_);
      }
      launchSpmvAtyInteract(dAt_.view(), vecsAty_, dst(), partDX_.get(), partInter_.get(), stream_);
      launchMeshDecide(dst(), mv, partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(), dAt_.nPartials(), stream_);
      return;
    }
    launchSpmvAtyPartial(dAt_.view(), vecs_, dst(), buf, stream_);
    launchMeshPushPartial(buf, F_.n, dst(), mv, stream_);
    launchMeshReduceInteract(vecsCol_, dst(), mv, buf, partDX_.get(), partInter_.get(), nb, stream_);
    launchMeshDecide(dst(), mv, partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(), nb, stream_);
    return;
  }
  if (persistent_) {
    launchSmallTrials(dA_.view(), dAt_.view(), vecs_, dst(), partDY_.get(), partDX_.get(), partInter_.get(), gridBar_.get(), smallGrid_, 1,
                      smallMode(), stream_, sw_.barrierTimeoutMs, sw_.fault == 1 && smallLaunches_ == 0, smallLaunches_ == 0, ++smallSeq_, primalInA_);
    ++smallLaunches_;
    return;
  }
  if (!sharded_ && fused_) {
    // single GPU, 2 launches per trial: A x+ (+ dual step), then A' y+ (+ movement / interaction partials, grid
    // barrier, decision, the NEXT trial's primal step); after a host push of the state, a stand-alone primal step first
    if (needPrimal_) {
      const DevState* in = dst();
      stPar_ ^= 1;
      launchDecidePrimal(vecs_, in, dst(), partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(), dAt_.nPartials(), stream_);
      needPrimal_ = false;
    }
    const DevState* st = dst();
    hipEvent_t* ev = nullptr;
    if (profile_) {
      while ((int32_t)profEvents_.size() < 4 * (profTrialsQueued_ + 1)) {
        hipEvent_t e;
        PDLP_HIP(hipEventCreate(&e));
        profEvents_.push_back(e);
      }
      ev = &profEvents_[4 * profTrialsQueued_++];
      PDLP_HIP(hipEventRecord(ev[0], stream_));
    }
    launchSpmvAxDual(dA_.view(), vecs_, st, partDY_.get(), stream_);
    if (ev) PDLP_HIP(hipEventRecord(ev[1], stream_));
    stPar_ ^= 1;
    launchSpmvAtyFusedPrimal(dAt_.view(), vecs_, st, dst(), partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(),
                             gridBar_.get(), stream_, sw_.barrierTimeoutMs, sw_.fault == 2 ? 12 : 0);
    if (ev) {
      PDLP_HIP(hipEventRecord(ev[2], stream_));
      PDLP_HIP(hipEventRecord(ev[3], stream_));
    }
    return;
  }
  if (!sharded_) {
    // single GPU: 3 launches per trial — [decision of the previous trial + primal step], A x+ (+ dual step),
    // A' y+ (+ movement / interaction partials); the state alternates between the two slots
    const DevState* stIn = dst();
    stPar_ ^= 1;
    DevState* st = dst();
    launchDecidePrimal(vecs_, stIn, st, partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(),
                       dAt_.nPartials(), stream_, hasQoff_ ? partQ_.get() : nullptr, hasQoff_ ? dQ_.nPartials() : 0);
    hipEvent_t* ev = nullptr;
    if (profile_) {
      while ((int32_t)profEvents_.size() < 4 * (profTrialsQueued_ + 1)) {
        hipEvent_t e;
        PDLP_HIP(hipEventCreate(&e));
        profEvents_.push_back(e);
      }
      ev = &profEvents_[4 * profTrialsQueued_++];
      PDLP_HIP(hipEventRecord(ev[0], stream_));
    }
    launchSpmvAxDual(dA_.view(), vecs_, st, partDY_.get(), stream_);
    if (ev) PDLP_HIP(hipEventRecord(ev[1], stream_));
    launchSpmvAtyInteract(dAt_.view(), vecs_, st, partDX_.get(), partInter_.get(), stream_);
    if (ev) {
      PDLP_HIP(hipEventRecord(ev[2], stream_));
      PDLP_HIP(hipEventRecord(ev[3], stream_));
    }
    if (hasQoff_) launchSpmvQxInteract(dQ_.view(), vecs_, st, partQ_.get(), stream_);  // N x+ and dx . N dx: the fourth launch of a QP trial
    return;
  }
  // row-block sharded, RCCL exchange: A_g' y_g partials are summed over the ranks together
  // with the local sum (dy)^2 in one all-reduce of n+1 doubles
  launchPrimalStep(vecs_, dst(), stream_);
  launchSpmvAxDual(dA_.view(), vecs_, dst(), partDY_.get(), stream_);
  double* buf = commBuf_.get();
  launchSpmvAtyPartial(dAt_.view(), vecs_, dst(), buf, stream_);
  launchReduceTo(partDY_.get(), dA_.nPartials(), buf + F_.n, dst(), stream_);
  comm_->allReduceSum(buf, (size_t)F_.n + 1, stream_);
  const int32_t nb = vecBlocks(F_.n);
  launchInteract(vecs_, dst(), buf, partDX_.get(), partInter_.get(), nb, stream_);
  launchDecide(dst(), nullptr, 0, partDX_.get(), partInter_.get(), nb, buf + F_.n, stream_);
}

// The batch of kGraphTrials (even) trials as one hipGraph.  Capturing enqueues nothing; the state-slot
// parity the capture started with is restored afterwards and remembered for the launches.
void Solver::captureGraph() {
  if (graphExec_) return;
  static_assert(kGraphTrials % 2 == 0, "the graph must leave the state-slot parity unchanged");
  hipGraph_t graph = nullptr;
  graphPar_ = stPar_;
  const bool savedNeed = needPrimal_;
  needPrimal_ = false;  // (the stand-alone primal step after a host push is never part of the graph)
  PDLP_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < kGraphTrials; ++i) enqueueTrial();
  PDLP_HIP(hipStreamEndCapture(stream_, &graph));
  needPrimal_ = savedNeed;
  PDLP_HIP(hipGraphInstantiate(&graphExec_, graph, nullptr, nullptr, 0));
  (void)hipGraphDestroy(graph);
  graphTrials_ = kGraphTrials;
  stPar_ = graphPar_;
}

// `todo` trials towards the next halt.  Trials queued behind the halt are early-exit kernels, so every form carries a
// few spare ones for rejected trials: the persistent launch runs up to todo + 8, the captured graph holds 42 for a
// period of 40; single launches (the first ten iterations, profile mode) carry none — the caller looks at the state.
void Solver::enqueueBatch(int32_t todo) {
  if (persistent_ && !profile_) {  // the whole stretch to the next check (plus spare trials for rejections) in one launch
    launchSmallTrials(dA_.view(), dAt_.view(), vecs_, dst(), partDY_.get(), partDX_.get(), partInter_.get(), gridBar_.get(),
                      smallGrid_, todo + 8, smallMode(), stream_, sw_.barrierTimeoutMs, sw_.fault == 1 && smallLaunches_ == 0,
                      smallLaunches_ == 0, ++smallSeq_, primalInA_);
    ++smallLaunches_;
    return;
  }
  if (useGraph_ && !persistent_ && !profile_ && (!sharded_ || meshMode_) && todo >= kGraphMinTodo) {
    if (!graphExec_) captureGraph();
    while (todo >= kGraphMinTodo) {
      if (stPar_ != graphPar_ || (fused_ && needPrimal_)) {  // the graph starts from the state slot it was captured with, after a primal step
        enqueueTrial();
        --todo;
        continue;
      }
      PDLP_HIP(hipGraphLaunch(graphExec_, stream_));  // an even number of trials: the slot parity is unchanged
      todo = todo > graphTrials_ ? todo - graphTrials_ : 0;
    }
  }
  for (int i = 0; i < todo; ++i) enqueueTrial();
}

void Solver::runUntilHalt() {
  for (;;) {
    int64_t remaining = (int64_t)hostState_->haltIter - hostState_->nIter;
    if (remaining < 1) remaining = 1;
    if (remaining > 4 * kCheckInterval) remaining = 4 * kCheckInterval;
    int32_t todo = (int32_t)remaining;
    const int32_t trialsBefore = hostState_->nTrials;
    BarrierRound round{*this, beginBarrierRound()};
    std::unique_lock<std::mutex>& gate = round.gate;
    enqueueBatch(todo);
    const int32_t iterBefore = hostState_->nIter;
    endBarrierRound(gate);
    syncState();
    if (profile_) profCollect(hostState_->nTrials - trialsBefore);
    if (hostState_->halted) return;
    // The reference's step-size search is a `while (!accepted)` loop: with NaN / Inf in the data it never ends.
    // Here every stop checks that the search still makes progress.
    if (hostState_->nIter == iterBefore && hostState_->nTrials - trialsBefore > 0) {
      if (++stalledRounds_ >= 50)
        throw std::runtime_error("pdlp_mi355x: the adaptive step-size search does not terminate (no trial step accepted in " +
                                 std::to_string(hostState_->nTrials - stalledSince_) + " trials: NaN or Inf in the problem data?)");
    } else {
      stalledRounds_ = 0;
      stalledSince_ = hostState_->nTrials;
    }
    if (timeIsUp()) return;
    // a stretch without check iterations (check_interval beyond the host's 160-iteration rounds): keep the tabulated
    // powers of the step rule ahead of the trial counter (the fused / persistent kernels take no other)
    if (hostState_->powRed && hostState_->nTrials + 4096 >= hostState_->powBase + hostState_->powCount) pushState();
  }
}

int32_t Solver::nextCheckIter(int32_t it) const {
  const int32_t interval = opt_.check_interval > 0 ? opt_.check_interval : kCheckInterval;
  int64_t next;
  if (it + 1 < 10) next = it + 1;
  else next = ((int64_t)it / interval + 1) * interval;
  const int64_t last = (int64_t)opt_.iter_limit - 1;
  if (last > it && last < next) next = last;
  if (next > INT_MAX) next = INT_MAX;
  return (int32_t)next;
}

// ---- check iteration -----------------------------------------------------------
// PDHG_Compute_Average_Iterate, cupdlp_step.c:377-420
void Solver::computeAverage() {
  const double ps = hostState_->sumPrimalStep > 0.0 ? 1.0 / hostState_->sumPrimalStep : 1.0;
  const double ds = hostState_->sumDualStep > 0.0 ? 1.0 / hostState_->sumDualStep : 1.0;
  // pending average update + the averages of the own columns / rows in one pass; the pending weights come from the
  // host's copy of the state (every caller has synchronised it) and are cleared here: the next pushState carries that
  launchFlushScale(vecsCol_, CheckGate(), hostState_->cur, hostState_->avgW, hostState_->avgWx, ps, ds, xAvg_.get() + c0_, yAvgl(), stream_);
  hostState_->avgW = 0.0;
  hostState_->avgWx = 0.0;
  launchClearAvgW(dst(), stream_);  // the device's record too: a caller that stops here (terminate, stage) leaves host and device agreeing
  if (meshMode_) mesh_->allGather(xAvg_.get(), false, stream_);
  deviceAx(xAvg_.get(), axAvg_.get());
  deviceATy(yAvgl(), atyAvg_.get());
  if (hasQoff_) launchSpmvPlain(dQ_.view(), xAvg_.get(), nxAvg_.get(), stream_);
}

// PDHG_Compute_Residuals + PDHG_Compute_Infeas_Residuals (cupdlp_solver.c:433-529)
// for the current and the average iterate: four fused passes, one D2H of 28 doubles.
void Solver::computeResiduals() {
  const int c = hostState_->cur;
  const int32_t nbM = vecBlocks(std::max(mLoc_, 1)), nbN = vecBlocks(std::max(nLoc_, 1));
  double* part = statPart_.get();
  const size_t co = (size_t)c0_;  // column statistics run on the own column slice (everything unless mesh-sharded)
  const int sc = F_.scaled ? 1 : 0;
  static_assert(kStatRowAvg == kStatRowCur + kRowStats && kStatColAvg == kStatColCur + kColStats, "current first, then average");
  // both iterates per pass (shared vectors read once), one reduction launch for all 30 quantities
  launchRowStats2(vecs_, CheckGate(), c, axAvg_.get(), yAvgl(), rowScale_.get(), sc, part + (size_t)kStatRowCur * statStride_, statStride_,
                  nbM, stream_);
  launchColStats2(vecsCol_, CheckGate(), c, atyAvg_.get() + co, xAvg_.get() + co, colScale_.get() + co, hasQoff_ ? nxAvg_.get() : nullptr,
                  sc, slackPos_.get() + co, slackNeg_.get() + co, slackPosAvg_.get() + co, slackNegAvg_.get() + co,
                  part + (size_t)kStatColCur * statStride_, statStride_, nbN, stream_);
  launchFinalReduce2(part, statStride_, 2 * kRowStats, nbM, 2 * kColStats, nbN, statOut_.get(), CheckGate(), stream_);
  if (sharded_) sumOverRanks(statOut_.get(), meshMode_ ? kStatTotal : 2 * kRowStats);
  PDLP_HIP(hipMemcpyAsync(hostStats_, statOut_.get(), sizeof(double) * kStatTotal, hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  if (meshMode_) {
    mesh_->checkError(stream_);
    mesh_->verifyReplicated(x_[c].get(), F_.n, stream_);  // every rank gathers from ITS copy of x
    if (colblock_) mesh_->verifyReplicated(y_[c].get(), F_.m, stream_);  // ... and, in this layout, from its copy of y
  }

  auto fill = [&](Residuals& r, const double* rs, const double* cs) {
    // QP (cs[10] = 1/2 x'Qx, zero for an LP): primal c'x + 1/2 x'Qx, dual b'y + l's+ - u's- - 1/2 x'Qx
    r.pObj = (qdiag_.size() ? cs[0] + cs[10] : cs[0]) * F_.sense + F_.offset;
    r.pFeas = std::sqrt(rs[0]);
    r.dObj = (qdiag_.size() ? ((rs[1] + cs[1]) - cs[2]) - cs[10] : (rs[1] + cs[1] - cs[2])) * F_.sense + F_.offset;
    r.dFeas = std::sqrt(cs[3]);
    r.gap = r.pObj - r.dObj;
    r.relGap = std::fabs(r.pObj - r.dObj) / (1.0 + std::fabs(r.pObj) + std::fabs(r.dObj));
    double dScale = std::sqrt(rs[2] + cs[4] + cs[5]);  // ||(y, s+, s-)||, cupdlp_solver.c:230-237
    if (dScale < 1e-8) dScale = 1.0;
    r.pInfObj = (r.dObj - F_.offset) / F_.sense / dScale;
    r.pInfRes = std::sqrt(cs[6]) / dScale;
    double pScale = std::sqrt(cs[7]);  // ||x||, :328-332
    if (pScale < 1e-8) pScale = 1.0;
    r.dInfObj = (r.pObj - F_.offset) / F_.sense / pScale;
    r.dInfRes = std::sqrt(rs[3] + cs[8] + cs[9]) / pScale;
  };
  fill(cur_, hostStats_ + kStatRowCur, hostStats_ + kStatColCur);
  fill(avg_, hostStats_ + kStatRowAvg, hostStats_ + kStatColAvg);
}

bool Solver::checkTermination(const Residuals& r) const {  // cupdlp_solver.c:797-841
  return (r.pFeas < opt_.primal_tol * (1.0 + F_.normRhs)) && (r.dFeas < opt_.dual_tol * (1.0 + F_.normCost)) &&
         (r.relGap < opt_.gap_tol);
}

bool Solver::checkInfeasibility() {  // cupdlp_solver.c:710-795
  bool t = false;
  auto primalInf = [&](const Residuals& r) { return r.pInfObj > 0.0 && r.pInfRes < feasTol_ * r.pInfObj; };
  auto dualInf = [&](const Residuals& r) { return r.dInfObj < 0.0 && r.dInfRes < -feasTol_ * r.dInfObj; };
  if (primalInf(cur_)) t = true;
  if (dualInf(cur_)) t = true;
  if (primalInf(avg_)) t = true;
  if (dualInf(avg_)) t = true;
  return t;
}

// PDHG_Restart_Iterate_GPU (cupdlp_proj.c:88-148) with PDHG_Check_Restart_GPU
// (cupdlp_restart.c:3-124) and PDHG_Compute_Step_Size_Ratio (cupdlp_step.c:147-176).
void Solver::restartIterate() {
  if (!restartOn_) return;
  DevState& s = *hostState_;
  auto score = [](double beta, double p, double d, double g) { return std::sqrt(beta * p * p + d * d / beta + g * g); };
  const int32_t it = s.nIter;
  if (it == iLastRestartIter_) {
    pFeasLR_ = cur_.pFeas; dFeasLR_ = cur_.dFeas; gapLR_ = cur_.gap;
    pFeasLC_ = cur_.pFeas; dFeasLC_ = cur_.dFeas; gapLC_ = cur_.gap;
    return;
  }
  const double muCur = score(s.beta, cur_.pFeas, cur_.dFeas, cur_.gap);
  const double muAvg = score(s.beta, avg_.pFeas, avg_.dFeas, avg_.gap);
  const bool toCurrent = muCur < muAvg;
  const double muCand = toCurrent ? muCur : muAvg;
  bool restart = true;
  if ((it - iLastRestartIter_) >= 0.36 * it) {
    // artificial restart
  } else {
    const double muLR = score(s.beta, pFeasLR_, dFeasLR_, gapLR_);
    if (muCand < 0.2 * muLR) {
      // sufficient decay
    } else {
      const double muLC = score(s.beta, pFeasLC_, dFeasLC_, gapLC_);
      if (!(muCand < 0.8 * muLR && muCand > muLC)) restart = false;  // necessary decay
    }
  }
  const Residuals& cand = toCurrent ? cur_ : avg_;
  pFeasLC_ = cand.pFeas; dFeasLC_ = cand.dFeas; gapLC_ = cand.gap;
  if (!restart) return;

  const int c = s.cur;
  const int32_t n = F_.n;
  s.sumPrimalStep = 0.0;
  s.sumDualStep = 0.0;
  xSum_.zero(stream_);
  ySum_.zero(stream_);
  if (!toCurrent) {
    pFeasLR_ = avg_.pFeas; dFeasLR_ = avg_.dFeas; gapLR_ = avg_.gap;
    PDLP_HIP(hipMemcpyAsync(x_[c].get(), xAvg_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
    PDLP_HIP(hipMemcpyAsync(y_[c].get(), yAvg_.get(), sizeof(double) * yLen_, hipMemcpyDeviceToDevice, stream_));  // (colblock: the full, all-gathered average)
    PDLP_HIP(hipMemcpyAsync(ax_[c].get(), axAvg_.get(), sizeof(double) * mLoc_, hipMemcpyDeviceToDevice, stream_));
    PDLP_HIP(hipMemcpyAsync(aty_[c].get(), atyAvg_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
    if (hasQoff_) PDLP_HIP(hipMemcpyAsync(nx_[c].get(), nxAvg_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
  } else {
    pFeasLR_ = cur_.pFeas; dFeasLR_ = cur_.dFeas; gapLR_ = cur_.gap;
  }
  // primal weight update
  const double mean = std::sqrt(s.primalStep * s.dualStep);
  const int32_t nbN = vecBlocks(std::max(nLoc_, 1)), nbM = vecBlocks(std::max(mLoc_, 1));
  launchDiffNorm2(x_[c].get() + c0_, xLast_.get() + c0_, nLoc_, partDX_.get(), nbN, stream_);
  const double dP = std::sqrt(reduceScalar(partDX_.get(), nbN, meshMode_));
  launchDiffNorm2(yl(c), yLast_.get(), mLoc_, partDX_.get(), nbM, stream_);
  const double dD = std::sqrt(reduceScalar(partDX_.get(), nbM, true));
  if (std::fmin(dP, dD) > 1e-10) {
    // (pdlp_detmath.h: the same bits as the device-driven restart, pdlp_check.hip k_restart_finish)
    const double lg = 0.5 * pdlp_det_log(dD / dP) + 0.5 * pdlp_det_log(std::sqrt(s.beta));
    s.beta = pdlp_det_exp(lg) * pdlp_det_exp(lg);
  }
  s.primalStep = mean / std::sqrt(s.beta);
  s.dualStep = s.primalStep * s.beta;
  s.eta = std::sqrt(s.primalStep * s.dualStep);
  if (adaptive_) {
    s.tau = s.eta / std::sqrt(s.beta);
    s.sigma = s.eta * std::sqrt(s.beta);
  } else {
    s.tau = s.primalStep;
    s.sigma = s.dualStep;
  }
  PDLP_HIP(hipMemcpyAsync(xLast_.get(), x_[c].get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
  PDLP_HIP(hipMemcpyAsync(yLast_.get(), yl(c), sizeof(double) * mLoc_, hipMemcpyDeviceToDevice, stream_));
  iLastRestartIter_ = it;
  ++nRestarts_;
  log(2, "Restart at iter %d to %s: beta = %g\n", it, toCurrent ? "current" : "average", s.beta);
  // The reference recomputes the residuals here (cupdlp_proj.c:145); the values
  // are overwritten by the next check before anything reads them, so we don't.
}

// One line of the iteration log (the reference prints every 100th check, the last iteration and at the time limit).
void Solver::logCheckLine(int32_t it, const Residuals& cur, const Residuals& avg, double t, int& logSinceHeader) const {
  if (opt_.log_level <= 0 || rank_ != 0) return;
  if (logSinceHeader >= 50) {
    logLine(opt_, 1, "%9s  %15s  %15s   %8s  %10s  %8s %7s\n", "Iter", "Primal.Obj", "Dual.Obj", "Gap", "Primal.Inf",
           "Dual.Inf", "Time");
    logSinceHeader = 0;
  }
  const Residuals& r = it == 0 ? cur : avg;
  logLine(opt_, 1, "%9d  %+15.8e  %+15.8e  %+8.2e  %10.2e  %8.2e %6.2fs [%c]\n", it, r.pObj, r.dObj, r.relGap,
         r.pFeas / (1.0 + F_.normRhs), r.dFeas / (1.0 + F_.normCost), t, it == 0 ? 'L' : 'A');
  ++logSinceHeader;
}

// ---- check iterations on the device (pdlp_kernels.hpp CheckCtl, pdlp_check.hip) -------------------------------------
static_assert(sizeof(ResidualsDev) == sizeof(Residuals), "Residuals is mirrored on the device");

// parameters of the loop + the host's mirrors of the restart bookkeeping -> device
void Solver::uploadCtl(bool terminate, int64_t iterLim) {
  CheckCtl& c = *hostCtl_;
  memset(&c, 0, sizeof(c));
  c.primalTolAbs = opt_.primal_tol * (1.0 + F_.normRhs);
  c.dualTolAbs = opt_.dual_tol * (1.0 + F_.normCost);
  c.gapTol = opt_.gap_tol;
  c.feasTol = feasTol_;
  c.sense = F_.sense;
  c.offset = F_.offset;
  c.terminate = terminate ? 1 : 0;
  c.restartOn = restartOn_ ? 1 : 0;
  c.interval = opt_.check_interval > 0 ? opt_.check_interval : kCheckInterval;
  c.iterLimit = (int32_t)std::min<int64_t>(iterLim, INT_MAX);
  c.optIterLimit = opt_.iter_limit;
  c.qp = qdiag_.size() ? 1 : 0;
  c.adaptive = adaptive_ ? 1 : 0;
  memcpy(&c.cur, &cur_, sizeof(Residuals));
  memcpy(&c.avg, &avg_, sizeof(Residuals));
  c.pFeasLR = pFeasLR_; c.dFeasLR = dFeasLR_; c.gapLR = gapLR_;
  c.pFeasLC = pFeasLC_; c.dFeasLC = dFeasLC_; c.gapLC = gapLC_;
  c.iLastRestartIter = iLastRestartIter_; c.nRestarts = nRestarts_; c.nChecks = nChecks_;
  c.termCode = termCode_; c.termIterate = termIterate_;
  c.lastCheckIter = -1;
  PDLP_HIP(hipMemcpyAsync(dCtl_.get(), hostCtl_, sizeof(CheckCtl), hipMemcpyHostToDevice, stream_));
}

// device -> the host's mirrors (the stream is idle)
void Solver::downloadCtl() {
  PDLP_HIP(hipMemcpyAsync(hostCtl_, dCtl_.get(), sizeof(CheckCtl), hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
  const CheckCtl& c = *hostCtl_;
  memcpy(&cur_, &c.cur, sizeof(Residuals));
  memcpy(&avg_, &c.avg, sizeof(Residuals));
  pFeasLR_ = c.pFeasLR; dFeasLR_ = c.dFeasLR; gapLR_ = c.gapLR;
  pFeasLC_ = c.pFeasLC; dFeasLC_ = c.dFeasLC; gapLC_ = c.gapLC;
  iLastRestartIter_ = c.iLastRestartIter; nRestarts_ = c.nRestarts; nChecks_ = c.nChecks;
  if (c.terminated) { termCode_ = c.termCode; termIterate_ = c.termIterate; }
}

// One check iteration behind whatever is queued: flush + averages, A xAvg, A'yAvg, the two statistics passes, their
// reduction, then the scalar logic and the restart — every kernel a no-op unless the device has halted at a scheduled
// iteration (checkDue).  10 launches, no host synchronisation.
void Solver::enqueueCheckDevice() {
  DevState* st = dst();
  const CheckGate g{st, dCtl_.get()};
  if (persistent_ && checkSmall_) {
    // the small-LP form: the whole check is ONE launch of the trial loop's workgroups (pdlp_check.hip k_check_small)
    CheckRecord* rec = hostRing_ + (checkSeq_ % kRingSlots);
    rec->ran = 0;
    ++checkSeq_;
    const RestartVecs rv{xAvg_.get(), yAvg_.get(), axAvg_.get(), atyAvg_.get(), nullptr, xLast_.get(), yLast_.get()};
    launchCheckSmall(dA_.view(), dAt_.view(), vecs_, st, dCtl_.get(), rec, rv, rowScale_.get(), colScale_.get(), F_.scaled ? 1 : 0,
                     slackPos_.get(), slackNeg_.get(), slackPosAvg_.get(), slackNegAvg_.get(), statPart_.get(), statStride_, statOut_.get(),
                     partDX_.get(), partRestartY_.get(), checkBar_.get(), smallGrid_, ++checkSmallSeq_, sw_.barrierTimeoutMs, stream_);
    needPrimal_ = true;
    return;
  }
  if (sharded_) {
    // Row-block sharded (mesh exchange, two-all-gathers layout): the kernels of the host-driven check (computeAverage,
    // computeResiduals, restartIterate) on this rank's rows and columns, gated like the single-GPU ones, with the SAME
    // collectives in between — enqueued, not waited for.  The collectives themselves are never gated (their epochs are
    // counted by the host on every rank alike): a check that is not due exchanges buffers nobody reads.  Every rank
    // holds the same all-reduced statistics and norms, so every rank takes the same decisions (bit-identical state, as
    // in the hot loop).
    const int32_t nbM = vecBlocks(std::max(mLoc_, 1)), nbN = vecBlocks(std::max(nLoc_, 1));
    const size_t co = (size_t)c0_;
    const int sc = F_.scaled ? 1 : 0;
    double* part = statPart_.get();
    launchFlushScale(vecsCol_, g, 0, 0.0, 0.0, 1.0, 1.0, xAvg_.get() + co, yAvgl(), stream_);
    mesh_->allGather(xAvg_.get(), false, stream_);
    launchSpmvPlain(dA_.view(), xAvg_.get(), axAvg_.get(), stream_, g);
    mesh_->allGather(yAvg_.get(), true, stream_);  // (yAvgl() = this rank's rows inside the full-length vector)
    launchSpmvPlain(dAt_.view(), yAvg_.get(), atyAvg_.get() + co, stream_, g);
    mesh_->allGather(atyAvg_.get(), false, stream_);
    launchRowStats2(vecs_, g, 0, axAvg_.get(), yAvgl(), rowScale_.get(), sc, part + (size_t)kStatRowCur * statStride_, statStride_, nbM,
                    stream_);
    launchColStats2(vecsCol_, g, 0, atyAvg_.get() + co, xAvg_.get() + co, colScale_.get() + co, nullptr, sc, slackPos_.get() + co,
                    slackNeg_.get() + co, slackPosAvg_.get() + co, slackNegAvg_.get() + co, part + (size_t)kStatColCur * statStride_,
                    statStride_, nbN, stream_);
    mesh_->reduce2AllReduce(part, statStride_, 2 * kRowStats, nbM, 2 * kColStats, nbN, statOut_.get(), g, stream_);
    CheckRecord* rec = hostRing_ + (checkSeq_ % kRingSlots);
    rec->ran = 0;
    ++checkSeq_;
    launchCheckDecide(st, dCtl_.get(), statOut_.get(), rec, stream_);
    double* xs[2] = {x_[0].get(), x_[1].get()};
    double* as[2] = {aty_[0].get(), aty_[1].get()};
    double* ys[2] = {y_[0].get(), y_[1].get()};
    launchRestartCopyFull(st, dCtl_.get(), xs, as, ys, xAvg_.get(), atyAvg_.get(), yAvg_.get(), F_.n, yLen_, stream_);
    const RestartVecs rv{xAvg_.get() + co, yAvgl(), axAvg_.get(), atyAvg_.get() + co, nullptr, xLast_.get() + co, yLast_.get()};
    launchRestartVec(vecsCol_, st, dCtl_.get(), rv, partDX_.get(), nbN, partRestartY_.get(), nbM, stream_);
    // the two norms of the primal-weight update: this rank's partials -> one scalar each (reduceScalar's kernel) -> summed
    // over the ranks in rank order -> k_restart_finish takes them as partial arrays of length one
    double* norms = statOut_.get() + kStatTotal + 2;
    mesh_->normsAllReduce(partDX_.get(), nbN, partRestartY_.get(), nbM, norms, stream_);
    launchRestartFinish(st, dCtl_.get(), norms, 1, norms + 1, 1, rec, stream_);
    return;
  }
  if (!persistent_ && !fused_)  // 3-launch loop: the decision of the last trial may still be pending
    launchDecide(st, partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(), dAt_.nPartials(), nullptr, stream_, true,
                 hasQoff_ ? partQ_.get() : nullptr, hasQoff_ ? dQ_.nPartials() : 0);
  launchFlushScale(vecs_, g, 0, 0.0, 0.0, 1.0, 1.0, xAvg_.get(), yAvg_.get(), stream_);
  launchSpmvPlain(dA_.view(), xAvg_.get(), axAvg_.get(), stream_, g);
  launchSpmvPlain(dAt_.view(), yAvg_.get(), atyAvg_.get(), stream_, g);
  if (hasQoff_) launchSpmvPlain(dQ_.view(), xAvg_.get(), nxAvg_.get(), stream_, g);
  const int32_t nbM = vecBlocks(std::max(mLoc_, 1)), nbN = vecBlocks(std::max(nLoc_, 1));
  double* part = statPart_.get();
  const int sc = F_.scaled ? 1 : 0;
  launchRowStats2(vecs_, g, 0, axAvg_.get(), yAvg_.get(), rowScale_.get(), sc, part + (size_t)kStatRowCur * statStride_, statStride_, nbM,
                  stream_);
  launchColStats2(vecs_, g, 0, atyAvg_.get(), xAvg_.get(), colScale_.get(), hasQoff_ ? nxAvg_.get() : nullptr, sc, slackPos_.get(),
                  slackNeg_.get(), slackPosAvg_.get(), slackNegAvg_.get(), part + (size_t)kStatColCur * statStride_, statStride_, nbN,
                  stream_);
  launchFinalReduce2(part, statStride_, 2 * kRowStats, nbM, 2 * kColStats, nbN, statOut_.get(), g, stream_);
  CheckRecord* rec = hostRing_ + (checkSeq_ % kRingSlots);
  rec->ran = 0;
  ++checkSeq_;
  launchCheckDecide(st, dCtl_.get(), statOut_.get(), rec, stream_);
  const RestartVecs rv{xAvg_.get(), yAvg_.get(), axAvg_.get(), atyAvg_.get(), hasQoff_ ? nxAvg_.get() : nullptr, xLast_.get(), yLast_.get()};
  launchRestartVec(vecs_, st, dCtl_.get(), rv, partDX_.get(), nbN, partRestartY_.get(), nbM, stream_);
  launchRestartFinish(st, dCtl_.get(), partDX_.get(), nbN, partRestartY_.get(), nbM, rec, stream_);
  needPrimal_ = true;  // (2-launch trial: the next batch starts with a stand-alone primal step, as after a host push)
}

// The records of the checks that have run since the last look: iteration log, restart notes.
void Solver::processRecords(bool terminate, int64_t iterLim, int& logSinceHeader) {
  for (; checkSeen_ < checkSeq_; ++checkSeen_) {
    const CheckRecord& r = hostRing_[checkSeen_ % kRingSlots];
    if (!r.ran) continue;  // the check was not due (a period that needed another batch; the queue behind a termination)
    if (opt_.log_level > 0 && rank_ == 0) {
      Residuals cur, avg;
      memcpy(&cur, &r.cur, sizeof(cur));
      memcpy(&avg, &r.avg, sizeof(avg));
      if ((r.it % (kCheckInterval * 100) == 0) || (terminate && r.it == iterLim - 1)) logCheckLine(r.it, cur, avg, elapsed(), logSinceHeader);
      if (r.restartKind) log(2, "Restart at iter %d to %s: beta = %g\n", r.it, r.restartKind == 1 ? "current" : "average", r.beta);
    }
  }
}

// PDHG_Solve (cupdlp_solver.c:899-1215) with the check iterations on the device.  The host enqueues units of
// [trial batch to the next scheduled check][check] — `ahead` of them, doubling from 1 up to ~25 ms of queued work —
// and only then synchronises: termination, time limit, the log and the tabulated powers of the step rule are looked
// after once per round instead of once per check.  What the device does between two looks is exactly what the
// host-driven loop (doSolve) does: same kernels for the trials, same statistics, the same scalar arithmetic.
void Solver::doSolveDevice(bool terminate, int32_t target) {
  DevState& s = *hostState_;
  const int64_t iterLim = terminate ? (int64_t)opt_.iter_limit : (int64_t)target;
  const int32_t interval = opt_.check_interval > 0 ? opt_.check_interval : kCheckInterval;
  if (s.nIter >= iterLim) return;
  uploadCtl(terminate, iterLim);
  checkSeen_ = checkSeq_;
  auto onSchedule = [&](int64_t it) { return it < 10 || it % interval == 0 || (terminate && it == (int64_t)opt_.iter_limit - 1); };
  auto haltAfter = [&](int64_t it) {
    int64_t h = nextCheckIter((int32_t)it);
    if (!terminate && h > iterLim) h = iterLim;
    return h;
  };
  // entry: a check is due right here (the device "halts" at the current iteration), or the device runs on to the next one
  if (onSchedule(s.nIter)) {
    s.haltIter = s.nIter;
    s.halted = 1;
  } else {
    s.haltIter = (int32_t)haltAfter(s.nIter);
    s.halted = 0;
  }
  pushState(false);
  int logSinceHeader = 50;
  int32_t ahead = 1, aheadMax = 16;
  bool timeUp = false;
  for (;;) {
    const auto roundBeg = std::chrono::steady_clock::now();
    const int32_t iterBefore = s.nIter, trialsBefore = s.nTrials;
    const int64_t seq0 = checkSeq_;
    BarrierRound round{*this, beginBarrierRound()};
    std::unique_lock<std::mutex>& gate = round.gate;
    int64_t itExp = s.nIter, haltExp = s.haltIter;
    if (s.halted) {  // (entry only: every batch below is followed by its check)
      enqueueCheckDevice();
      haltExp = haltAfter(itExp);
    }
    int32_t units = 0;
    int64_t queuedTrials = 0;  // (the tabulated powers of the step rule reach 4096 trials beyond the last refresh)
    for (int32_t u = 0; u < ahead && queuedTrials < 3000; ++u) {
      int64_t todo = haltExp - itExp;
      if (todo < 1) todo = 1;
      if (todo > 4 * kCheckInterval) todo = 4 * kCheckInterval;
      // single launches carry their own spare trials (a rejected trial must not cost a host round trip)
      const bool batched = persistent_ || (useGraph_ && todo >= kGraphMinTodo);
      enqueueBatch((int32_t)todo + (batched ? 0 : todo >= 8 ? 2 : 1));
      enqueueCheckDevice();
      ++units;
      queuedTrials += todo + 8;
      itExp = std::min(itExp + todo, haltExp);
      if (itExp >= iterLim || (terminate && itExp >= iterLim - 1)) break;  // the target / the check that ends the solve
      if (itExp == haltExp) haltExp = haltAfter(itExp);
    }
    endBarrierRound(gate);
    syncState();
    if (meshMode_) {  // (sharded: the exchange's error flag and the checksum guard of the replicated iterates, once per round)
      mesh_->checkError(stream_);
      mesh_->verifyReplicated(x_[s.cur].get(), F_.n, stream_);
      if (colblock_) mesh_->verifyReplicated(y_[s.cur].get(), F_.m, stream_);
    }
    processRecords(terminate, iterLim, logSinceHeader);
    bool over = false;  // a check of this round has ended the solve (everything queued behind it was a no-op)
    for (int64_t q = seq0; q < checkSeq_; ++q) over = over || (hostRing_[q % kRingSlots].ran && hostRing_[q % kRingSlots].terminated);
    if (over) break;
    if (!terminate && s.nIter >= iterLim) break;
    // The reference's step-size search is a `while (!accepted)` loop: with NaN / Inf in the data it never ends.
    if (s.nIter == iterBefore && s.nTrials - trialsBefore > 0) {
      if (++stalledRounds_ >= 50)
        throw std::runtime_error("pdlp_mi355x: the adaptive step-size search does not terminate (no trial step accepted in " +
                                 std::to_string(s.nTrials - stalledSince_) + " trials: NaN or Inf in the problem data?)");
    } else {
      stalledRounds_ = 0;
      stalledSince_ = s.nTrials;
    }
    if (timeIsUp()) { timeUp = true; break; }
    // tabulated powers of the step rule: kept ahead of the trial counter while the stream is idle
    if (s.powRed && s.nTrials + 4096 >= s.powBase + s.powCount) pushState(false);
    // queue depth: ~25 ms of work, at most 16 units
    const double roundMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - roundBeg).count();
    // (sharded: every rank must queue the same units per round — the rounds end in collectives — so the depth does not
    // follow this rank's clock there: 1, 2, 4, 4, ...)
    if (sharded_) aheadMax = 4;
    else if (units > 0 && roundMs > 0.0) aheadMax = std::max(1, std::min(16, (int32_t)(25.0 * units / roundMs)));
    ahead = std::min(ahead * 2, aheadMax);
  }
  downloadCtl();
  if (timeUp) {
    // The time limit: the reference checks at once and stops (cupdlp_solver.c:953-962).  The device stands right behind
    // a check (fresh residuals) unless a period was cut short by rejected trials — then one host-driven check.  The
    // same when that last check went on to RESTART: its records describe the iterate before the restart, the vectors
    // that post-solve returns are the restarted ones (the host-driven loop stops in front of the restart).
    const bool behindRestart = s.nIter == hostCtl_->lastCheckIter && hostCtl_->restartKind != 0;
    if (s.nIter != hostCtl_->lastCheckIter || hostCtl_->restartKind != 0) {
      computeAverage();
      computeResiduals();
      ++nChecks_;
    }
    // (right behind a restart the running sums are zero: the "average" would be the zero vector.  The reference never looks
    // at it there — it stops in front of the restart — so the current iterate stands in for both in the log line and in the
    // termination tests.)
    if (behindRestart) avg_ = cur_;
    logCheckLine(s.nIter, cur_, avg_, elapsed(), logSinceHeader);
    if (terminate) {
      if (checkTermination(cur_)) { termIterate_ = 0; termCode_ = PDLP_TERM_OPTIMAL; }
      else if (checkTermination(avg_)) { termIterate_ = 1; termCode_ = PDLP_TERM_OPTIMAL; }
      else if (checkInfeasibility()) termCode_ = PDLP_TERM_INFEASIBLE_OR_UNBOUNDED;
      else termCode_ = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT;
    }
  }
}

// PDHG_Solve, cupdlp_solver.c:899-1215.  `terminate` false = fixed-work timing loop.
void Solver::doSolve(bool terminate, int32_t target) {
  DevState& s = *hostState_;
  const int64_t iterLim = terminate ? (int64_t)opt_.iter_limit : (int64_t)target;
  int logSinceHeader = 50;
  for (;;) {
    const int32_t it = s.nIter;
    if (it >= iterLim) break;
    const double t = elapsed();
    const bool timeUp = timeIsUp();
    // Every stop of the device is a check iteration of the reference schedule
    // (nIter < 10, nIter % 40 == 0, last iteration, or time limit exceeded) — except the entry of
    // a fixed-work timing loop at an iteration that is not on the schedule: a continuous run would
    // not check there either, so neither does the measurement.
    if (!terminate) {
      const int32_t interval = opt_.check_interval > 0 ? opt_.check_interval : kCheckInterval;
      if (!(it < 10 || it % interval == 0)) {
        int64_t halt = nextCheckIter(it);
        if (halt > iterLim) halt = iterLim;
        s.haltIter = (int32_t)halt;
        s.halted = 0;
        pushState();
        runUntilHalt();
        continue;
      }
    }
    computeAverage();
    computeResiduals();
    ++nChecks_;
    if ((it % (kCheckInterval * 100) == 0) || it == iterLim - 1 || timeUp) logCheckLine(it, cur_, avg_, t, logSinceHeader);
    if (terminate) {
      if (checkTermination(cur_)) { termIterate_ = 0; termCode_ = PDLP_TERM_OPTIMAL; break; }
      if (checkTermination(avg_)) { termIterate_ = 1; termCode_ = PDLP_TERM_OPTIMAL; break; }
      if (checkInfeasibility()) { termCode_ = PDLP_TERM_INFEASIBLE_OR_UNBOUNDED; break; }
      if (timeUp) { termCode_ = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT; break; }
      if (it >= iterLim - 1) { termCode_ = PDLP_TERM_TIMELIMIT_OR_ITERLIMIT; break; }
    }
    restartIterate();
    int64_t halt = nextCheckIter(it);
    if (!terminate && halt > iterLim) halt = iterLim;
    s.haltIter = (int32_t)halt;
    s.halted = 0;
    pushState(false);  // runUntilHalt only reads hostState_ until its own synchronisation: one host round trip less per check
    runUntilHalt();
  }
}

void Solver::run(pdlp_result_t* R) {
  reset();
  solveBeg_ = std::chrono::steady_clock::now();
  if (hasStart_) log(1, "Hot starting with given column primal values and row dual values\n");
  if (devCheck_ && !profile_) doSolveDevice(true, 0);
  else doSolve(true, 0);
  solveSeconds_ = elapsed();
  if (opt_.log_level > 0 && rank_ == 0) {
    const Residuals& r = (termCode_ == PDLP_TERM_OPTIMAL && termIterate_ == 1) ? avg_ : cur_;
    const char* what = termCode_ == PDLP_TERM_OPTIMAL
                           ? (termIterate_ ? "Optimal average solution." : "Optimal current solution.")
                           : termCode_ == PDLP_TERM_INFEASIBLE_OR_UNBOUNDED ? "Infeasible or unbounded."
                                                                            : "Time or iteration limit reached.";
    logLine(opt_, 1, "\n%-27s %s\n%27s %+15.8e\n%27s %+15.8e\n%27s %8.2e / %8.2e\n%27s %8.2e / %8.2e\n%27s %8.2e\n%27s %d\n\n",
           "Solving information:", what, "Primal objective:", r.pObj, "Dual objective:", r.dObj,
           "Primal infeas (abs/rel):", r.pFeas, r.pFeas / (1.0 + F_.normRhs), "Dual infeas (abs/rel):", r.dFeas,
           r.dFeas / (1.0 + F_.normCost), "Duality gap (rel):", r.relGap, "Number of iterations:",
           hostState_->nIter);
  }
  if (R) postsolve(R);
}

void Solver::iterate(int32_t nIters, pdlp_iter_stats_t* st) {
  syncState();
  profAxMs_ = profAtyMs_ = 0.0;
  profLaunches_ = 0;
  const int32_t it0 = hostState_->nIter, tr0 = hostState_->nTrials, ck0 = nChecks_, rs0 = nRestarts_;
  solveBeg_ = std::chrono::steady_clock::now();
  const double savedLimit = opt_.time_limit;
  opt_.time_limit = INFINITY;
  hipEvent_t e0, e1;
  PDLP_HIP(hipEventCreate(&e0));
  PDLP_HIP(hipEventCreate(&e1));
  PDLP_HIP(hipEventRecord(e0, stream_));
  if (devCheck_ && !profile_) doSolveDevice(false, it0 + nIters);
  else doSolve(false, it0 + nIters);
  PDLP_HIP(hipEventRecord(e1, stream_));
  PDLP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PDLP_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  opt_.time_limit = savedLimit;
  if (st) {
    memset(st, 0, sizeof(*st));
    st->iters = hostState_->nIter - it0;
    st->trials = hostState_->nTrials - tr0;
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

// PDHG_PostSolve, cupdlp_solver.c:1281-1435 (un-scale, un-permute, un-negate)
void Solver::postsolve(pdlp_result_t* R) {
  const int32_t n = F_.n, m = F_.m, n0 = F_.n0;
  const bool useAvg = (termCode_ == PDLP_TERM_OPTIMAL && termIterate_ == 1);
  const int c = hostState_->cur;
  std::vector<double> x(n), sp(n), sn(n), y(m, 0.0), ax(m, 0.0);
  (useAvg ? xAvg_ : x_[c]).download(x.data(), n, stream_);
  (useAvg ? slackPosAvg_ : slackPos_).download(sp.data(), n, stream_);
  (useAvg ? slackNegAvg_ : slackNeg_).download(sn.data(), n, stream_);
  PDLP_HIP(hipMemcpyAsync(y.data() + r0_, useAvg ? yAvgl() : yl(c), sizeof(double) * mLoc_, hipMemcpyDeviceToHost, stream_));
  (useAvg ? axAvg_ : ax_[c]).download(ax.data() + r0_, mLoc_, stream_);
  PDLP_HIP(hipStreamSynchronize(stream_));
  if (sharded_) {  // assemble the row-sharded (and, with the mesh, column-sliced) vectors on every rank
    gatherToHost(useAvg ? yAvgl() : yl(c), r0_, r1_, true, y);
    gatherToHost((useAvg ? axAvg_ : ax_[c]).get(), r0_, r1_, true, ax);
    if (meshMode_) {
      gatherToHost((useAvg ? slackPosAvg_ : slackPos_).get() + c0_, c0_, c1_, false, sp);
      gatherToHost((useAvg ? slackNegAvg_ : slackNeg_).get() + c0_, c0_, c1_, false, sn);
    }
  }
  if (F_.scaled) {
    for (int32_t j = 0; j < n; ++j) { x[j] /= F_.colScale[j]; sp[j] *= F_.colScale[j]; sn[j] *= F_.colScale[j]; }
    for (int32_t i = 0; i < m; ++i) { y[i] /= F_.rowScale[i]; ax[i] *= F_.rowScale[i]; }
  }
  bool cv = false, cd = false, rv = false, rd = false;
  if (R->col_value) { std::copy(x.begin(), x.begin() + n0, R->col_value); cv = true; }
  if (R->row_value) {
    for (int32_t i = 0, j = 0; i < m; ++i) {
      double v = ax[F_.rowNewIdx[i]];
      if (F_.rowKind[i] == kRowLeq) v = -v;
      else if (F_.rowKind[i] == kRowBound) { v = v + x[n0 + j]; ++j; }
      R->row_value[i] = v;
    }
    rv = true;
  }
  if (R->col_dual) {
    for (int32_t j = 0; j < n0; ++j) R->col_dual[j] = (sp[j] - sn[j]) * F_.sense;
    cd = true;
  }
  if (R->row_dual) {
    for (int32_t i = 0; i < m; ++i) {
      double v = y[F_.rowNewIdx[i]] * F_.sense;
      if (F_.rowKind[i] == kRowLeq) v = -v;
      R->row_dual[i] = v;
    }
    rd = true;
  }
  const Residuals& r = useAvg ? avg_ : cur_;
  R->value_valid = cv && rv;
  R->dual_valid = cd && rd;
  R->term_code = termCode_;
  R->term_iterate = termIterate_;
  R->num_iter = hostState_->nIter;
  R->num_trials = adaptive_ ? hostState_->nTrials : 0;  // nStepSizeIter only counts adaptive trials (cupdlp_step.c:238)
  R->num_restarts = nRestarts_;
  R->primal_obj = r.pObj; R->dual_obj = r.dObj; R->primal_feas = r.pFeas; R->dual_feas = r.dFeas;
  R->rel_gap = r.relGap; R->norm_rhs = F_.normRhs; R->norm_cost = F_.normCost;
  R->setup_seconds = setupSeconds_;
  R->solve_seconds = solveSeconds_;
}

// ---- test / measurement hooks -----------------------------------------------------
std::pair<double*, int64_t> Solver::lookup(const std::string& name) {
  const int c = hostState_->cur, u = c ^ 1;
  const int64_t n = F_.n, m = mLoc_;
  if (name == "x") return {x_[c].get(), n};
  if (name == "y") return {yl(c), m};
  if (name == "ax") return {ax_[c].get(), m};
  if (name == "aty") return {aty_[c].get(), n};
  if (name == "x_next") return {x_[u].get(), n};
  if (name == "y_next") return {yl(u), m};
  if (name == "ax_next") return {ax_[u].get(), m};
  if (name == "aty_next") return {aty_[u].get(), n};
  if (name == "x_avg") return {xAvg_.get(), n};
  if (name == "y_avg") return {yAvgl(), m};
  if (name == "ax_avg") return {axAvg_.get(), m};
  if (name == "aty_avg") return {atyAvg_.get(), n};
  if (name == "x_sum") return {xSum_.get(), n};
  if (name == "y_sum") return {ySum_.get(), m};
  if (name == "cost") return {cost_.get(), n};
  if (name == "rhs") return {rhs_.get(), m};
  if (name == "lower") return {lower_.get(), n};
  if (name == "upper") return {upper_.get(), n};
  if (name == "col_scale") return {colScale_.get(), n};
  if (name == "row_scale") return {rowScale_.get(), m};
  if (name == "slack_pos") return {slackPos_.get(), n};
  if (name == "slack_neg") return {slackNeg_.get(), n};
  if (hasQoff_ && name == "nx") return {nx_[c].get(), n};
  if (hasQoff_ && name == "nx_next") return {nx_[u].get(), n};
  throw std::runtime_error("unknown vector name: " + name);
}

void Solver::getVector(const std::string& name, double* host, int64_t len) {
  syncState();
  if (name == "steps") {  // {tau, sigma, beta, eta, primalStep, dualStep, sumPrimalStep, avgW}
    const DevState& s = *hostState_;
    const double v[8] = {s.tau, s.sigma, s.beta, s.eta, s.primalStep, s.dualStep, s.sumPrimalStep, s.avgW};
    for (int64_t i = 0; i < len && i < 8; ++i) host[i] = v[i];
    return;
  }
  auto [p, l] = lookup(name);
  if (l != len) throw std::runtime_error("length mismatch for vector " + name);
  PDLP_HIP(hipMemcpyAsync(host, p, sizeof(double) * len, hipMemcpyDeviceToHost, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
}

void Solver::setVector(const std::string& name, const double* host, int64_t len) {
  syncState();
  if (name == "steps") {  // {tau, sigma, beta}
    if (len < 3) throw std::runtime_error("steps needs tau, sigma, beta");
    DevState& s = *hostState_;
    s.tau = host[0]; s.sigma = host[1]; s.beta = host[2];
    s.eta = std::sqrt(s.tau * s.sigma);
    s.primalStep = s.tau; s.dualStep = s.sigma;
    pushState();
    return;
  }
  auto [p, l] = lookup(name);
  if (l != len) throw std::runtime_error("length mismatch for vector " + name);
  PDLP_HIP(hipMemcpyAsync(p, host, sizeof(double) * len, hipMemcpyHostToDevice, stream_));
  PDLP_HIP(hipStreamSynchronize(stream_));
}

void Solver::stage(const std::string& name, double* out, int32_t cap) {
  syncState();
  const int c = hostState_->cur;
  auto put = [&](int i, double v) { if (out && i < cap) out[i] = v; };
  if (name == "profile_on" || name == "profile_off") {
    profile_ = name == "profile_on";
  } else if (name == "ax") {
    deviceAx(x_[c].get(), ax_[c].get());
  } else if (name == "aty") {
    deviceATy(yl(c), aty_[c].get());
  } else if (name == "trial") {
    hostState_->haltIter = INT_MAX;
    hostState_->halted = 0;
    pushState();
    enqueueTrial();
    syncState();
    const DevState& s = *hostState_;
    put(0, s.dX2); put(1, s.dY2); put(2, s.inter); put(3, (double)s.lastAccepted);
    put(4, s.tau); put(5, s.sigma); put(6, s.eta); put(7, s.movement); put(8, s.limit); put(9, s.qint);
  } else if (name == "mesh_phases") {  // {X, P, S} average wait in us, then the three wait counts (since the last call)
    double us[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
    if (meshMode_) mesh_->phaseStats(us, cnt, stream_);
    for (int k = 0; k < 3; ++k) { put(k, us[k]); put(3 + k, cnt[k]); }
  } else if (name == "mesh_hop_us") {  // bring-up: microseconds per scalar all-reduce over the mesh (one flag hop out + one back), 200 rounds
    if (meshMode_) {
      hipEvent_t e0, e1;
      PDLP_HIP(hipEventCreate(&e0));
      PDLP_HIP(hipEventCreate(&e1));
      for (int k = 0; k < 8; ++k) mesh_->allReduceScalars(statOut_.get(), 4, stream_);
      PDLP_HIP(hipEventRecord(e0, stream_));
      for (int k = 0; k < 200; ++k) mesh_->allReduceScalars(statOut_.get(), 4, stream_);
      PDLP_HIP(hipEventRecord(e1, stream_));
      PDLP_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      PDLP_HIP(hipEventElapsedTime(&ms, e0, e1));
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      mesh_->checkError(stream_);
      put(0, (double)ms * 1e3 / 200.0);
    } else {
      put(0, -1.0);
    }
  } else if (name == "trial_launches") {  // kernels per trial step of the hot loop (2 = fused decision + primal step)
    // mesh, two all-gathers: 9 with single-block wait kernels, 7 with consumers that wait themselves, 5 with an exchange
    // per launch (fusedWait 0 / 1 / 2); round-1 layout: 9 / 8
    const int fw = meshMode_ ? mesh_->args().fusedWait : 0;
    put(0, meshMode_ ? (colblock_ ? 9.0 - 2.0 * fw : 9.0 - (fw ? 1.0 : 0.0)) : sharded_ ? 7.0 : persistent_ ? 0.0 : fused_ ? 2.0 : 3.0);  // 0: one persistent launch per batch
  } else if (name == "trial_barriers") {  // grid barriers per trial of the persistent loop (0: no persistent loop)
    put(0, !persistent_ ? 0.0 : primalInA_ ? 2.0 : 3.0);
  } else if (name == "check_launches") {  // kernels of one device-driven check iteration (1: the one-launch form of small LPs;
    // sharded: 12 gated kernels + three all-gathers of 4 launches + two scalar all-reduces = 26, with an exchange per
    // launch (fusedWait 2) 9 gated kernels + three all-gathers + the two reductions with their all-reduces = 14;
    // 0: the host drives the checks)
    put(0, !devCheck_ ? 0.0 : sharded_ ? (mesh_ && mesh_->args().fusedWait == 2 ? 14.0 : 26.0) : persistent_ && checkSmall_ ? 1.0 : 10.0);
  } else if (name == "barrier_fallbacks") {  // times a launch with grid barriers gave up and the loop went on with plain launches
    put(0, (double)barrierFallbacks_);
  } else if (name == "uniform_bound_columns") {  // columns whose lower / upper bound the fused launch takes from a scalar of their block
    put(0, (double)uniLowerCols_); put(1, (double)uniUpperCols_);
  } else if (name == "exchange") {  // 0 = not sharded, 1 = RCCL all-reduce, 2 = direct xGMI mesh (partials), 3 = mesh, two all-gathers
    put(0, !sharded_ ? 0.0 : !meshMode_ ? 1.0 : colblock_ ? 3.0 : 2.0);
  } else if (name == "residuals") {
    computeAverage();
    computeResiduals();
    pushState();  // (computeAverage consumed the pending average weights in the host's copy of the state only)
    put(0, cur_.pObj); put(1, cur_.dObj); put(2, cur_.pFeas); put(3, cur_.dFeas);
    put(4, avg_.pObj); put(5, avg_.dObj); put(6, avg_.pFeas); put(7, avg_.dFeas);
    put(8, cur_.pInfObj); put(9, cur_.pInfRes); put(10, cur_.dInfObj); put(11, cur_.dInfRes);
  } else {
    throw std::runtime_error("unknown stage: " + name);
  }
  PDLP_HIP(hipStreamSynchronize(stream_));
}

double Solver::timeKernel(const std::string& name, int32_t reps) {
  syncState();
  if (reps < 1) reps = 1;
  launchFlushAverage(vecsCol_, dst(), stream_);
  hostState_->avgW = 0.0;
  hostState_->avgWx = 0.0;
  const int32_t savedHalt = hostState_->haltIter;
  hostState_->haltIter = INT_MAX;
  hostState_->halted = 0;
  pushState();
  DeviceArray<double> big;
  size_t bigCount = 0;
  if (name == "copy") {
    bigCount = (size_t)64 << 20;  // 2 x 512 MiB: beyond the 256 MiB Infinity Cache
    big.alloc(2 * bigCount);
    big.zero(stream_);
  }
  auto once = [&]() {
    if (name == "spmv_ax") launchSpmvAxDual(dA_.view(), vecs_, dst(), partDY_.get(), stream_);
    else if (name == "spmv_aty") {
      if (!sharded_) launchSpmvAtyInteract(dAt_.view(), vecs_, dst(), partDX_.get(), partInter_.get(), stream_);
      else if (colblock_) launchSpmvAtyInteract(dAt_.view(), vecsAty_, dst(), partDX_.get(), partInter_.get(), stream_);
      else launchSpmvAtyPartial(dAt_.view(), vecs_, dst(), commBuf_.get(), stream_);
    } else if (name == "primal_step") launchPrimalStep(vecs_, dst(), stream_);
    else if (name == "decide_primal") {  // (pending is set by the kernel itself: from the second launch on it decides too)
      const DevState* in = dst();
      stPar_ ^= 1;
      launchDecidePrimal(vecs_, in, dst(), partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(),
                         dAt_.nPartials(), stream_);
    }
    else if (name == "decide")
      launchDecide(dst(), partDY_.get(), dA_.nPartials(), partDX_.get(), partInter_.get(), dAt_.nPartials(), nullptr, stream_);
    else if (name == "trial") enqueueTrial();
    else if (name == "spmv_ax_plain") launchSpmvPlain(dA_.view(), x_[0].get(), tmpM_.get(), stream_);
    else if (name == "spmv_ax_plain_nolong") {  // A x without the segment tasks of its long majors
      MatView v = dA_.view();
      v.lng.nTasks = 0;
      launchSpmvPlain(v, x_[0].get(), tmpM_.get(), stream_);
    }
    else if (name == "spmv_aty_plain") launchSpmvPlain(dAt_.view(), y_[0].get(), commBuf_.get(), stream_);
    else if (name == "copy")
      PDLP_HIP(hipMemcpyAsync(big.get() + bigCount, big.get(), sizeof(double) * bigCount, hipMemcpyDeviceToDevice, stream_));
    else throw std::runtime_error("unknown kernel: " + name);
  };
  for (int i = 0; i < 3; ++i) once();  // warm-up
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
  syncState();
  hostState_->haltIter = savedHalt;
  pushState();
  return (double)ms / reps;
}

}  // namespace pdlp
