// pdlp_kernels.hpp — launch interface of the gfx950 kernels (pdlp_kernels.hip).
//
// Every per-trial kernel reads its step sizes, the current/next buffer parity
// and the "halted" flag from a DevState record in HBM, so a whole batch of
// trial steps can be enqueued (or replayed from a hipGraph) without the host
// knowing which trials get accepted: the accept/reject logic of
// PDHG_Update_Iterate_Adaptive_Step_Size (cupdlp_step.c:215-310) runs on the
// device in k_decide.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pdlp_env.hpp"

namespace pdlp {

// Nonzeros staged through LDS per work block of the CSR-adaptive SpMV: 2048 for big operands (8 loads in
// flight per lane), 512 below 2^18 nonzeros — a small operand then still spreads over tens of blocks and its
// longest majors take the block-wide path (80bau3b: 23.2 -> 18.3 us/iteration).  At 1M nonzeros the small chunk
// is slower in the loop (A x 8.9 -> 9.5 us), hence the cut at 2^18.
constexpr int kChunk = 2048;
constexpr int kChunkSmall = 512;
constexpr int64_t kChunkSmallBelowNnz = 1 << 18;
inline int32_t spmvChunkFor(int64_t nnz) { return nnz < kChunkSmallBelowNnz ? kChunkSmall : kChunk; }
constexpr int kSpmvThreads = 256;
constexpr int kMaxMajorsPerBlock = 2048;
constexpr int kVecThreads = 256;

// Device-resident solver state (one per solver; lives in HBM).
struct DevState {
  double eta;            // dStepSizeUpdate of the NEXT trial = sqrt(tau*sigma)
  double beta;           // stepsize->dBeta (primal weight squared)
  double tau, sigma;     // step sizes of the next trial
  double primalStep, dualStep;        // stepsize->dPrimalStep / dDualStep
  double sumPrimalStep, sumDualStep;  // stepsize->dSum*Step
  double avgW;           // weight of the accepted iterate not yet added to ySum (the dual-step epilogue adds it)
  double avgWx;          // ... not yet added to xSum (the next primal step adds it)
  double dX2, dY2, inter, movement, limit;  // last trial (diagnostics / tests)
  double qint;           // QP with off-diagonal Hessian entries: dx . N dx of the last trial
  int32_t nIter;         // timers->nIter
  int32_t nTrials;       // stepsize->nStepSizeIter
  int32_t cur;           // parity of the current iterate buffers (nIter % 2 in the reference)
  int32_t halted;        // set when nIter reaches haltIter: remaining queued kernels no-op
  int32_t haltIter;      // next iteration at which the host must run a check
  int32_t adaptive;      // PDHG_ADAPTIVE_LINESEARCH (1) or fixed step (0)
  int32_t lastAccepted;
  // (k+1)^-0.3 and (k+1)^-0.6 for the next trial counters k = powBase .. powBase+powCount-1,
  // tabulated by the host (glibc pow) so that the step-size update is bit-reproducible on the CPU
  int32_t powBase;
  int32_t powCount;
  int32_t commError;     // a mesh exchange wait timed out (sharded path)
  int32_t pending;       // single-GPU loop: a trial has been computed whose accept/reject decision is still to be taken
  int32_t pad_;
  const double* powRed;
  const double* powGrow;
};

// ---- check / restart control on the device (round 4) -----------------------------------------------------------
// The reference's check iteration (PDHG_Solve, cupdlp_solver.c:975-1069: residuals of both iterates, termination
// tests, PDHG_Check_Restart_GPU cupdlp_restart.c:3-124, PDHG_Compute_Step_Size_Ratio cupdlp_step.c:147-176) used to
// come back to the host every 40 iterations: 30 statistics down, the decision on the host, a state record up, a
// stand-alone primal step — ~240 us of idle queue per check at 1M x 1M next to 173 us of kernels.  Now the whole
// check is a sequence of kernels BEHIND the trial batch: every kernel of it first asks checkDue() (the device has
// halted at a scheduled iteration and the solve is not over), k_check_decide holds the scalar logic, and the host
// enqueues several [batch][check] units before it looks at the pinned CheckRecords.  A check that is not due
// (a period that needed more spare trials than were queued; the queue behind a termination) costs early-exit
// kernels and changes nothing.
struct ResidualsDev {  // = Residuals (pdlp_solver.hpp)
  double pObj, dObj, gap, relGap, pFeas, dFeas, pInfObj, pInfRes, dInfObj, dInfRes;
};
struct CheckCtl {
  // parameters of the running loop (host, before the first unit)
  double primalTolAbs, dualTolAbs;  // tol * (1 + ||b||), tol * (1 + ||c||)   cupdlp_solver.c:797-841
  double gapTol, feasTol, sense, offset;
  int32_t terminate;     // 1: Solver::run (termination tests), 0: fixed-work loop of Solver::iterate
  int32_t restartOn;
  int32_t interval;      // check interval (CUPDLP_RELEASE_INTERVAL 40 unless the caller asked otherwise)
  int32_t iterLimit;     // terminate: opt.iter_limit, else the target iteration of the fixed-work loop
  int32_t optIterLimit;  // opt.iter_limit (the schedule's "last iteration" rule uses it in both modes)
  int32_t qp;            // the objective has a 1/2 x'Qx term (statistic 10 of the column pass)
  int32_t adaptive;
  int32_t pad_;
  // state (host mirrors: Solver::cur_, avg_, pFeasLR_ ...; uploaded before, downloaded after a device-driven loop)
  ResidualsDev cur, avg;
  double pFeasLR, dFeasLR, gapLR, pFeasLC, dFeasLC, gapLC;
  int32_t iLastRestartIter, nRestarts, nChecks;
  int32_t termCode, termIterate;
  int32_t terminated;    // the solve is over: everything still queued is a no-op
  int32_t restartKind;   // of the check in flight: 0 none, 1 to the current, 2 to the average iterate
  int32_t lastCheckIter;
};
// What the host reads of an executed check (pinned host memory, written by the check's scalar kernels)
struct CheckRecord {
  int32_t ran, it, terminated, termCode, termIterate, restartKind, nRestarts, nChecks, nTrials, pad_;
  double beta;
  ResidualsDev cur, avg;
};

struct SpmvMat {
  const int32_t* beg;       // [nMajor+1]
  const int32_t* idx;       // [nnz]
  const double* val;        // [nnz]
  const int32_t* blockBeg;  // [4*nBlocks] stream plan: first and end major, first and end entry of each work block
  int32_t nMajor;
  int32_t nBlocks;
  int32_t partOffset;       // first slot of this matrix in the per-block partial arrays
  int32_t chunk;            // kChunk or kChunkSmall: the work plan's block size
};

// Long majors (pdlp_host.hpp LongPlan), device pointers.  A major that is too long for the left-to-right
// lanes (more than the stream plan's chunk; more than 256 nonzeros in the slab layout) is cut into SEGMENT
// TASKS of segLen nonzeros (512, doubled until a major has at most 64 of them).  One wave per task: lane l
// accumulates the entries l, l+64, ... of the segment in ascending order, then the 64-lane shuffle tree; the
// segment sums of a major are added left to right and its epilogue runs once.  The tasks are extra workgroups
// at the end of the SpMV grid (W = 4 waves per 256-thread block, 16 per 1024-thread block), so they run next to
// the streams instead of behind them.  A major with at most W segments never straddles two workgroups (the task
// list is padded with idle tasks): its segment sums meet in LDS.  A longer one spans workgroups: segment sums go
// to HBM, every wave takes a ticket of the major, the wave with the LAST ticket finishes it.  The reduction
// contributions of long major c go to their own slot of the partial arrays (slotBase + c): which wave finishes
// last does not matter, results are deterministic.
constexpr int kLongSegment = 512;   // nonzeros per segment task (8 per lane)
constexpr int kLongMaxSegments = 64;
constexpr int kLongSlotCap = 2048;  // more long majors than this: contributions are summed in fixed groups (k_long_groups)
struct LongTask {      // 32 bytes: one scalar load per wave
  int32_t pBeg, pEnd;  // entries of the segment
  int32_t c;           // long-major index (-1: idle task, padding)
  int32_t first;       // first segment-sum slot of the major (stream layout: = its first task)
  int32_t nSeg;        // segments of the major
  int32_t major;       // index of the major in the result vector
  int32_t contained;   // 1: all segments in this workgroup, consecutive waves (LDS), 0: spanning (HBM slots + tickets)
  int32_t seg;         // this task's segment of the major; its sum goes to slot first + seg
};
struct LongMat {
  const int32_t* idx;        // the CSR arrays the long majors live in (the operand's, or the slab layout's side copy)
  const double* val;
  const LongTask* tasks;     // [nTasks]
  double* segSum;            // [segment slots] (spanning majors)
  uint32_t* ticket;          // [nLong] zero between launches
  double* contrib;           // [2*nLong] when nLong > kLongSlotCap (then k_long_groups fills the slots), else nullptr
  int32_t nLong, nTasks;
  int32_t slotBase;          // first slot of the long majors in the partial arrays
  int32_t nSlots;            // nLong, or the number of groups
  int32_t groupSize;         // long majors per slot (1 unless nLong > kLongSlotCap)
  // tasks per workgroup (the plan's group, pdlp_host.hpp planLong): 4 in the stream layout; 16, 8, ... 1 in the slab
  // layout — as many task workgroups as the device has CUs where there are enough tasks, so that every CU carries the
  // same extra load next to its streaming block (DeviceMatrix::uploadPlans)
  int32_t taskGroup;
};

// Slab layout (pdlp_host.hpp SlabLayout), device pointers.  One 1024-thread block = 16 waves, each
// owning the consecutive majors [waveBeg[gw], waveBeg[gw+1]) (cut by work, pdlp_host.hpp slabPartition) and its own
// sorted entry list.
constexpr int kSlabThreads = 1024;
constexpr int kSlabMaxRows = 16384;  // majors per block (LDS accumulators: 128 KB of the 160 KB)
struct SlabMat {
  const int32_t* wavePtr;    // [16*nBlocks+1] entry offsets per wave
  const uint32_t* ent;       // [nnz] (localMajor << minorBits | minor)
  const double* val;         // [nnz]
  const uint32_t* longMask;  // [ceil(nMajor/32)+1] bit r: major r is a long one
  const int32_t* waveBeg;    // [16*nBlocks+1] first major of every wave
  int32_t nMajor, nBlocks, rowsPerBlock, minorBits;  // rowsPerBlock: the most majors any block owns (LDS accumulators)
  // 1: no block barrier per 64-entry group.  The barrier keeps the CU's waves on the same slab of the gathered vector
  // (a random matrix needs that: 50 -> 58 us at 1M x 1M without); an operand whose blocks touch little of the gathered
  // vector anyway runs faster free (block-angular LP of bench.py --config c: 31.8 -> 29.0 us).  Chosen per operand by
  // timing both at set-up (tuneXcdMap).
  int32_t noPace;
};
// Slab width for operands whose row blocks touch few 2^14-entry stretches of the gathered vector densely (network
// blocks, staircases): shorter runs of equal majors per 64-entry group, more lanes adding (bench.py --config c, A x:
// 44.0 -> 41.1 us).  (Round 3 also built LDS staging of those stretches — bit-identical, measured SLOWER, 46.7 vs
// 33.2 us on config c: two block barriers and a dependent load chain per tile cost more than the L2 gathers they
// replace; removed in round 4, the code is in the history at commit 22b5412.)
constexpr int kSlabTileLog2 = 14;

// One operand matrix of the iteration: a plain CSR stream or the slab layout for the majors that are summed
// left to right, plus the segment tasks of the long ones.
struct MatView {
  SpmvMat csr;
  SlabMat slab;
  LongMat lng;
  int32_t useSlab;
  int32_t nPartials;  // slab.nBlocks (if used) + csr.nBlocks + lng.nSlots
  int32_t xcdMap;     // 1: XCD x owns a contiguous range of work blocks, 0: round robin (chosen per operand at setup)
  // Fused trial on an operand with long majors: > 0 = their segment tasks run as that many EXTRA workgroups of the fused
  // launch, resident next to the streaming blocks (two 1024-thread blocks per CU at 64 registers) and counted by its grid
  // barrier; 0 = the streaming blocks run the task passes themselves behind their stream (where two blocks per CU do
  // not fit).  Set by the solver from fusedCoTaskBlocks().
  int32_t coTaskBlocks;
  int32_t touchTail;     // the fused trial touches the operands of a block's columns beyond the register-held ones before its barrier
};

// Vectors of the iteration (device pointers). Pairs are double-buffered by parity.
struct IterVecs {
  double* x[2];
  double* y[2];
  double* ax[2];
  double* aty[2];
  double* xSum;
  double* ySum;
  const double* cost;
  const double* rhs;
  const double* lower;
  const double* upper;
  const double* qdiag;  // diagonal of Q (QP prox step) or nullptr for an LP
  double* nx[2];        // N x for the off-diagonal part N of Q (explicit gradient term), by parity; nullptr without one
  int32_t n, m;
  int32_t nEqs;       // GLOBAL count of equality rows
  int32_t rowOffset;  // global index of local row 0 (0 unless sharded)
  // 1: c, l, u of the primal step are read with ordinary loads (they stay in the Infinity Cache next to the matrices);
  // 0: non-temporal like every vector touched once per kernel.  Set by the solver from the operands' sizes (constCached).
  int32_t constCached;
  // Round 6: where ALL columns of a block of the transposed operand's slab partition share one lower (upper) bound —
  // x >= 0 without an upper bound is the rule in LPs, and an infinite bound stays infinite under column scaling — the
  // fused trial's tail takes the value from here instead of loading it per column: colBlockUni[b] bit 0 = lower, bit 1 =
  // upper uniform in logical block b, colBlockBounds[2 b], [2 b + 1] the values (launchBlockBounds; nullptr: none).
  const int32_t* colBlockUni;
  const double* colBlockBounds;
  int32_t lowerUniform;  // every block's bit 0 is set and the values agree: the fused slab kernel's ULO instantiation
};
// The 256 MB Infinity Cache holds the two matrix copies (24 bytes per nonzero) for the whole solve; the constant vectors of
// the primal step (24 bytes per column) join them only where that leaves room to spare: at 216 MB (1M x 1M / 8M nonzeros)
// the fused launch gained 1.3 us and the other launch lost as much, with outliers of +7 us (round 6, measurements
// section 14); at 176 MB (config c) the fused launch gained 5.5 us.
inline int32_t constCached(int64_t nnz, int64_t n) { return 24 * nnz + 24 * n <= (int64_t)200 * 1000 * 1000 ? 1 : 0; }

// ---- HiPDLP path (solver="hipdlp"): restarted Halpern PDHG, hipdlp/pdhg.cc:961-1018 ----------
// Step sizes and the Halpern counter live in HBM so that a block of 40 steps replays from a
// hipGraph; they only change at restarts (host, check iterations).
struct HalpernState {
  double tau, sigma;  // stepsize_.primal_step / dual_step
  double rho;         // params_.halpern_gamma (1 = full reflection)
  int32_t hIter;      // halpern_iteration_ at the start of the block
  int32_t halted;     // the device-driven loop has ended (converged, iteration limit): every kernel queued behind is a no-op
  // ---- the scalars of PDLPSolver (pdhg.hpp) that the check iteration reads and writes: on the device the block's
  // decision kernel updates them (pdlp_halpernfn.hpp halpernDecide), in the host-driven loop (sharded, profile mode) the
  // same function runs on the host's copy ----
  double eta, omega, primalWeight, bestPrimalWeight, bestGap, errSum, lastErr;
  double fpe, initialFpe, lastTrialFpe;
  double normRhs, normCost, offset, tol;
  long long iters, iterLimit;
  int32_t run;        // gate word of the block's kernels: 1 while the loop runs (= !halted)
  int32_t runFpe0;    // gate word of the initial fixed-point error behind the block's first step: a restart came before
  int32_t doRestart;  // gate word of the restart copies (anchor and current iterate <- pdhg iterate of the last major step)
  int32_t converged;  // gate word of the output copy
  int32_t pid, terminate, nRestarts, nChecks;
  int32_t termStatus;
  int32_t fpe0Pending;  // the next block starts behind a restart (runFpe0 is this word while the loop runs, 0 once it has halted)
};
// One line of the check, written by the device into pinned host memory (a ring): what the host logs and returns.
struct HalpernRecord {
  long long iters;
  double pObj, dObj, gap, relGap, pFeas, dFeas, fpe, primalWeight;
  int32_t restarted, converged;
};
constexpr int kHalpernRing = 32;
// slots of the statistics vector of one check (pdlp_halpern.cpp): three sums of the fixed-point error, the six sums of
// checkConvergence, the three sums of the initial fixed-point error, the two restart distances
constexpr int kHSlotFpe = 0, kHSlotCheck = 3, kHSlotFpe0 = 9, kHSlotDiff = 12, kHStatOut = 16;
struct HalpernVecs {
  double* xc; double* yc;          // x_current_, y_current_
  double* xn; double* yn;          // x_next_, y_next_ (pdhg iterate of the last MAJOR step)
  double* rx; double* ry;          // reflected_x_ (every step), reflected_y_ (major steps)
  const double* xa; const double* ya;  // anchors
  double* slack;                   // halpern_dual_slack_next_ (major steps)
  const double* cost; const double* lower; const double* upper;
  const double* rowLower; const double* rowUpper;
  const HalpernState* hs;
  int32_t kOff;    // k_offset of this step inside the block (1..40)
  int32_t major;   // is_major
};
// A' y_current fused with the primal projection, reflection and Halpern blend (steps 5+1+4x)
void launchHalpernPrimal(const MatView& At, const HalpernVecs& h, hipStream_t s);
// A reflected_x fused with the dual projection, reflection and Halpern blend (steps 2+3+4y)
void launchHalpernDual(const MatView& A, const HalpernVecs& h, hipStream_t s);

// ---- per-trial kernels ----------------------------------------------------
// Single-GPU loop, first launch of a trial: takes the accept/reject decision of the PREVIOUS trial (if one is
// pending: every block re-reduces the per-block partials in the fixed order of k_decide and comes to the same
// result), then the primal step of this trial with the new step sizes.  Reads *stIn, block 0 writes *stOut
// (the two slots alternate from trial to trial, so no block can read a half-written state).
void launchDecidePrimal(const IterVecs& v, const DevState* stIn, DevState* stOut, const double* partDY, int32_t nDY,
                        const double* partDX, const double* partInter, int32_t nDX, hipStream_t s,
                        const double* partQ = nullptr, int32_t nQ = 0);
// QP with off-diagonal Hessian entries: nx_next = N x_next fused with the partials of dx . N dx (the third SpMV of a trial)
void launchSpmvQxInteract(const MatView& N, const IterVecs& v, const DevState* st, double* partQ, hipStream_t s);
void launchPrimalStep(const IterVecs& v, const DevState* st, hipStream_t s);
// ax_next = A x_next fused with the dual step; writes per-block sum (dy)^2 to partDY[block]
void launchSpmvAxDual(const MatView& A, const IterVecs& v, const DevState* st, double* partDY, hipStream_t s);
// aty_next = A' y_next fused with movement/interaction partials
void launchSpmvAtyInteract(const MatView& At, const IterVecs& v, const DevState* st, double* partDX,
                           double* partInter, hipStream_t s);
// sharded variant: partial A_g' y_next into out[n] (no epilogue)
void launchSpmvAtyPartial(const MatView& At, const IterVecs& v, const DevState* st, double* out, hipStream_t s);
// sharded: aty_next = reduced; movement/interaction partials
void launchInteract(const IterVecs& v, const DevState* st, const double* atyReduced, double* partDX,
                    double* partInter, int32_t nBlocks, hipStream_t s);
// sums partials[0..count) deterministically into *out (one block)
void launchReduceTo(const double* partials, int32_t count, double* out, const DevState* st, hipStream_t s);
// accept/reject + step-size update; dyGlobal != nullptr -> use *dyGlobal instead of partDY;
// onlyIfPending: the flush of the single-GPU loop (no-op unless st->pending)
void launchDecide(DevState* st, const double* partDY, int32_t nDY, const double* partDX, const double* partInter,
                  int32_t nDX, const double* dyGlobal, hipStream_t s, bool onlyIfPending = false,
                  const double* partQ = nullptr, int32_t nQ = 0);

// ---- fused trial (single GPU, either layout): 2 launches ------------------------------------------------------
// aty_next = A' y_next with the movement / interaction partials as above, then — inside the same launch — a grid
// barrier, the accept/reject decision (every block re-reduces the partials in the fixed order of k_decide), and
// the NEXT trial's primal step on the columns the block owns: x, x+, A'y are still in registers, A'y+ in LDS, and
// c, l, u, xSum were fetched while the matrix streamed.  Reads *stIn, block 0 writes *stOut (the other slot).
// Needs every block of the grid resident at once (one 1024-thread block per CU): fusedAtyBlocksResident() tells
// how many the device takes; the solver falls back to the 3-launch trial otherwise.  bar: one zeroed 8-byte
// arrival word per block + one timeout flag (a wait that does not end within ~1 s sets commError instead of
// hanging the device); the words must be zeroed whenever the trial counter starts again (Solver::reset).
inline size_t gridBarWords(int nBlocks) { return (size_t)nBlocks + 8; }
int fusedAtyBlocksResident(const MatView& At, int device);
int fusedAtyBlocks(const MatView& At);  // blocks of the fused launch (= arrival words of the barrier)
// Task workgroups the fused launch can carry next to its streaming blocks (0: none / not resident together)
int fusedCoTaskBlocks(const MatView& At, int device);
// timeoutMs: how long the barrier waits for a block that is not resident (a shared device); then the trial stays
// undecided, *stOut carries commError = 3 and the caller falls back to the 3-launch trial.  faultTrial (tests): the
// trial that raises the trial counter to this value expects one block too many (0: none).
void launchSpmvAtyFusedPrimal(const MatView& At, const IterVecs& v, const DevState* stIn, DevState* stOut,
                              const double* partDY, int32_t nDY, double* partDX, double* partInter,
                              unsigned long long* bar, hipStream_t s, int32_t timeoutMs = 1000, int32_t faultTrial = 0);

// ---- small LPs: a batch of trials as ONE persistent launch (pdlp_small.hip) -------------------------------------
// Both operands in the stream layout, no long majors.  smallTrialsGrid: workgroups of the launch (0: does not
// qualify) and, in *resident, how many the device holds at once (the grid barrier needs all of them resident).
// bar: smallBarWords(grid) zeroed words.  The launch runs at most maxTrials trials and stops early when the device
// halts; the state record is read from and written back to *st.  mode 0: agent-scope accesses on all XCDs, every
// workgroup sweeps the arrival words (512-entry blocks only).  mode 1 (512-entry blocks only): only every eighth of
// 8 * grid workgroups works (one XCD, one coherent L2: no agent-scope traffic); the launch checks that placement and,
// if it does not hold, changes nothing and sets commError = 2 in *st — the caller then goes on with another mode.
// mode 2: all XCDs, XCD-hierarchical barrier (pdlp_devfn.hpp hierBarrier) — what hundreds of workgroups need.
int smallTrialsGrid(const MatView& A, const MatView& At, int32_t n, int device, int* resident, bool primalInA = false);
// Every launch begins with a roll call of its working workgroups (pdlp_devfn.hpp rollCall): if they are not all resident
// within timeoutMs, the launch changes nothing but commError = 3 in *st (failRollCall: a test asks for exactly that).
// seq: number of this launch since the caller zeroed `bar` (1, 2, ...): the roll call counts cumulatively.
// primalInA: two barriers per trial — phase A's gathers recompute x+ themselves (partDY then needs 2 * A.nPartials slots).
void launchSmallTrials(const MatView& A, const MatView& At, const IterVecs& v, DevState* st, double* partDY, double* partDX,
                       double* partInter, unsigned long long* bar, int32_t grid, int32_t maxTrials, int mode, hipStream_t s,
                       int32_t timeoutMs = 1000, bool failRollCall = false, bool selfTest = false, unsigned long long seq = 1,
                       bool primalInA = false);
constexpr int kSmallHierWords = 4 * 16 * 32;  // the XCD-hierarchical barrier's words (pdlp_devfn.hpp HierBar)
// arrival words, timeout flag, XCC ids of the placement check; behind them (256-byte aligned) the hierarchical barrier's words
// ... and, last, the words of the XCD-local mode's coherence self-test (grid test words, grid arrival words, flag, failure word)
inline __host__ __device__ size_t smallBarWords(int grid) { return ((2 * (size_t)grid + 16 + 31) / 32) * 32 + kSmallHierWords + 2 * (size_t)grid + 8; }

// ---- check-iteration kernels --------------------------------------------------------------------------------------
// Every launcher takes a CheckGate.  {nullptr, nullptr}: host-driven check (sharded paths, stage("residuals"),
// profile mode) — the kernel always runs and takes the buffer parity / weights from its arguments.  Otherwise the
// kernel runs only if checkDue(st, cc) (pdlp_devfn.hpp) and reads parity, weights and step sums from *st.
struct CheckGate {
  const DevState* st = nullptr;
  const CheckCtl* cc = nullptr;
  const int32_t* flag = nullptr;  // HiPDLP's device-driven loop: the kernel runs only while this word is non-zero
};
void launchFlushAverage(const IterVecs& v, DevState* st, hipStream_t s);
void launchClearAvgW(DevState* st, hipStream_t s);  // the pending average weights of *st have been consumed
void launchScaleCopy(double* dst, const double* src, double a, int32_t len, hipStream_t s);  // dst = a*src
// one pass: pending average update (weights w for y, wx for x) and the averages xAvg = xSum * ps, yAvg = ySum * ds
// (PDHG_Compute_Average_Iterate, cupdlp_step.c:377-420)
void launchFlushScale(const IterVecs& v, CheckGate g, int cur, double w, double wx, double ps, double ds, double* xAvg, double* yAvg,
                      hipStream_t s);
// Row / column statistics of the current AND the average iterate in one pass each (quantities: current first), and
// the final reduction of both in one launch.  v: the iterate vectors of the rows / of the own column slice.
//  row  0: sum ((ax-b) projected) * rowScale)^2   primal residual^2   (cupdlp_solver.c:12-67)
//       1: sum y*b                                dual objective part (:80)
//       2: sum y^2                                ray norm part       (:230)
//       3: sum ((ax projected)*rowScale)^2        dual-infeasibility constraint part (:339-345)
//  col  0: sum c*x   1: sum sp*lowerF   2: sum sn*upperF   3: sum ((r-sp+sn)*colScale)^2
//       4: sum sp^2  5: sum sn^2        6: sum ((aty+sp-sn)*colScale)^2
//       7: sum x^2   8: sum (min(x,0)*hasLower/colScale)^2    9: sum (max(x,0)*hasUpper/colScale)^2
//      10: sum 1/2 x (Qx) (QP only; the reduced cost then is c + Q x - A'y; nx = N x for the off-diagonal part)
constexpr int kRowStats = 4;
constexpr int kColStats = 11;
void launchRowStats2(const IterVecs& v, CheckGate g, int cur, const double* axA, const double* yA, const double* rowScale, int scaled,
                     double* partials, int32_t stride, int32_t nBlocks, hipStream_t s);
void launchColStats2(const IterVecs& v, CheckGate g, int cur, const double* atyA, const double* xA, const double* colScale,
                     const double* nxA, int scaled, double* spC, double* snC, double* spA, double* snA, double* partials,
                     int32_t stride, int32_t nBlocks, hipStream_t s);
void launchFinalReduce2(const double* partials, int32_t stride, int32_t nQ0, int32_t nBlocks0, int32_t nQ1, int32_t nBlocks1,
                        double* out, CheckGate g, hipStream_t s);
void launchSpmvPlain(const MatView& A, const double* in, double* out, hipStream_t s, CheckGate g = CheckGate());
void launchFill(double* dst, double value, int32_t len, hipStream_t s);
void launchProjectBounds(double* x, const double* lower, const double* upper, int32_t n, hipStream_t s);
void launchMulInPlace(double* x, const double* y, int32_t len, hipStream_t s);   // x *= y
void launchDivInPlace(double* x, const double* y, int32_t len, hipStream_t s);   // x /= y
// out[q] = sum_{b<nBlocks} partials[q*stride+b], q < nQ (deterministic)
void launchFinalReduce(const double* partials, int32_t stride, int32_t nBlocks, int32_t nQ, double* out,
                       hipStream_t s, const int32_t* gate = nullptr);  // gate: the kernel runs only while this device word is non-zero
// partials of ||a-b||^2
void launchDiffNorm2(const double* a, const double* b, int32_t len, double* partials, int32_t nBlocks,
                     hipStream_t s, const int32_t* gate = nullptr);
// partials of a.b
void launchDot(const double* a, const double* b, int32_t len, double* partials, int32_t nBlocks, hipStream_t s);

// ---- the whole device-driven check of a small LP as ONE launch (pdlp_check.hip k_check_small) ----
// For the LPs of the persistent trial loop with at most 64 workgroups: the phases of the launch sequence below separated by
// grid barriers inside one launch of `grid` resident workgroups; same statistics grids, same scalar logic, same bits.
// bar: grid + 8 words zeroed when the solve starts; seq = 1, 2, ... counts the launches since.  A roll call that fails
// (shared device) leaves everything untouched and sets commError = 3.
int checkSmallResident(const MatView& A, const MatView& At, int device);
struct RestartVecs;
void launchCheckSmall(const MatView& A, const MatView& At, const IterVecs& v, DevState* st, CheckCtl* cc, CheckRecord* rec,
                      const RestartVecs& r, const double* rowScale, const double* colScale, int scaled, double* spC, double* snC,
                      double* spA, double* snA, double* statPart, int32_t statStride, double* statOut, double* partX, double* partY,
                      unsigned long long* bar, int32_t grid, unsigned long long seq, int32_t timeoutMs, hipStream_t s);

// ---- the scalar side of a device-driven check (pdlp_check.hip) ------------------------------------------------------
// stat: the 2*kRowStats + 2*kColStats statistics (launchFinalReduce2).  Residuals of both iterates, termination
// tests, the restart decision (-> cc->restartKind); rec: pinned host record of this check.
void launchCheckDecide(DevState* st, CheckCtl* cc, const double* stat, CheckRecord* rec, hipStream_t s);
// If the check decided to restart: running sums cleared, average -> current iterate (restartKind 2), the partials of
// ||x - xLast||^2 (nbX blocks, partX) and ||y - yLast||^2 (nbY blocks, partY) in the grids of launchDiffNorm2, and
// xLast / yLast <- the restarted iterate (PDHG_Restart_Iterate_GPU, cupdlp_proj.c:88-148).  v / vCol: rows / own columns.
struct RestartVecs {
  const double* xAvg; const double* yAvg; const double* axAvg; const double* atyAvg; const double* nxAvg;
  double* xLast; double* yLast;
};
void launchRestartVec(const IterVecs& v, const DevState* st, const CheckCtl* cc, const RestartVecs& r, double* partX, int32_t nbX,
                      double* partY, int32_t nbY, hipStream_t s);
// Row-block sharded solve, restart to the average: launchRestartVec above works on a rank's own columns and rows; the
// REPLICATED vectors (x and A'y full length on every rank; y too in the two-all-gathers layout) are copied whole here —
// xAvg / atyAvg / yAvg are full length on every rank after the check's all-gathers.  No-op unless the check is due and
// decided restartKind 2.
void launchRestartCopyFull(const DevState* st, const CheckCtl* cc, double* const x[2], double* const aty[2], double* const yFull[2],
                           const double* xAvg, const double* atyAvg, const double* yAvgFull, int32_t n, int32_t yLen, hipStream_t s);
// Primal-weight update after a restart (PDHG_Compute_Step_Size_Ratio, cupdlp_step.c:147-176) from the two partial
// arrays, then — restart or not — the next halt iteration of the reference's check schedule and the device runs on.
void launchRestartFinish(DevState* st, CheckCtl* cc, const double* partX, int32_t nbX, const double* partY, int32_t nbY,
                         CheckRecord* rec, hipStream_t s);

int32_t vecBlocks(int32_t len);  // grid size used by the vector/statistics kernels
void launchAddInt(int32_t* v, int32_t d, int64_t len, hipStream_t s);  // v[i] += d (set-up: rebasing index arrays of a shard)

// ---- set-up: which slab width suits an operand ----
// lo/hi/cnt [nBlocks]: column span and entry count of each slab block's short majors (INT_MAX / -1 / 0 for a block without
// any); hist (nullptr: none) [8 * nTiles], zeroed by the caller: entries per (XCD of the contiguous map, tile of 2^tileLog2
// minors) — which XCD's streaming blocks gather from which stretch of the vector (pdlp_host.hpp xcdTileOwners)
// uni[b] / bounds[2 b .. 2 b + 1] of IterVecs::colBlockUni / colBlockBounds for the nBlocks logical blocks of a slab
// partition (waveBeg: its prefix array of majors, 16 waves per block): equality of the bit patterns, so -0.0 != 0.0
void launchBlockBounds(const double* lower, const double* upper, const int32_t* waveBeg, int32_t nBlocks, int32_t* uni, double* bounds,
                       hipStream_t s);
void launchBlockSpan(const int32_t* beg, const int32_t* idx, const int32_t* waveBeg, int32_t nBlocks, int32_t longLimit, int32_t* lo,
                     int32_t* hi, int32_t* cnt, int32_t tileLog2, int32_t nTiles, int32_t* hist, hipStream_t s);

}  // namespace pdlp
