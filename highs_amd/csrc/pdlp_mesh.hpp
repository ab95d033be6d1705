// pdlp_mesh.hpp — direct xGMI exchange for the row-block sharded PDHG loop
// ("the performance path is a direct full-mesh exchange").
//
// One rank per GPU on one node: either one PROCESS per GPU (a launcher, pdlp_mi355x_create_sharded) or
// one host THREAD per GPU inside a single process (pdlp_mi355x_solve with num_devices > 1, i.e. what
// Highs::run() reaches).  Every rank owns an ARENA of uncached device memory that all peers map —
// through HIP IPC across processes, through peer access inside one process; kernels write straight into the
// peers' arenas over xGMI (write-through system-scope stores), wait until those
// stores have landed, and then announce the data by storing a monotonically
// increasing epoch into a per-sender flag; the consumer kernel spins on its own
// flags (system-scope loads, bounded by a wall-clock timeout) before it reads the
// payload with system-scope loads.  No RCCL, no host round trip, no cache fences,
// nothing that a hipGraph cannot replay.
//
// Per trial step (rank g owns row block [r0,r1) and column slice [c0,c1)):
//   X: x+[c0:c1) is pushed into every peer's recvX      (all-gather of x+)
//   P: the partial A_g' y+ slice of owner h is pushed into h's recvP[g]; h adds
//      the G contributions in RANK ORDER                 (reduce-scatter of A'y+)
//   S: {dX^2, dY^2, interaction} partials go to every peer's mailbox; every rank
//      adds the G triples in rank order and takes the identical accept/reject
//      decision                                          (all-reduce of 3 scalars)
// All sums are in a fixed order, so every rank holds bit-identical x, step sizes
// and control flow, run after run.
//
// The reference has no counterpart: Ax_multi_gpu / ATy_multi_gpu are exit(1)
// stubs (cupdlp_linalg.c:420-423,453-456).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "pdlp_kernels.hpp"

namespace pdlp {

constexpr int kMeshMaxRanks = 16;
constexpr int kMeshMailDoubles = 64;  // capacity of one generic scalar all-reduce

// Flag kinds: one array of per-sender epochs each.
enum MeshFlag : int { kFlagX = 0, kFlagP = 1, kFlagS = 2, kFlagGen = 3, kFlagBar = 4, kNumMeshFlags = 5 };

// Mutable exchange state in (ordinary) device memory of the owning rank.
struct MeshState {
  long long seq;        // executed trial steps so far == epoch of the hot-loop flags
  int32_t error;        // set by a kernel whose wait timed out (host turns it into an exception)
  int32_t pad_;
  uint32_t counter[4];  // "last block signals" tickets
  // time spent waiting for the peers' flags in the hot loop, per exchange (X all-gather, P reduce-scatter,
  // S scalars): 100 MHz wall-clock ticks and number of waits (measurement only; bench.py reports them)
  unsigned long long waitTicks[3];
  unsigned long long waitCount[3];
};

// Lives in device memory (the kernels index its arrays dynamically); the host keeps a copy.
struct MeshView {
  char* arena[kMeshMaxRanks];  // arena base of every rank as mapped into THIS process (own = local)
  int32_t G, g;
  int32_t colOff[kMeshMaxRanks + 1];
  int32_t rowOff[kMeshMaxRanks + 1];
  int64_t offFlags, offMailHot, offMailGen, offRecvX, offRecvP, offRecvY;
  int64_t sliceMax;  // doubles per recvP slot
  int64_t waitTicks; // wall-clock (100 MHz) budget of one wait
  MeshState* ms;
};

// Kernel argument (by value): the scalars every mesh kernel needs right away, so that its first
// instructions do not chase pointers through the view.
struct MeshArgs {
  const MeshView* v;  // device copy of the view (arena pointers, offsets, partitions)
  MeshState* ms;
  int32_t G, g;
  int64_t waitTicks;
  int32_t fences;  // PDLP_MI355X_MESH_FENCES=1: system-scope release before every flag store, acquire after every wait
  // 1: the kernels that consume an all-gather wait for the peers' flags themselves (every block polls; two launches
  // less per trial).  2 (round 6): producer and consumer of an exchange are ONE kernel — push, epoch, wait, copy: five
  // launches per trial, one per all-gather of a check, the statistics' reduction with its all-reduce.  2 is taken when
  // every rank has a GPU of its own — with ranks folded onto one device (the tests of this repository's one-GPU box) a
  // spinning grid per rank could keep the producers it waits for off the CUs, so there the wait stays a single-block
  // kernel of its own (0).  PDLP_MI355X_MESH_FUSED_WAIT=0|1|2 forces one.
  int32_t fusedWait;
};

// Host side: arena allocation, IPC rendezvous through a POSIX shared-memory
// segment named after the 128-byte communicator id, generic collectives.
class Mesh {
 public:
  // Collective over the `world` ranks (all on one node).  Throws on failure.
  Mesh(int32_t rank, int32_t world, const void* id128, int32_t n, int32_t m, const std::vector<int32_t>& rowOff,
       hipStream_t s);
  ~Mesh();
  Mesh(const Mesh&) = delete;
  Mesh& operator=(const Mesh&) = delete;

  const MeshView& view() const { return v_; }
  const MeshArgs& args() const { return args_; }
  int32_t c0() const { return v_.colOff[v_.g]; }
  int32_t c1() const { return v_.colOff[v_.g + 1]; }

  // vec[lo_h:hi_h) of every rank h -> vec of every rank (partition = colOff or rowOff)
  void allGather(double* vec, bool byRows, hipStream_t s);
  // dst[c0:c1) = sum over ranks (rank order) of their partial[c0:c1)
  void reduceScatterCols(const double* partial, double* dst, hipStream_t s);
  // buf[0:k) = sum over ranks (rank order), k <= kMeshMailDoubles; identical bits on every rank
  void allReduceScalars(double* buf, int32_t k, hipStream_t s);
  // the statistics of a sharded check: out[q] = fixed-order sum of quantity q's per-block partials (launchFinalReduce2,
  // gated by g), then summed over the ranks — one launch with fusedWait == 2, else the two steps one after the other
  void reduce2AllReduce(const double* partials, int32_t stride, int32_t nQ0, int32_t nBlocks0, int32_t nQ1, int32_t nBlocks1, double* out,
                        CheckGate g, hipStream_t s);
  // norms[0], norms[1] = fixed-order sums of the two partial arrays, then summed over the ranks
  void normsAllReduce(const double* partX, int32_t nX, const double* partY, int32_t nY, double* norms, hipStream_t s);
  // throws unless every rank holds bit-identical copies of vec[0:len) (collective; syncs the stream)
  void verifyReplicated(const double* vec, int64_t len, hipStream_t s);
  // average microseconds a hot-loop wait for the peers took since the last call, per exchange {X, P, S}, and
  // the number of waits (syncs the stream; resets the counters)
  void phaseStats(double usPerWait[3], double count[3], hipStream_t s);
  // throws if a kernel reported a timed-out wait (call after a stream sync)
  void checkError(hipStream_t s);
  // exchange self-test (pattern all-gather + reduce-scatter + scalars); false = mismatch
  bool selfTest(hipStream_t s);
  // host-level agreement: true iff every rank passed `ok`
  bool allAgree(bool ok);

 private:
  void construct(int32_t rank, int32_t world, const void* id128, int32_t n, int32_t m,
                 const std::vector<int32_t>& rowOff, hipStream_t s);
  void release() noexcept;
  void hostBarrier(int slot, double timeoutSec);
  MeshView v_{};
  MeshView* dView_ = nullptr;
  MeshArgs args_{};
  void* arena_ = nullptr;
  size_t arenaBytes_ = 0;
  MeshState* state_ = nullptr;
  double* testV_ = nullptr;  // selfTest() scratch (kept for the object's lifetime)
  double* testP_ = nullptr;
  void* shm_ = nullptr;
  size_t shmBytes_ = 0;
  std::string shmName_;
  long long epoch_ = 0;  // generic collectives (host-counted, identical on every rank)
  int agreeRound_ = 0;
  int32_t n_ = 0;
  bool setupOk_ = false;  // arenas exported and mapped on this rank
  bool ipcMapped_[kMeshMaxRanks] = {};  // peer arena mapped through HIP IPC (another process), not peer access
};

// ---- hot-loop kernels (mesh flavour of enqueueTrial) -----------------------------
// vc = column-sliced view of the iteration vectors (pointers offset by c0, n = c1-c0);
// vf = the full-length view.
int32_t meshGrid(int64_t len);  // grid size of the mesh kernels for a vector of `len`
void launchMeshPrimalStep(const IterVecs& vc, const DevState* st, const MeshArgs& dmv, hipStream_t s);
void launchMeshWaitCopyX(const IterVecs& vf, const DevState* st, const MeshArgs& dmv, hipStream_t s);
void launchMeshPushPartial(const double* partial, int32_t n, const DevState* st, const MeshArgs& dmv, hipStream_t s);
void launchMeshReduceInteract(const IterVecs& vc, const DevState* st, const MeshArgs& dmv, const double* partial,
                              double* partDX, double* partInter, int32_t nBlocks, hipStream_t s);
void launchMeshDecide(DevState* st, const MeshArgs& dmv, const double* partDY, int32_t nDY, const double* partDX,
                      const double* partInter, int32_t nDX, hipStream_t s);
// "Two all-gathers" layout (row block for A x, column block for A'y): the dual step's y+[r0:r1) is pushed into
// every peer's recvY (flag P stands for "Y" there), then recvY -> yNext outside the own rows.  yNextFull = the
// full-length y of the next parity on this rank.
void launchMeshPushY(const IterVecs& vf, const double* const yFull[2], const DevState* st, const MeshArgs& dmv, hipStream_t s);
void launchMeshWaitCopyY(double* const yFull[2], int32_t m, const DevState* st, const MeshArgs& dmv, hipStream_t s);
// MeshArgs::fusedWait == 2 (round 6): the X exchange (primal step on the own slice + push + epoch + wait + copy of the
// peers' slices) and the Y exchange (push of the own rows + epoch + wait + copy) as ONE launch each
void launchMeshPrimalX(const IterVecs& vc, double* const xFull[2], int32_t nFull, DevState* st, const MeshArgs& dmv, hipStream_t s);
void launchMeshY(double* const yFull[2], int32_t m, DevState* st, const MeshArgs& dmv, hipStream_t s);

// One sharded HiPDLP step (pdhg.cc:961-1018 over row-block shards): hFull = full-length / local-row
// pointers, hCol = the same with the column vectors offset to the own slice (rx stays full-length).
void launchMeshHalpernStep(const MatView& A, const MatView& At, const HalpernVecs& hFull, const HalpernVecs& hCol,
                           int32_t n, int32_t nLoc, double* partial, const MeshArgs& ma, hipStream_t s);

}  // namespace pdlp
