// pdlp_comm.cpp — RCCL (xGMI) all-reduce for the row-block sharded path.
// librccl is dlopen'ed on first use so the single-GPU path (and a CPU-only
// import of the library for symbol checks) never needs it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "pdlp_solver.hpp"

namespace pdlp {

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  });
  if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy)
    throw std::runtime_error("pdlp_mi355x: librccl.so could not be loaded (needed for num GPUs > 1)");
  return r;
}

void check(ncclResult_t rc, const char* what) {
  if (rc != ncclSuccess) {
    Rccl& r = rccl();
    throw std::runtime_error(std::string("RCCL error in ") + what + ": " +
                             (r.GetErrorString ? r.GetErrorString(rc) : "?"));
  }
}
}  // namespace

void Comm::uniqueId(void* id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId must be 128 bytes");
  ncclUniqueId id;
  check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
}

Comm::Comm(int32_t rank, int32_t world, const void* id128) : rank_(rank), world_(world) {
  if (!id128) throw std::runtime_error("sharded solver needs an ncclUniqueId");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  check(rccl().CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  comm_ = c;
}

Comm::~Comm() {
  if (comm_) (void)rccl().CommDestroy((ncclComm_t)comm_);
}

void Comm::allReduceSum(double* buf, size_t count, hipStream_t s) {
  check(rccl().AllReduce(buf, buf, count, ncclDouble, ncclSum, (ncclComm_t)comm_, s), "ncclAllReduce");
}

}  // namespace pdlp
