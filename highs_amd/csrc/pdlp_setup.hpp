// pdlp_setup.hpp — GPU-side problem preparation: the work that
// the reference does on the host before the first PDHG iteration —
// formulateLP_highs (CupdlpWrapper.cpp:280-448), Ruiz + Pock-Chambolle scaling
// (cupdlp_scaling.c), both matrix orientations (cupdlp_cs.c:189) — plus this
// library's slab layouts, done on the device.  At 1M x 1M / 8M nnz the host
// path costs ~1.2 s on the GPU box (the reference: ~1.5 s), i.e. thousands of
// GPU iterations; here it is a few radix sorts and streaming passes.
//
// The results are BIT-IDENTICAL to the host path (pdlp_host.cpp), which is
// itself bit-identical to the oracle and the reference: every reduction whose
// order matters (Pock-Chambolle row/column sums) is done by one thread per
// major in the reference's traversal order; max-reductions are order-free;
// sqrt and division are IEEE-correct on gfx950.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "pdlp_device.hpp"
#include "pdlp_host.hpp"
#include "pdlp_kernels.hpp"

namespace pdlp {

// Compressed matrix in HBM; `major[p]` = major index of entry p (kept for the
// scaling passes and the slab-layout sort).
struct DeviceCsrData {
  DeviceArray<int32_t> beg, idx, major;
  DeviceArray<double> val;
  int32_t nMajor = 0, nMinor = 0;
  int64_t nnz = 0;
};

struct DeviceProblem {
  int32_t n = 0, m = 0, n0 = 0, nEqs = 0;
  int64_t nnz = 0;
  bool scaled = false;
  double offset = 0.0, sense = 1.0;
  DeviceCsrData A;   // rows, ascending column
  DeviceCsrData At;  // columns, ascending row
  DeviceArray<double> cost, rhs, lower, upper, colScale, rowScale;
  DeviceArray<double> qdiag;      // QP only (cuPDLP-C form): diagonal of Q with the sense, scaled with the columns
  DeviceArray<double> rowUpper;   // HiPDLP form only (rhs then holds the row lower bounds)
  DeviceArray<uint8_t> rowIsEq;   // HiPDLP form only, per permuted row
  // host copies of what the host side of the solver needs
  std::vector<int32_t> rowKind, rowNewIdx;
  std::vector<double> hColScale, hRowScale;
  double normCost = 0, normRhs = 0, matNormInf = 0;
  double sumCost2 = 0, sumRhs2 = 0;  // left-to-right sums of the SCALED c, b (PDHG_Init_Step_Sizes)
};

// Options of the HiPDLP form (pdlp_host.hpp formulateHipdlp / scaleHipdlp); nullptr = cuPDLP-C form.
struct HipdlpSetup {
  bool ruiz = true, pc = true, l2 = false;
  int ruizIters = 10;
};
// Formulate + scale + both orientations on the device.  Throws std::runtime_error.
void gpuPrepare(const pdlp_problem_t& P, bool doScale, hipStream_t s, DeviceProblem& out,
                const HipdlpSetup* hipdlp = nullptr);

// Slab layout (pdlp_host.hpp SlabLayout) built on the device from a device CSR.
struct DeviceSlabLayout {
  int32_t rowsPerBlock = 0, nBlocks = 0, minorBits = 0, nLong = 0;  // rowsPerBlock: most majors in one block
  int64_t nnzShort = 0;
  std::vector<int32_t> hostWaveBeg;  // the partition (pdlp_host.hpp slabPartition), computed on the host from the major starts
  DeviceArray<int32_t> wavePtr, waveBeg;
  DeviceArray<uint32_t> ent, longMask;
  DeviceArray<double> val;
  DeviceCsrData longCsr;            // compacted long majors (major[] unused)
  DeviceArray<int32_t> longMap;     // compact index -> major
  std::vector<int32_t> hostLongBeg; // for the stream plan of the side kernel
};
// gpuSlabPartition: the partition alone (nBlocks, minorBits, rowsPerBlock, hostWaveBeg, waveBeg); gpuBuildSlabLayout
// computes it itself when `out` does not hold one yet.
void gpuSlabPartition(const DeviceCsrData& M, int32_t longLimit, int32_t majorCost, hipStream_t s, DeviceSlabLayout& out);
void gpuBuildSlabLayout(const DeviceCsrData& M, int32_t longLimit, int32_t slabWidthLog2, int32_t majorCost, hipStream_t s,
                        DeviceSlabLayout& out);

}  // namespace pdlp
