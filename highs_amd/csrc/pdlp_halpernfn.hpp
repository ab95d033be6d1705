// pdlp_halpernfn.hpp — the scalar logic of HiPDLP's check iteration, ONCE, for the host and for the device: the decision
// kernel of the device-driven loop (pdlp_halpern.hip k_h_decide) and the host-driven loop (sharded solves, profile mode;
// pdlp_halpern.cpp) call the same function on the same sums, so the two cannot differ in a bit.  Reference:
// hipdlp/pdhg.cc:578-707 (the block loop), :709-739 (fixed-point error), :1474-1527 (checkConvergence), :901-927
// (checkRestartCriteria, factors restart.hpp:91-93), :1979-2049 (updatePrimalWeightAtRestart, PID).  log / exp are the
// plain-arithmetic functions of pdlp_detmath.h (the same bits on host and device; the reference calls libm).
#pragma once
#include <math.h>

#include "pdlp_detmath.h"
#include "pdlp_kernels.hpp"

namespace pdlp {

#if defined(__HIPCC__) || defined(__HIP__)
#define PDLP_HD __host__ __device__ inline
#else
#define PDLP_HD inline
#endif

PDLP_HD double halpernFpe(const HalpernState& s, const double* h) {  // computeFixedPointError from its three sums
  const double dn = h[0], pn = h[1], cross = h[2];
  const double movement = pn * s.omega + dn / s.omega;
  const double interaction = 2.0 * s.eta * cross;
  const double v = movement + interaction;
  return sqrt(v > 0.0 ? v : 0.0);
}

// checkConvergence on the six sums (row sums rs[0..1], column sums cs[0..3])
PDLP_HD bool halpernResiduals(const HalpernState& s, const double* stat, HalpernRecord& r) {
  const double* rs = stat + kHSlotCheck;
  const double* cs = rs + 2;
  r.pFeas = sqrt(rs[0]);
  r.dFeas = sqrt(cs[0]);
  r.pObj = s.offset + cs[1];
  r.dObj = ((s.offset + rs[1]) + cs[2]) - cs[3];
  const double gap = r.pObj - r.dObj;
  r.gap = fabs(gap);
  r.relGap = fabs(gap) / (1.0 + fabs(r.pObj) + fabs(r.dObj));
  return r.pFeas < s.tol * (1.0 + s.normRhs) && r.dFeas < s.tol * (1.0 + s.normCost) && r.relGap < s.tol;
}

// updatePrimalWeightAtRestart (k_p 0.99, k_i 0.01, k_d 0, i_smooth 0.3); dist2: |x_next - x_anchor|^2, |y_next - y_anchor|^2
PDLP_HD void halpernUpdateWeight(HalpernState& s, const HalpernRecord& r, const double* dist2) {
  const double primalDist = sqrt(dist2[0]), dualDist = sqrt(dist2[1]);
  const double relP = r.pFeas / (1.0 + s.normRhs), relD = r.dFeas / (1.0 + s.normCost);
  const double ratio = relP > 0.0 ? relD / relP : 1e300;
  if (primalDist > 1e-16 && dualDist > 1e-16 && primalDist < 1e12 && dualDist < 1e12 && ratio > 1e-8 && ratio < 1e8) {
    const double err = pdlp_det_log(dualDist) - pdlp_det_log(primalDist) - pdlp_det_log(s.primalWeight);
    s.errSum = 0.3 * s.errSum + err;
    const double dErr = err - s.lastErr;
    s.primalWeight *= pdlp_det_exp(0.99 * err + 0.01 * s.errSum + 0.0 * dErr);
    s.lastErr = err;
  } else {
    s.primalWeight = s.bestPrimalWeight;
    s.errSum = 0.0;
    s.lastErr = 0.0;
  }
  // |log10(relD / relP)| as log / ln 10 (0.4342944819032518 = 1 / ln 10)
  const double gap = (relP > 0.0 && relD > 0.0) ? fabs(pdlp_det_log(relD / relP) * 0.4342944819032518) : s.bestGap;
  if (gap < s.bestGap) { s.bestGap = gap; s.bestPrimalWeight = s.primalWeight; }
  const double eta = sqrt(s.tau * s.sigma);
  s.tau = eta / s.primalWeight;
  s.sigma = eta * s.primalWeight;
  s.omega = sqrt(s.primalWeight * s.primalWeight);  // params_.omega = primal_weight_, then RestartScheme::updateBeta
}

// The end of one block of 40 steps: fixed-point errors, convergence, restart criteria, primal weight.  `stat`: the
// statistics vector of the check (slots kHSlot*; summed over the ranks when sharded).  Returns the record of the check.
PDLP_HD HalpernRecord halpernDecide(HalpernState& s, const double* stat) {
  constexpr int kInterval = 40;  // PDHG_CHECK_INTERVAL, pdhg.cc:32
  HalpernRecord r;
  if (s.fpe0Pending) s.initialFpe = halpernFpe(s, stat + kHSlotFpe0);
  s.fpe = halpernFpe(s, stat + kHSlotFpe);
  s.hIter += kInterval;
  s.iters += kInterval;
  const bool converged = halpernResiduals(s, stat, r);
  s.nChecks += 1;
  r.iters = s.iters; r.fpe = s.fpe; r.primalWeight = s.primalWeight; r.restarted = 0; r.converged = converged ? 1 : 0;
  s.doRestart = 0;
  s.runFpe0 = 0;
  s.fpe0Pending = 0;
  if (converged && s.terminate) {
    s.converged = 1; s.halted = 1; s.run = 0; s.termStatus = 0;
    return r;
  }
  bool restart = false;  // checkRestartCriteria
  if (s.iters == kInterval) restart = true;
  else if (s.iters > kInterval) {
    if (s.fpe <= 0.2 * s.initialFpe) restart = true;
    else if (s.fpe <= 0.8 * s.initialFpe && s.fpe > s.lastTrialFpe) restart = true;
    else if ((double)s.hIter >= 0.36 * (double)s.iters) restart = true;
  }
  s.lastTrialFpe = s.fpe;
  if (restart) {
    if (s.pid) halpernUpdateWeight(s, r, stat + kHSlotDiff);
    s.hIter = 0;
    s.lastTrialFpe = INFINITY;
    s.nRestarts += 1;
    s.doRestart = 1;
    r.restarted = 1;
  }
  if (s.iters >= s.iterLimit) {  // (the restart copies of the last block still run: gated by doRestart, not by run)
    s.halted = 1; s.run = 0;
    if (s.terminate) s.termStatus = 1;
  }
  s.fpe0Pending = restart ? 1 : 0;
  s.runFpe0 = (restart && !s.halted) ? 1 : 0;
  return r;
}

}  // namespace pdlp
