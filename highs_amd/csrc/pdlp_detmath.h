/* pdlp_detmath.h — exp and log in plain IEEE double arithmetic (+, -, *, /, bit moves; no libm, no fused
 * multiply-add), so that the HOST, the DEVICE and the test oracle's device-order mode compute the same bits.
 *
 * Why: since round 4 the restart's primal-weight update (PDHG_Compute_Step_Size_Ratio, cupdlp_step.c:147-176:
 * beta = exp(2 * (0.5 log(dD/dP) + 0.5 log(sqrt(beta))))) runs on the device, behind the check iteration, without a
 * host round trip.  The device's libm (ocml) and the host's (glibc) round exp/log differently in the last place —
 * and glibc's own variants differ between CPUs — so a solve that uses either is not reproducible across the two.
 * These two functions are fdlibm's e_log.c / e_exp.c (FreeBSD msun / SunPro): table-free argument reduction (to
 * [sqrt(2)/2, sqrt(2)) for log, to |r| <= ln2/2 for exp) + minimax polynomial, with fdlibm's coefficients (Lg1..Lg7,
 * P1..P5, ln2_hi / ln2_lo, 1/ln2) digit for digit; error below 1 ulp (checked against long-double libm in
 * tests/test_host.py), every operation a single correctly rounded IEEE operation in a fixed order.  Compile with
 * -ffp-contract=off.
 *   ====================================================
 *   Copyright (C) 1993, 2004 by Sun Microsystems, Inc. All rights reserved.
 *   Developed at SunPro / SunSoft, a Sun Microsystems, Inc. business.
 *   Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice
 *   is preserved.
 *   ====================================================
 * (The notice covers the algorithm and constants taken from fdlibm; this restatement — bit-level reduction without the
 * high / low word macros, no errno / exception paths — is this repository's.)
 *
 * The test oracle does NOT include this file: its device-order mode has a separately written restatement
 * (oracle/det_math.h), and tests/test_host.py compares the two bit for bit (pdlp_mi355x_det_exp_log). */
#ifndef PDLP_DETMATH_H_
#define PDLP_DETMATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PDLP_DET_FN __host__ __device__ static inline
#else
#define PDLP_DET_FN static inline
#endif

PDLP_DET_FN uint64_t pdlp_det_bits(double x) {
  uint64_t u;
  memcpy(&u, &x, sizeof u);
  return u;
}
PDLP_DET_FN double pdlp_det_from_bits(uint64_t u) {
  double x;
  memcpy(&x, &u, sizeof x);
  return x;
}

/* natural logarithm of a positive finite x (x <= 0 or NaN: NaN; +inf: +inf) */
PDLP_DET_FN double pdlp_det_log(double x) {
  const double ln2Hi = 6.93147180369123816490e-01; /* high part of ln 2: the low 32 bits are zero, k * ln2Hi is exact */
  const double ln2Lo = 1.90821492927058770002e-10;
  /* minimax coefficients of (log(1+f) - log(1-f)) / s - 2 in z = s^2 on [0, 0.1716^2], s = f / (2 + f) */
  const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
               L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
               L7 = 1.479819860511658591e-01;
  uint64_t u = pdlp_det_bits(x);
  int k = 0;
  if (!(x > 0.0)) return pdlp_det_from_bits(0x7ff8000000000000ull); /* <= 0 or NaN */
  if ((u >> 52) == 0x7ffu) return x;                                /* +inf */
  if ((u >> 52) == 0) {                                             /* subnormal: scale into the normal range */
    x *= 18014398509481984.0; /* 2^54 */
    u = pdlp_det_bits(x);
    k = -54;
  }
  /* x = 2^k * m with m in [sqrt(2)/2, sqrt(2)) */
  {
    const uint64_t mant = u & 0x000fffffffffffffull;
    const int e = (int)(u >> 52) - 1023;
    const int up = mant >= 0x6a09e667f3bcdull ? 1 : 0; /* mantissa of sqrt(2) */
    k += e + up;
    x = pdlp_det_from_bits(mant | ((uint64_t)(1023 - up) << 52));
  }
  {
    const double f = x - 1.0;
    const double dk = (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (L2 + w * (L4 + w * L6));
    const double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    /* log(1+f) = f - hfsq + s (hfsq + R) */
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2Hi - ((hfsq - (s * (hfsq + R) + dk * ln2Lo)) - f);
  }
}

/* e^x for finite x (overflow: +inf, underflow: 0 / subnormal) */
PDLP_DET_FN double pdlp_det_exp(double x) {
  const double ln2Hi = 6.93147180369123816490e-01, ln2Lo = 1.90821492927058770002e-10;
  const double invLn2 = 1.44269504088896338700e+00;
  /* minimax coefficients of r (e^r + 1) / (e^r - 1) = 2 + P1 r^2 + P2 r^4 + ... on |r| <= ln2 / 2 */
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double hi, lo, r, t, c, y;
  int k;
  if (x != x) return x;
  if (x > 709.782712893383973096) return pdlp_det_from_bits(0x7ff0000000000000ull);
  if (x < -745.13321910194110842) return 0.0;
  /* x = k ln2 + r, |r| <= ln2 / 2, r kept as hi - lo */
  {
    const double q = invLn2 * x;
    k = (int)(q < 0.0 ? q - 0.5 : q + 0.5);
    t = (double)k;
    hi = x - t * ln2Hi; /* exact: t has at most 11 bits, ln2Hi 21 trailing zero bits */
    lo = t * ln2Lo;
    r = hi - lo;
  }
  t = r * r;
  c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((r * c) / (c - 2.0) - r);
  y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  /* y in [sqrt(2)/2 - eps, sqrt(2) + eps): scale by 2^k in two exact steps (the result may be subnormal) */
  if (k > 1000) {
    y *= pdlp_det_from_bits((uint64_t)(1023 + 1000) << 52);
    k -= 1000;
  } else if (k < -1000) {
    y *= pdlp_det_from_bits((uint64_t)(1023 - 1000) << 52);
    k += 1000;
  }
  return y * pdlp_det_from_bits((uint64_t)(1023 + k) << 52);
}

#endif /* PDLP_DETMATH_H_ */
