/*
 * det_math.h — TEST INFRASTRUCTURE ONLY: the oracle's OWN restatement of exp and log in plain IEEE double
 * arithmetic, used by the device-order mode of pdlp_oracle.c for the restart's primal-weight update
 * (cupdlp_step.c:147-176).  Nothing here is included from, or shared with, the product (highs_amd/): the product
 * computes the same two functions in highs_amd/csrc/pdlp_detmath.h, and a bit-exact test can only pass if two
 * separately written sources agree (tests/test_host.py also compares the two on 1.4 M points, and each against
 * long-double libm).
 *
 * Algorithm: FreeBSD/SunPro fdlibm, e_log.c and e_exp.c (the scheme every libm descends from):
 *   log: x = 2^k (1 + f), sqrt(2)/2 <= 1 + f < sqrt(2);  s = f / (2 + f);  log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)),
 *        R = Lg1 s^2 + ... + Lg7 s^14 evaluated as an even and an odd part in s^4;  log x = k ln2_hi + (... + k ln2_lo)
 *   exp: x = k ln2 + r, |r| <= ln2 / 2, r = hi - lo;  c = r - r^2 (P1 + ... + P5 r^8);
 *        exp r = 1 + (hi - (lo - r c / (2 - c)));  result scaled by 2^k
 * The coefficients Lg1..Lg7, P1..P5, ln2_hi, ln2_lo, 1/ln2 are fdlibm's.
 *   ====================================================
 *   Copyright (C) 1993, 2004 by Sun Microsystems, Inc. All rights reserved.
 *   Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice
 *   is preserved.
 *   ====================================================
 * Every operation below is ONE correctly rounded +, -, *, / (compile with -ffp-contract=off).
 */
#ifndef ORACLE_DET_MATH_H_
#define ORACLE_DET_MATH_H_
#include <stdint.h>
#include <string.h>

static inline uint64_t o_bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
static inline double o_dbl(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }
static inline double o_pow2(int e) { return o_dbl((uint64_t)(1023 + e) << 52); } /* 2^e, -1022 <= e <= 1023 */

static const double o_ln2_hi = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000 */
static const double o_ln2_lo = 1.90821492927058770002e-10; /* 0x3dea39ef35793c76 */
static const double o_inv_ln2 = 1.44269504088896338700e+00;
static const double o_Lg[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
                               1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01};
static const double o_P[5] = {1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05,
                              -1.65339022054652515390e-06, 4.13813679705723846039e-08};

static inline double o_det_log(double x) {
  if (!(x > 0.0)) return o_dbl(0x7ff8000000000000ull);
  uint64_t u = o_bits(x);
  if ((u >> 52) == 0x7ff) return x;
  int k = 0;
  if ((u >> 52) == 0) { x = x * o_pow2(54); u = o_bits(x); k = -54; }
  const uint64_t frac = u & ((1ull << 52) - 1);
  const int above = frac >= 0x6a09e667f3bcdull; /* 1.frac >= sqrt 2: halve the mantissa, raise the exponent */
  k += (int)(u >> 52) - 1023 + above;
  const double m = o_dbl(frac | ((uint64_t)(above ? 1022 : 1023) << 52));
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  double even = o_Lg[5];           /* Lg2 + w (Lg4 + w Lg6), times w */
  even = o_Lg[3] + w * even;
  even = o_Lg[1] + w * even;
  even = w * even;
  double odd = o_Lg[6];            /* Lg1 + w (Lg3 + w (Lg5 + w Lg7)), times z */
  odd = o_Lg[4] + w * odd;
  odd = o_Lg[2] + w * odd;
  odd = o_Lg[0] + w * odd;
  odd = z * odd;
  const double R = odd + even;
  const double half_f2 = 0.5 * f * f;
  const double sR = s * (half_f2 + R);
  if (k == 0) return f - (half_f2 - sR);
  const double dk = (double)k;
  return dk * o_ln2_hi - ((half_f2 - (sR + dk * o_ln2_lo)) - f);
}

static inline double o_det_exp(double x) {
  if (x != x) return x;
  if (x > 709.782712893383973096) return o_dbl(0x7ff0000000000000ull);
  if (x < -745.13321910194110842) return 0.0;
  const double q = o_inv_ln2 * x;
  int k = (int)(q < 0.0 ? q - 0.5 : q + 0.5); /* nearest integer, ties away from zero */
  const double dk = (double)k;
  const double hi = x - dk * o_ln2_hi;
  const double lo = dk * o_ln2_lo;
  const double r = hi - lo;
  const double r2 = r * r;
  double p = o_P[4];
  for (int i = 3; i >= 0; --i) p = o_P[i] + r2 * p;
  const double c = r - r2 * p;
  const double rc = r * c;
  if (k == 0) return 1.0 - (rc / (c - 2.0) - r);
  double y = 1.0 - ((lo - rc / (2.0 - c)) - hi);
  if (k > 1000) { y = y * o_pow2(1000); k -= 1000; }
  else if (k < -1000) { y = y * o_pow2(-1000); k += 1000; }
  return y * o_pow2(k);
}
#endif
