/* Stand-in for the HConfig.h that the reference's CMake generates
 * (highs/HConfig.h.in).  Only the two switches the cuPDLP-C C core reads:
 * CPU backend, 32-bit HighsInt.  TEST INFRASTRUCTURE (oracle/_ref build). */
#ifndef HCONFIG_H_
#define HCONFIG_H_
#define CUPDLP_CPU
/* #undef CUPDLP_GPU */
/* #undef HIGHSINT64 */
#endif
