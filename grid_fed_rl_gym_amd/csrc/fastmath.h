// fastmath.h -- the three elementary functions of the environment prologue, written out.
//
// The load noise of one env step is 43 Box-Muller pairs per instance on the IEEE-123 feeder; with libm's log / sincos
// (general arguments: ~120 / ~250 vector instructions each) that phase was bound by FP64 VALU issue.  The arguments
// here are special -- a uniform in (0, 1], an angle that is a known fraction of a turn, a divisor that is the same
// for every instance -- so range reduction is exact and short:
//   gs_log01(u)            natural logarithm, u in (0, 1] (any positive normal double in fact)
//   gs_sincos_turns(t,..)  sin and cos of 2 pi t, |t| < 2^30: the quadrant comes from 4 t exactly
//   gs_fmod_pos(x, d, rd)  fmod(x, d) for x >= 0, exactly (the remainder of an fma, then one correction step)
//   gs_div_by(x, d, rd)    x / d with rd = RN(1 / d) supplied by the host: correctly rounded (Markstein's
//                          theorem: q = RN(x rd) is faithful, r = x - d q is exact in an fma, RN(q + r rd) = RN(x / d))
// Kernels follow fdlibm's k_sin / k_cos / e_log polynomials; max error ~1 ulp (tests/test_fastmath.py compiles this
// header for the host with g++ and compares with libm / true division on 10^6 points: the arithmetic is the same on
// both sides because every multiply-add is an explicit fma).
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD static inline
#endif
// No implicit contraction inside these functions: which multiply the compiler would fuse into which add depends on the
// code around the inlined copy, and two kernels that inline the same function must give the same bits (the step
// kernels with and without the fused checks are compared bit for bit).  Every fused multiply-add is written out.
#ifdef __clang__
#define GS_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define GS_NO_CONTRACT
#endif

GS_HD double gs_log01(double x) {
  GS_NO_CONTRACT
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int k;
  double m = __builtin_frexp(x, &k);                 // x = m 2^k, m in [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }       // m in [sqrt(1/2), sqrt(2))
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double R = Lg7;
  R = __builtin_fma(R, z, Lg6); R = __builtin_fma(R, z, Lg5); R = __builtin_fma(R, z, Lg4);
  R = __builtin_fma(R, z, Lg3); R = __builtin_fma(R, z, Lg2); R = __builtin_fma(R, z, Lg1);
  R *= z;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  // log(1 + f) = f - hfsq + s (hfsq + R);  log x = k ln2 + log(1 + f)
  return __builtin_fma(dk, ln2_hi, f - (hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)));
}

GS_HD void gs_sincos_turns(double t, double* sn, double* cs) {
  GS_NO_CONTRACT
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double q4 = 4.0 * t;                          // exact
  const double n = __builtin_rint(q4);
  const double x = (q4 - n) * 1.57079632679489661923;  // (q4 - n) exact, in [-1/2, 1/2]; x in [-pi/4, pi/4]
  const double z = x * x;
  double ps = S6;
  ps = __builtin_fma(ps, z, S5); ps = __builtin_fma(ps, z, S4); ps = __builtin_fma(ps, z, S3);
  ps = __builtin_fma(ps, z, S2); ps = __builtin_fma(ps, z, S1);
  const double s = __builtin_fma(x * z, ps, x);
  double pc = C6;
  pc = __builtin_fma(pc, z, C5); pc = __builtin_fma(pc, z, C4); pc = __builtin_fma(pc, z, C3);
  pc = __builtin_fma(pc, z, C2); pc = __builtin_fma(pc, z, C1);
  const double c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const int q = (int)n & 3;                           // two's complement: right for negative n as well
  const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}

// x / d for a divisor d shared by many numerators, rd = RN(1 / d) (computed once, by a true division).
GS_HD double gs_div_by(double x, double d, double rd) {
  GS_NO_CONTRACT
  const double q = x * rd;
  const double r = __builtin_fma(-d, q, x);
  const double q1 = __builtin_fma(r, rd, q);
  return (__builtin_fabs(q) < __builtin_inf()) ? q1 : q;      // an infinite or NaN quotient stays what it is
}

// fmod(x, d) for x >= 0, d > 0, rd = RN(1 / d), x / d < 2^52: q is the true quotient's floor or one off, x - d q is
// exact in the fma (it is a multiple of ulp(x) below 2 d), and one conditional step brings it into [0, d).
GS_HD double gs_fmod_pos(double x, double d, double rd) {
  GS_NO_CONTRACT
  const double q = __builtin_floor(x * rd);
  double r = __builtin_fma(-d, q, x);
  if (r < 0.0) r += d;
  else if (r >= d) r -= d;
  return r;
}
