// Cheaper instruction sequences that return the SAME floats as the reference's statements, each proven exhaustively over
// every finite float argument by tests/cpp/exact_math_check.cpp (run by tests/test_mvs_cpu.py, 2^32 cases, seconds):
//
//  * x / c for a double x that holds a float and the constants c = pi, 2 pi (SphereToImage, sensors/Equirectangular.h:84-85):
//    q = x * RN(1/c), one exact-remainder correction r = fma(-q, c, x), q' = fma(r, RN(1/c), q) is the correctly rounded
//    quotient (Markstein) — three multiply-add class instructions instead of the ~12-instruction IEEE double division
//    (v_div_scale x2, v_rcp_f64, Newton steps, v_div_fmas, v_div_fixup).  Only the sign of a zero quotient can differ, and
//    every caller adds the quotient to 0.5.
//  * (float)sqrt((double)v) for a float v (CamToSphere, :50-51): equal to the correctly rounded float square root, because
//    rounding a 53-bit square root to 24 bits never double-rounds (53 >= 2 * 24 + 2).
//  * x / c in FLOAT for c = (float)cols, (float)rows (ImageToSphere, :102-103): the same correction in single precision gives
//    the correctly rounded quotient whenever it is a normal number; a subnormal quotient (|x / c| < 2^-126) can differ in its
//    last bits and -0 becomes +0, and both callers immediately form (q - 1) resp. (0.5 - q), which absorbs either.
//    exact_math_check.cpp counts, for every finite float x and several sizes, the x where the value AFTER that step differs: 0.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PVLM_XHD __host__ __device__
#else
#define PVLM_XHD
#endif

namespace pvlm_exact {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * 3.14159265358979323846;
constexpr double kInvPi = 1.0 / kPi;        // correctly rounded at compile time
constexpr double kInvTwoPi = 1.0 / kTwoPi;

PVLM_XHD inline double div_const(double x, double c, double inv_c) {
  const double q = x * inv_c;
  const double r = fma(-q, c, x);
  return fma(r, inv_c, q);
}
PVLM_XHD inline double div_pi(float x) { return div_const((double)x, kPi, kInvPi); }
PVLM_XHD inline double div_two_pi(float x) { return div_const((double)x, kTwoPi, kInvTwoPi); }

PVLM_XHD inline float div_f32(float x, float c, float inv_c) {
  const float q = x * inv_c;
  const float r = fmaf(-q, c, x);
  return fmaf(r, inv_c, q);
}

// sqrtf is llvm.sqrt.f32 on the device, which hipcc lowers to the correctly rounded sequence by default
// (-fhip-fp32-correctly-rounded-divide-sqrt); HIP's __fsqrt_rn is NOT: without OCML_BASIC_ROUNDED_OPERATIONS it is the 1-ulp
// v_sqrt_f32 (found the hard way: 0.5 % of the PatchMatch pixels changed).
PVLM_XHD inline float sqrt_via_double(float v) { return sqrtf(v); }

}  // namespace pvlm_exact
