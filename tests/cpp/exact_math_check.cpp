// Exhaustive proof of the two identities panovlm_amd/csrc/pvlm_exact_math.h relies on: for EVERY finite float (2^32 - 2^24
// bit patterns) the cheap sequence gives the same double / float as the reference's statement.  Test infrastructure;
// includes the product header so that the code under test is the code that ships.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <omp.h>
#include "../../panovlm_amd/csrc/pvlm_exact_math.h"
int main() {
  const double PI = 3.14159265358979323846, TWO_PI = 2.0 * PI;
  long long bad_div = 0, bad_sqrt = 0, n = 0;
#pragma omp parallel for reduction(+ : bad_div, bad_sqrt, n) schedule(static)
  for (long long b = 0; b < (1ll << 32); ++b) {
    uint32_t u = (uint32_t)b; float f; std::memcpy(&f, &u, 4);
    if (!std::isfinite(f)) continue;
    ++n;
    const double x = (double)f;
    const double a1 = x / TWO_PI, a2 = pvlm_exact::div_two_pi(f);
    const double b1 = x / PI, b2 = pvlm_exact::div_pi(f);
    // compared the way the callers use them: 0.5 + q (a signed zero may differ, the sum does not)
    if ((0.5 + a1) != (0.5 + a2) || (a1 != a2 && !(a1 == 0 && a2 == 0))) ++bad_div;
    if ((0.5 - b1) != (0.5 - b2) || (b1 != b2 && !(b1 == 0 && b2 == 0))) ++bad_div;
    if (f >= 0) { const float s1 = (float)std::sqrt(x), s2 = pvlm_exact::sqrt_via_double(f); if (std::memcmp(&s1, &s2, 4)) ++bad_sqrt; }
  }
  printf("floats %lld  division mismatches %lld  sqrt mismatches %lld\n", n, bad_div, bad_sqrt);
  // float division by an image size, judged after the step that follows it in ImageToSphere: (2 u / cols - 1), (0.5 - v / rows)
  long long bad_f32 = 0, raw_f32 = 0;
  const int sizes[] = {5760};
  for (int c : sizes) {
    const float cf = (float)c, rc = 1.0f / cf;
#pragma omp parallel for reduction(+ : bad_f32, raw_f32) schedule(static)
    for (long long b = 0; b < (1ll << 32); ++b) {
      uint32_t u = (uint32_t)b; float f; std::memcpy(&f, &u, 4);
      if (!std::isfinite(f)) continue;
      const float a1 = f / cf, a2 = pvlm_exact::div_f32(f, cf, rc);
      if (std::memcmp(&a1, &a2, 4)) { ++raw_f32; if ((a1 - 1) != (a2 - 1) || (0.5 - a1) != (0.5 - a2) || std::fabs(a1) >= 1.1754944e-38f) ++bad_f32; }
    }
  }
  printf("float division by 5760: %lld subnormal or signed-zero quotients differ, %lld visible\n", raw_f32, bad_f32);
  return (bad_div || bad_sqrt || bad_f32) ? 1 : 0;
}
