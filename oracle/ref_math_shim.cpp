// ORACLE / _ref — compiles the REAL reference header /root/reference/base/Math.h (FastAtan2,
// Square) where it lies; nothing of it is copied into this repo. Math.h has no #includes of its
// own (its includers provide <cmath>/<cfloat>/<algorithm>), so this shim supplies exactly those
// standard headers and exports C symbols for ctypes. Output: oracle/_ref/libref_math.so
// (git-ignored, travels to the GPU box). Every other reference file on the hot path needs
// Eigen/PCL/OpenCV/Ceres (absent here) and is therefore unbuildable in this image.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include REF_MATH_H

extern "C" {
float ref_fast_atan2_f(float y, float x) { return FastAtan2<float>(y, x); }
double ref_fast_atan2_d(double y, double x) { return FastAtan2<double>(y, x); }
void ref_fast_atan2_f_vec(long n, const float* y, const float* x, float* o) { for (long i = 0; i < n; ++i) o[i] = FastAtan2<float>(y[i], x[i]); }
void ref_fast_atan2_d_vec(long n, const double* y, const double* x, double* o) { for (long i = 0; i < n; ++i) o[i] = FastAtan2<double>(y[i], x[i]); }
double ref_square_d(double a) { return Square(a); }
}
