// Motion compensation of a LiDAR sweep (between the two EstimatePose passes of main.cpp:415-432):
//   Velodyne::UndistortCloud   sensors/Velodyne.cpp:1642-1674 — point i of n gets the share i / n of the motion from the sweep's start to its end:
//                              q_sc = Identity.slerp(i / n, q_se), t_sc = (i / n) t_se, p <- q_sc p + t_sc
//   SlerpPose                  base/Geometry.hpp:572-583
// The quaternion routines are Eigen's, restated ([recalled]: Quaternion(Matrix3), slerp with its 1 - epsilon threshold, q * v as v + w uv + u x uv,
// toRotationMatrix) — host/device, double; compiled with -ffp-contract=off like the reference's x86-64 build.
#pragma once
#include <cfloat>
#include <cmath>

#ifndef PVLM_UD
#if defined(__HIPCC__)
#define PVLM_UD __host__ __device__ inline
#else
#define PVLM_UD inline
#endif
#endif

namespace pvlm_undistort {

struct Quat { double x, y, z, w; };

PVLM_UD Quat quat_of_matrix(const double* m) {   // row-major 3x3 rotation
  Quat q;
  double t = (m[0] + m[4]) + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t; t = 0.5 / t;
    q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    return q;
  }
  int i = 0;
  if (m[4] > m[0]) i = 1;
  if (m[8] > m[4 * i]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
  double c[3];
  c[i] = 0.5 * t; t = 0.5 / t;
  q.w = (m[3 * k + j] - m[3 * j + k]) * t;
  c[j] = (m[3 * j + i] + m[3 * i + j]) * t;
  c[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  q.x = c[0]; q.y = c[1]; q.z = c[2];
  return q;
}

// Identity.slerp(t, b) needs, of b, only what does not depend on t: the per-sweep part ...
struct SlerpFromIdentity { Quat b; double theta, sin_theta; int linear, negate; };
PVLM_UD SlerpFromIdentity slerp_prepare(const Quat& b) {
  SlerpFromIdentity s;
  s.b = b;
  const double d = ((0.0 * b.x + 0.0 * b.y) + 0.0 * b.z) + 1.0 * b.w, a = fabs(d);
  s.linear = a >= 1.0 - DBL_EPSILON;
  s.theta = s.linear ? 0.0 : acos(a);
  s.sin_theta = s.linear ? 1.0 : sin(s.theta);
  s.negate = d < 0.0;
  return s;
}
// ... and the per-point part
PVLM_UD Quat slerp_at(const SlerpFromIdentity& s, double t) {
  double scale0, scale1;
  if (s.linear) { scale0 = 1.0 - t; scale1 = t; }
  else { scale0 = sin((1.0 - t) * s.theta) / s.sin_theta; scale1 = sin(t * s.theta) / s.sin_theta; }
  if (s.negate) scale1 = -scale1;
  return Quat{scale0 * 0.0 + scale1 * s.b.x, scale0 * 0.0 + scale1 * s.b.y, scale0 * 0.0 + scale1 * s.b.z, scale0 * 1.0 + scale1 * s.b.w};
}

PVLM_UD void rotate(const Quat& q, const double* v, double* out) {
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c2[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
  for (int k = 0; k < 3; ++k) out[k] = (v[k] + q.w * uv[k]) + c2[k];
}

PVLM_UD void matrix_of_quat(const Quat& q, double* m) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  m[0] = 1.0 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
  m[3] = txy + twz; m[4] = 1.0 - (txx + tzz); m[5] = tyz - twx;
  m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1.0 - (txx + tyy);
}

// what a sweep's points share: the motion from the sweep's end back to its start, in the start frame (:1647-1649)
struct Sweep { SlerpFromIdentity s; double t_se[3]; };
PVLM_UD Sweep sweep_of(const double* R_wl, const double* t_wl, const double* R_we, const double* t_we) {
  double R_se[9];
  Sweep w;
  const double d[3] = {t_we[0] - t_wl[0], t_we[1] - t_wl[1], t_we[2] - t_wl[2]};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) R_se[3 * r + c] = (R_wl[r] * R_we[c] + R_wl[3 + r] * R_we[3 + c]) + R_wl[6 + r] * R_we[6 + c];
    w.t_se[r] = (R_wl[r] * d[0] + R_wl[3 + r] * d[1]) + R_wl[6 + r] * d[2];
  }
  w.s = slerp_prepare(quat_of_matrix(R_se));
  return w;
}
// point i of n (:1656-1661)
PVLM_UD void undistort_point(const Sweep& w, int i, int n, const float* in, float* out) {
  const double ratio = (double)(1.f * (float)i / (float)n);
  const Quat q = slerp_at(w.s, ratio);
  const double p[3] = {(double)in[0], (double)in[1], (double)in[2]};
  double r[3];
  rotate(q, p, r);
  for (int k = 0; k < 3; ++k) out[k] = (float)(r[k] + ratio * w.t_se[k]);
}

}  // namespace pvlm_undistort
