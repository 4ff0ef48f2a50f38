// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// CPU restatement of the motion compensation between the two EstimatePose passes:
//   /root/reference/lidar_mapping/LidarOdometry.cpp:189-263   LidarOdometry::UndistortLidars (which pose ends a sweep; the PCD export is I/O and left out)
//   /root/reference/sensors/Velodyne.cpp:1635-1674            Velodyne::UndistortCloud (per point: slerp of the end-to-start rotation by the point's position in the sweep)
//   /root/reference/base/Geometry.hpp:572-583                 SlerpPose
// Third-party arithmetic restated ([recalled], Eigen 3.4 is not in this image) — statement for statement where Eigen's statements are known:
//   Eigen::Quaternion(Matrix3)      -> quat_of_matrix   (the trace / largest-diagonal branches of QuaternionBase::operator=(MatrixBase))
//   Eigen::Quaternion::slerp        -> slerp            (dot, 1 - epsilon threshold, acos / sin, sign of the dot on scale1)
//   Eigen::Quaternion::operator*(v) -> rotate           (uv = 2 u x v;  v + w uv + u x uv)
//   Eigen::Quaternion::toRotationMatrix -> matrix_of_quat
//   Matrix4d::inverse() of a rigid pose -> the rigid inverse (R^T, -R^T t): Eigen's 4x4 cofactor kernel rounds differently in the last bit;
//                                          the product uses the same rigid inverse, the tolerance of the GPU test (1e-6) covers a real Eigen build
// "parity unpinned": the reference has no tests for these; cross-checked against scipy's Rotation / Slerp (tests/test_undistort_cpu.py).
#pragma once
#include <cmath>
#include <cfloat>
#include <vector>

namespace oracle {
namespace undistort {

struct Quat { double x, y, z, w; };
struct Pose { double R[9]; double t[3]; };   // row-major rotation, world <- sensor

inline Quat quat_of_matrix(const double* m) {   // m row-major 3x3
  Quat q;
  double t = (m[0] + m[4]) + m[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[7] - m[5]) * t;
    q.y = (m[2] - m[6]) * t;
    q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    double c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m[3 * k + j] - m[3 * j + k]) * t;
    c[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    c[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}

inline Quat slerp(const Quat& a, double t, const Quat& b) {
  const double one = 1.0 - DBL_EPSILON;
  const double d = ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w;
  const double absD = std::fabs(d);
  double scale0, scale1;
  if (absD >= one) { scale0 = 1.0 - t; scale1 = t; }
  else {
    const double theta = std::acos(absD), sinTheta = std::sin(theta);
    scale0 = std::sin((1.0 - t) * theta) / sinTheta;
    scale1 = std::sin(t * theta) / sinTheta;
  }
  if (d < 0.0) scale1 = -scale1;
  return Quat{scale0 * a.x + scale1 * b.x, scale0 * a.y + scale1 * b.y, scale0 * a.z + scale1 * b.z, scale0 * a.w + scale1 * b.w};
}

inline void rotate(const Quat& q, const double* v, double* out) {
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  for (double& c : uv) c += c;
  const double c2[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
  for (int k = 0; k < 3; ++k) out[k] = (v[k] + q.w * uv[k]) + c2[k];
}

inline void matrix_of_quat(const Quat& q, double* m) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  m[0] = 1.0 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
  m[3] = txy + twz; m[4] = 1.0 - (txx + tzz); m[5] = tyz - twx;
  m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1.0 - (txx + tyy);
}

inline Pose inverse(const Pose& p) {
  Pose o;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o.R[3 * r + c] = p.R[3 * c + r];
  for (int r = 0; r < 3; ++r) o.t[r] = -((o.R[3 * r] * p.t[0] + o.R[3 * r + 1] * p.t[1]) + o.R[3 * r + 2] * p.t[2]);
  return o;
}
inline Pose mul(const Pose& a, const Pose& b) {
  Pose o;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o.R[3 * r + c] = (a.R[3 * r] * b.R[c] + a.R[3 * r + 1] * b.R[3 + c]) + a.R[3 * r + 2] * b.R[6 + c];
    o.t[r] = ((a.R[3 * r] * b.t[0] + a.R[3 * r + 1] * b.t[1]) + a.R[3 * r + 2] * b.t[2]) + a.t[r];
  }
  return o;
}

// base/Geometry.hpp:572-583
inline Pose SlerpPose(const Pose& pose_w1, const Pose& pose_w2, double ratio) {
  const Pose T_21 = mul(inverse(pose_w2), pose_w1);
  const Quat q_21 = quat_of_matrix(T_21.R);
  const Quat q_s1 = slerp(Quat{0, 0, 0, 1}, ratio, q_21);
  Pose T_s1;
  matrix_of_quat(q_s1, T_s1.R);
  for (int k = 0; k < 3; ++k) T_s1.t[k] = T_21.t[k] * ratio;
  return mul(pose_w1, inverse(T_s1));
}

// sensors/Velodyne.cpp:1642-1674 on a cloud of n points (x, y, z, intensity), in place; returns false when the scan has no pose or no points
inline bool UndistortCloud(float* cloud, long n, const Pose& T_wl, bool pose_valid, const Pose& T_we) {
  if (!pose_valid) return false;
  double Rt[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = T_wl.R[3 * c + r];
  double R_se[9], t_se[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) R_se[3 * r + c] = (Rt[3 * r] * T_we.R[c] + Rt[3 * r + 1] * T_we.R[3 + c]) + Rt[3 * r + 2] * T_we.R[6 + c];
    const double d[3] = {T_we.t[0] - T_wl.t[0], T_we.t[1] - T_wl.t[1], T_we.t[2] - T_wl.t[2]};
    t_se[r] = (Rt[3 * r] * d[0] + Rt[3 * r + 1] * d[1]) + Rt[3 * r + 2] * d[2];
  }
  const Quat q_se = quat_of_matrix(R_se);
  if (n <= 0) return false;
  for (long i = 0; i < n; ++i) {
    const double ratio = 1.f * i / n;                        // float: 1.f * i, then / (float)size
    const Quat q_sc = slerp(Quat{0, 0, 0, 1}, ratio, q_se);
    const double t_sc[3] = {ratio * t_se[0], ratio * t_se[1], ratio * t_se[2]};
    const double p[3] = {cloud[4 * i], cloud[4 * i + 1], cloud[4 * i + 2]};
    double r[3];
    rotate(q_sc, p, r);
    for (int k = 0; k < 3; ++k) cloud[4 * i + k] = (float)(r[k] + t_sc[k]);
  }
  return true;
}

// lidar_mapping/LidarOdometry.cpp:206-241: the pose that ends scan i's sweep, or "leave the scan as it is" (false).  pose_ok[k] = IsPoseValid(), ok[k] = valid.
// The conditions are upstream's, as written (`!IsPoseValid() && !valid` skips a neighbour only when BOTH fail; the backward search tests lidars[i].valid;
// idx <= 0 gives up).
inline bool SweepEndPose(const std::vector<Pose>& poses, const std::vector<char>& pose_ok, const std::vector<char>& ok, int i, float gap_time, Pose* out) {
  const int n = (int)poses.size();
  const double lidar_duration = 0.1;
  if (!pose_ok[i] || !ok[i]) return false;
  if (i < n - 1) {
    int idx = i + 1;
    while (idx < n && !pose_ok[idx] && !ok[idx]) idx++;
    if (idx >= n) return false;
    *out = SlerpPose(poses[i], poses[idx], lidar_duration / ((idx - i) * (lidar_duration + gap_time)));
    return true;
  }
  int idx = i - 1;
  while (idx >= 0 && !pose_ok[idx] && !ok[i]) idx--;
  if (idx <= 0) return false;
  Pose pose = SlerpPose(poses[idx], poses[i], 1.0 - lidar_duration / ((idx - i) * (lidar_duration + gap_time)));
  const Pose T_cs = mul(inverse(poses[i]), pose);
  *out = mul(poses[i], T_cs);
  return true;
}

}  // namespace undistort
}  // namespace oracle
