// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// Restates the angle-axis helpers of Ceres 2.0.0 `include/ceres/rotation.h` (un-vendored
// third-party dependency of the reference; version pinned by /root/reference/README.md:22-26).
// Call sites in the reference: base/CostFunction.h:595-601,659-665,802-808,869-875 (the P_r chain),
// :372-400,469-483 (Plane2Plane_Global / PlaneIOUResidual), lidar_mapping/LidarOdometry.cpp:31,105.
// The formulas below are the published Ceres algorithms ([recalled] — Ceres sources are absent
// from this image): Rodrigues with the first-order branch at theta^2 <= DBL_EPSILON, Shoemake
// rotation-matrix -> quaternion, quaternion -> angle-axis via atan2 with the q0<0 flip.
// "parity unpinned": the reference holds no test vectors for these.
#pragma once
#include <limits>
#include "jet.hpp"

namespace oracle {

// R is COLUMN-MAJOR 3x3 (Eigen default, what `R.data()` hands to ceres in CostFunction.h:595).
template <typename T>
inline void AngleAxisToRotationMatrix(const T* aa, T* R) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  const T kOne(1.0);
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const T c = cos(theta), s = sin(theta);
    R[0] = c + wx * wx * (kOne - c);
    R[1] = wz * s + wx * wy * (kOne - c);
    R[2] = -wy * s + wx * wz * (kOne - c);
    R[3] = wx * wy * (kOne - c) - wz * s;
    R[4] = c + wy * wy * (kOne - c);
    R[5] = wx * s + wy * wz * (kOne - c);
    R[6] = wy * s + wx * wz * (kOne - c);
    R[7] = -wx * s + wy * wz * (kOne - c);
    R[8] = c + wz * wz * (kOne - c);
  } else {
    R[0] = kOne;   R[1] = aa[2];  R[2] = -aa[1];
    R[3] = -aa[2]; R[4] = kOne;   R[5] = aa[0];
    R[6] = aa[1];  R[7] = -aa[0]; R[8] = kOne;
  }
}

template <typename T>
inline void QuaternionToAngleAxis(const T* q, T* aa) {
  const T& q1 = q[1]; const T& q2 = q[2]; const T& q3 = q[3];
  const T s2 = q1 * q1 + q2 * q2 + q3 * q3;
  if (s2 > T(0.0)) {
    const T s = sqrt(s2);
    const T& c = q[0];
    const T two_theta = T(2.0) * ((c < T(0.0)) ? atan2(-s, -c) : atan2(s, c));
    const T k = two_theta / s;
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  } else {
    const T k(2.0);
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  }
}

// R column-major: element (r,c) = R[r + 3*c].
template <typename T>
inline void RotationMatrixToQuaternion(const T* R, T* q) {
  auto M = [&](int r, int c) -> const T& { return R[r + 3 * c]; };
  const T trace = M(0, 0) + M(1, 1) + M(2, 2);
  if (trace >= 0.0) {
    T t = sqrt(trace + T(1.0));
    q[0] = T(0.5) * t;
    t = T(0.5) / t;
    q[1] = (M(2, 1) - M(1, 2)) * t;
    q[2] = (M(0, 2) - M(2, 0)) * t;
    q[3] = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    T t = sqrt(M(i, i) - M(j, j) - M(k, k) + T(1.0));
    q[i + 1] = T(0.5) * t;
    t = T(0.5) / t;
    q[0] = (M(k, j) - M(j, k)) * t;
    q[j + 1] = (M(j, i) + M(i, j)) * t;
    q[k + 1] = (M(k, i) + M(i, k)) * t;
  }
}

template <typename T>
inline void RotationMatrixToAngleAxis(const T* R, T* aa) {
  T q[4];
  RotationMatrixToQuaternion(R, q);
  QuaternionToAngleAxis(q, aa);
}

template <typename T>
inline void AngleAxisRotatePoint(const T* aa, const T* pt, T* result) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T c = cos(theta), s = sin(theta);
    const T ti = T(1.0) / theta;
    const T w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
    const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - c);
    const T r0 = pt[0] * c + wxp[0] * s + w[0] * tmp;
    const T r1 = pt[1] * c + wxp[1] * s + w[1] * tmp;
    const T r2 = pt[2] * c + wxp[2] * s + w[2] * tmp;
    result[0] = r0; result[1] = r1; result[2] = r2;
  } else {
    const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    const T r0 = pt[0] + wxp[0], r1 = pt[1] + wxp[1], r2 = pt[2] + wxp[2];
    result[0] = r0; result[1] = r1; result[2] = r2;
  }
}

}  // namespace oracle
