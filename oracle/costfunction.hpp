// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// CPU restatement of the six residual functors on the hot path, evaluated like
// ceres::AutoDiffCostFunction<F,1,3,3,3,3> does (Jet<12> seeded on the 4 parameter blocks):
//   /root/reference/base/CostFunction.h:567-619  Point2Plane_Meter
//   /root/reference/base/CostFunction.h:630-729  Point2Plane_Angle
//   /root/reference/base/CostFunction.h:769-829  Point2Line_Meter
//   /root/reference/base/CostFunction.h:836-934  Point2Line_Angle
//   /root/reference/base/CostFunction.h:350-425  Plane2Plane_Global
//   /root/reference/base/CostFunction.h:433-507  PlaneIOUResidual
//   /root/reference/base/CostFunction.h:294-348  Plane2Plane_Relative, :509-565 PlaneRelativeIOUResidual (calibration mode)
// plus ceres::HuberLoss + the Ceres corrector ([recalled] Ceres 2.0.0 loss_function.cc,
// corrector.cc; call sites util/Optimization.cpp:513-517,336-340).
// "parity unpinned": the reference ships no tests for these; cross-checked here against an
// independent torch.float64 autograd transcription (tests/test_oracle_crosscheck.py).
#pragma once
#include <algorithm>
#include "geometry.hpp"
#include "rotation.hpp"

namespace oracle {

// P_r = R(aa_rw) R(-aa_nw) (P_n - t_nw) + t_rw, computed through the reference's
// matrix -> angle-axis -> Rodrigues round trip (CostFunction.h:585-604).
template <typename T>
inline void TransformNeighborToRef(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw,
                                   const double* curr_point, T* point_ref) {
  T aa_rn[3], aa_wn[3], vec_tmp[3];
  T point[3] = {T(curr_point[0]), T(curr_point[1]), T(curr_point[2])};
  aa_wn[0] = T(-1.0) * aa_nw[0]; aa_wn[1] = T(-1.0) * aa_nw[1]; aa_wn[2] = T(-1.0) * aa_nw[2];
  T R_rw[9], R_wn[9], R_rn[9];
  AngleAxisToRotationMatrix(aa_rw, R_rw);
  AngleAxisToRotationMatrix(aa_wn, R_wn);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      R_rn[r + 3 * c] = R_rw[r + 0] * R_wn[0 + 3 * c] + R_rw[r + 3] * R_wn[1 + 3 * c] + R_rw[r + 6] * R_wn[2 + 3 * c];
  RotationMatrixToAngleAxis(R_rn, aa_rn);
  AngleAxisRotatePoint(aa_rn, point, point_ref);
  AngleAxisRotatePoint(aa_rn, t_nw, vec_tmp);
  point_ref[0] = point_ref[0] - vec_tmp[0] + t_rw[0];
  point_ref[1] = point_ref[1] - vec_tmp[1] + t_rw[1];
  point_ref[2] = point_ref[2] - vec_tmp[2] + t_rw[2];
}

struct Point2Plane_Meter {
  double plane[4], curr_point[3], weight;
  template <typename T>
  bool operator()(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw, T* residual) const {
    T point_ref[3];
    TransformNeighborToRef(aa_rw, t_rw, aa_nw, t_nw, curr_point, point_ref);
    T pc[4] = {T(plane[0]), T(plane[1]), T(plane[2]), T(plane[3])};
    residual[0] = T(weight) * PointToPlaneDistance(pc, point_ref, true);
    return true;
  }
};

// shared tail of the two *_Angle functors (CostFunction.h:695-717, :899-920)
template <typename T>
inline T NormalizedAngle(const T* point_ref, const T* point_projected, bool normalize_distance) {
  if (normalize_distance) {
    T norm = sqrt(point_projected[0] * point_projected[0] + point_projected[1] * point_projected[1] +
                  point_projected[2] * point_projected[2]);
    T ratio = (norm - T(1.0)) / norm;
    T c[3] = {ratio * point_projected[0], ratio * point_projected[1], ratio * point_projected[2]};
    T vec1[3] = {point_projected[0] - c[0], point_projected[1] - c[1], point_projected[2] - c[2]};
    T vec2[3] = {point_ref[0] - c[0], point_ref[1] - c[1], point_ref[2] - c[2]};
    return VectorAngle3D(vec1, vec2);
  }
  return VectorAngle3D(point_ref, point_projected);
}

struct Point2Plane_Angle {
  double plane[4], curr_point[3], weight;  // weight is stored but NOT applied (CostFunction.h:630-729)
  bool normalize_distance;
  template <typename T>
  bool operator()(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw, T* residual) const {
    T point_ref[3];
    TransformNeighborToRef(aa_rw, t_rw, aa_nw, t_nw, curr_point, point_ref);
    T pc[4] = {T(plane[0]), T(plane[1]), T(plane[2]), T(plane[3])};
    T pp[3];
    T dis = PointToPlaneDistance(pc, point_ref, true);
    if (dis < T(1e-3)) { residual[0] = T(0.0); return true; }
    pp[0] = point_ref[0] - dis * pc[0];
    pp[1] = point_ref[1] - dis * pc[1];
    pp[2] = point_ref[2] - dis * pc[2];
    if (abs(pc[0] * pp[0] + pc[1] * pp[1] + pc[2] * pp[2] + pc[3]) > 1e-4) {
      pp[0] = point_ref[0] + dis * pc[0];
      pp[1] = point_ref[1] + dis * pc[1];
      pp[2] = point_ref[2] + dis * pc[2];
    }
    residual[0] = NormalizedAngle(point_ref, pp, normalize_distance);
    return true;
  }
};

struct Point2Line_Meter {
  double line_point[3], line_direction[3], curr_point[3], weight;
  // ctor semantics CostFunction.h:778-783: direction = (a - b).normalized()
  void SetLine(const double* a, const double* b) {
    double d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 > 0.0) { const double n = std::sqrt(n2); d[0] /= n; d[1] /= n; d[2] /= n; }
    for (int k = 0; k < 3; ++k) { line_point[k] = a[k]; line_direction[k] = d[k]; }
  }
  template <typename T>
  bool operator()(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw, T* residual) const {
    T point_ref[3];
    TransformNeighborToRef(aa_rw, t_rw, aa_nw, t_nw, curr_point, point_ref);
    T line[6] = {T(line_point[0]), T(line_point[1]), T(line_point[2]),
                 T(line_direction[0]), T(line_direction[1]), T(line_direction[2])};
    residual[0] = T(weight) * PointToLineDistance3D(point_ref, line);
    return true;
  }
};

struct Point2Line_Angle {
  double line_point[3], line_direction[3], curr_point[3], weight;  // weight NOT applied (:836-934)
  bool normalize_distance;
  void SetLine(const double* a, const double* b) {
    double d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 > 0.0) { const double n = std::sqrt(n2); d[0] /= n; d[1] /= n; d[2] /= n; }
    for (int k = 0; k < 3; ++k) { line_point[k] = a[k]; line_direction[k] = d[k]; }
  }
  template <typename T>
  bool operator()(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw, T* residual) const {
    T point_ref[3];
    TransformNeighborToRef(aa_rw, t_rw, aa_nw, t_nw, curr_point, point_ref);
    T x0 = T(line_point[0]), y0 = T(line_point[1]), z0 = T(line_point[2]);
    T nx = T(line_direction[0]), ny = T(line_direction[1]), nz = T(line_direction[2]);
    T k = nx * (point_ref[0] - x0) + ny * (point_ref[1] - y0) + nz * (point_ref[2] - z0);
    T pp[3] = {k * nx + x0, k * ny + y0, k * nz + z0};
    T dis = sqrt((point_ref[0] - pp[0]) * (point_ref[0] - pp[0]) + (point_ref[1] - pp[1]) * (point_ref[1] - pp[1]) +
                 (point_ref[2] - pp[2]) * (point_ref[2] - pp[2]));
    if (dis < T(1e-3)) { residual[0] = T(0.0); return true; }
    residual[0] = NormalizedAngle(point_ref, pp, normalize_distance);
    return true;
  }
};

// helper for the camera<-LiDAR chain of the two plane functors (CostFunction.h:369-403,466-486):
// p_ref = R(aa_rw) ( R(-aa_nw) p + t_wn ) + t_rw, t_wn = -R(-aa_nw) t_nw
template <typename T>
inline void TransformViaWorld(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw, const double* p, T* out) {
  T aa_wn[3] = {T(-1.0) * aa_nw[0], T(-1.0) * aa_nw[1], T(-1.0) * aa_nw[2]};
  T t_wn[3];
  AngleAxisRotatePoint(aa_wn, t_nw, t_wn);
  t_wn[0] *= T(-1.0); t_wn[1] *= T(-1.0); t_wn[2] *= T(-1.0);
  T pt[3] = {T(p[0]), T(p[1]), T(p[2])};
  T pw[3];
  AngleAxisRotatePoint(aa_wn, pt, pw);
  pw[0] += t_wn[0]; pw[1] += t_wn[1]; pw[2] += t_wn[2];
  AngleAxisRotatePoint(aa_rw, pw, out);
  out[0] += t_rw[0]; out[1] += t_rw[1]; out[2] += t_rw[2];
}

struct Plane2Plane_Global {
  double plane_ref[3], point_a[3], point_b[3], weight;
  // ctor normalises plane_ref (CostFunction.h:357-362)
  void SetPlane(const double* p) {
    const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    const double n = std::sqrt(n2);
    for (int k = 0; k < 3; ++k) plane_ref[k] = n2 > 0.0 ? p[k] / n : p[k];
  }
  template <typename T>
  bool operator()(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw, T* residual) const {
    T pa[3], pb[3];
    TransformViaWorld(aa_rw, t_rw, aa_nw, t_nw, point_a, pa);
    TransformViaWorld(aa_rw, t_rw, aa_nw, t_nw, point_b, pb);
    T a = pa[1] * pb[2] - pa[2] * pb[1];
    T b = pa[2] * pb[0] - pa[0] * pb[2];
    T c = pa[0] * pb[1] - pa[1] * pb[0];
    T plane1[3] = {a, b, c};
    T plane2[3] = {T(plane_ref[0]), T(plane_ref[1]), T(plane_ref[2])};
    residual[0] = T(weight) * PlaneAngle<T>(plane2, plane1);
    return true;
  }
};

struct PlaneIOUResidual {
  double ref_plane[4], middle_neighbor[3], middle_ref[3], angle, weight;
  // LiDAR-LiDAR / mapping ctor (CostFunction.h:453-460): ref_plane = plane / |plane.xyz|
  void SetPlane(const double* p) {
    const double n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    for (int k = 0; k < 4; ++k) ref_plane[k] = p[k] / n;
  }
  template <typename T>
  bool operator()(const T* aa_rw, const T* t_rw, const T* aa_nw, const T* t_nw, T* residual) const {
    T mr[3];
    TransformViaWorld(aa_rw, t_rw, aa_nw, t_nw, middle_neighbor, mr);
    T pc[4] = {T(ref_plane[0]), T(ref_plane[1]), T(ref_plane[2]), T(ref_plane[3])};
    T ip[3] = {T(middle_ref[0]), T(middle_ref[1]), T(middle_ref[2])};
    T np[3];
    ProjectPointToPlane(mr, pc, np, true);
    T curr = VectorAngle3D(np, ip);
    if (curr < T(angle)) residual[0] = T(0.0);
    else residual[0] = T(weight) * (curr - T(angle));
    return true;
  }
};

// Plane2Plane_Relative (base/CostFunction.h:294-348) and PlaneRelativeIOUResidual (:509-565): the calibration-mode functors of
// CameraLidarOptimizer::Optimize(line_pairs, T_cl) (joint_optimization/CameraLidarOptimizer.cpp:32-87) — ONE pose (aa_cl, t_cl).
struct Plane2Plane_Relative {
  double plane_ref[3], point_a[3], point_b[3], weight;
  void SetPlane(const double* p) {                                  // ctor: plane_ref.normalize() (:306)
    const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    const double n = std::sqrt(n2);
    for (int k = 0; k < 3; ++k) plane_ref[k] = n2 > 0.0 ? p[k] / n : p[k];
  }
  template <typename T>
  bool operator()(const T* aa_cl, const T* t_cl, T* residual) const {
    T pa0[3] = {T(point_a[0]), T(point_a[1]), T(point_a[2])}, pa[3];
    AngleAxisRotatePoint(aa_cl, pa0, pa);
    pa[0] += t_cl[0]; pa[1] += t_cl[1]; pa[2] += t_cl[2];
    T pb0[3] = {T(point_b[0]), T(point_b[1]), T(point_b[2])}, pb[3];
    AngleAxisRotatePoint(aa_cl, pb0, pb);
    pb[0] += t_cl[0]; pb[1] += t_cl[1]; pb[2] += t_cl[2];
    T a = pa[1] * pb[2] - pa[2] * pb[1];
    T b = pa[2] * pb[0] - pa[0] * pb[2];
    T c = pa[0] * pb[1] - pa[1] * pb[0];
    T lidar_line_plane[3] = {a, b, c};
    T img_plane_norm[3] = {T(plane_ref[0]), T(plane_ref[1]), T(plane_ref[2])};
    residual[0] = T(weight) * PlaneAngle<T>(img_plane_norm, lidar_line_plane) * T(180.0) / T(M_PI);
    return true;
  }
};

struct PlaneRelativeIOUResidual {
  double ref_plane[4], middle_neighbor[3], middle_ref[3], angle, weight;
  void SetPlane(const double* p) {                                  // ctor: ref_plane = plane / |plane.xyz| (:523)
    const double n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    for (int k = 0; k < 4; ++k) ref_plane[k] = p[k] / n;
  }
  template <typename T>
  bool operator()(const T* aa_cl, const T* t_cl, T* residual) const {
    T pn[3] = {T(middle_neighbor[0]), T(middle_neighbor[1]), T(middle_neighbor[2])}, mt[3];
    AngleAxisRotatePoint(aa_cl, pn, mt);
    mt[0] += t_cl[0]; mt[1] += t_cl[1]; mt[2] += t_cl[2];
    T pc[4] = {T(ref_plane[0]), T(ref_plane[1]), T(ref_plane[2]), T(ref_plane[3])};
    T rp[3] = {T(middle_ref[0]), T(middle_ref[1]), T(middle_ref[2])};
    T np[3];
    ProjectPointToPlane(mt, pc, np, true);
    T curr = VectorAngle3D(np, rp);
    if (curr < T(angle)) residual[0] = T(0.0);
    else residual[0] = T(weight) * (curr - T(angle));
    return true;
  }
};

// PanoramaReprojResidual_1Angle (base/CostFunction.h:218-247): angle between the keypoint's bearing and the
// direction of the 3-D point in the camera frame; 3 parameter blocks (aa_cw, t_cw, point_3d).  Added by
// AddCameraResidual (util/Optimization.cpp:172-222) with HuberLoss(4 deg).
struct PanoramaReprojResidual_1Angle {
  double point_sphere[3], weight;
  // ctor: point_sphere.normalize() (:227-230; Eigen divides by the norm)
  void SetBearing(const double* p) {
    const double n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    for (int k = 0; k < 3; ++k) point_sphere[k] = n > 0.0 ? p[k] / n : p[k];
  }
  template <typename T>
  bool operator()(const T* aa_cw, const T* t_cw, const T* point_3d, T* residual) const {
    T pc[3];
    AngleAxisRotatePoint(aa_cw, point_3d, pc);
    pc[0] += t_cw[0]; pc[1] += t_cw[1]; pc[2] += t_cw[2];
    T norm = sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    T dot = pc[0] * T(point_sphere[0]) + pc[1] * T(point_sphere[1]) + pc[2] * T(point_sphere[2]);
    residual[0] = T(weight) * acos(dot / norm);
    return true;
  }
};

// AutoDiffCostFunction<PanoramaReprojResidual_1Angle,1,3,3,3>::Evaluate; J = 1x9 [d/daa_cw | d/dt_cw | d/dX]
inline bool AutoDiffEvaluateReproj(const PanoramaReprojResidual_1Angle& f, const double* aa, const double* t, const double* X,
                                   double* residual, double* J) {
  if (!J) return f(aa, t, X, residual);
  typedef Jet<9> JT;
  JT p[3][3];
  const double* src[3] = {aa, t, X};
  for (int b = 0; b < 3; ++b)
    for (int k = 0; k < 3; ++k) p[b][k] = JT(src[b][k], b * 3 + k);
  JT r;
  const bool ok = f(p[0], p[1], p[2], &r);
  *residual = r.a;
  for (int k = 0; k < 9; ++k) J[k] = r.v[k];
  return ok;
}

// AutoDiffCostFunction<F,1,3,3>::Evaluate (the calibration-mode functors): params = {aa_cl, t_cl}; J = 1x6 [d/daa_cl | d/dt_cl]
template <typename F>
inline bool AutoDiffEvaluateRelative(const F& f, const double* aa, const double* t, double* residual, double* J) {
  if (!J) return f(aa, t, residual);
  typedef Jet<6> JT;
  JT p[2][3];
  const double* src[2] = {aa, t};
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 3; ++k) p[b][k] = JT(src[b][k], b * 3 + k);
  JT r;
  const bool ok = f(p[0], p[1], &r);
  *residual = r.a;
  for (int k = 0; k < 6; ++k) J[k] = r.v[k];
  return ok;
}

// AutoDiffCostFunction<F,1,3,3,3,3>::Evaluate: params = {aa_rw, t_rw, aa_nw, t_nw};
// J is the 1x12 row [d/daa_rw | d/dt_rw | d/daa_nw | d/dt_nw]; J may be null (cost only).
template <typename F>
inline bool AutoDiffEvaluate(const F& f, const double* aa_r, const double* t_r, const double* aa_n, const double* t_n,
                             double* residual, double* J) {
  if (!J) {
    double r;
    const bool ok = f(aa_r, t_r, aa_n, t_n, &r);
    *residual = r;
    return ok;
  }
  typedef Jet<12> JT;
  JT p[4][3];
  const double* src[4] = {aa_r, t_r, aa_n, t_n};
  for (int b = 0; b < 4; ++b)
    for (int k = 0; k < 3; ++k) p[b][k] = JT(src[b][k], b * 3 + k);
  JT r;
  const bool ok = f(p[0], p[1], p[2], p[3], &r);
  *residual = r.a;
  for (int k = 0; k < 12; ++k) J[k] = r.v[k];
  return ok;
}

// The same evaluation in x87 extended precision (JetT<long double, 12>: 64-bit significand): the reference's own statements,
// 11 more bits.  The arbiter where the double evaluation of upstream's formula is the less accurate side — acos(cos) of
// the *_Angle functors has an error of ~eps/r near r -> 0 (base/Geometry.hpp:450-466) — so that the parity tests can assert
// 1e-6 against the exact value of the formula instead of widening the gate by that error.  Results rounded to double.
template <typename F>
inline bool AutoDiffEvaluateExt(const F& f, const double* aa_r, const double* t_r, const double* aa_n, const double* t_n,
                                double* residual, double* J) {
  typedef JetT<long double, 12> JT;
  JT p[4][3];
  const double* src[4] = {aa_r, t_r, aa_n, t_n};
  for (int b = 0; b < 4; ++b)
    for (int k = 0; k < 3; ++k) p[b][k] = JT((long double)src[b][k], b * 3 + k);
  JT r;
  const bool ok = f(p[0], p[1], p[2], p[3], &r);
  *residual = (double)r.a;
  if (J) for (int k = 0; k < 12; ++k) J[k] = (double)r.v[k];
  return ok;
}

// The quantity the *_Angle functors branch on (`dis < 1e-3 -> residual 0`, CostFunction.h:680-684, :893-897), in extended
// precision: lets a test COUNT the blocks that sit on the threshold instead of hiding them in a tolerance.
inline double BranchDistanceExt(const Point2Plane_Angle& f, const double* aa_r, const double* t_r, const double* aa_n, const double* t_n) {
  long double a[4][3];
  const double* src[4] = {aa_r, t_r, aa_n, t_n};
  for (int b = 0; b < 4; ++b) for (int k = 0; k < 3; ++k) a[b][k] = src[b][k];
  long double point_ref[3];
  TransformNeighborToRef(a[0], a[1], a[2], a[3], f.curr_point, point_ref);
  long double pc[4] = {f.plane[0], f.plane[1], f.plane[2], f.plane[3]};
  return (double)PointToPlaneDistance(pc, point_ref, true);
}
inline double BranchDistanceExt(const Point2Line_Angle& f, const double* aa_r, const double* t_r, const double* aa_n, const double* t_n) {
  long double a[4][3];
  const double* src[4] = {aa_r, t_r, aa_n, t_n};
  for (int b = 0; b < 4; ++b) for (int k = 0; k < 3; ++k) a[b][k] = src[b][k];
  long double q[3];
  TransformNeighborToRef(a[0], a[1], a[2], a[3], f.curr_point, q);
  const long double x0 = f.line_point[0], y0 = f.line_point[1], z0 = f.line_point[2], nx = f.line_direction[0], ny = f.line_direction[1], nz = f.line_direction[2];
  const long double k = nx * (q[0] - x0) + ny * (q[1] - y0) + nz * (q[2] - z0);
  const long double pp[3] = {k * nx + x0, k * ny + y0, k * nz + z0};
  return (double)std::sqrt((q[0] - pp[0]) * (q[0] - pp[0]) + (q[1] - pp[1]) * (q[1] - pp[1]) + (q[2] - pp[2]) * (q[2] - pp[2]));
}

// ceres::HuberLoss(a)::Evaluate — rho[0..2] for s = r^2 ([recalled] loss_function.cc)
inline void HuberLossEvaluate(double a, double s, double* rho) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

}  // namespace oracle
