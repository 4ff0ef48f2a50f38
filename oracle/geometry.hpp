// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// CPU restatement of the geometry helpers the hot path uses from the reference:
//   /root/reference/base/Geometry.hpp:198-211  PointToLineDistance3D
//   /root/reference/base/Geometry.hpp:220-260  FormLine (PCA, eigenvalue-ratio test)
//   /root/reference/base/Geometry.hpp:275-283  PointToPlaneDistance
//   /root/reference/base/Geometry.hpp:301-316  ProjectPointToPlane
//   /root/reference/base/Geometry.hpp:328-336  FormPlane(p1,p2,p3)
//   /root/reference/base/Geometry.hpp:345-373  FormPlane(points, tol)  (10x3 least squares)
//   /root/reference/base/Geometry.hpp:450-466  VectorAngle3D
//   /root/reference/base/Geometry.hpp:471-485  PlaneAngle
//   /root/reference/base/Math.h:31-35          Square
// Third-party arithmetic restated ([recalled], Eigen 3.4 is not in this image):
//   Eigen::ColPivHouseholderQR (computeInPlace + _solve_impl) -> qr_colpiv_solve_nx3 below;
//   Eigen::SelfAdjointEigenSolver<Matrix3d> -> cyclic Jacobi (any backward-stable symmetric
//   eigensolver yields the same accept/reject decision away from the threshold).
// "parity unpinned": the reference has no tests / golden vectors for these; cross-checked here
// against numpy.linalg.lstsq / eigh (tests/test_oracle_crosscheck.py).
#pragma once
#include <cfloat>
#include <cmath>
#include "jet.hpp"

namespace oracle {

template <typename T> inline T Square(const T& a) { return a * a; }

template <typename T>
inline T PointToLineDistance3D(const T* point, const T* line) {
  T x0 = line[0], y0 = line[1], z0 = line[2], nx = line[3], ny = line[4], nz = line[5];
  T k = (nx * (point[0] - x0) + ny * (point[1] - y0) + nz * (point[2] - z0)) / (Square(nx) + Square(ny) + Square(nz));
  T pp[3] = {k * nx + x0, k * ny + y0, k * nz + z0};
  return sqrt(Square(pp[0] - point[0]) + Square(pp[1] - point[1]) + Square(pp[2] - point[2]));
}

template <typename T>
inline T PointToPlaneDistance(const T* plane, const T* point, const bool normalized = false) {
  if (!normalized)
    return abs(plane[0] * point[0] + plane[1] * point[1] + plane[2] * point[2] + plane[3]) /
           sqrt(Square(plane[0]) + Square(plane[1]) + Square(plane[2]));
  return abs(plane[0] * point[0] + plane[1] * point[1] + plane[2] * point[2] + plane[3]);
}

template <typename T>
inline void ProjectPointToPlane(const T* point, const T* plane, T* pp, const bool normalized = false) {
  T dis = PointToPlaneDistance(plane, point, normalized);
  T t = normalized ? dis : dis / sqrt(Square(plane[0]) + Square(plane[1]) + Square(plane[2]));
  pp[0] = point[0] - t * plane[0];
  pp[1] = point[1] - t * plane[1];
  pp[2] = point[2] - t * plane[2];
  if (abs(plane[0] * pp[0] + plane[1] * pp[1] + plane[2] * pp[2] + plane[3]) > 1e-4) {
    pp[0] = point[0] + t * plane[0];
    pp[1] = point[1] + t * plane[1];
    pp[2] = point[2] + t * plane[2];
  }
}

// FormPlane(p1,p2,p3) — Geometry.hpp:328-336. out = (a,b,c,d), NOT normalised.
template <typename T>
inline void FormPlane3(const T* p1, const T* p2, const T* p3, T* out) {
  T a = ((p2[1] - p1[1]) * (p3[2] - p1[2]) - (p2[2] - p1[2]) * (p3[1] - p1[1]));
  T b = ((p2[2] - p1[2]) * (p3[0] - p1[0]) - (p2[0] - p1[0]) * (p3[2] - p1[2]));
  T c = ((p2[0] - p1[0]) * (p3[1] - p1[1]) - (p2[1] - p1[1]) * (p3[0] - p1[0]));
  T d = -(a * p1[0] + b * p1[1] + c * p1[2]);
  out[0] = a; out[1] = b; out[2] = c; out[3] = d;
}

template <typename T>
inline T VectorAngle3D(const T* v1, const T* v2, const bool normalized = false) {
  T cos_angle = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2];
  if (!normalized) {
    T n1 = sqrt(Square(v1[0]) + Square(v1[1]) + Square(v1[2]));
    T n2 = sqrt(Square(v2[0]) + Square(v2[1]) + Square(v2[2]));
    cos_angle = cos_angle / (n1 * n2);
  }
  if (cos_angle >= T(1.0)) return T(0);
  else if (cos_angle <= T(-1.0)) return T(M_PI);
  else return acos(cos_angle);
}

template <typename T>
inline T PlaneAngle(const T* p1, const T* p2, const bool normalized = false) {
  T cos_angle = abs(p1[0] * p2[0] + p1[1] * p2[1] + p1[2] * p2[2]);
  if (!normalized) {
    T n1 = sqrt(Square(p1[0]) + Square(p1[1]) + Square(p1[2]));
    T n2 = sqrt(Square(p2[0]) + Square(p2[1]) + Square(p2[2]));
    cos_angle = cos_angle / (n1 * n2);
  }
  if (cos_angle >= T(1.0)) return T(0);
  else return acos(cos_angle);
}

// ---------------------------------------------------------------------------------------------
// Column-pivoting Householder QR least squares for an n x 3 system A x = b (n <= 16), restating
// Eigen 3.4 ColPivHouseholderQR::computeInPlace / _solve_impl. A is row-major n x 3 (destroyed).
// Plain sequential sums (Eigen's SIMD reductions reorder these at the ulp level).
// ---------------------------------------------------------------------------------------------
inline void qr_colpiv_solve_nx3(int n, double* A, double* b, double* x) {
  const int cols = 3;
  const double eps = DBL_EPSILON;
  double normsU[3], normsD[3], hcoef[3];
  int perm[3] = {0, 1, 2};
  for (int j = 0; j < cols; ++j) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += A[i * 3 + j] * A[i * 3 + j];
    normsD[j] = normsU[j] = std::sqrt(s);
  }
  double maxn = normsU[0]; if (normsU[1] > maxn) maxn = normsU[1]; if (normsU[2] > maxn) maxn = normsU[2];
  const double threshold_helper = (maxn * eps) * (maxn * eps) / double(n);
  const double norm_downdate_threshold = std::sqrt(eps);
  int nonzero_pivots = cols;
  int transp[3];
  for (int k = 0; k < cols; ++k) {
    int big = k; double bigv = normsU[k];
    for (int j = k + 1; j < cols; ++j) if (normsU[j] > bigv) { bigv = normsU[j]; big = j; }
    const double big_sq = bigv * bigv;
    if (nonzero_pivots == cols && big_sq < threshold_helper * double(n - k)) nonzero_pivots = k;
    transp[k] = big;
    if (k != big) {
      for (int i = 0; i < n; ++i) { double t = A[i * 3 + k]; A[i * 3 + k] = A[i * 3 + big]; A[i * 3 + big] = t; }
      double t = normsU[k]; normsU[k] = normsU[big]; normsU[big] = t;
      t = normsD[k]; normsD[k] = normsD[big]; normsD[big] = t;
    }
    // makeHouseholderInPlace on A[k..n-1][k]
    double tailSq = 0.0;
    for (int i = k + 1; i < n; ++i) tailSq += A[i * 3 + k] * A[i * 3 + k];
    const double c0 = A[k * 3 + k];
    double tau, beta;
    if (tailSq <= DBL_MIN) {
      tau = 0.0; beta = c0;
      for (int i = k + 1; i < n; ++i) A[i * 3 + k] = 0.0;
    } else {
      beta = std::sqrt(c0 * c0 + tailSq);
      if (c0 >= 0.0) beta = -beta;
      const double den = c0 - beta;
      for (int i = k + 1; i < n; ++i) A[i * 3 + k] = A[i * 3 + k] / den;
      tau = (beta - c0) / beta;
    }
    A[k * 3 + k] = beta;
    hcoef[k] = tau;
    // apply H_k to the remaining columns
    if (tau != 0.0) {
      for (int j = k + 1; j < cols; ++j) {
        double tmp = 0.0;
        for (int i = k + 1; i < n; ++i) tmp += A[i * 3 + k] * A[i * 3 + j];
        tmp += A[k * 3 + j];
        A[k * 3 + j] -= tau * tmp;
        for (int i = k + 1; i < n; ++i) A[i * 3 + j] -= tau * A[i * 3 + k] * tmp;
      }
    }
    // column-norm downdate (LAPACK LAWN 176 as in Eigen)
    for (int j = k + 1; j < cols; ++j) {
      if (normsU[j] != 0.0) {
        double temp = std::fabs(A[k * 3 + j]) / normsU[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        const double ratio = normsU[j] / normsD[j];
        const double temp2 = temp * ratio * ratio;
        if (temp2 <= norm_downdate_threshold) {
          double s = 0.0;
          for (int i = k + 1; i < n; ++i) s += A[i * 3 + j] * A[i * 3 + j];
          normsD[j] = std::sqrt(s);
          normsU[j] = normsD[j];
        } else {
          normsU[j] *= std::sqrt(temp);
        }
      }
    }
  }
  // permutation indices from the transposition sequence
  for (int k = 0; k < cols; ++k) { int t = perm[k]; perm[k] = perm[transp[k]]; perm[transp[k]] = t; }
  // c = Q^T b : apply H_0 .. H_{nonzero_pivots-1}
  for (int k = 0; k < nonzero_pivots; ++k) {
    const double tau = hcoef[k];
    if (tau != 0.0) {
      double tmp = 0.0;
      for (int i = k + 1; i < n; ++i) tmp += A[i * 3 + k] * b[i];
      tmp += b[k];
      b[k] -= tau * tmp;
      for (int i = k + 1; i < n; ++i) b[i] -= tau * A[i * 3 + k] * tmp;
    }
  }
  // back substitution on the leading nonzero_pivots x nonzero_pivots upper triangle
  double c[3] = {0.0, 0.0, 0.0};
  for (int i = nonzero_pivots - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < nonzero_pivots; ++j) s -= A[i * 3 + j] * c[j];
    c[i] = s / A[i * 3 + i];
  }
  x[0] = x[1] = x[2] = 0.0;
  for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
}

// FormPlane(points, tolerance) — Geometry.hpp:345-373. pts = n x 3 row-major (n <= 16).
// Returns false (plane = 0) when any point is farther than `tolerance` from the fitted plane.
inline bool FormPlaneLSQ(const double* pts, int n, double tolerance, double* plane) {
  double A[16 * 3], b[16], nrm[3];
  for (int i = 0; i < n; ++i) { A[i * 3] = pts[i * 3]; A[i * 3 + 1] = pts[i * 3 + 1]; A[i * 3 + 2] = pts[i * 3 + 2]; b[i] = -1.0; }
  qr_colpiv_solve_nx3(n, A, b, nrm);
  const double len = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
  const double d = 1.0 / len;
  // Eigen normalize(): v /= norm  (no-op when the squared norm is 0)
  if (len * len > 0.0) { nrm[0] /= len; nrm[1] /= len; nrm[2] /= len; }
  if (tolerance > 0) {
    for (int i = 0; i < n; ++i) {
      const double dist = std::fabs((nrm[0] * pts[i * 3] + nrm[1] * pts[i * 3 + 1]) + nrm[2] * pts[i * 3 + 2] + d);
      if (dist > tolerance) { plane[0] = plane[1] = plane[2] = plane[3] = 0.0; return false; }
    }
  }
  plane[0] = nrm[0]; plane[1] = nrm[1]; plane[2] = nrm[2]; plane[3] = d;
  return true;
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi. Input S (row-major, symmetric).
// Output eigenvalues ascending in w[3], eigenvectors in columns of V (row-major 3x3).
inline void eig_sym3_jacobi(const double* S, double* w, double* V) {
  double a[3][3] = {{S[0], S[1], S[2]}, {S[3], S[4], S[5]}, {S[6], S[7], S[8]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[p][q];
        if (apq == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0);
        const double s = t * c;
        const int r = 3 - p - q;
        const double app = a[p][p], aqq = a[q][q];
        a[p][p] = app - t * apq;
        a[q][q] = aqq + t * apq;
        a[p][q] = a[q][p] = 0.0;
        const double arp = a[r][p], arq = a[r][q];
        a[r][p] = a[p][r] = c * arp - s * arq;
        a[r][q] = a[q][r] = s * arp + c * arq;
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (d[idx[j]] > d[idx[j + 1]]) { int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
  for (int i = 0; i < 3; ++i) {
    w[i] = d[idx[i]];
    for (int k = 0; k < 3; ++k) V[k * 3 + i] = v[k][idx[i]];
  }
}

// FormLine(points, tolerance, dis_threshold) — Geometry.hpp:220-260. pts = n x 3 row-major.
// Returns true and fills line[6] = (center, unit direction) when the points form a line
// (largest eigenvalue > tolerance * middle eigenvalue, and every point within dis_threshold
// when dis_threshold > 0); otherwise returns false and zeroes line.
inline bool FormLinePCA(const double* pts, int n, double tolerance, double dis_threshold, double* line) {
  double cx = 0.0, cy = 0.0, cz = 0.0;
  for (int i = 0; i < n; ++i) { cx = cx + pts[i * 3]; cy = cy + pts[i * 3 + 1]; cz = cz + pts[i * 3 + 2]; }
  cx = cx / double(n); cy = cy / double(n); cz = cz / double(n);
  double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double d[3] = {pts[i * 3] - cx, pts[i * 3 + 1] - cy, pts[i * 3 + 2] - cz};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) S[r * 3 + c] = S[r * 3 + c] + d[r] * d[c];
  }
  double w[3], V[9];
  eig_sym3_jacobi(S, w, V);
  for (int k = 0; k < 6; ++k) line[k] = 0.0;
  if (!(w[2] > tolerance * w[1])) return false;
  double dir[3] = {V[2], V[5], V[8]};
  const double len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  if (len * len > 0.0) { dir[0] /= len; dir[1] /= len; dir[2] /= len; }
  double l[6] = {cx, cy, cz, dir[0], dir[1], dir[2]};
  if (dis_threshold > 0.0) {
    for (int i = 0; i < n; ++i)
      if (PointToLineDistance3D(pts + i * 3, l) > dis_threshold) return false;
  }
  for (int k = 0; k < 6; ++k) line[k] = l[k];
  return true;
}

}  // namespace oracle
