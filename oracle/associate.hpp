// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// CPU restatement of the LiDAR<->LiDAR and camera<->LiDAR association code of the reference:
//   /root/reference/lidar_mapping/LidarFeatureAssociate.cpp:19-111   FindNeighbors
//   /root/reference/lidar_mapping/LidarFeatureAssociate.cpp:120-197  FindAssociations
//   /root/reference/lidar_mapping/LidarFeatureAssociate.cpp:219-236  TransformLines
//   /root/reference/lidar_mapping/LidarFeatureAssociate.cpp:442-476  AssociateLine2Line
//   /root/reference/lidar_mapping/LidarFeatureAssociate.cpp:478-548  AssociatePoint2Line
//   /root/reference/lidar_mapping/LidarFeatureAssociate.cpp:550-630  AssociatePoint2Plane
//   /root/reference/sensors/Velodyne.cpp:1850-1859                   World2Local / Local2World
//   /root/reference/joint_optimization/CameraLidarLineAssociate.cpp:340-475  AssociateByAngle
//   /root/reference/joint_optimization/CameraLidarLineAssociate.cpp:628-715  Filter
//   /root/reference/joint_optimization/CameraLidarLineAssociate.cpp:754-876  UniqueLinePair
// Third-party arithmetic restated ([recalled]; PCL 1.10 / FLANN 1.9.x absent from this image):
//   pcl::KdTreeFLANN::nearestKSearch = exact k-NN, squared L2 accumulated in float32 in the
//   order ((dx*dx)+dy*dy)+dz*dz (flann::L2_Simple), results ascending; equal distances are
//   returned in tree-traversal order upstream — here ties break by ascending target index
//   (documented deviation; synthetic data avoids exact ties);
//   pcl::transformPointCloud(float cloud, Matrix4d) = per coordinate
//   float(((m0*x + m1*y) + m2*z) + m3) evaluated in double.
// "parity unpinned": the reference has no tests / fixtures for these; the k-NN sets are
// cross-checked against scipy.spatial.cKDTree and the fits against numpy (tests/).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#include "equirect.hpp"
#include "geometry.hpp"

namespace oracle {

struct Scan {
  int id = 0;
  bool valid = true;
  double R_wl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
  double t_wl[3] = {0, 0, 0};
  // world-frame float32 clouds, xyz interleaved; tag = intensity field
  std::vector<float> surfFlat, surfFlat_tag;
  std::vector<float> surfLessFlat, surfLessFlat_tag;
  std::vector<float> cornerLessSharp;
  std::vector<std::vector<int>> point_to_segment;  // per corner point, ascending segment ids
  std::vector<int> segment_size;                   // edge_segmented[s].size()
  std::vector<double> segment_coeffs;              // 6 per segment, LOCAL frame (point, unit dir)
  std::vector<double> end_points;                  // 2x3 per segment, LOCAL frame

  bool IsPoseValid() const {
    bool zero = true;
    for (int k = 0; k < 9; ++k) if (std::fabs(R_wl[k]) > 1e-12) zero = false;
    for (int k = 0; k < 3; ++k) if (std::isinf(t_wl[k]) || std::isnan(t_wl[k])) return false;
    return !zero;
  }
  // Velodyne.cpp:1850-1853: R_wl^T * p - R_wl^T * t_wl
  void World2Local(const double* pw, double* pl) const {
    for (int i = 0; i < 3; ++i) {
      const double a = (R_wl[0 * 3 + i] * pw[0] + R_wl[1 * 3 + i] * pw[1]) + R_wl[2 * 3 + i] * pw[2];
      const double b = (R_wl[0 * 3 + i] * t_wl[0] + R_wl[1 * 3 + i] * t_wl[1]) + R_wl[2 * 3 + i] * t_wl[2];
      pl[i] = a - b;
    }
  }
  void Local2World(const double* pl, double* pw) const {
    for (int i = 0; i < 3; ++i)
      pw[i] = ((R_wl[i * 3] * pl[0] + R_wl[i * 3 + 1] * pl[1]) + R_wl[i * 3 + 2] * pl[2]) + t_wl[i];
  }
};

// exact k-NN, float32, FLANN L2_Simple accumulation order, ascending, ties by index.
// Returns false when the target cloud has fewer than k points.
inline bool KnnBrute(const float* tgt, int nt, const float* q, int k, int* idx, float* sqd) {
  if (nt < k) return false;
  int cnt = 0;
  for (int j = 0; j < nt; ++j) {
    const float dx = q[0] - tgt[3 * j], dy = q[1] - tgt[3 * j + 1], dz = q[2] - tgt[3 * j + 2];
    float d = 0.0f;
    d += dx * dx; d += dy * dy; d += dz * dz;
    if (cnt == k && !(d < sqd[k - 1])) continue;
    int pos = cnt < k ? cnt : k - 1;
    while (pos > 0 && d < sqd[pos - 1]) { sqd[pos] = sqd[pos - 1]; idx[pos] = idx[pos - 1]; --pos; }
    sqd[pos] = d; idx[pos] = j;
    if (cnt < k) ++cnt;
  }
  return true;
}

struct Point2Plane { double point[3]; double plane[4]; int query_index; int nn[10]; };

// LidarFeatureAssociate.cpp:550-630
inline std::vector<Point2Plane> AssociatePoint2Plane(const Scan& ref, const Scan& nei, double plane_tolerance,
                                                     float dist_threshold, std::vector<int>* knn_dump = nullptr) {
  std::vector<Point2Plane> out;
  const float sq_thr = dist_threshold * dist_threshold;
  const int K = 10;
  const int nt = int(ref.surfLessFlat.size() / 3), nq = int(nei.surfFlat.size() / 3);
  if (knn_dump) knn_dump->assign(size_t(nq) * K, -1);
  for (int i = 0; i < nq; ++i) {
    int idx[K]; float sqd[K];
    const float* q = &nei.surfFlat[3 * i];
    if (!KnnBrute(ref.surfLessFlat.data(), nt, q, K, idx, sqd)) continue;
    if (knn_dump) for (int j = 0; j < K; ++j) (*knn_dump)[size_t(i) * K + j] = idx[j];
    if (sqd[K - 1] > sq_thr) continue;
    double pl[K * 3];
    size_t same = 0;
    for (int j = 0; j < K; ++j) {
      same += (ref.surfLessFlat_tag[idx[j]] == nei.surfFlat_tag[i]);
      const double pw[3] = {ref.surfLessFlat[3 * idx[j]], ref.surfLessFlat[3 * idx[j] + 1], ref.surfLessFlat[3 * idx[j] + 2]};
      ref.World2Local(pw, &pl[3 * j]);
    }
    if (same < size_t(K)) continue;
    double plane[4], line[6];
    const bool plane_ok = FormPlaneLSQ(pl, K, plane_tolerance, plane);
    const bool is_line = FormLinePCA(pl, K, 3.0, 0.0, line);
    if (!plane_ok || is_line) continue;
    Point2Plane a;
    const double qw[3] = {q[0], q[1], q[2]};
    nei.World2Local(qw, a.point);
    for (int k = 0; k < 4; ++k) a.plane[k] = plane[k];
    a.query_index = i;
    for (int j = 0; j < K; ++j) a.nn[j] = idx[j];
    out.push_back(a);
  }
  return out;
}

struct Point2Line { double point[3], a[3], b[3]; int query_index; };

// LidarFeatureAssociate.cpp:478-548
inline std::vector<Point2Line> AssociatePoint2Line(const Scan& ref, const Scan& nei, float dist_threshold) {
  std::vector<Point2Line> out;
  const float sq_thr = dist_threshold * dist_threshold;
  const int K = 5;
  const int nt = int(ref.cornerLessSharp.size() / 3), nq = int(nei.cornerLessSharp.size() / 3);
  for (int i = 0; i < nq; ++i) {
    int idx[K]; float sqd[K];
    const float* q = &nei.cornerLessSharp[3 * i];
    if (!KnnBrute(ref.cornerLessSharp.data(), nt, q, K, idx, sqd)) continue;
    if (sqd[K - 1] > sq_thr) continue;
    double pts[K * 3];
    for (int j = 0; j < K; ++j)
      for (int c = 0; c < 3; ++c) pts[3 * j + c] = ref.cornerLessSharp[3 * idx[j] + c];
    double line[6];
    if (!FormLinePCA(pts, K, 10.0, 0.05, line)) continue;
    double pa[3], pb[3];
    for (int c = 0; c < 3; ++c) { pa[c] = 0.1 * line[3 + c] + line[c]; pb[c] = -0.1 * line[3 + c] + line[c]; }
    Point2Line a;
    const double qw[3] = {q[0], q[1], q[2]};
    nei.World2Local(qw, a.point);
    ref.World2Local(pa, a.a);
    ref.World2Local(pb, a.b);
    a.query_index = i;
    out.push_back(a);
  }
  return out;
}

// LidarFeatureAssociate.cpp:219-236 (T = [R_wl | t_wl])
inline std::vector<double> TransformLines(const std::vector<double>& coeffs, const double* R, const double* t) {
  std::vector<double> out(coeffs.size());
  for (size_t s = 0; s < coeffs.size() / 6; ++s) {
    const double* c = &coeffs[6 * s];
    for (int i = 0; i < 3; ++i) {
      out[6 * s + i] = ((R[i * 3] * c[0] + R[i * 3 + 1] * c[1]) + R[i * 3 + 2] * c[2]) + t[i];
      out[6 * s + 3 + i] = (R[i * 3] * c[3] + R[i * 3 + 1] * c[4]) + R[i * 3 + 2] * c[5];
    }
  }
  return out;
}

struct Line2Line { int neighbor_line_idx, ref_line_idx; double p1[3], p2[3]; };

// LidarFeatureAssociate.cpp:120-197. line_matrix is row-major [n_nei_seg x n_ref_seg].
inline std::vector<Line2Line> FindAssociations(const Scan& ref, const Scan& nei, const std::vector<double>& ref_world,
                                               const std::vector<double>& nei_world, const std::vector<int>& line_matrix) {
  std::map<int, Line2Line> m;
  const int nr = int(ref.segment_size.size()), nn = int(nei.segment_size.size());
  for (int s = 0; s < nn; ++s) {
    if (nr == 0) continue;
    int max_col = 0, max_count = line_matrix[size_t(s) * nr];
    for (int c = 1; c < nr; ++c)
      if (line_matrix[size_t(s) * nr + c] > max_count) { max_count = line_matrix[size_t(s) * nr + c]; max_col = c; }
    // int vs size_t comparison as upstream: max_count is promoted to unsigned
    if (size_t(max_count) < size_t(nei.segment_size[s]) / 2) continue;
    const double* dref = &ref_world[6 * max_col + 3];
    const double* dnei = &nei_world[6 * s + 3];
    if (PlaneAngle(dref, dnei) * 180.0 / M_PI > 7) continue;
    const double* dloc = &ref.segment_coeffs[6 * max_col + 3];
    const double* ploc = &ref.segment_coeffs[6 * max_col];
    Line2Line a;
    a.neighbor_line_idx = s; a.ref_line_idx = max_col;
    for (int c = 0; c < 3; ++c) { a.p1[c] = 0.1 * dloc[c] + ploc[c]; a.p2[c] = -0.1 * dloc[c] + ploc[c]; }
    auto it = m.find(max_col);
    if (it == m.end()) m.insert({max_col, a});
    else {
      const double d1 = PointToLineDistance3D(&nei_world[6 * it->second.neighbor_line_idx], &ref_world[6 * max_col]);
      const double d2 = PointToLineDistance3D(&nei_world[6 * s], &ref_world[6 * max_col]);
      if (d2 < d1) it->second = a;
    }
  }
  std::vector<Line2Line> out;
  for (auto& kv : m) out.push_back(kv.second);
  return out;
}

// LidarFeatureAssociate.cpp:442-476. votes_out (optional) receives the vote matrix.
inline std::vector<Line2Line> AssociateLine2Line(const Scan& ref, const Scan& nei, float dist_threshold,
                                                 std::vector<int>* votes_out = nullptr) {
  std::vector<Line2Line> out;
  if (ref.segment_size.empty() || nei.segment_size.empty()) return out;
  const std::vector<double> nei_world = TransformLines(nei.segment_coeffs, nei.R_wl, nei.t_wl);
  const std::vector<double> ref_world = TransformLines(ref.segment_coeffs, ref.R_wl, ref.t_wl);
  const int nr = int(ref.segment_size.size()), nn = int(nei.segment_size.size());
  std::vector<int> votes(size_t(nn) * nr, 0);
  const int nc = int(nei.cornerLessSharp.size() / 3);
  for (int i = 0; i < nc; ++i) {
    const double p[3] = {nei.cornerLessSharp[3 * i], nei.cornerLessSharp[3 * i + 1], nei.cornerLessSharp[3 * i + 2]};
    for (int s = 0; s < nr; ++s) {
      const double d = PointToLineDistance3D(p, &ref_world[6 * s]);
      if (d > dist_threshold) continue;  // double vs float threshold promoted to double
      for (int ns : nei.point_to_segment[i]) votes[size_t(ns) * nr + s] += 1;
    }
  }
  if (votes_out) *votes_out = votes;
  return FindAssociations(ref, nei, ref_world, nei_world, votes);
}

// LidarFeatureAssociate.cpp:238-317: every one of the 5 nearest ref corner points must lie on ONE ref segment; the
// line end points are the segment's LOCAL coefficients +- 0.1 * direction (not transformed, as upstream).
inline std::vector<Point2Line> AssociatePoint2LineSegmentKNN(const Scan& ref, const Scan& nei, float dist_threshold) {
  std::vector<Point2Line> out;
  if (ref.segment_size.empty() || nei.segment_size.empty()) return out;
  const float sq_thr = dist_threshold * dist_threshold;
  const int K = 5;
  const int nt = int(ref.cornerLessSharp.size() / 3), nq = int(nei.cornerLessSharp.size() / 3);
  for (int i = 0; i < nq; ++i) {
    int idx[K]; float sqd[K];
    const float* q = &nei.cornerLessSharp[3 * i];
    if (!KnnBrute(ref.cornerLessSharp.data(), nt, q, K, idx, sqd)) continue;
    if (sqd[K - 1] > sq_thr) continue;
    std::map<size_t, size_t> seg_count;
    for (int j = 0; j < K; ++j) for (int sid : ref.point_to_segment[idx[j]]) seg_count[size_t(sid)]++;
    for (const auto& kv : seg_count) {
      if (kv.second < size_t(K - 0)) continue;
      const double* l = &ref.segment_coeffs[6 * kv.first];
      Point2Line a;
      for (int c = 0; c < 3; ++c) { a.a[c] = 0.1 * l[3 + c] + l[c]; a.b[c] = -0.1 * l[3 + c] + l[c]; }
      const double qw[3] = {q[0], q[1], q[2]};
      nei.World2Local(qw, a.point);
      a.query_index = i;
      out.push_back(a);
    }
  }
  return out;
}

// LidarFeatureAssociate.cpp:319-383: nearest ref segment line (world) by point-to-line distance, brute force.
inline std::vector<Point2Line> AssociatePoint2LineSegment(const Scan& ref, const Scan& nei, float dist_threshold) {
  std::vector<Point2Line> out;
  if (ref.segment_size.empty() || nei.segment_size.empty()) return out;
  const std::vector<double> ref_world = TransformLines(ref.segment_coeffs, ref.R_wl, ref.t_wl);
  const int nq = int(nei.cornerLessSharp.size() / 3), nr = int(ref.segment_size.size());
  for (int i = 0; i < nq; ++i) {
    const double p[3] = {nei.cornerLessSharp[3 * i], nei.cornerLessSharp[3 * i + 1], nei.cornerLessSharp[3 * i + 2]};
    double min_distance = std::numeric_limits<double>::max();
    int seg = -1;
    for (int s = 0; s < nr; ++s) {
      const double d = PointToLineDistance3D(p, &ref_world[6 * s]);
      if (d < min_distance) { min_distance = d; seg = s; }
    }
    if (!(min_distance <= dist_threshold)) continue;
    const double* l = &ref.segment_coeffs[6 * seg];
    Point2Line a;
    for (int c = 0; c < 3; ++c) { a.a[c] = 0.1 * l[3 + c] + l[c]; a.b[c] = -0.1 * l[3 + c] + l[c]; }
    nei.World2Local(p, a.point);
    a.query_index = i;
    out.push_back(a);
  }
  return out;
}

// LidarFeatureAssociate.cpp:385-440: votes from the 5-NN's segments (>= 3 of the 5 on one ref segment), then FindAssociations.
inline std::vector<Line2Line> AssociateLine2LineKNN(const Scan& ref, const Scan& nei, float dist_threshold, std::vector<int>* votes_out = nullptr) {
  std::vector<Line2Line> out;
  if (ref.segment_size.empty() || nei.segment_size.empty()) return out;
  const float sq_thr = dist_threshold * dist_threshold;
  const int K = 5;
  const std::vector<double> nei_world = TransformLines(nei.segment_coeffs, nei.R_wl, nei.t_wl);
  const std::vector<double> ref_world = TransformLines(ref.segment_coeffs, ref.R_wl, ref.t_wl);
  const int nr = int(ref.segment_size.size()), nn = int(nei.segment_size.size());
  std::vector<int> votes(size_t(nn) * nr, 0);
  const int nt = int(ref.cornerLessSharp.size() / 3), nq = int(nei.cornerLessSharp.size() / 3);
  for (int i = 0; i < nq; ++i) {
    int idx[K]; float sqd[K];
    if (!KnnBrute(ref.cornerLessSharp.data(), nt, &nei.cornerLessSharp[3 * i], K, idx, sqd)) continue;
    if (sqd[K - 1] > sq_thr) continue;
    std::map<size_t, size_t> seg_count;
    for (int j = 0; j < K; ++j) for (int sid : ref.point_to_segment[idx[j]]) seg_count[size_t(sid)]++;
    for (const auto& kv : seg_count) {
      if (kv.second < size_t(K - 2)) continue;
      for (int ns : nei.point_to_segment[i]) votes[size_t(ns) * nr + kv.first] += 1;
    }
  }
  if (votes_out) *votes_out = votes;
  return FindAssociations(ref, nei, ref_world, nei_world, votes);
}

// ---------------------------------------------------------------------------------------------
// FindNeighbors — LidarFeatureAssociate.cpp:19-111 (exact k-NN / radius search over scan centres,
// float32 centres as PointXYZI; FLANN radius search keeps dist < r^2 and returns ascending).
// ---------------------------------------------------------------------------------------------
inline std::vector<std::vector<int>> FindNeighbors(const std::vector<Scan>& lidars, int neighbor_size) {
  std::vector<std::vector<int>> all;
  std::vector<float> centers; std::vector<int> owner;
  for (size_t i = 0; i < lidars.size(); ++i) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    centers.push_back(float(lidars[i].t_wl[0])); centers.push_back(float(lidars[i].t_wl[1])); centers.push_back(float(lidars[i].t_wl[2]));
    owner.push_back(int(i));
  }
  const int nc = int(owner.size());
  for (size_t i = 0; i < lidars.size(); ++i) {
    std::vector<int> neighbors;
    if (lidars[i].IsPoseValid()) {
      const float q[3] = {float(lidars[i].t_wl[0]), float(lidars[i].t_wl[1]), float(lidars[i].t_wl[2])};
      std::vector<std::pair<float, int>> d(nc);
      for (int j = 0; j < nc; ++j) {
        const float dx = q[0] - centers[3 * j], dy = q[1] - centers[3 * j + 1], dz = q[2] - centers[3 * j + 2];
        float s = 0.0f; s += dx * dx; s += dy * dy; s += dz * dz;
        d[j] = {s, j};
      }
      std::stable_sort(d.begin(), d.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
      const int k = std::min(neighbor_size, nc);
      for (int j = 0; j < k; ++j) neighbors.push_back(d[j].second);
      if (!neighbors.empty()) neighbors.erase(neighbors.begin());
      for (int& n : neighbors) n = owner[n];
      std::set<int> nset(neighbors.begin(), neighbors.end());
      int ni = int(i) - 1;
      while (ni >= 0 && !lidars[ni].IsPoseValid()) ni--;
      if (ni >= 0 && nset.count(ni) == 0) neighbors.push_back(ni);
      ni = int(i) + 1;
      while (ni < int(lidars.size()) && !lidars[ni].IsPoseValid()) ni++;
      if (ni < int(lidars.size()) && nset.count(ni) == 0) neighbors.push_back(ni);
      const float r2 = float(20.0 * 20.0);
      const int loop_length = 200;
      for (int j = 0; j < nc; ++j) {
        if (!(d[j].first < r2)) break;
        const int n_idx = owner[d[j].second];
        int same_loop = 0;
        for (int v : nset) {
          if (std::abs(n_idx - v) <= loop_length) same_loop++;
          if (same_loop >= 2) break;
        }
        if (same_loop < 2 && nset.count(n_idx) == 0) { neighbors.push_back(n_idx); nset.insert(n_idx); }
      }
    } else {
      for (int j = -neighbor_size / 2; j <= neighbor_size / 2; j++) neighbors.push_back(int(i) - j);
    }
    all.push_back(neighbors);
  }
  return all;
}

// ---------------------------------------------------------------------------------------------
// Camera <-> LiDAR line association by angle — CameraLidarLineAssociate.cpp:340-475 (+Filter
// :628-715 with (false,true), + UniqueLinePair :754-876 when !multiple_association).
// ---------------------------------------------------------------------------------------------
struct CameraLidarLinePair {
  float image_line[4];
  double lidar_line_start[3], lidar_line_end[3];
  int image_line_id, lidar_line_id;
  float angle;
  float weight;
};

inline void TransformPoint4d(const double* T, const double* p, double* o) {  // (T * p.homogeneous()).hnormalized(), T row-major 4x4
  double h[4];
  for (int i = 0; i < 4; ++i) h[i] = ((T[i * 4] * p[0] + T[i * 4 + 1] * p[1]) + T[i * 4 + 2] * p[2]) + T[i * 4 + 3] * 1.0;
  for (int i = 0; i < 3; ++i) o[i] = h[i] / h[3];
}

// General 4x4 inverse by Gauss-Jordan with partial pivoting (Eigen uses a cofactor/SSE path
// for 4x4 .inverse(); ulp-level differences only).
inline void Invert4x4(const double* A, double* Ai) {
  double m[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { m[i][j] = A[i * 4 + j]; m[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r) if (std::fabs(m[r][c]) > std::fabs(m[piv][c])) piv = r;
    if (piv != c) for (int j = 0; j < 8; ++j) std::swap(m[c][j], m[piv][j]);
    const double d = m[c][c];
    for (int j = 0; j < 8; ++j) m[c][j] /= d;
    for (int r = 0; r < 4; ++r) if (r != c) { const double f = m[r][c]; for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j]; }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Ai[i * 4 + j] = m[i][4 + j];
}

struct ByAngleDebug { std::vector<int> votes; };  // [n_lines x n_segments] vote counts

inline std::vector<CameraLidarLinePair> FilterByLength(const std::vector<CameraLidarLinePair>& pairs, int rows, int cols) {
  const float min_len = 100, max_len = 2000;
  std::vector<CameraLidarLinePair> good;
  Equirectangular eq(rows, cols);
  for (const CameraLidarLinePair& p : pairs) {
    const float a[3] = {float(p.lidar_line_start[0]), float(p.lidar_line_start[1]), float(p.lidar_line_start[2])};
    const float b[3] = {float(p.lidar_line_end[0]), float(p.lidar_line_end[1]), float(p.lidar_line_end[2])};
    float pa[2], pb[2];
    eq.CamToImage(a, pa);
    eq.CamToImage(b, pb);
    std::vector<float> seg = eq.BreakToSegments(pa, pb, 100);
    float len = 0;
    const size_t n = seg.size() / 2;
    for (size_t i = 0; i + 1 < n; i++) {
      if (std::abs(seg[2 * i] - seg[2 * (i + 1)]) > 0.8 * cols) continue;
      const float dx = seg[2 * i] - seg[2 * (i + 1)], dy = seg[2 * i + 1] - seg[2 * (i + 1) + 1];
      len += std::sqrt(dx * dx + dy * dy);
    }
    if (len < min_len) continue;
    if (len > max_len) continue;
    good.push_back(p);
  }
  return good;
}

inline std::vector<CameraLidarLinePair> UniqueLinePair(const std::vector<CameraLidarLinePair>& pairs, const float* lines,
                                                       const std::vector<double>& lidar_endpoints_cam) {
  struct PairScore { int idx; float score; };
  std::map<int, PairScore> i2l, l2i;
  for (const CameraLidarLinePair& pr : pairs) {
    const int il = pr.image_line_id, ll = pr.lidar_line_id;
    const float score = pr.angle;
    auto a = i2l.find(il);
    auto b = l2i.find(ll);
    const bool ha = a != i2l.end(), hb = b != l2i.end();
    if (!ha && !hb) { i2l.insert({il, {ll, score}}); l2i.insert({ll, {il, score}}); }
    if (ha && !hb) {
      if (score < a->second.score) { l2i.erase(l2i.find(a->second.idx)); a->second = {ll, score}; l2i.insert({ll, {il, score}}); }
    }
    if (!ha && hb) {
      if (score < b->second.score) { i2l.erase(i2l.find(b->second.idx)); b->second = {il, score}; i2l.insert({il, {ll, score}}); }
    }
    if (ha && hb) {
      const float sa = a->second.score, sb = b->second.score;
      if (score < std::min(sa, sb)) {
        i2l.erase(b->second.idx); l2i.erase(a->second.idx);
        i2l.erase(a); l2i.erase(b);
        i2l.insert({il, {ll, score}}); l2i.insert({ll, {il, score}});
      } else if (score > sa && score < sb) {
        // note: upstream evaluates the three cases as independent `if`s; after case 1 fires the
        // iterators are dead, but cases 2/3 cannot also hold then, so else-if is equivalent.
        i2l.erase(i2l.find(b->second.idx)); l2i.erase(b);
      } else if (score < sa && score > sb) {
        l2i.erase(l2i.find(a->second.idx)); i2l.erase(a);
      }
    }
  }
  std::vector<CameraLidarLinePair> out;
  for (auto& kv : i2l) {
    CameraLidarLinePair lp;
    for (int k = 0; k < 4; ++k) lp.image_line[k] = lines[4 * kv.first + k];
    for (int k = 0; k < 3; ++k) { lp.lidar_line_start[k] = lidar_endpoints_cam[6 * kv.second.idx + k]; lp.lidar_line_end[k] = lidar_endpoints_cam[6 * kv.second.idx + 3 + k]; }
    lp.image_line_id = kv.first; lp.lidar_line_id = kv.second.idx; lp.angle = kv.second.score; lp.weight = 1;
    out.push_back(lp);
  }
  return out;
}

// lines: n_lines x 4 float (x1,y1,x2,y2 pixels). cloud_local: LiDAR-frame float corner points (xyz).
// T_cl row-major 4x4. Output pairs have LiDAR endpoints back in the LiDAR frame (:468-474).
inline std::vector<CameraLidarLinePair> AssociateByAngle(int rows, int cols, const float* lines, int n_lines,
                                                         const Scan& lidar_local, const double* T_cl, bool multiple_association,
                                                         ByAngleDebug* dbg = nullptr) {
  const int n_seg = int(lidar_local.segment_size.size());
  const int n_pts = int(lidar_local.cornerLessSharp.size() / 3);
  const std::vector<float>& pc = lidar_local.cornerLessSharp;
  std::vector<float> range(n_pts);
  std::vector<float> cam(size_t(n_pts) * 3);
  for (int i = 0; i < n_pts; ++i) {
    const float x = pc[3 * i], y = pc[3 * i + 1], z = pc[3 * i + 2];
    range[i] = x * x + y * y + z * z;
    for (int r = 0; r < 3; ++r)
      cam[3 * i + r] = float(T_cl[r * 4] * double(x) + T_cl[r * 4 + 1] * double(y) + T_cl[r * 4 + 2] * double(z) + T_cl[r * 4 + 3]);
  }
  std::vector<double> ends(size_t(n_seg) * 6), planes(size_t(n_seg) * 4);
  const double zero3[3] = {0, 0, 0};
  for (int s = 0; s < n_seg; ++s) {
    TransformPoint4d(T_cl, &lidar_local.end_points[6 * s], &ends[6 * s]);
    TransformPoint4d(T_cl, &lidar_local.end_points[6 * s + 3], &ends[6 * s + 3]);
    double pl[4];
    FormPlane3(&ends[6 * s], &ends[6 * s + 3], zero3, pl);
    const double n = std::sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2] + pl[3] * pl[3]);
    if (n * n > 0.0) for (int k = 0; k < 4; ++k) pl[k] /= n;
    for (int k = 0; k < 4; ++k) planes[4 * s + k] = pl[k];
  }
  const double thr = 3.0 / 180.0 * M_PI;
  Equirectangular eq(rows, cols);
  std::vector<CameraLidarLinePair> pairs;
  if (dbg) dbg->votes.assign(size_t(n_lines) * n_seg, 0);
  for (int li = 0; li < n_lines; ++li) {
    const float* l = &lines[4 * li];
    const double px1[2] = {l[0], l[1]}, px2[2] = {l[2], l[3]};
    double p1[3], p2[3], ip[4];
    eq.ImageToCam(px1, 1.0, p1);
    eq.ImageToCam(px2, 1.0, p2);
    FormPlane3(p1, p2, zero3, ip);
    {
      const double n = std::sqrt(ip[0] * ip[0] + ip[1] * ip[1] + ip[2] * ip[2] + ip[3] * ip[3]);
      if (n * n > 0.0) for (int k = 0; k < 4; ++k) ip[k] /= n;
    }
    const double p4[3] = {(p1[0] + p2[0]) / 2.0, (p1[1] + p2[1]) / 2.0, (p1[2] + p2[2]) / 2.0};
    const double scope = VectorAngle3D(p1, p4);
    std::map<size_t, size_t> seg_count;
    for (int i = 0; i < n_pts; ++i) {
      if (range[i] > 15 * 15) continue;
      const double p[3] = {cam[3 * i], cam[3 * i + 1], cam[3 * i + 2]};
      double pp[3];
      ProjectPointToPlane(p, ip, pp, true);
      if (VectorAngle3D(p, pp) >= thr) continue;
      if (VectorAngle3D(p4, pp) >= scope + thr) continue;
      for (int sid : lidar_local.point_to_segment[i]) seg_count[size_t(sid)]++;
    }
    for (auto& kv : seg_count) {
      if (dbg) dbg->votes[size_t(li) * n_seg + kv.first] = int(kv.second);
      if (kv.second < size_t(lidar_local.segment_size[kv.first]) / 2) continue;
      const size_t s = kv.first;
      const double angle = PlaneAngle(ip, &planes[4 * s], true);
      if (angle > thr) continue;
      const double mid[3] = {(ends[6 * s] + ends[6 * s + 3]) / 2.0, (ends[6 * s + 1] + ends[6 * s + 4]) / 2.0, (ends[6 * s + 2] + ends[6 * s + 5]) / 2.0};
      double midp[3];
      ProjectPointToPlane(mid, ip, midp, true);
      if (VectorAngle3D(midp, p4) > scope) continue;
      const float angle2 = float(VectorAngle3D(mid, midp));
      if (angle2 > thr / 2.0) continue;
      const float score = float(angle + angle2);
      CameraLidarLinePair lp;
      for (int k = 0; k < 4; ++k) lp.image_line[k] = l[k];
      for (int k = 0; k < 3; ++k) { lp.lidar_line_start[k] = ends[6 * s + k]; lp.lidar_line_end[k] = ends[6 * s + 3 + k]; }
      lp.image_line_id = li; lp.lidar_line_id = int(s); lp.angle = score; lp.weight = 1;
      pairs.push_back(lp);
    }
  }
  pairs = FilterByLength(pairs, rows, cols);
  if (!multiple_association) pairs = UniqueLinePair(pairs, lines, ends);
  double T_lc[16];
  Invert4x4(T_cl, T_lc);
  for (CameraLidarLinePair& lp : pairs) {
    double a[3], b[3];
    TransformPoint4d(T_lc, lp.lidar_line_start, a);
    TransformPoint4d(T_lc, lp.lidar_line_end, b);
    for (int k = 0; k < 3; ++k) { lp.lidar_line_start[k] = a[k]; lp.lidar_line_end[k] = b[k]; }
  }
  return pairs;
}

}  // namespace oracle
