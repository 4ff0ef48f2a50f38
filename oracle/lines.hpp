// ORACLE — test infrastructure only (see oracle/__init__.py).  CPU restatement of the LINE branch of the reference's LiDAR
// feature extraction, the step that produces the segments AssociateLine2Line consumes:
//   Velodyne::EdgeToLine     sensors/Velodyne.cpp:1269-1324
//   ExtractLineFeatures      sensors/LidarLineExtraction.cpp:296-389   seed (point + 2 of its 4 nearest neighbours), grow both ends
//   ExpandLine               sensors/LidarLineExtraction.cpp:10-70
//   FuseLineSegments         sensors/LidarLineExtraction.cpp:177-250   neighbour graph of segments, connected groups
//   FindNeighbors            sensors/LidarLineExtraction.cpp:72-110
//   FuseLines                sensors/LidarLineExtraction.cpp:113-175
//   FilterLineByScan / FilterLineByLength  sensors/LidarLineExtraction.cpp:254-294
//   FormLine / FurthestPoints / ProjectPoint2Line3D / PointToLineDistance3D / PlaneAngle   base/Geometry.hpp
// Written to follow the reference statement by statement (std::set iteration orders, float / double conversions, the
// quirks: `all_segment` is never filled upstream, so its duplicate test never fires; FindNeighbors leaves a line with an
// empty neighbour list un-fused).
//
// PARITY UNPINNED, with ONE DELIBERATE, DOCUMENTED REDEFINITION: FuseLines fits a fused group with
// pcl::SACSegmentation (SACMODEL_LINE, SAC_RANSAC, 0.02 m, no refinement; :150-160).  Its 2-point samples come from a
// generator inside PCL, so no restatement can reproduce its choice.  Here — and in the product
// (panovlm_amd/host/pvlm_features.cpp) — the RANSAC is replaced by the limit it approximates: the EXHAUSTIVE 2-point
// maximum-consensus line (all pairs i < j of the group's points, inliers = points with squared distance to the line
// through the pair < 0.02^2, first pair in (i, j) order among equal counts).  Every RANSAC run returns the consensus set
// of SOME pair, so for any PCL seed: |RANSAC inliers| <= |exhaustive inliers|, both sets lie within 0.02 m of a line
// through two of the group's points, and both are rejected at <= 4 inliers.  tests/test_lines_cpu.py checks those
// properties for random pair choices.
// Third-party behaviour restated from memory [recalled]: pcl::KdTreeFLANN::nearestKSearch = exact k nearest, ascending
// float squared distance accumulated as (dx*dx + dy*dy) + dz*dz, equal distances by ascending index (FLANN: traversal
// order); Eigen fixed-size 3-vector norm = sqrt(x*x + (y*y + z*z)).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <set>
#include <vector>

#include "features.hpp"
#include "geometry.hpp"

namespace oracle {

using LVec3 = std::array<double, 3>;
using LVec6 = std::array<double, 6>;

struct LineFeatures {
  std::vector<FPoint> cornerBeforeFilter;
  std::vector<std::vector<FPoint>> edge_segmented;
  std::vector<LVec6> segment_coeffs;              // LiDAR-local (point, unit direction); all zero when FormLine(., 5.0) refused the fused group
  std::vector<LVec3> end_points;                  // 2 per segment
  std::vector<std::set<int>> point_to_segment;    // per point of the filtered cornerLessSharp
};

inline LVec3 PclPoint2Vec(const FPoint& p) { return {(double)p.x, (double)p.y, (double)p.z}; }   // base/common.h:62-65
inline bool IsZero6(const LVec6& l) { for (double v : l) if (v != 0.0) return false; return true; }
inline double Norm3(const LVec3& a, const LVec3& b) {
  const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
  return std::sqrt(x * x + (y * y + z * z));
}

// pcl::KdTreeFLANN<PointXYZI>::nearestKSearch(query, k, idx, sqd)  [recalled, see header]
inline void NearestK(const std::vector<FPoint>& cloud, const FPoint& q, int k, std::vector<int>& idx, std::vector<float>& sqd) {
  std::vector<std::pair<float, int>> d(cloud.size());
  for (size_t i = 0; i < cloud.size(); ++i) {
    const float dx = q.x - cloud[i].x, dy = q.y - cloud[i].y, dz = q.z - cloud[i].z;
    float s = 0.0f;
    s += dx * dx; s += dy * dy; s += dz * dz;
    d[i] = {s, (int)i};
  }
  const size_t kk = std::min<size_t>((size_t)k, d.size());
  std::partial_sort(d.begin(), d.begin() + (std::ptrdiff_t)kk, d.end());
  idx.resize(kk); sqd.resize(kk);
  for (size_t i = 0; i < kk; ++i) { idx[i] = d[i].second; sqd[i] = d[i].first; }
}

// FormLine(points, tolerance, dis_threshold)  base/Geometry.hpp:220-260 — zero vector when the points do not form a line
inline LVec6 FormLineV(const std::vector<LVec3>& pts, double tolerance, double dis_threshold = 0) {
  std::vector<double> flat(pts.size() * 3);
  for (size_t i = 0; i < pts.size(); ++i) { flat[3 * i] = pts[i][0]; flat[3 * i + 1] = pts[i][1]; flat[3 * i + 2] = pts[i][2]; }
  LVec6 l{};
  FormLinePCA(flat.data(), (int)pts.size(), tolerance, dis_threshold, l.data());
  return l;
}

// FurthestPoints(eigen_vector<Vector3d>)  base/Geometry.hpp:594-617
inline bool FurthestPointsV(const std::vector<LVec3>& p, int& idx1, int& idx2, double& distance) {
  if (p.size() <= 1) return false;
  distance = -1; idx1 = idx2 = -1;
  for (int i = 0; i < (int)p.size() - 1; i++)
    for (int j = i + 1; j < (int)p.size(); j++) {
      const double cur = Norm3(p[i], p[j]);
      if (cur > distance) { idx1 = i; idx2 = j; distance = cur; }
    }
  return !(idx1 < 0 || idx2 < 0);
}
// FurthestPoints(pcl::PointCloud)  base/Geometry.hpp:619-645: float squared distances, one sqrt at the end
inline bool FurthestPointsC(const std::vector<FPoint>& p, int& idx1, int& idx2, double& distance) {
  if (p.size() <= 1) return false;
  distance = -1; idx1 = idx2 = -1;
  for (int i = 0; i < (int)p.size() - 1; i++)
    for (int j = i + 1; j < (int)p.size(); j++) {
      const double cur = PointDistanceSquare(p[i], p[j]);
      if (cur > distance) { idx1 = i; idx2 = j; distance = cur; }
    }
  distance = std::sqrt(distance);
  return !(idx1 < 0 || idx2 < 0);
}

// sensors/LidarLineExtraction.cpp:10-70
inline bool ExpandLine(const std::vector<FPoint>& cloud, const FPoint& start_point, std::set<int>& line_points_idx) {
  bool expand = false;
  std::vector<LVec3> line_points;
  for (const int& id : line_points_idx) line_points.emplace_back(PclPoint2Vec(cloud[id]));
  double line_length = 0;
  int a, b;
  FurthestPointsV(line_points, a, b, line_length);
  LVec6 line_coeff = FormLineV(line_points, 3.0);
  std::vector<int> neighbor_idx;
  std::vector<float> neighbor_sq_distance;
  NearestK(cloud, start_point, 5, neighbor_idx, neighbor_sq_distance);
  for (int i = 1; i < (int)neighbor_idx.size(); i++) {
    const int& neighbor = neighbor_idx[i];
    if (line_points_idx.count(neighbor) > 0) continue;
    if (neighbor_sq_distance[i] > (line_length / 2) * (line_length / 2)) break;
    line_points.emplace_back(PclPoint2Vec(cloud[neighbor]));
    double curr_line_length = -1;
    for (LVec3& p : line_points) curr_line_length = std::max(curr_line_length, Norm3(p, line_points[line_points.size() - 1]));
    curr_line_length = std::max(curr_line_length, line_length);
    LVec6 curr_line_coeff;
    if (curr_line_length < 2) {
      curr_line_coeff = FormLineV(line_points, 5.0, 0.07);
      if (IsZero6(curr_line_coeff)) { line_points.pop_back(); continue; }
    } else {
      curr_line_coeff = FormLineV(line_points, 20.0);
      const double line_angle = PlaneAngle<double>(&curr_line_coeff[3], &line_coeff[3]) * 180.0 / M_PI;
      if (IsZero6(curr_line_coeff) || line_angle > 1) { line_points.pop_back(); continue; }
    }
    expand = true;
    line_points_idx.insert(neighbor);
    line_length = curr_line_length;
    line_coeff = curr_line_coeff;
  }
  return expand;
}

// sensors/LidarLineExtraction.cpp:72-110
inline std::vector<int> FindNeighborLines(std::vector<bool>& fused, const std::vector<std::vector<int>>& neighbor_idx, size_t line_idx) {
  std::set<int> group;
  std::vector<int> stack;
  std::vector<int> neighbors = neighbor_idx[line_idx];
  std::vector<int> group_lines;
  if (neighbors.empty()) { group_lines.push_back((int)line_idx); return group_lines; }
  std::vector<bool> visited(neighbor_idx.size(), false);
  stack.insert(stack.end(), neighbors.begin(), neighbors.end());
  group.insert(neighbors.begin(), neighbors.end());
  group.insert((int)line_idx);
  visited[line_idx] = true;
  while (!stack.empty()) {
    int idx = stack[stack.size() - 1];
    stack.pop_back();
    if (visited[idx]) continue;
    visited[idx] = true;
    neighbors = neighbor_idx[idx];
    stack.insert(stack.end(), neighbors.begin(), neighbors.end());
    group.insert(neighbors.begin(), neighbors.end());
  }
  for (const int& g : group) { group_lines.push_back(g); fused[g] = true; }
  return group_lines;
}

// The redefinition of pcl::SACSegmentation (see header): exhaustive 2-point maximum consensus, squared threshold test
// |(p - p_i) x (p_j - p_i)|^2 < thr^2 |p_j - p_i|^2 in double, cross-product components squared and summed as (x + y) + z.
inline std::vector<int> LineConsensus(const std::vector<FPoint>& cloud, double threshold) {
  const int n = (int)cloud.size();
  int best = 0, bi = -1, bj = -1;
  const double t2 = threshold * threshold;
  auto inlier = [&](const LVec3& o, const LVec3& d, double len2, const FPoint& q) {
    const double x = (double)q.x - o[0], y = (double)q.y - o[1], z = (double)q.z - o[2];
    const double cx = y * d[2] - z * d[1], cy = z * d[0] - x * d[2], cz = x * d[1] - y * d[0];
    return (cx * cx + cy * cy) + cz * cz < t2 * len2;
  };
  for (int i = 0; i < n - 1; ++i) {
    const LVec3 o = PclPoint2Vec(cloud[i]);
    for (int j = i + 1; j < n; ++j) {
      const LVec3 d = {(double)cloud[j].x - o[0], (double)cloud[j].y - o[1], (double)cloud[j].z - o[2]};
      const double len2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
      if (!(len2 > 0.0)) continue;
      int c = 0;
      for (int k = 0; k < n; ++k) c += inlier(o, d, len2, cloud[k]) ? 1 : 0;
      if (c > best) { best = c; bi = i; bj = j; }
    }
  }
  std::vector<int> in;
  if (bi < 0) return in;
  const LVec3 o = PclPoint2Vec(cloud[bi]);
  const LVec3 d = {(double)cloud[bj].x - o[0], (double)cloud[bj].y - o[1], (double)cloud[bj].z - o[2]};
  const double len2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
  for (int k = 0; k < n; ++k) if (inlier(o, d, len2, cloud[k])) in.push_back(k);
  return in;
}

// sensors/LidarLineExtraction.cpp:113-175 (optimize = true is the only call, :241)
inline void FuseLines(const std::vector<std::vector<FPoint>>& segments, const std::vector<LVec6>& line_coeffs, const std::vector<int>& neighbors,
                      std::vector<FPoint>& cloud_fused, LVec6& line_coeff) {
  if (neighbors.empty()) return;
  if (neighbors.size() == 1) { cloud_fused = segments[neighbors[0]]; line_coeff = line_coeffs[neighbors[0]]; return; }
  std::vector<FPoint> cloud;
  std::set<int> point_id;
  for (const int& n : neighbors)
    for (const FPoint& p : segments[n]) {
      if (point_id.count(int(p.intensity)) > 0) continue;
      point_id.insert(int(p.intensity));
      cloud.push_back(p);
    }
  const std::vector<int> inliers = LineConsensus(cloud, 0.02);
  if (inliers.size() <= 4) { cloud_fused.clear(); line_coeff = LVec6{}; return; }
  std::vector<LVec3> line_points;
  for (const int& idx : inliers) { cloud_fused.push_back(cloud[idx]); line_points.emplace_back(PclPoint2Vec(cloud[idx])); }
  line_coeff = FormLineV(line_points, 5.0);
}

// sensors/LidarLineExtraction.cpp:177-250
inline void FuseLineSegments(std::vector<std::vector<FPoint>>& points_each_line, std::vector<LVec6>& line_coeffs) {
  const double angle_threshold = 3.0 / 180.0 * M_PI;
  std::vector<std::set<int>> segment_ids;
  for (const auto& seg : points_each_line) {
    std::set<int> ids;
    for (const auto& p : seg) ids.insert(int(p.intensity));
    segment_ids.push_back(ids);
  }
  std::vector<FPoint> line_centers;                                  // pcl::PointXYZ(coeff[0], coeff[1], coeff[2]): double -> float
  for (const LVec6& coeff : line_coeffs) line_centers.push_back(FPoint{(float)coeff[0], (float)coeff[1], (float)coeff[2], 0.f});
  std::vector<std::vector<int>> neighbor_idx(line_centers.size());
  for (size_t i = 0; i < line_centers.size(); i++) {
    std::vector<int> curr_neighbor;
    std::vector<float> sq_dist;
    NearestK(line_centers, line_centers[i], 5, curr_neighbor, sq_dist);
    for (size_t j = 0; j < curr_neighbor.size(); j++) {
      if (sq_dist[j] > 1) break;
      int n_idx = curr_neighbor[j];
      double distance = PointToLineDistance3D(line_coeffs[i].data(), line_coeffs[n_idx].data());
      if (distance > 0.2) continue;
      distance = PointToLineDistance3D(line_coeffs[n_idx].data(), line_coeffs[i].data());
      if (distance > 0.2) continue;
      double line_angle = PlaneAngle<double>(&line_coeffs[i][3], &line_coeffs[n_idx][3]);
      if (line_angle > angle_threshold) continue;
      const std::set<int>& curr_ids = segment_ids[i];
      const std::set<int>& neighbor_ids = segment_ids[n_idx];
      size_t num_same = 0;
      for (const int& id : curr_ids) num_same += neighbor_ids.count(id);
      if (num_same <= 2) continue;
      neighbor_idx[i].push_back(n_idx);
    }
  }
  std::vector<std::vector<FPoint>> lines_fused;
  std::vector<LVec6> line_coeffs_fused;
  std::vector<bool> fused(line_centers.size(), false);
  for (size_t line_idx = 0; line_idx < line_centers.size(); line_idx++) {
    if (fused[line_idx]) continue;
    std::vector<int> group_lines = FindNeighborLines(fused, neighbor_idx, line_idx);
    std::vector<FPoint> cloud_fused;
    LVec6 coeff_fused{};
    FuseLines(points_each_line, line_coeffs, group_lines, cloud_fused, coeff_fused);
    if (!cloud_fused.empty()) { lines_fused.push_back(cloud_fused); line_coeffs_fused.push_back(coeff_fused); }
  }
  points_each_line.swap(lines_fused);
  line_coeffs.swap(line_coeffs_fused);
}

// sensors/LidarLineExtraction.cpp:254-294
inline void FilterLineByLength(std::vector<std::vector<FPoint>>& points_each_line, std::vector<LVec6>& line_coeffs) {
  float distance_threshold = 0.3;
  std::vector<std::vector<FPoint>> good_lines;
  std::vector<LVec6> good_line_coeffs;
  for (size_t line_id = 0; line_id < points_each_line.size(); line_id++) {
    double max_distance = -1;
    int start = -1, end = -1;
    FurthestPointsC(points_each_line[line_id], start, end, max_distance);
    if (max_distance > distance_threshold) { good_lines.push_back(points_each_line[line_id]); good_line_coeffs.push_back(line_coeffs[line_id]); }
  }
  points_each_line = good_lines;
  line_coeffs = good_line_coeffs;
}
inline void FilterLineByScan(const std::vector<std::pair<int, int>>& point_idx_to_image, std::vector<std::vector<FPoint>>& points_each_line,
                             std::vector<LVec6>& line_coeffs) {
  std::vector<std::vector<FPoint>> good_lines;
  std::vector<LVec6> good_line_coeffs;
  for (size_t line_id = 0; line_id < points_each_line.size(); line_id++) {
    const std::vector<FPoint>& line_points = points_each_line[line_id];
    std::set<size_t> scan_ids;
    for (const auto& p : line_points) scan_ids.insert((size_t)point_idx_to_image[static_cast<int>(p.intensity)].first);
    if (scan_ids.size() >= line_points.size() / 2 && scan_ids.size() >= 3) { good_lines.push_back(line_points); good_line_coeffs.push_back(line_coeffs[line_id]); }
  }
  points_each_line = good_lines;
  line_coeffs = good_line_coeffs;
}

// sensors/LidarLineExtraction.cpp:296-389
inline void ExtractLineFeatures(const std::vector<FPoint>& edge_points, const std::vector<std::pair<int, int>>& point_idx_to_image,
                                std::vector<std::vector<FPoint>>& points_each_line, std::vector<LVec6>& line_coeffs) {
  if (edge_points.empty()) return;
  std::vector<unsigned char> visited(edge_points.size(), 0);
  for (size_t idx = 0; idx < edge_points.size(); idx++) {
    if (visited[idx]) continue;
    visited[idx] = true;
    std::vector<int> pointSearchInd;
    std::vector<float> pointSearchSqDis;
    NearestK(edge_points, edge_points[idx], 5, pointSearchInd, pointSearchSqDis);
    std::vector<std::set<int>> all_segment;            // never filled upstream either: the duplicate test below never fires
    std::vector<LVec3> line_points;
    for (int i = 1; i < (int)pointSearchInd.size() - 1; i++) {
      for (int j = i + 1; j < (int)pointSearchInd.size(); j++) {
        line_points.clear();
        int idx1 = pointSearchInd[i];
        int idx2 = pointSearchInd[j];
        bool skip = false;
        for (const std::set<int>& seg : all_segment)
          if (seg.count(idx1) > 0 && seg.count(idx2) > 0) { skip = true; break; }
        if (skip) continue;
        line_points.push_back(PclPoint2Vec(edge_points[idx]));
        line_points.push_back(PclPoint2Vec(edge_points[idx1]));
        line_points.push_back(PclPoint2Vec(edge_points[idx2]));
        if (IsZero6(FormLineV(line_points, 5.0))) continue;
        std::set<int> curr_segment = {(int)idx, idx1, idx2};
        int line_start = 0, line_end = 0;
        double line_length = 0;
        FurthestPointsV(line_points, line_start, line_end, line_length);
        bool expand1 = true, expand2 = true;
        while (expand1 || expand2) {
          FPoint tmp{(float)line_points[line_start][0], (float)line_points[line_start][1], (float)line_points[line_start][2], 0.f};
          expand1 = ExpandLine(edge_points, tmp, curr_segment);
          tmp = FPoint{(float)line_points[line_end][0], (float)line_points[line_end][1], (float)line_points[line_end][2], 0.f};
          expand2 = ExpandLine(edge_points, tmp, curr_segment);
          line_points.clear();
          for (const auto& id : curr_segment) line_points.emplace_back(PclPoint2Vec(edge_points[id]));
          FurthestPointsV(line_points, line_start, line_end, line_length);
        }
        if (curr_segment.size() >= 5) {
          std::vector<FPoint> cloud;
          for (const int& s : curr_segment) { visited[s] = true; cloud.push_back(edge_points[s]); }
          points_each_line.push_back(cloud);
          line_coeffs.push_back(FormLineV(line_points, 1.0));
        }
      }
    }
  }
  FuseLineSegments(points_each_line, line_coeffs);
  FilterLineByScan(point_idx_to_image, points_each_line, line_coeffs);
  FilterLineByLength(points_each_line, line_coeffs);
}

// ProjectPoint2Line3D(Vector3d, const double* line)  base/Geometry.hpp:180-191
inline LVec3 ProjectPoint2Line3D(const LVec3& point, const double* line) {
  const double x0 = line[0], y0 = line[1], z0 = line[2], nx = line[3], ny = line[4], nz = line[5];
  const double k = (nx * (point[0] - x0) + ny * (point[1] - y0) + nz * (point[2] - z0)) / (nx * nx + ny * ny + nz * nz);
  return {k * nx + x0, k * ny + y0, k * nz + z0};
}

// Velodyne::EdgeToLine  sensors/Velodyne.cpp:1269-1324.  In: f.cornerLessSharp / f.cornerSharp as ExtractEdgeFeatures2 left them
// (intensity = index into cloud_scan).  Out: the filtered clouds in f, the segments in L.
inline void EdgeToLine(ScanFeatures& f, LineFeatures& L) {
  L = LineFeatures();
  L.cornerBeforeFilter = f.cornerLessSharp;
  ExtractLineFeatures(f.cornerLessSharp, f.point_idx_to_image, L.edge_segmented, L.segment_coeffs);
  for (size_t i = 0; i < L.edge_segmented.size(); i++) {
    int start = -1, end = -1;
    double distance;
    FurthestPointsC(L.edge_segmented[i], start, end, distance);
    L.end_points.push_back(ProjectPoint2Line3D(PclPoint2Vec(L.edge_segmented[i][start]), L.segment_coeffs[i].data()));
    L.end_points.push_back(ProjectPoint2Line3D(PclPoint2Vec(L.edge_segmented[i][end]), L.segment_coeffs[i].data()));
  }
  f.cornerLessSharp.clear();
  std::map<int, int> id_to_idx;
  for (size_t seg_id = 0; seg_id < L.edge_segmented.size(); seg_id++)
    for (const FPoint& p : L.edge_segmented[seg_id]) {
      auto it = id_to_idx.find(int(p.intensity));
      if (it != id_to_idx.end()) L.point_to_segment[it->second].insert((int)seg_id);
      else {
        id_to_idx[int(p.intensity)] = (int)f.cornerLessSharp.size();
        f.cornerLessSharp.push_back(p);
        L.point_to_segment.push_back(std::set<int>{(int)seg_id});
      }
    }
  std::vector<FPoint> tmp = f.cornerSharp;
  f.cornerSharp.clear();
  std::set<int> idx;
  for (const FPoint& p : f.cornerLessSharp) idx.insert(int(p.intensity));
  for (const FPoint& p : tmp) {
    if (idx.count((int)p.intensity) == 0) continue;
    f.cornerSharp.emplace_back(p);
  }
}

}  // namespace oracle
