// ORACLE — test infrastructure only (see oracle/__init__.py).  CPU restatement of the PLANAR branch of the reference's
// LiDAR feature extraction, the step in front of AssociatePoint2Plane:
//   Velodyne::ReOrderVLP            sensors/Velodyne.cpp:371-526   raw firing order -> ring order + range image
//   Velodyne::Segmentation          sensors/Velodyne.cpp:1438-1586 range-image BFS labelling, small segments removed
//   Velodyne::ExtractFeatures       sensors/Velodyne.cpp:531-760   ADAPTIVE curvature (:623-657) + per-sector sort (:707-723)
//   Velodyne::ExtractEdgeFeatures2  sensors/Velodyne.cpp:883-1000  greedy edge picks (they change cloudState, which the planar picks read)
//   Velodyne::ExtractPlaneFeatures2 sensors/Velodyne.cpp:1098-1189 surfFlat (<= 4 per sector) + surfLessFlat through pcl::VoxelGrid
// Written to follow the reference statement by statement, float for float: the file has `using namespace std`
// (sensors/Velodyne.cpp:7), so sqrt / atan / atan2 / acos / abs / sin / cos on float arguments are the FLOAT overloads.
//
// PARITY UNPINNED (the reference has no tests or vectors for this path, and PCL is not in the image).  Two third-party
// behaviours are restated from memory and marked [recalled]:
//   * pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.10 filters/impl/voxel_grid.hpp): voxel index from
//     floor(x * inverse_leaf) - min_b, std::sort of (idx, point index) pairs by idx only, centroid of all fields summed
//     in sorted order (CentroidPoint: Vector3f sum / n, float intensity sum / n), output in ascending voxel order.
//   * Eigen 3.4 fixed-size 3-vector reductions (dot, squaredNorm) without vectorisation: c0 + (c1 + c2).
// std::sort is libstdc++'s here as in a reference build; with equal keys (curvature ties, points of one voxel) its
// order is deterministic for a given library but not specified by the language.
// EdgeToLine / ExtractLineFeatures (sensors/LidarLineExtraction.cpp) live in oracle/lines.hpp; ExtractFeatures runs them
// (between the edge and the planar picks, as upstream) when it is handed a LineFeatures object, otherwise cornerLessSharp
// stays the cloud BEFORE that filter (the reference keeps it as cornerBeforeFilter, sensors/Velodyne.cpp:1271).
// Two out-of-bounds reads of the reference (undefined behaviour there) are made safe, see "UB" below.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace oracle {

struct FPoint { float x, y, z, intensity; };

// sensors/Velodyne.h:57-66
enum : int { F_NORMAL = 0x01, F_LESS_SHARP = 0x02, F_SHARP = 0x04, F_FLAT = 0x08, F_GROUND = 0x10, F_DISABLE = 0x20, F_OCCLUDED = 0x40 };

struct ScanFeatures {
  int n_scans = 16, horizon = 1800;
  bool valid = true;
  std::vector<FPoint> cloud_scan;                     // ring order; intensity = ring id
  std::vector<float> range_image;                     // n_scans x horizon, row-major, 0 = empty
  std::vector<std::pair<int, int>> point_idx_to_image;
  std::vector<int> image_to_point_idx;                // n_scans x horizon, -1 = empty
  std::vector<int> scanStartInd, scanEndInd;
  std::vector<float> curvature;
  std::vector<int> state, sortInd, left, right;
  std::vector<FPoint> cornerSharp, cornerLessSharp, surfFlat, surfLessFlat;
};

struct LineFeatures;                                        // oracle/lines.hpp
inline void EdgeToLine(ScanFeatures& f, LineFeatures& L);   // oracle/lines.hpp (sensors/Velodyne.cpp:1269-1324)

inline float FSquare(float a) { return a * a; }
// base/Geometry.hpp:38-40
inline float PointDistanceSquare(const FPoint& a, const FPoint& b) { return FSquare(a.x - b.x) + FSquare(a.y - b.y) + FSquare(a.z - b.z); }

// sensors/Velodyne.cpp:170-211
inline int VerticalAngleToScanID(float vertical_angle, int max_scan) {
  if (!(vertical_angle == vertical_angle)) return -1;   // NaN (point at the origin): int(NaN) is undefined upstream; on x86-64 it comes out negative = rejected
  int scanID = -1;
  if (max_scan == 16) {
    scanID = int((vertical_angle + 15) / 2 + 0.5);
    if (scanID > (max_scan - 1) || scanID < 0) scanID = -1;
  } else if (max_scan == 32) {
    scanID = int((vertical_angle + 92.0 / 3.0) * 3.0 / 4.0);
    if (scanID > (max_scan - 1) || scanID < 0) scanID = -1;
  } else if (max_scan == 64) {
    if (vertical_angle >= -8.83) scanID = int((2 - vertical_angle) * 3.0 + 0.5);
    else scanID = max_scan / 2 + int((-8.83 - vertical_angle) * 2.0 + 0.5);
    if (vertical_angle > 2 || vertical_angle < -24.33 || scanID > 50 || scanID < 0) scanID = -1;
  }
  return scanID;
}

// sensors/Velodyne.cpp:371-526.  cloud: LoadLidar's output (camera-style axes).
inline void ReOrderVLP(const std::vector<FPoint>& cloud, int n_scans, int horizon, ScanFeatures& f) {
  f = ScanFeatures();
  f.n_scans = n_scans; f.horizon = horizon;
  if (n_scans <= 0 || horizon <= 0) return;   // not a range image (upstream would divide by zero / never leave the column loop)
  f.image_to_point_idx.assign((size_t)n_scans * horizon, -1);
  f.scanStartInd.assign(n_scans, 0); f.scanEndInd.assign(n_scans, 0);
  if (n_scans != 16 && n_scans != 32 && n_scans != 64) return;
  const double horizon_resolution = 2.0 * M_PI / horizon;
  f.range_image.assign((size_t)n_scans * horizon, 0.f);
  const int cloudSize = (int)cloud.size();
  if (cloudSize == 0) return;   // the reference reads cloud.points[0] unconditionally; LoadLidar invalidates scans under 4000 points
  double start_ori = std::atan2(cloud[0].x, cloud[0].z);
  if (start_ori < 0) start_ori += 2 * M_PI;
  // scan_order (std::map, :407-414): firing position of a ring; operator[] on a missing key (-1 before the first point,
  // any ring when n_scans != 16) inserts 0
  auto scan_order = [&](int id) -> int { if (n_scans != 16 || id < 0) return 0; return id <= 7 ? 2 * id : 2 * id - 15; };
  bool cross_z_axis = false;
  double last_ori = -1;
  std::vector<std::vector<FPoint>> laserCloudScans(n_scans);
  std::vector<std::vector<std::pair<size_t, size_t>>> point_idx_to_col(n_scans);
  int col_offset = 0;
  int last_col = 0, last_scan = -1;
  size_t cloud_scan_count = 0;
  for (int i = 0; i < cloudSize; i++) {
    FPoint point = cloud[i];
    float vertical_angle = std::atan(-point.y / std::sqrt(point.x * point.x + point.z * point.z)) * 180 / M_PI;
    int scanID = VerticalAngleToScanID(vertical_angle, n_scans);
    if (scanID == -1) continue;
    double ori = std::atan2(point.x, point.z);
    if (ori < 0) ori += 2 * M_PI;
    if (ori < last_ori && cross_z_axis == false) {
      int reliable = 0;
      int reliable_threshold = n_scans;
      for (int idx = i + 1; idx < i + n_scans + 1 && idx < cloudSize; idx++) {
        double angle = std::atan2(cloud[idx].x, cloud[idx].z);
        if (angle < 0) angle += 2 * M_PI;
        reliable += (angle < last_ori);
        if (reliable >= reliable_threshold) break;
      }
      cross_z_axis = (reliable >= reliable_threshold);
    }
    ori += 2 * M_PI * cross_z_axis;
    int row_index = scanID;
    int col_index = std::round((ori - start_ori) / horizon_resolution);
    if (scan_order(scanID) < scan_order(last_scan)) {
      col_offset = (last_col == col_index);
      last_col = col_index + col_offset;
    }
    last_scan = scanID;
    col_index += col_offset;
    while (col_index >= horizon) col_index -= horizon;
    if (col_index < 0) continue;
    point.intensity = scanID;
    f.range_image[(size_t)row_index * horizon + col_index] = std::sqrt(point.x * point.x + point.y * point.y + point.z * point.z);
    point_idx_to_col[scanID].push_back(std::pair<size_t, size_t>(laserCloudScans[scanID].size(), col_index));
    laserCloudScans[scanID].push_back(point);
    cloud_scan_count++;
    last_ori = ori;
  }
  f.point_idx_to_image.resize(cloud_scan_count);
  for (int i = 0; i < n_scans; i++) {
    for (const auto& idx_col : point_idx_to_col[i]) {
      const size_t point_idx = f.cloud_scan.size() + idx_col.first;
      f.point_idx_to_image[point_idx] = std::pair<int, int>(i, (int)idx_col.second);
      f.image_to_point_idx[(size_t)i * horizon + idx_col.second] = (int)point_idx;
    }
    f.scanStartInd[i] = (int)(f.cloud_scan.size() + 5);
    f.cloud_scan.insert(f.cloud_scan.end(), laserCloudScans[i].begin(), laserCloudScans[i].end());
    f.scanEndInd[i] = (int)(f.cloud_scan.size() - 6);
  }
}

// sensors/Velodyne.cpp:1438-1586 ("Fast Range Image Segmentation", LeGO-LOAM style labelling)
inline void Segmentation(ScanFeatures& f) {
  const int N_SCANS = f.n_scans, horizon_scans = f.horizon;
  std::vector<int> label_image((size_t)N_SCANS * horizon_scans, 0);
  auto label = [&](int r, int c) -> int& { return label_image[(size_t)r * horizon_scans + c]; };
  auto range = [&](int r, int c) -> float { return f.range_image[(size_t)r * horizon_scans + c]; };
  std::vector<uint16_t> queueIndX((size_t)N_SCANS * horizon_scans), queueIndY((size_t)N_SCANS * horizon_scans);
  std::vector<uint16_t> allPushedIndX((size_t)N_SCANS * horizon_scans), allPushedIndY((size_t)N_SCANS * horizon_scans);
  const int8_t nb[4][2] = {{-1, 0}, {0, 1}, {0, -1}, {1, 0}};
  float segmentAlphaX = 0.2 / 180.0 * M_PI;
  float segmentAlphaY = 2.0 / 180.0 * M_PI;
  float segmentTheta = 20.0 / 180.0 * M_PI;
  int label_count = 1;
  for (int row = 0; row < N_SCANS; ++row)
    for (int col = 0; col < horizon_scans; ++col)
      if (label(row, col) == 0) {
        float d1, d2, alpha, angle;
        int fromIndX, fromIndY, thisIndX, thisIndY;
        std::vector<char> lineCountFlag(N_SCANS, 0);
        queueIndX[0] = row; queueIndY[0] = col;
        int queueSize = 1, queueStartInd = 0, queueEndInd = 1;
        allPushedIndX[0] = row; allPushedIndY[0] = col;
        int allPushedIndSize = 1;
        while (queueSize > 0) {
          fromIndX = queueIndX[queueStartInd];
          fromIndY = queueIndY[queueStartInd];
          --queueSize;
          ++queueStartInd;
          label(fromIndX, fromIndY) = label_count;
          for (int k = 0; k < 4; ++k) {
            thisIndX = fromIndX + nb[k][0];
            thisIndY = fromIndY + nb[k][1];
            if (thisIndX < 0 || thisIndX >= N_SCANS) continue;
            if (thisIndY < 0) thisIndY = horizon_scans - 1;
            if (thisIndY >= horizon_scans) thisIndY = 0;
            if (label(thisIndX, thisIndY) != 0) continue;
            d1 = std::max(range(fromIndX, fromIndY), range(thisIndX, thisIndY));
            d2 = std::min(range(fromIndX, fromIndY), range(thisIndX, thisIndY));
            alpha = nb[k][0] == 0 ? segmentAlphaX : segmentAlphaY;
            angle = std::atan2(d2 * std::sin(alpha), (d1 - d2 * std::cos(alpha)));
            if (angle > segmentTheta) {
              queueIndX[queueEndInd] = thisIndX; queueIndY[queueEndInd] = thisIndY;
              ++queueSize; ++queueEndInd;
              label(thisIndX, thisIndY) = label_count;
              lineCountFlag[thisIndX] = 1;
              allPushedIndX[allPushedIndSize] = thisIndX; allPushedIndY[allPushedIndSize] = thisIndY;
              ++allPushedIndSize;
            }
          }
        }
        bool feasibleSegment = false;
        if (allPushedIndSize >= 30) feasibleSegment = true;
        else if (allPushedIndSize >= 5) {
          int lineCount = 0;
          for (int i = 0; i < N_SCANS; ++i) if (lineCountFlag[i]) ++lineCount;
          if (lineCount >= 3) feasibleSegment = true;
        }
        if (feasibleSegment) ++label_count;
        else for (int i = 0; i < allPushedIndSize; ++i) label(allPushedIndX[i], allPushedIndY[i]) = INT16_MAX;
      }
  std::vector<std::vector<FPoint>> laserCloudScans(N_SCANS);
  std::vector<std::pair<int, int>> point_idx_to_image_new;
  std::vector<int> image_to_point_idx_new((size_t)N_SCANS * horizon_scans, -1);
  int count = 0;
  for (size_t idx = 0; idx < f.cloud_scan.size(); idx++) {
    const auto rc = f.point_idx_to_image[idx];
    if (label(rc.first, rc.second) == INT16_MAX) continue;
    point_idx_to_image_new.push_back(rc);
    image_to_point_idx_new[(size_t)rc.first * horizon_scans + rc.second] = count;
    count++;
    laserCloudScans[(int)f.cloud_scan[idx].intensity].push_back(f.cloud_scan[idx]);
  }
  f.cloud_scan.clear();
  f.point_idx_to_image.swap(point_idx_to_image_new);
  f.image_to_point_idx.swap(image_to_point_idx_new);
  for (int i = 0; i < N_SCANS; i++) {
    f.scanStartInd[i] = (int)(f.cloud_scan.size() + 5);
    f.cloud_scan.insert(f.cloud_scan.end(), laserCloudScans[i].begin(), laserCloudScans[i].end());
    f.scanEndInd[i] = (int)(f.cloud_scan.size() - 6);
  }
}

// pcl::VoxelGrid<PointXYZI> with setLeafSize(l, l, l), downsample_all_data = true, no filter field  [recalled, see header]
inline std::vector<FPoint> VoxelGrid(const std::vector<FPoint>& in, float leaf) {
  std::vector<FPoint> out;
  if (in.empty()) return out;
  const float inv = 1.f / leaf;
  float mn[3] = {in[0].x, in[0].y, in[0].z}, mx[3] = {in[0].x, in[0].y, in[0].z};
  for (const FPoint& p : in) {
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
  }
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) return in;   // "Leaf size is too small": PCL returns the input unchanged
  int min_b[3], max_b[3], div_b[3];
  for (int k = 0; k < 3; ++k) { min_b[k] = (int)std::floor(mn[k] * inv); max_b[k] = (int)std::floor(mx[k] * inv); div_b[k] = max_b[k] - min_b[k] + 1; }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  struct Item { unsigned idx, pt; bool operator<(const Item& o) const { return idx < o.idx; } };
  std::vector<Item> items;
  items.reserve(in.size());
  for (unsigned i = 0; i < in.size(); ++i) {
    const FPoint& p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const int i0 = (int)(std::floor(p.x * inv) - (float)min_b[0]), i1 = (int)(std::floor(p.y * inv) - (float)min_b[1]),
              i2 = (int)(std::floor(p.z * inv) - (float)min_b[2]);
    items.push_back(Item{(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), i});
  }
  std::sort(items.begin(), items.end(), std::less<Item>());
  for (size_t a = 0; a < items.size();) {
    size_t b = a + 1;
    while (b < items.size() && items[b].idx == items[a].idx) ++b;
    float sx = 0, sy = 0, sz = 0, si = 0;
    for (size_t k = a; k < b; ++k) { const FPoint& p = in[items[k].pt]; sx += p.x; sy += p.y; sz += p.z; si += p.intensity; }
    const float n = (float)(b - a);
    out.push_back(FPoint{sx / n, sy / n, sz / n, si / n});
    a = b;
  }
  return out;
}

// Velodyne::ExtractFeatures with method == ADAPTIVE (sensors/Velodyne.cpp:531-760), up to and including ExtractPlaneFeatures2
inline void ExtractFeatures(ScanFeatures& f, float max_curvature, float intersect_angle_threshold, bool segment, LineFeatures* lines = nullptr) {
  if (!f.valid || f.cloud_scan.empty()) return;
  int cloudSize = (int)f.cloud_scan.size();
  if (segment) Segmentation(f);
  if (f.cloud_scan.size() < cloudSize * 0.1) { f.valid = false; return; }
  cloudSize = (int)f.cloud_scan.size();
  const size_t neighbor_size = 5;
  const int N_SCANS = f.n_scans;
  f.curvature.assign(cloudSize, -1.f);
  f.state.assign(cloudSize, F_NORMAL);
  f.sortInd.resize(cloudSize);
  f.left.assign(cloudSize, -1); f.right.assign(cloudSize, -1);
  for (int i = 0; i < cloudSize; ++i) f.sortInd[i] = i;
  std::vector<float> cloudDistance(cloudSize);
  for (int i = 0; i < cloudSize; ++i) cloudDistance[i] = f.range_image[(size_t)f.point_idx_to_image[i].first * f.horizon + f.point_idx_to_image[i].second];
  const std::vector<FPoint>& P = f.cloud_scan;
  // ADAPTIVE curvature :623-657.  Statements kept as written, including the two that test left_idx where right_idx was
  // meant (:637 and :642).  UB: the right-hand walk / the summation can index past the end of the cloud on the last
  // ring; here the walk stops at the end of the cloud and a window that would leave it gives no curvature (-1).
  for (int scan_idx = 0; scan_idx < N_SCANS; scan_idx++) {
    if ((size_t)(f.scanEndInd[scan_idx] - f.scanStartInd[scan_idx]) < neighbor_size) continue;   // int -> size_t, as compiled
    for (int idx = f.scanStartInd[scan_idx]; idx <= f.scanEndInd[scan_idx]; idx++) {
      int left_idx = idx - (int)neighbor_size;
      int right_idx = idx + (int)neighbor_size;
      while (left_idx >= f.scanStartInd[scan_idx] && PointDistanceSquare(P[left_idx], P[idx]) < 0.0064) left_idx--;
      while (left_idx <= f.scanEndInd[scan_idx] && right_idx < cloudSize && PointDistanceSquare(P[right_idx], P[idx]) < 0.0064) right_idx++;
      int max_diff_idx = std::max(idx - left_idx, right_idx - idx);
      left_idx = idx - max_diff_idx;
      right_idx = idx + max_diff_idx;
      if (!(left_idx >= f.scanStartInd[scan_idx] - 5 && left_idx <= f.scanEndInd[scan_idx] + 5)) continue;
      if (right_idx >= cloudSize) continue;   // UB in the reference
      float diff_depth = 0;
      for (int i = left_idx; i <= right_idx; i++) diff_depth += cloudDistance[i];
      diff_depth -= (right_idx - left_idx + 1) * cloudDistance[idx];
      diff_depth /= (right_idx - left_idx);
      f.curvature[idx] = std::abs(diff_depth);
      f.left[idx] = left_idx;
      f.right[idx] = right_idx;
    }
  }
  // per-sector sort :707-723
  for (int i = 0; i < N_SCANS; i++) {
    if (f.scanEndInd[i] - f.scanStartInd[i] < 6) continue;
    for (int j = 0; j < 6; j++) {
      int sp = f.scanStartInd[i] + (f.scanEndInd[i] - f.scanStartInd[i]) * j / 6;
      int ep = f.scanStartInd[i] + (f.scanEndInd[i] - f.scanStartInd[i]) * (j + 1) / 6 - 1;
      const float* curv = f.curvature.data();
      std::sort(f.sortInd.begin() + sp, f.sortInd.begin() + ep + 1, [curv](int a, int b) -> bool { return curv[a] < curv[b]; });
    }
  }
  // ---- ExtractEdgeFeatures2 :883-1000
  for (int i = 0; i < N_SCANS; i++) {
    if (f.scanEndInd[i] - f.scanStartInd[i] < 6) continue;
    for (int j = 0; j < 6; j++) {
      int sp = f.scanStartInd[i] + (f.scanEndInd[i] - f.scanStartInd[i]) * j / 6;
      int ep = f.scanStartInd[i] + (f.scanEndInd[i] - f.scanStartInd[i]) * (j + 1) / 6 - 1;
      int largestPickedNum = 0;
      for (int k = ep; k >= sp; k--) {
        int ind = f.sortInd[k];
        if (f.state[ind] != F_NORMAL) continue;
        if (f.curvature[ind] > max_curvature || f.curvature[ind] < 0.1) continue;
        const FPoint& a = P[ind]; const FPoint& l = P[f.left[ind]]; const FPoint& r = P[f.right[ind]];
        const float bx = l.x - r.x, by = l.y - r.y, bz = l.z - r.z;
        const float dot = a.x * bx + (a.y * by + a.z * bz);          // Eigen 3-vector reduction order [recalled]
        const float nb = std::sqrt(bx * bx + (by * by + bz * bz));
        float view_angle = std::acos(std::abs(dot) / (f.range_image[(size_t)f.point_idx_to_image[ind].first * f.horizon + f.point_idx_to_image[ind].second] * nb));
        view_angle *= (180.0 / M_PI);
        if (view_angle < intersect_angle_threshold || view_angle > 180 - intersect_angle_threshold) continue;
        largestPickedNum++;
        if (largestPickedNum <= 3) {
          f.state[ind] = F_SHARP;
          FPoint p = P[ind]; p.intensity = ind;
          f.cornerSharp.push_back(p); f.cornerLessSharp.push_back(p);
        } else if (largestPickedNum <= 30) {
          f.state[ind] = F_LESS_SHARP;
          FPoint p = P[ind]; p.intensity = ind;
          f.cornerLessSharp.push_back(p);
        } else break;
        for (int l2 = 1; ind + l2 <= f.scanEndInd[i]; l2++) {
          if (l2 <= 5 && PointDistanceSquare(P[ind + l2], P[ind + l2 - 1]) > 0.05) break;
          else if (l2 > 5 && PointDistanceSquare(P[ind + l2], P[ind]) > 0.0036) break;
          f.state[ind + l2] |= F_DISABLE;
        }
        for (int l2 = 1; ind - l2 >= f.scanStartInd[i]; l2++) {
          if (l2 <= 5 && PointDistanceSquare(P[ind - l2], P[ind - l2 + 1]) > 0.05) break;
          else if (l2 > 5 && PointDistanceSquare(P[ind - l2], P[ind]) > 0.0036) break;
          f.state[ind - l2] |= F_DISABLE;
        }
      }
    }
  }
  // EdgeToLine :752 — filters cornerLessSharp / cornerSharp down to the members of line segments; does not touch cloudState
  if (lines) EdgeToLine(f, *lines);
  // ---- ExtractPlaneFeatures2 :1098-1189
  for (int i = 0; i < N_SCANS; i++) {
    if (f.scanEndInd[i] - f.scanStartInd[i] < 6) continue;
    std::vector<FPoint> surfPointsLessFlatScan;
    for (int j = 0; j < 6; j++) {
      int sp = f.scanStartInd[i] + (f.scanEndInd[i] - f.scanStartInd[i]) * j / 6;
      int ep = f.scanStartInd[i] + (f.scanEndInd[i] - f.scanStartInd[i]) * (j + 1) / 6 - 1;
      int smallestPickedNum = 0;
      for (int k = sp; k <= ep; k++) {
        int ind = f.sortInd[k];
        if (f.state[ind] != F_NORMAL && f.state[ind] != F_GROUND) continue;
        if (f.curvature[ind] > 0.02) continue;
        FPoint pt = P[ind];
        pt.intensity = f.state[ind];
        f.surfFlat.push_back(pt);
        if (f.state[ind] == F_NORMAL) surfPointsLessFlatScan.push_back(pt);
        f.state[ind] |= F_FLAT;
        smallestPickedNum++;
        for (int l2 = 1; ind + l2 <= f.scanEndInd[i]; l2++) {
          if (l2 <= 5 && PointDistanceSquare(P[ind + l2], P[ind + l2 - 1]) > 0.05) break;
          else if (l2 > 5 && PointDistanceSquare(P[ind + l2], P[ind]) > 0.0036) break;
          f.state[ind + l2] |= F_DISABLE;
        }
        for (int l2 = 1; ind - l2 >= f.scanStartInd[i]; l2++) {
          if (l2 <= 5 && PointDistanceSquare(P[ind - l2], P[ind - l2 + 1]) > 0.05) break;
          else if (l2 > 5 && PointDistanceSquare(P[ind - l2], P[ind]) > 0.0036) break;
          f.state[ind - l2] |= F_DISABLE;
        }
        if (smallestPickedNum >= 4) break;
      }
      for (int k = sp; k <= ep; k++)
        if ((f.state[k] & F_NORMAL) > 0 && (f.state[k] & F_DISABLE) == 0 && f.curvature[k] < 0.3) surfPointsLessFlatScan.push_back(P[k]);
    }
    std::vector<FPoint> ds = VoxelGrid(surfPointsLessFlatScan, 0.2f);
    f.surfLessFlat.insert(f.surfLessFlat.end(), ds.begin(), ds.end());
  }
  for (FPoint& p : f.surfLessFlat) p.intensity = F_NORMAL;
  std::vector<FPoint> ground_cloud;
  for (int i = 0; i < cloudSize; i++) if ((f.state[i] & F_GROUND) > 0) ground_cloud.push_back(P[i]);
  std::vector<FPoint> ground_ds = VoxelGrid(ground_cloud, 0.2f);
  for (FPoint& p : ground_ds) p.intensity = F_GROUND;
  f.surfLessFlat.insert(f.surfLessFlat.end(), ground_ds.begin(), ground_ds.end());
}

}  // namespace oracle
