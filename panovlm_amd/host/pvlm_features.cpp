// LiDAR feature extraction in front of the hot path, planar branch — host side.
//   Velodyne::ReOrderVLP       sensors/Velodyne.cpp:371-526
//   Velodyne::Segmentation     sensors/Velodyne.cpp:1438-1586
//   Velodyne::ExtractFeatures  sensors/Velodyne.cpp:531-760 (ADAPTIVE) -> ExtractEdgeFeatures2 :883-1000, ExtractPlaneFeatures2 :1098-1189
// Scan-by-scan host forms of every stage (one scan is 28.8 k points; the column state machine of the re-ordering, the
// component labelling of the range image, greedy picks with non-maximum suppression along a ring).  The per-point / per-ring
// stages also exist as batched HIP kernels (csrc/pvlm_ring.hip): ExtractFeaturesBatch at the end of this file runs those and keeps only the sort-dependent picks here; the clouds are what
// Velodyne::DeviceScan uploads for the GPU association.  The arithmetic is the reference's float arithmetic
// (`using namespace std` there: the float overloads of sqrt / atan / atan2 / acos / sin / cos), compiled with
// -ffp-contract=off.  Two third-party behaviours are restated from memory [recalled — PCL is not in the image]:
// pcl::VoxelGrid<PointXYZI>::applyFilter of PCL 1.10 (voxel index floor(x * inverse_leaf) - min_b, std::sort of
// (voxel, point) pairs by voxel only, centroid summed in sorted order and divided by the count, output in ascending
// voxel order) and Eigen 3.4's unvectorised 3-vector reductions (dot, squaredNorm: c0 + (c1 + c2)).  std::sort is
// libstdc++'s, as in a reference build: with equal keys its order is deterministic for a given library only.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <thread>

#include "pvlm_host.hpp"
#include "../csrc/pvlm_workers.h"

namespace pvlm {
namespace {

inline float Dist2(const PointXYZI& a, const PointXYZI& b) {   // base/Geometry.hpp:38-40
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;
}

// sensors/Velodyne.cpp:170-211
int RingOfElevation(float deg, int rings) {
  if (!(deg == deg)) return -1;   // a return at the sensor origin has no elevation (upstream converts the NaN to int: undefined, negative in practice)
  int id = -1;
  if (rings == 16) {
    id = int((deg + 15) / 2 + 0.5);
    if (id > 15 || id < 0) id = -1;
  } else if (rings == 32) {
    id = int((deg + 92.0 / 3.0) * 3.0 / 4.0);
    if (id > 31 || id < 0) id = -1;
  } else if (rings == 64) {
    id = deg >= -8.83 ? int((2 - deg) * 3.0 + 0.5) : 32 + int((-8.83 - deg) * 2.0 + 0.5);
    if (deg > 2 || deg < -24.33 || id > 50 || id < 0) id = -1;
  }
  return id;
}

inline double Azimuth(const PointXYZI& p) {   // atan2(float, float) in [0, 2 pi)
  double a = std::atan2(p.x, p.z);
  if (a < 0) a += 2 * M_PI;
  return a;
}

// position of a ring in the VLP-16 firing sequence (the std::map of :407-414; a missing key reads as 0)
inline int FiringSlot(int ring, int rings) { return (rings != 16 || ring < 0) ? 0 : (ring <= 7 ? 2 * ring : 2 * ring - 15); }

// pcl::VoxelGrid<PointXYZI>, leaf x leaf x leaf, all fields averaged ([recalled], see the header of this file)
struct VoxelKey { unsigned cell, point; };
inline bool operator<(const VoxelKey& a, const VoxelKey& b) { return a.cell < b.cell; }

void VoxelGridAppend(const PointCloud& in, float leaf, float tag, PointCloud& out) {
  if (in.empty()) return;
  const float inv = 1.f / leaf;
  float lo[3] = {in[0].x, in[0].y, in[0].z}, hi[3] = {in[0].x, in[0].y, in[0].z};
  for (const PointXYZI& p : in) {
    const float v[3] = {p.x, p.y, p.z};
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); }
  }
  int64_t cells = 1;
  for (int k = 0; k < 3; ++k) cells *= (int64_t)((hi[k] - lo[k]) * inv) + 1;
  if (cells > (int64_t)INT32_MAX) {                                  // PCL: leaf too small for the extent -> input passed through
    for (PointXYZI p : in) { p.intensity = tag; out.push_back(p); }
    return;
  }
  int base[3], dim[3];
  for (int k = 0; k < 3; ++k) { base[k] = (int)std::floor(lo[k] * inv); dim[k] = (int)std::floor(hi[k] * inv) - base[k] + 1; }
  const int stride_y = dim[0], stride_z = dim[0] * dim[1];
  std::vector<VoxelKey> keys;
  keys.reserve(in.size());
  for (unsigned i = 0; i < in.size(); ++i) {
    const PointXYZI& p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const int cx = (int)(std::floor(p.x * inv) - (float)base[0]);
    const int cy = (int)(std::floor(p.y * inv) - (float)base[1]);
    const int cz = (int)(std::floor(p.z * inv) - (float)base[2]);
    keys.push_back(VoxelKey{(unsigned)(cx + cy * stride_y + cz * stride_z), i});
  }
  std::sort(keys.begin(), keys.end(), std::less<VoxelKey>());
  size_t first = 0;
  while (first < keys.size()) {
    size_t last = first;
    float sum[3] = {0, 0, 0};
    while (last < keys.size() && keys[last].cell == keys[first].cell) {
      const PointXYZI& p = in[keys[last].point];
      sum[0] += p.x; sum[1] += p.y; sum[2] += p.z;
      ++last;
    }
    const float n = (float)(last - first);
    out.push_back(PointXYZI{sum[0] / n, sum[1] / n, sum[2] / n, tag});   // the caller overwrites intensity with the class tag (:1177-1178, :1186-1187)
    first = last;
  }
}

struct StageTimer {   // wall clock of a host stage into pvlm::StageSeconds()
  const char* name; std::chrono::steady_clock::time_point t0;
  explicit StageTimer(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~StageTimer() { AddStageSeconds(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
};

// PVLM_FEATURE_PROFILE=1: thread-seconds of the host stages of ExtractFeaturesBatch on stderr (tools/feature_batch_bench.py)
struct FeatureProfile {
  std::atomic<long long> ns[8];
  bool on;
  FeatureProfile() : on(std::getenv("PVLM_FEATURE_PROFILE") != nullptr) { for (auto& v : ns) v = 0; }
  void Report(const char* what, double wall_ms, double ring_ms, double producer_ms) {
    static const char* name[8] = {"layout", "sector sort", "edge picks", "EdgeToLine", "plane picks", "voxel grid", "", ""};
    fprintf(stderr, "feature_profile %s wall_ms %.2f ring_batch_events_ms %.2f producer_wall_ms %.2f thread_ms:", what, wall_ms, ring_ms, producer_ms);
    for (int k = 0; k < 6; ++k) fprintf(stderr, " [%s] %.2f", name[k], 1e-6 * (double)ns[k].exchange(0));
    fprintf(stderr, "\n");
  }
};
FeatureProfile& Profile() { static FeatureProfile p; return p; }
struct ProfileSpan {
  int slot; std::chrono::steady_clock::time_point t0;
  explicit ProfileSpan(int k) : slot(Profile().on ? k : -1) { if (slot >= 0) t0 = std::chrono::steady_clock::now(); }
  void Next(int k) { Stop(); if (Profile().on) { slot = k; t0 = std::chrono::steady_clock::now(); } }
  void Stop() { if (slot >= 0) Profile().ns[slot] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); slot = -1; }
  ~ProfileSpan() { Stop(); }
};

// union-find over range-image cells
struct DisjointSets {
  std::vector<int> parent;
  explicit DisjointSets(size_t n) : parent(n) { std::iota(parent.begin(), parent.end(), 0); }
  int Find(int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; }
  void Union(int a, int b) { a = Find(a); b = Find(b); if (a != b) parent[std::max(a, b)] = std::min(a, b); }   // root = smallest cell = the BFS seed
};

}  // namespace

void Velodyne::ReOrderVLP() {
  if (!valid) return;
  if (!cloud_scan.empty()) return;
  RingLayout& L = layout_;
  L = RingLayout();
  if (N_SCANS <= 0 || horizon_scans <= 0) { fprintf(stderr, "ReOrderVLP: %d rings x %d columns is not a range image\n", N_SCANS, horizon_scans); return; }
  L.image_to_point_idx.assign((size_t)N_SCANS * horizon_scans, -1);
  L.scanStartInd.assign(N_SCANS, 0);
  L.scanEndInd.assign(N_SCANS, 0);
  if (N_SCANS != 16 && N_SCANS != 32 && N_SCANS != 64) { fprintf(stderr, "only support velodyne with 16, 32 or 64 scan line!\n"); return; }
  L.range_image.assign((size_t)N_SCANS * horizon_scans, 0.f);
  const int n = (int)cloud.size();
  if (n == 0) return;
  const double column_width = 2.0 * M_PI / horizon_scans;
  const double first_azimuth = Azimuth(cloud[0]);
  // pass 1 (sequential: the wrap detection and the column correction carry state from point to point): ring + column
  std::vector<int> ring_of(n, -1), col_of(n, -1);
  std::vector<int> ring_count(N_SCANS, 0);
  bool wrapped = false;          // the sweep has passed the +z axis once (:416-417, :447-461)
  double prev_azimuth = -1;
  int shift = 0, prev_col = 0, prev_ring = -1;
  for (int i = 0; i < n; ++i) {
    const PointXYZI& p = cloud[i];
    const float elevation = std::atan(-p.y / std::sqrt(p.x * p.x + p.z * p.z)) * 180 / M_PI;
    const int ring = RingOfElevation(elevation, N_SCANS);
    if (ring < 0) continue;
    double azimuth = Azimuth(p);
    if (azimuth < prev_azimuth && !wrapped) {
      // believed only when the next N_SCANS returns all lie behind the previous azimuth as well
      int behind = 0;
      for (int j = i + 1; j < i + N_SCANS + 1 && j < n && behind < N_SCANS; ++j) behind += Azimuth(cloud[j]) < prev_azimuth ? 1 : 0;
      wrapped = behind >= N_SCANS;
    }
    azimuth += 2 * M_PI * (wrapped ? 1 : 0);
    int col = std::round((azimuth - first_azimuth) / column_width);
    if (FiringSlot(ring, N_SCANS) < FiringSlot(prev_ring, N_SCANS)) {   // a new firing sequence started: same column as the last one?
      shift = prev_col == col ? 1 : 0;
      prev_col = col + shift;
    }
    prev_ring = ring;
    col += shift;
    while (col >= horizon_scans) col -= horizon_scans;
    if (col < 0) continue;
    ring_of[i] = ring; col_of[i] = col;
    ring_count[ring]++;
    L.range_image[(size_t)ring * horizon_scans + col] = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
    prev_azimuth = azimuth;
  }
  // pass 2: stable counting sort by ring
  std::vector<int> ring_begin(N_SCANS + 1, 0);
  for (int r = 0; r < N_SCANS; ++r) ring_begin[r + 1] = ring_begin[r] + ring_count[r];
  const int kept = ring_begin[N_SCANS];
  cloud_scan.assign(kept, PointXYZI{0, 0, 0, 0});
  L.point_idx_to_image.assign(kept, std::pair<int, int>(0, 0));
  std::vector<int> cursor(ring_begin.begin(), ring_begin.end() - 1);
  for (int i = 0; i < n; ++i) {
    const int ring = ring_of[i];
    if (ring < 0) continue;
    const int dst = cursor[ring]++;
    cloud_scan[dst] = PointXYZI{cloud[i].x, cloud[i].y, cloud[i].z, (float)ring};
    L.point_idx_to_image[dst] = std::pair<int, int>(ring, col_of[i]);
    L.image_to_point_idx[(size_t)ring * horizon_scans + col_of[i]] = dst;
  }
  for (int r = 0; r < N_SCANS; ++r) { L.scanStartInd[r] = ring_begin[r] + 5; L.scanEndInd[r] = ring_begin[r + 1] - 6; }
}

void Velodyne::Segmentation() {
  RingLayout& L = layout_;
  const int rows = N_SCANS, cols = horizon_scans;
  const size_t cells = (size_t)rows * cols;
  const float alpha_x = 0.2 / 180.0 * M_PI, alpha_y = 2.0 / 180.0 * M_PI, theta = 20.0 / 180.0 * M_PI;
  const float sin_x = std::sin(alpha_x), cos_x = std::cos(alpha_x), sin_y = std::sin(alpha_y), cos_y = std::cos(alpha_y);
  auto joined = [&](float a, float b, bool same_row) {
    const float far = std::max(a, b), near = std::min(a, b);
    const float angle = same_row ? std::atan2(near * sin_x, far - near * cos_x) : std::atan2(near * sin_y, far - near * cos_y);
    return angle > theta;
  };
  // components of the symmetric "joined" relation between 4-neighbours (columns wrap around) = the BFS labels of :1463-1530
  DisjointSets sets(cells);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const int here = r * cols + c;
      const int right = r * cols + (c + 1 == cols ? 0 : c + 1);
      if (right != here && joined(L.range_image[here], L.range_image[right], true)) sets.Union(here, right);
      if (r + 1 < rows && joined(L.range_image[here], L.range_image[here + cols], false)) sets.Union(here, here + cols);
    }
  // per component: size and the rows that hold a cell OTHER than the seed (the BFS flags the row of every pushed cell;
  // the seed — the first cell in raster order — is never pushed)
  std::vector<int> size(cells, 0);
  std::vector<uint64_t> row_mask(cells, 0);
  for (size_t cell = 0; cell < cells; ++cell) {
    const int root = sets.Find((int)cell);
    size[root]++;
    if ((int)cell != root) row_mask[root] |= 1ull << (cell / cols);
  }
  auto keep = [&](int root) {
    if (size[root] >= 30) return true;
    if (size[root] >= 5) return __builtin_popcountll(row_mask[root]) >= 3;
    return false;
  };
  // drop the points of rejected components, ring order preserved
  PointCloud kept;
  kept.reserve(cloud_scan.size());
  std::vector<std::pair<int, int>> kept_rc;
  std::vector<int> cell_to_point(cells, -1);
  std::vector<int> ring_count(rows, 0);
  for (size_t i = 0; i < cloud_scan.size(); ++i) {
    const std::pair<int, int> rc = L.point_idx_to_image[i];
    const int cell = rc.first * cols + rc.second;
    if (!keep(sets.Find(cell))) continue;
    cell_to_point[cell] = (int)kept.size();
    kept.push_back(cloud_scan[i]);
    kept_rc.push_back(rc);
    ring_count[(int)cloud_scan[i].intensity]++;
  }
  cloud_scan.swap(kept);
  L.point_idx_to_image.swap(kept_rc);
  L.image_to_point_idx.swap(cell_to_point);
  int begin = 0;
  for (int r = 0; r < rows; ++r) { L.scanStartInd[r] = begin + 5; begin += ring_count[r]; L.scanEndInd[r] = begin - 6; }
}

void Velodyne::ExtractFeatures(float max_curvature, float intersect_angle_threshold, int method, bool segment, ExtractionTrace* trace, bool edge_to_line) {
  if (!valid) return;
  if (method != ADAPTIVE) throw std::invalid_argument("ExtractFeatures: only the ADAPTIVE method (config/Room.txt:32) is mirrored");
  if (cloud_scan.empty()) { fprintf(stderr, "cloud_scan is empty, call Reorder first\n"); return; }
  if (!cornerLessSharp.empty() || !surfLessFlat.empty()) return;
  if (N_SCANS > 64) throw std::invalid_argument("ExtractFeatures: at most 64 rings");
  const size_t before = cloud_scan.size();
  if (segment) Segmentation();
  if (cloud_scan.size() < before * 0.1) { fprintf(stderr, "LiDAR data %d has something wrong\n", id); valid = false; return; }
  const RingLayout& L = layout_;
  const int n = (int)cloud_scan.size();
  const PointCloud& P = cloud_scan;
  std::vector<float> curvature(n, -1.f), range(n);
  std::vector<int> left(n, -1), right(n, -1);
  for (int i = 0; i < n; ++i) range[i] = L.range_image[(size_t)L.point_idx_to_image[i].first * horizon_scans + L.point_idx_to_image[i].second];

  // ---- curvature over a window grown until both ends are >= 8 cm away (:623-657).  Kept as upstream, including the
  // right-hand walk that is guarded by the LEFT index and the window check that tests the left end twice; where
  // upstream would read past the end of the cloud (undefined behaviour) the walk stops and the point gets no curvature.
  for (int ring = 0; ring < N_SCANS; ++ring) {
    const int lo = L.scanStartInd[ring], hi = L.scanEndInd[ring];
    if ((size_t)(hi - lo) < (size_t)5) continue;
    for (int i = lo; i <= hi; ++i) {
      int a = i - 5, b = i + 5;
      while (a >= lo && Dist2(P[a], P[i]) < 0.0064) --a;
      while (a <= hi && b < n && Dist2(P[b], P[i]) < 0.0064) ++b;
      const int half = std::max(i - a, b - i);
      a = i - half; b = i + half;
      if (a < lo - 5 || a > hi + 5 || b >= n) continue;
      float acc = 0;
      for (int k = a; k <= b; ++k) acc += range[k];
      acc -= (b - a + 1) * range[i];
      acc /= (b - a);
      curvature[i] = std::abs(acc);
      left[i] = a; right[i] = b;
    }
  }
  PickFeatures(max_curvature, intersect_angle_threshold, PickInputs{curvature.data(), range.data(), left.data(), right.data(), nullptr, nullptr, nullptr}, trace, edge_to_line);
}

void Velodyne::PickFeatures(float max_curvature, float intersect_angle_threshold, const PickInputs& in, ExtractionTrace* trace, bool edge_to_line) {
  const RingLayout& L = layout_;
  const int n = (int)cloud_scan.size();
  const PointCloud& P = cloud_scan;
  const float* curvature = in.curvature;
  const float* range = in.range;
  auto left_of = [&](int i) { return in.left ? in.left[i] : (in.half_window[i] < 0 ? -1 : i - in.half_window[i]); };
  auto right_of = [&](int i) { return in.right ? in.right[i] : (in.half_window[i] < 0 ? -1 : i + in.half_window[i]); };
  std::vector<int> state(n, POINT_NORMAL), order(n);
  if (in.sorted) std::copy(in.sorted, in.sorted + n, order.begin());
  else std::iota(order.begin(), order.end(), 0);
  // the six sectors of a ring (:707-723) — the same integer arithmetic everywhere below
  auto sector = [&](int ring, int j, int* sp, int* ep) {
    const int lo = L.scanStartInd[ring], span = L.scanEndInd[ring] - lo;
    *sp = lo + span * j / 6;
    *ep = lo + span * (j + 1) / 6 - 1;
  };
  auto usable = [&](int ring) { return L.scanEndInd[ring] - L.scanStartInd[ring] >= 6; };
  ProfileSpan span(1);
  {
    // sectors that arrive sorted (pvlm_ring_result::sorted: distinct curvatures, so the order is the one any sort returns) are taken as they are;
    // the others — equal curvatures, whose order is libstdc++'s — are in index order and sorted here, as the per-scan path sorts all of them
    const float* c = curvature;
    for (int ring = 0; ring < N_SCANS; ++ring) {
      if (!usable(ring)) continue;
      for (int j = 0; j < 6; ++j) {
        if (in.sorted && !in.sector_host[ring * 6 + j]) continue;
        int sp, ep; sector(ring, j, &sp, &ep);
        std::sort(order.begin() + sp, order.begin() + ep + 1, [c](int x, int y) { return c[x] < c[y]; });
      }
    }
  }
  // non-maximum suppression around a picked point (:969-986, :1140-1155)
  auto suppress = [&](int ind, int ring) {
    const int lo = L.scanStartInd[ring], hi = L.scanEndInd[ring];
    for (int l = 1; ind + l <= hi; ++l) {
      if (l <= 5 ? Dist2(P[ind + l], P[ind + l - 1]) > 0.05 : Dist2(P[ind + l], P[ind]) > 0.0036) break;
      state[ind + l] |= POINT_DISABLE;
    }
    for (int l = 1; ind - l >= lo; ++l) {
      if (l <= 5 ? Dist2(P[ind - l], P[ind - l + 1]) > 0.05 : Dist2(P[ind - l], P[ind]) > 0.0036) break;
      state[ind - l] |= POINT_DISABLE;
    }
  };

  // ---- ExtractEdgeFeatures2 (:883-1000): per sector, from the largest curvature down, at most 3 sharp + 27 less sharp
  cornerSharp.clear(); cornerLessSharp.clear();
  span.Next(2);
  for (int ring = 0; ring < N_SCANS; ++ring) {
    if (!usable(ring)) continue;
    for (int j = 0; j < 6; ++j) {
      int sp, ep; sector(ring, j, &sp, &ep);
      int picked = 0;
      for (int k = ep; k >= sp; --k) {
        const int ind = order[k];
        if (state[ind] != POINT_NORMAL) continue;
        if (curvature[ind] > max_curvature || curvature[ind] < 0.1) continue;
        // incidence angle between the beam and the local surface direction (left - right window ends), degrees
        const PointXYZI &a = P[ind], &l = P[left_of(ind)], &r = P[right_of(ind)];
        const float bx = l.x - r.x, by = l.y - r.y, bz = l.z - r.z;
        const float along = a.x * bx + (a.y * by + a.z * bz);
        const float blen = std::sqrt(bx * bx + (by * by + bz * bz));
        float view_angle = std::acos(std::abs(along) / (range[ind] * blen));
        view_angle *= (180.0 / M_PI);
        if (view_angle < intersect_angle_threshold || view_angle > 180 - intersect_angle_threshold) continue;
        ++picked;
        if (picked > 30) break;
        PointXYZI p = a;
        p.intensity = ind;
        if (picked <= 3) { state[ind] = POINT_SHARP; cornerSharp.push_back(p); }
        else state[ind] = POINT_LESS_SHARP;
        cornerLessSharp.push_back(p);
        suppress(ind, ring);
      }
    }
  }
  // ---- EdgeToLine (:752, :1269-1324): line segments from the edge points; does not touch the per-point state
  span.Next(3);
  if (edge_to_line) EdgeToLine();
  span.Next(4);
  // ---- ExtractPlaneFeatures2 (:1098-1189)
  surfFlat.clear(); surfLessFlat.clear();
  PointCloud ring_less_flat;
  for (int ring = 0; ring < N_SCANS; ++ring) {
    if (!usable(ring)) continue;
    ring_less_flat.clear();
    for (int j = 0; j < 6; ++j) {
      int sp, ep; sector(ring, j, &sp, &ep);
      int picked = 0;
      for (int k = sp; k <= ep && picked < 4; ++k) {
        const int ind = order[k];
        if (state[ind] != POINT_NORMAL && state[ind] != POINT_GROUND) continue;
        if (curvature[ind] > 0.02) continue;
        PointXYZI p = P[ind];
        p.intensity = state[ind];
        surfFlat.push_back(p);
        if (state[ind] == POINT_NORMAL) ring_less_flat.push_back(p);
        state[ind] |= POINT_FLAT;
        ++picked;
        suppress(ind, ring);
      }
      for (int k = sp; k <= ep; ++k)      // note: k is a point index here, not a position in the sorted order (as upstream)
        if ((state[k] & POINT_NORMAL) > 0 && (state[k] & POINT_DISABLE) == 0 && curvature[k] < 0.3) ring_less_flat.push_back(P[k]);
    }
    span.Next(5);
    VoxelGridAppend(ring_less_flat, 0.2f, (float)POINT_NORMAL, surfLessFlat);
    span.Next(4);
  }
  PointCloud ground;
  for (int i = 0; i < n; ++i) if ((state[i] & POINT_GROUND) > 0) ground.push_back(P[i]);
  span.Next(5);
  VoxelGridAppend(ground, 0.2f, (float)POINT_GROUND, surfLessFlat);
  span.Stop();
  InvalidateDevice();
  if (trace) {
    trace->curvature.assign(curvature, curvature + n); trace->state = std::move(state); trace->sort_ind = std::move(order);
    trace->left_neighbor.resize((size_t)n); trace->right_neighbor.resize((size_t)n);
    for (int i = 0; i < n; ++i) { trace->left_neighbor[(size_t)i] = left_of(i); trace->right_neighbor[(size_t)i] = right_of(i); }
  }
}

void Velodyne::AssemblePicks(const pvlm_ring_result& r, ExtractionTrace* trace, bool edge_to_line, const pvlm_line_grow_result* grown) {
  const int n = (int)cloud_scan.size();
  const PointCloud& P = cloud_scan;
  ProfileSpan span(2);
  cornerSharp.clear(); cornerLessSharp.clear();
  for (int ring = 0; ring < N_SCANS; ++ring) {
    const int* list = r.corner + (size_t)ring * 181;
    for (int q = 0; q < list[0]; ++q) {
      const int ind = list[1 + q] & 0x7FFFFFFF;
      PointXYZI p = P[(size_t)ind];
      p.intensity = ind;
      if (list[1 + q] < 0) cornerSharp.push_back(p);
      cornerLessSharp.push_back(p);
    }
  }
  span.Next(3);
  if (edge_to_line) EdgeToLine(grown);
  span.Next(4);
  surfFlat.clear(); surfLessFlat.clear();
  size_t centroids = 0;
  for (int ring = 0; ring < N_SCANS; ++ring) centroids += (size_t)r.voxel_span[2 * ring + 1];
  surfLessFlat.reserve(centroids);
  for (int ring = 0; ring < N_SCANS; ++ring) {
    const int* list = r.flat + (size_t)ring * 25;
    for (int q = 0; q < list[0]; ++q) {
      PointXYZI p = P[(size_t)list[1 + q]];
      p.intensity = (float)(r.state[list[1 + q]] & (POINT_NORMAL | POINT_GROUND));       // the class the point had when it was picked (:1128)
      surfFlat.push_back(p);
    }
    const float* v = r.voxels + 4 * (size_t)r.voxel_span[2 * ring];
    for (int q = 0; q < r.voxel_span[2 * ring + 1]; ++q) surfLessFlat.push_back(PointXYZI{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
  }
  span.Stop();
  InvalidateDevice();
  if (trace) {
    trace->curvature.assign(r.curvature, r.curvature + n); trace->sort_ind.assign(r.sorted, r.sorted + n);
    trace->state.resize((size_t)n); trace->left_neighbor.resize((size_t)n); trace->right_neighbor.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
      trace->state[(size_t)i] = r.state[i];
      const int h = r.half_window[i];
      trace->left_neighbor[(size_t)i] = h < 0 ? -1 : i - h; trace->right_neighbor[(size_t)i] = h < 0 ? -1 : i + h;
    }
  }
}

// ---- the batch form: range-image stages, sector orders, picks and voxel grid on the GPU; EdgeToLine on the host ---------------------------------------------------------
void Velodyne::ExtractFeaturesBatch(const std::vector<Velodyne*>& scans, float max_curvature, float intersect_angle_threshold, int method, bool segment, bool edge_to_line,
                                    int num_threads, std::vector<ExtractionTrace>* traces) {
  if (method != ADAPTIVE) throw std::invalid_argument("ExtractFeatures: only the ADAPTIVE method (config/Room.txt:32) is mirrored");
  if (traces) traces->assign(scans.size(), ExtractionTrace());
  std::vector<size_t> todo;
  for (size_t k = 0; k < scans.size(); ++k) {
    Velodyne* v = scans[k];
    if (!v || !v->valid || !v->cloud_scan.empty() || v->cloud.empty()) continue;                  // ReOrderVLP returns early (:373-377)
    if (!v->cornerLessSharp.empty() || !v->surfLessFlat.empty()) continue;                         // ExtractFeatures would (:542-543)
    todo.push_back(k);
  }
  if (todo.empty()) return;
  // the device copies of the scans that are about to change, given back HERE, on the thread that owns the engine: the workers below call InvalidateDevice() too
  // (PickFeatures, AssemblePicks, EdgeToLine), concurrently with the producer's use of the same context — with nothing left to give back those calls are no-ops
  for (size_t k : todo) scans[k]->InvalidateDevice();
  const auto profile_t0 = std::chrono::steady_clock::now();
  const int rings = scans[todo[0]]->N_SCANS, horizon = scans[todo[0]]->horizon_scans;
  if (rings > 64) throw std::invalid_argument("ExtractFeatures: at most 64 rings");
  for (size_t k : todo) if (scans[k]->N_SCANS != rings || scans[k]->horizon_scans != horizon) throw std::invalid_argument("ExtractFeaturesBatch: the scans of a call share one range-image shape");
  std::vector<pvlm_raw_scan> raw(todo.size());
  for (size_t j = 0; j < todo.size(); ++j) { const PointCloud& c = scans[todo[j]]->cloud; raw[j] = pvlm_raw_scan{&c[0].x, (int)c.size(), (int)(sizeof(PointXYZI) / sizeof(float))}; }
  Engine& e = Engine::Default();
  // The scans go to the device as a sequence of batches: while the host threads pick the features of one batch, the next one is on the GPU (upload,
  // K16-K23, download: PCIe-bound) under the one thread that talks to the context.  A batch lives until the call returns (the picks read its pinned arrays).
  const size_t total = todo.size();
  static const size_t n_parts = [] { const char* v = std::getenv("PVLM_FEATURE_PARTS"); const long k = v ? std::atol(v) : 0; return k > 0 ? (size_t)k : (size_t)2; }();
  const size_t per_batch = total <= 48 ? total : std::max<size_t>(24, (total + n_parts - 1) / n_parts);
  // PVLM_FEATURE_PICKS=host: the picks and the voxel grid on the host threads (PickFeatures) instead of K24 — the A/B switch of tools/feature_batch_bench.py
  static const bool device_picks = [] { const char* v = std::getenv("PVLM_FEATURE_PICKS"); return !(v && std::strcmp(v, "host") == 0); }();
  std::atomic<long> scans_on_host{0}, scans_refused{0}, scans_grown_on_host{0};
  // grow: the line segments of the part's scans grown ahead on the GPU (K27, pvlm_line_grow_batch) from the edge picks the device made; grow_slot[j]: the scan's
  // position in that batch, -1 = none (picks left to the host, scan dropped, no edge points).  PVLM_EDGE_GROW=host: all growth on the host threads (round 5).
  struct Part { pvlm_ring_batch* batch = nullptr; bool on_host = false; pvlm_line_grow* grow = nullptr; std::vector<int> grow_slot; };
  std::vector<Part> parts((total + per_batch - 1) / per_batch);
  struct Release { pvlm_ctx* c; std::vector<Part>& p; ~Release() { for (Part& q : p) { pvlm_ring_batch_destroy(c, q.batch); pvlm_line_grow_destroy(c, q.grow); } } } release{e.ctx(), parts};
  static const bool device_grow = [] { const char* v = std::getenv("PVLM_EDGE_GROW"); return !(v && (std::strcmp(v, "host") == 0 || std::strcmp(v, "tasks") == 0)); }();
  double grow_ms = 0, grow_kernel_ms = 0; long long grow_tasks = 0;
  std::mutex gate; std::condition_variable published_cv;
  size_t published = 0;                        // scans (in `todo` order) whose batch is back from the device
  size_t grown_parts = 0;                      // parts whose line growth is back (or that have none)
  std::vector<size_t> deferred;                // scans assembled before their part's growth was back: their EdgeToLine is still to run
  size_t assembled = 0;                        // scans whose first pass is over
  bool production_failed = false;
  double ring_ms = 0, producer_ms = 0;
  auto produce = [&]() {
    const auto produce_t0 = std::chrono::steady_clock::now();
    StageTimer stage_timer_("  (inside feature extraction) range-image stages of all scans on the GPU (pvlm_ring_extract_batch), overlapped with the picks");
    try {
      for (size_t k = 0; k < parts.size(); ++k) {
        const size_t first = k * per_batch, count = std::min(per_batch, total - first);
        // traces (parity tests) read curvature / windows / order of every scan: bit 1 of `segment` keeps the per-point arrays in the download
        const pvlm_status rc = device_picks ? pvlm_ring_extract_batch_picks(e.ctx(), (int)count, raw.data() + first, rings, horizon, (segment ? 1 : 0) | (traces ? 2 : 0), max_curvature,
                                                                            intersect_angle_threshold, &parts[k].batch)
                                            : pvlm_ring_extract_batch(e.ctx(), (int)count, raw.data() + first, rings, horizon, segment ? 1 : 0, &parts[k].batch);
        if (rc == PVLM_ERR_REFUSED) {
          // The batch form is stricter than the reference: it refuses a batch with a non-finite coordinate (upstream such a point gets ring -1 and is
          // skipped, sensors/Velodyne.cpp:439-445) and gives up beyond its bounds on undecided segmentation edges.  Neither is a reason to fail
          // EstimatePose: these scans go through the per-scan host path, which skips such points exactly as upstream does.  Every OTHER error (a bad argument,
          // memory, a failed launch) is an error and propagates: a silent fall-back would turn it into a large unexplained slowdown.
          fprintf(stderr, "ExtractFeaturesBatch: %s — the per-scan host extraction takes these %zu scans\n", pvlm_last_error(e.ctx()), count);
          scans_refused += (long)count;
          parts[k].on_host = true;
        } else {
          e.Check(rc, "pvlm_ring_extract_batch");
          if (traces) {                             // the two images: device-resident, fetched for the parity tests only
            for (size_t j = first; j < first + count; ++j) {
              RingLayout& L = scans[todo[j]]->layout_;
              L.range_image.assign((size_t)rings * horizon, 0.f); L.image_to_point_idx.assign((size_t)rings * horizon, -1);
              e.Check(pvlm_ring_batch_fetch(e.ctx(), parts[k].batch, (int)(j - first), 1, nullptr, nullptr, L.range_image.data(), L.image_to_point_idx.data()), "pvlm_ring_batch_fetch");
            }
          }
          double ms8[8] = {0};
          if (pvlm_ring_batch_timing(parts[k].batch, ms8) == PVLM_OK) for (double v : ms8) ring_ms += v;
        }
        // the part's scans are the workers' from here on; the growth of the PREVIOUS part (it ran beside this part's range-image stages) comes back now, this
        // part's growth starts and runs beside the next part's stages
        { std::lock_guard<std::mutex> g(gate); published = first + count; }
        published_cv.notify_all();
        auto finish_growth = [&](size_t q) {
          Part& pq = parts[q];
          if (pq.grow) {
            const auto t0 = std::chrono::steady_clock::now();
            const pvlm_status gs = pvlm_line_grow_finish(e.ctx(), pq.grow);
            if (gs == PVLM_ERR_REFUSED) { fprintf(stderr, "ExtractFeaturesBatch: %s — the host threads grow the lines of these scans\n", pvlm_last_error(e.ctx())); pvlm_line_grow_destroy(e.ctx(), pq.grow); pq.grow = nullptr; }
            else e.Check(gs, "pvlm_line_grow_finish");
            pvlm_line_grow_result g0;
            if (pq.grow && pvlm_line_grow_scan(pq.grow, 0, &g0) == PVLM_OK) { grow_kernel_ms += g0.kernel_ms; grow_tasks += g0.tasks_run; }
            grow_ms += 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          }
          { std::lock_guard<std::mutex> g(gate); grown_parts = q + 1; }
          published_cv.notify_all();
        };
        if (k > 0) finish_growth(k - 1);
        if (!parts[k].on_host && edge_to_line && device_picks && device_grow) {
          // the edge points of every scan whose picks the device made, as AssemblePicks will lay them out (ring by ring, pick order), grown in one batch
          const auto grow_t0 = std::chrono::steady_clock::now();
          Part& part = parts[k];
          part.grow_slot.assign(count, -1);
          std::vector<std::vector<float>> edge(count);
          std::vector<pvlm_edge_cloud> clouds;
          for (size_t j = first; j < first + count; ++j) {
            pvlm_ring_result r;
            if (pvlm_ring_batch_scan(part.batch, (int)(j - first), &r) != PVLM_OK || !r.picks) continue;
            bool decided = r.n_kept > 0 && !(r.n_kept < r.n_reordered * 0.1);
            for (int q = 0; decided && q < rings; ++q) decided = r.ring_host[q] == 0;
            if (!decided) continue;
            const PointCloud& raw_cloud = scans[todo[j]]->cloud;
            std::vector<float>& xyz = edge[j - first];
            for (int ring = 0; ring < rings; ++ring) {
              const int* list = r.corner + (size_t)ring * 181;
              for (int q = 0; q < list[0]; ++q) {
                const PointXYZI& p = raw_cloud[(size_t)r.source[list[1 + q] & 0x7FFFFFFF]];
                xyz.push_back(p.x); xyz.push_back(p.y); xyz.push_back(p.z);
              }
            }
            if (xyz.empty() || xyz.size() / 3 > 65535) continue;
            part.grow_slot[j - first] = (int)clouds.size();
            clouds.push_back(pvlm_edge_cloud{xyz.data(), (int)(xyz.size() / 3), 3});
          }
          if (!clouds.empty()) e.Check(pvlm_line_grow_begin(e.ctx(), (int)clouds.size(), clouds.data(), &part.grow), "pvlm_line_grow_begin");      // copies the clouds
          grow_ms += 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - grow_t0).count();
        }
        if (k + 1 == parts.size()) finish_growth(k);
      }
      producer_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - produce_t0).count();
    } catch (...) {
      { std::lock_guard<std::mutex> g(gate); production_failed = true; }
      published_cv.notify_all();
      throw;
    }
  };
  // the lines of a scan whose clouds are assembled, from its part's growth (back by now)
  auto lines = [&](size_t j) {
    Velodyne& v = *scans[todo[j]];
    const Part& part = parts[j / per_batch];
    pvlm_line_grow_result grown; const pvlm_line_grow_result* pre = nullptr;
    const int slot = part.grow && !part.grow_slot.empty() ? part.grow_slot[j % per_batch] : -1;
    if (slot >= 0 && pvlm_line_grow_scan(part.grow, slot, &grown) == PVLM_OK) { pre = &grown; if (grown.status != 0) ++scans_grown_on_host; }
    ProfileSpan span(3);
    v.EdgeToLine(pre);
  };
  // first pass over a scan; returns true when its EdgeToLine waits for the part's growth (`lines` runs it)
  auto pick = [&](size_t j) -> bool {
    Velodyne& v = *scans[todo[j]];
    const Part& part = parts[j / per_batch];
    if (part.on_host) {
      v.ReOrderVLP();
      v.ExtractFeatures(max_curvature, intersect_angle_threshold, method, segment, traces ? &(*traces)[todo[j]] : nullptr, edge_to_line);
      return false;
    }
    ProfileSpan span(0);
    pvlm_ring_result r;
    if (pvlm_ring_batch_scan(part.batch, (int)(j % per_batch), &r) != PVLM_OK) throw std::runtime_error("pvlm_ring_batch_scan failed");
    RingLayout& L = v.layout_;
    const std::vector<float> keep_image = std::move(L.range_image); const std::vector<int> keep_index = std::move(L.image_to_point_idx);
    L = RingLayout();
    L.range_image = keep_image; L.image_to_point_idx = keep_index;
    L.scanStartInd.assign(rings, 0); L.scanEndInd.assign(rings, 0);
    // < 10 % of the re-ordered points survive the segmentation: the scan is dropped (:551-556); the cloud is left as Segmentation left it
    const int n = r.n_kept;
    v.cloud_scan.resize((size_t)n);
    L.point_idx_to_image.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
      const PointXYZI& p = v.cloud[(size_t)r.source[i]];
      const int ring = r.ring_col[i] >> 16;
      v.cloud_scan[(size_t)i] = PointXYZI{p.x, p.y, p.z, (float)ring};
      L.point_idx_to_image[(size_t)i] = std::pair<int, int>(ring, r.ring_col[i] & 0xFFFF);
    }
    int begin = 0;
    for (int q = 0; q < rings; ++q) { L.scanStartInd[q] = begin + 5; begin += r.ring_count[q]; L.scanEndInd[q] = begin - 6; }
    if (n < r.n_reordered * 0.1) { fprintf(stderr, "LiDAR data %d has something wrong\n", v.id); v.valid = false; return false; }
    if (n == 0) return false;
    span.Stop();
    bool decided = r.picks != 0;
    for (int q = 0; decided && q < rings; ++q) decided = r.ring_host[q] == 0;
    if (decided) {
      // the clouds now; the lines when the part's growth is back (K27 runs beside the next part's range-image stages) — EdgeToLine touches neither the per-point
      // state nor the planar clouds, so the order of the two does not matter
      const bool defer = edge_to_line && device_grow;
      v.AssemblePicks(r, traces ? &(*traces)[todo[j]] : nullptr, edge_to_line && !defer, nullptr);
      return defer;
    }
    if (r.picks) ++scans_on_host;                  // a ring the device left undecided (incidence angle at the threshold, bounds): the whole scan is picked here
    v.PickFeatures(max_curvature, intersect_angle_threshold, PickInputs{r.curvature, r.range, nullptr, nullptr, r.half_window, r.sorted, r.sector_host},
                   traces ? &(*traces)[todo[j]] : nullptr, edge_to_line);
    return false;
  };
  StageTimer stage_timer_picks_("  (inside feature extraction) picks, EdgeToLine, voxel grid (host, scan-parallel)");
  std::atomic<size_t> next{0};
  std::atomic<bool> producer_taken{false};
  std::mutex failure_lock;
  std::exception_ptr failure;
  auto work = [&]() {
    if (!producer_taken.exchange(true)) produce();              // the first thread to arrive feeds the device, then picks like the others
    for (size_t j = next++; j < total; j = next++) {
      {
        std::unique_lock<std::mutex> g(gate);
        published_cv.wait(g, [&] { return published > j || production_failed; });
        if (published <= j) break;
      }
      bool wants_lines = false;
      try { wants_lines = pick(j); }
      catch (...) { std::lock_guard<std::mutex> g(failure_lock); if (!failure) failure = std::current_exception(); }
      bool now = false;
      {
        std::lock_guard<std::mutex> g(gate);
        ++assembled;
        if (wants_lines) { now = grown_parts > j / per_batch; if (!now) deferred.push_back(j); }
      }
      published_cv.notify_all();
      if (now) {
        try { lines(j); }
        catch (...) { std::lock_guard<std::mutex> g(failure_lock); if (!failure) failure = std::current_exception(); }
      }
    }
    // the scans whose growth was not back when their clouds were done
    for (;;) {
      size_t j = 0; bool have = false;
      {
        std::unique_lock<std::mutex> g(gate);
        published_cv.wait(g, [&] {
          if (production_failed) return true;
          for (size_t q : deferred) if (grown_parts > q / per_batch) return true;
          return deferred.empty() && assembled >= total;
        });
        if (production_failed) return;
        for (size_t q = 0; q < deferred.size(); ++q) if (grown_parts > deferred[q] / per_batch) { j = deferred[q]; deferred[q] = deferred.back(); deferred.pop_back(); have = true; break; }
        if (!have) return;                                       // nothing deferred and every scan assembled
      }
      try { lines(j); }
      catch (...) { std::lock_guard<std::mutex> g(failure_lock); if (!failure) failure = std::current_exception(); }
    }
  };
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::max(num_threads, 1), total + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
  pvlm_run_workers(n_threads, work);
  if (failure) std::rethrow_exception(failure);
  if (scans_refused.load()) AddStageSeconds("  (inside feature extraction) scans of batches the device refused, extracted scan by scan on the host [count in calls]", 0.0);
  if (Profile().on) fprintf(stderr, "feature_profile scans picked on the host (a ring undecided on the device): %ld of %zu; scans of refused batches: %ld\n", scans_on_host.load(), total, scans_refused.load());
  if (Profile().on) fprintf(stderr, "feature_profile line growth on the GPU (K27): %.2f ms on the producer (kernels %.2f ms, %lld tasks); scans the device handed back to the host growth: %ld\n", grow_ms, grow_kernel_ms, grow_tasks, scans_grown_on_host.load());
  if (Profile().on) Profile().Report("ExtractFeaturesBatch", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - profile_t0).count(), ring_ms, producer_ms);
}

}  // namespace pvlm
