// pvlm_host_lidar_associate.cpp — part of the C++ host mirror (pvlm_host.hpp): lidar_mapping/LidarFeatureAssociate.cpp (FindNeighbors, AssociatePoint2Plane, AssociateLine2Line, FindAssociations, the k-NN variants) and the line tracks: lidar_mapping/LidarLineMatch.cpp:36-91, util/Tracks.cpp:58-196.
// Host logic only; every residual, Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host_internal.hpp"

namespace pvlm {

// ================================================================================================
// FindNeighbors — lidar_mapping/LidarFeatureAssociate.cpp:19-111 (scan centres as float32
// PointXYZI, exact k-NN / radius search as pcl::KdTreeFLANN returns them: ascending, L2_Simple)
// ================================================================================================
std::vector<std::vector<int>> FindNeighborsConsecutive(const std::vector<Velodyne>& lidars, const int neighbor_size) {
  std::vector<std::vector<int>> all;
  for (int i = 0; i < int(lidars.size()) - neighbor_size; i++) {
    std::vector<int> nb;
    for (int j = i + 1; j < (int)lidars.size() && j <= i + neighbor_size; j++) nb.push_back(j);
    all.push_back(nb);
  }
  return all;
}

std::vector<std::vector<int>> FindNeighbors(const std::vector<Velodyne>& lidars, const int neighbor_size) {
  StageTimer stage_timer_("  (inside) FindNeighbors (host)");
  std::vector<std::vector<int>> neighbors_all;
  std::vector<std::array<float, 3>> center;
  std::vector<int> owner;
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    const Vector3d& t = lidars[i].GetTranslation();
    center.push_back({float(t[0]), float(t[1]), float(t[2])});
    owner.push_back((int)i);
  }
  const int nc = (int)owner.size();
  // The distances and their order — the two searches of every scan — on the GPU for all scans at once (K28, pvlm_centre_orders: a row of sorted (distance, position)
  // keys per centre, the same keys the loop below builds and sorts) when the queries are the centres themselves: every scan with a valid pose is a valid scan, as
  // EstimatePose leaves them.  Fewer than PVLM_NEIGHBORS_GPU_MIN scans, more than 4096 or a pose that is valid on an invalid scan: the host's own distance loop and sort.
  const uint16_t* rows = nullptr;
  std::vector<uint16_t> row_store;
  std::vector<int> position((size_t)lidars.size(), -1);
  {
    // from PVLM_NEIGHBORS_GPU_MIN scans on (default 2048): the host's sort of 1593 keys per scan on 16 threads takes as long as the launch, its copies and the
    // first-use costs they meet in a fresh process (Floor: 5.9 against 6.1 ms per call); the host's share grows with n^2 log n (37 ms at 4000 scans), the launch's does not
    static const int gpu_min = [] { const char* v = std::getenv("PVLM_NEIGHBORS_GPU_MIN"); return v ? std::atoi(v) : 2048; }();
    bool same_sets = nc >= gpu_min && nc > 0;
    for (size_t i = 0; i < lidars.size() && same_sets; ++i) if (lidars[i].IsPoseValid() && !lidars[i].valid) same_sets = false;
    for (int j = 0; j < nc; ++j) position[(size_t)owner[(size_t)j]] = j;
    if (same_sets) {
      Engine& e = Engine::Default();
      row_store.resize((size_t)nc * (size_t)nc);
      const pvlm_status st = pvlm_centre_orders(e.ctx(), nc, &center[0][0], row_store.data());
      if (st != PVLM_ERR_CAPACITY) { e.Check(st, "pvlm_centre_orders"); rows = row_store.data(); }
    }
  }
  // every scan's list is independent of the others: scan-parallel (at Floor size — 1593 scans, all inside the 20 m radius of the
  // synthetic room — the serial loop was 0.1 s per call, four calls per EstimatePose)
  neighbors_all.assign(lidars.size(), std::vector<int>());
  auto one = [&](size_t i) {
    std::vector<int> neighbors;
    if (lidars[i].IsPoseValid()) {
      const Vector3d& t = lidars[i].GetTranslation();
      const float q[3] = {float(t[0]), float(t[1]), float(t[2])};
      // (squared distance, position) pairs sorted as one 64-bit word each: the distances are sums of squares (never negative, a NaN centre is no valid
      // pose), so their bit patterns order like the floats, and equal distances fall back to the position as the pair comparison did
      std::vector<uint64_t> own_keys;
      const uint16_t* order = rows ? rows + (size_t)position[i] * (size_t)nc : nullptr;
      if (!order) {
        own_keys.resize((size_t)nc);
        for (int j = 0; j < nc; ++j) {
          const float dx = q[0] - center[j][0], dy = q[1] - center[j][1], dz = q[2] - center[j][2];
          float s = 0.0f; s += dx * dx; s += dy * dy; s += dz * dz;
          uint32_t bits; std::memcpy(&bits, &s, 4);
          own_keys[(size_t)j] = ((uint64_t)bits << 32) | (uint32_t)j;
        }
        // one sorted list serves both searches below: nearestKSearch (its first neighbor_size entries) and radiusSearch (its
        // prefix within 20 m, also ascending) — ties in pcl's order = position
        std::sort(own_keys.begin(), own_keys.end());
      }
      std::vector<std::pair<float, int>> d((size_t)nc);
      if (order) {
        // the device's row holds the positions in key order; the distance of an entry by the same float chain as above
        for (int j = 0; j < nc; ++j) {
          const int c = (int)order[(size_t)j];
          const float dx = q[0] - center[(size_t)c][0], dy = q[1] - center[(size_t)c][1], dz = q[2] - center[(size_t)c][2];
          float s = 0.0f; s += dx * dx; s += dy * dy; s += dz * dz;
          d[(size_t)j] = {s, c};
        }
      } else
        for (int j = 0; j < nc; ++j) { const uint32_t bits = (uint32_t)(own_keys[(size_t)j] >> 32); float s; std::memcpy(&s, &bits, 4); d[(size_t)j] = {s, (int)(uint32_t)own_keys[(size_t)j]}; }
      for (int j = 0; j < std::min(neighbor_size, nc); ++j) neighbors.push_back(d[j].second);
      if (!neighbors.empty()) neighbors.erase(neighbors.begin());  // the first one is the scan itself
      for (int& n : neighbors) n = owner[n];
      // upstream's std::set of the chosen scans, as a sorted vector: "at least two members within loop_length of the candidate" looks at the members in
      // [candidate - loop_length, candidate + loop_length], which are consecutive (the set was walked from its start for each of the ~1 600 candidates)
      std::vector<int> nset(neighbors.begin(), neighbors.end());
      std::sort(nset.begin(), nset.end());
      nset.erase(std::unique(nset.begin(), nset.end()), nset.end());
      auto has = [&nset](int v) { return std::binary_search(nset.begin(), nset.end(), v); };
      int ni = (int)i - 1;
      while (ni >= 0 && !lidars[ni].IsPoseValid()) ni--;
      if (ni >= 0 && !has(ni)) neighbors.push_back(ni);
      ni = (int)i + 1;
      while (ni < (int)lidars.size() && !lidars[ni].IsPoseValid()) ni++;
      if (ni < (int)lidars.size() && !has(ni)) neighbors.push_back(ni);
      const float r2 = float(20.0 * 20.0);  // radiusSearch(20 m): FLANN keeps dist < r^2
      const int loop_length = 200;
      for (int j = 0; j < nc && d[j].first < r2; ++j) {
        const int n_idx = owner[d[j].second];
        int same_loop = 0;
        for (auto it = std::lower_bound(nset.begin(), nset.end(), n_idx - loop_length); it != nset.end() && *it <= n_idx + loop_length && same_loop < 2; ++it) same_loop++;
        if (same_loop < 2 && !has(n_idx)) { neighbors.push_back(n_idx); nset.insert(std::upper_bound(nset.begin(), nset.end(), n_idx), n_idx); }
      }
    } else {
      for (int j = -neighbor_size / 2; j <= neighbor_size / 2; j++) neighbors.push_back((int)i - j);
    }
    neighbors_all[i].swap(neighbors);
  };
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({pvlm_thread_cap(), lidars.size() / 64 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
  std::atomic<size_t> next{0};
  auto work = [&]() { for (size_t i = next++; i < lidars.size(); i = next++) one(i); };
  pvlm_run_workers(n_threads, work);
  return neighbors_all;
}

// ================================================================================================
// association wrappers
// ================================================================================================
std::vector<Point2Plane> AssociatePoint2Plane(const Velodyne& ref, const Velodyne& nei, double plane_tolerance, const float dist_threshold, bool) {
  std::vector<Point2Plane> out;
  if (!ref.IsInWorldCoordinate() || !nei.IsInWorldCoordinate()) { fprintf(stderr, "lidar %d / %d is not in world coordinate\n", ref.id, nei.id); return out; }
  Engine& e = Engine::Default();
  pvlm_scan* r = ref.DeviceScan(); pvlm_scan* n = nei.DeviceScan();
  pvlm_resset* rs = nullptr;
  // the one-pair form hands the records to the caller, as the reference's function does: the reference's own QR for every query
  // (PVLM_FLAG_ASSOC_EXACT_FIT — records bit-identical to the reference's); the adders, whose records stay on the device, use the certified fast fit
  e.Check(pvlm_assoc_point2plane(e.ctx(), 1, &r, &n, plane_tolerance, dist_threshold, PVLM_POINT2PLANE_METER, PVLM_FLAG_ASSOC_EXACT_FIT, 1.0, &rs), "pvlm_assoc_point2plane");
  int64_t m = 0;
  pvlm_resset_info(rs, &m, nullptr, nullptr, nullptr);
  std::vector<double> rows((size_t)std::max<int64_t>(m, 1) * 7);
  e.Check(pvlm_resset_download(e.ctx(), rs, nullptr, nullptr, nullptr, rows.data()), "pvlm_resset_download");
  pvlm_resset_destroy(e.ctx(), rs);
  out.resize((size_t)m);
  for (int64_t i = 0; i < m; ++i) {
    out[i].point = {rows[7 * i], rows[7 * i + 1], rows[7 * i + 2]};
    out[i].plane_coeff = {rows[7 * i + 3], rows[7 * i + 4], rows[7 * i + 5], rows[7 * i + 6]};
  }
  return out;
}

std::vector<Vector6d> TransformLines(const std::vector<Vector6d>& lc, const Matrix4d& T) {
  std::vector<Vector6d> out(lc.size());
  for (size_t s = 0; s < lc.size(); ++s) {
    const Vector6d& c = lc[s];
    for (int i = 0; i < 3; ++i) {
      out[s][i] = ((T[4 * i] * c[0] + T[4 * i + 1] * c[1]) + T[4 * i + 2] * c[2]) + T[4 * i + 3];
      out[s][3 + i] = (T[4 * i] * c[3] + T[4 * i + 1] * c[4]) + T[4 * i + 2] * c[5];
    }
  }
  return out;
}

static inline double PointToLineDistance3D(const double* p, const double* l) {  // base/Geometry.hpp:198-211
  const double k = (l[3] * (p[0] - l[0]) + l[4] * (p[1] - l[1]) + l[5] * (p[2] - l[2])) / (l[3] * l[3] + l[4] * l[4] + l[5] * l[5]);
  const double q[3] = {k * l[3] + l[0], k * l[4] + l[1], k * l[5] + l[2]};
  return std::sqrt((q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]) + (q[2] - p[2]) * (q[2] - p[2]));
}
// lidar_mapping/LidarFeatureAssociate.cpp:120-197; line_matrix row-major [nei segments x ref segments]
static std::vector<Line2Line> FindAssociationsOn(const Velodyne& ref, const Velodyne& nei, const std::vector<Vector6d>& ref_world,
                                                 const std::vector<Vector6d>& nei_world, const int* line_matrix);
std::vector<Line2Line> FindAssociations(const Velodyne& ref, const Velodyne& nei, const std::vector<Vector6d>& ref_world,
                                        const std::vector<Vector6d>& nei_world, const std::vector<int>& line_matrix) {
  return FindAssociationsOn(ref, nei, ref_world, nei_world, line_matrix.data());
}
// FindAssociations from its second statement on: (max_col, max_count) of every neighbour segment are given (best_col / best_count, one entry
// per row of the vote block: the arg-max loop of :126-130, taken on the host by FindAssociationsOn or on the device by pvlm_line2line_best_batch)
static std::vector<Line2Line> FindAssociationsBest(const Velodyne& ref, const Velodyne& nei, const std::vector<Vector6d>& ref_world,
                                                   const std::vector<Vector6d>& nei_world, const int* best_col, const int* best_count) {
  std::vector<Line2Line> m;                      // one entry per reference segment, ordered by it at the end (upstream: a std::map keyed by the segment)
  const int nr = (int)ref.edge_segmented.size(), nn = (int)nei.edge_segmented.size();
  for (int s = 0; s < nn && nr > 0; ++s) {
    const int max_col = best_col[s], max_count = best_count[s];
    if ((size_t)max_count < nei.edge_segmented[s].size() / 2) continue;
    if (PlaneAngle(&ref_world[max_col][3], &nei_world[s][3]) * 180.0 / M_PI > 7) continue;
    const Vector6d& loc = ref.segment_coeffs[max_col];
    Line2Line a;
    a.neighbor_line_idx = s; a.ref_line_idx = max_col;
    for (int c = 0; c < 3; ++c) { a.line_point1[c] = 0.1 * loc[3 + c] + loc[c]; a.line_point2[c] = -0.1 * loc[3 + c] + loc[c]; }
    auto it = std::find_if(m.begin(), m.end(), [max_col](const Line2Line& x) { return x.ref_line_idx == max_col; });
    if (it == m.end()) m.push_back(a);
    else {
      const double d1 = PointToLineDistance3D(nei_world[it->neighbor_line_idx].data(), ref_world[max_col].data());
      const double d2 = PointToLineDistance3D(nei_world[s].data(), ref_world[max_col].data());
      if (d2 < d1) *it = a;
    }
  }
  std::sort(m.begin(), m.end(), [](const Line2Line& x, const Line2Line& y) { return x.ref_line_idx < y.ref_line_idx; });   // keys are unique
  return m;
}
static std::vector<Line2Line> FindAssociationsOn(const Velodyne& ref, const Velodyne& nei, const std::vector<Vector6d>& ref_world,
                                                 const std::vector<Vector6d>& nei_world, const int* line_matrix) {
  const int nr = (int)ref.edge_segmented.size(), nn = (int)nei.edge_segmented.size();
  std::vector<int> col((size_t)std::max(nn, 1), 0), cnt((size_t)std::max(nn, 1), 0);
  for (int s = 0; s < nn && nr > 0; ++s) {
    int max_col = 0, max_count = line_matrix[(size_t)s * nr];
    for (int c = 1; c < nr; ++c) if (line_matrix[(size_t)s * nr + c] > max_count) { max_count = line_matrix[(size_t)s * nr + c]; max_col = c; }
    col[(size_t)s] = max_col; cnt[(size_t)s] = max_count;
  }
  return FindAssociationsBest(ref, nei, ref_world, nei_world, col.data(), cnt.data());
}

std::vector<Line2Line> AssociateLine2Line(const Velodyne& ref, const Velodyne& nei, const float dist_threshold, bool) {
  std::vector<Line2Line> out;
  if (!ref.IsInWorldCoordinate() || !nei.IsInWorldCoordinate()) { fprintf(stderr, "lidar %d / %d is not in world coordinate\n", ref.id, nei.id); return out; }
  if (ref.edge_segmented.empty() || nei.edge_segmented.empty()) return out;  // CheckLidarSegment
  const std::vector<Vector6d> nei_world = TransformLines(nei.segment_coeffs, nei.GetPose());
  const std::vector<Vector6d> ref_world = TransformLines(ref.segment_coeffs, ref.GetPose());
  std::vector<int> votes(ref.edge_segmented.size() * nei.edge_segmented.size(), 0);
  Engine& e = Engine::Default();
  e.Check(pvlm_line2line_votes(e.ctx(), ref.DeviceScan(), nei.DeviceScan(), dist_threshold, votes.data()), "pvlm_line2line_votes");
  return FindAssociations(ref, nei, ref_world, nei_world, votes);
}

// ---- k-NN based variants: LidarFeatureAssociate.cpp:238-440, :478-548 ------------------------------------------
namespace {
// 5-NN of every nei corner point in ref.cornerLessSharp (world-frame floats), on the GPU: pcl::KdTreeFLANN::nearestKSearch
// of :251-261 / :399-414 / :487-496.  idx/sqd are nq x 5; rows without 5 neighbours within the threshold carry +inf.
constexpr int kLineK = 5;
void CornerKnn(const Velodyne& ref, const Velodyne& nei, float dist_threshold, std::vector<int32_t>& idx, std::vector<float>& sqd) {
  const size_t nq = nei.cornerLessSharp.size();
  idx.assign(nq * kLineK, -1); sqd.assign(nq * kLineK, INFINITY);
  if (nq == 0 || ref.cornerLessSharp.size() < (size_t)kLineK) return;
  std::vector<float> q(nq * 3);
  for (size_t i = 0; i < nq; ++i) { q[3 * i] = nei.cornerLessSharp[i].x; q[3 * i + 1] = nei.cornerLessSharp[i].y; q[3 * i + 2] = nei.cornerLessSharp[i].z; }
  Engine& e = Engine::Default();
  e.Check(pvlm_knn(e.ctx(), ref.DeviceScan(), 1, q.data(), (int)nq, kLineK, dist_threshold, idx.data(), sqd.data()), "pvlm_knn");
}

// principal axis test of FormLine (base/Geometry.hpp:220-260): scatter matrix of the points, cyclic Jacobi rotations;
// a line when the largest eigenvalue exceeds tolerance x the middle one and every point is within dis_threshold of it.
bool FormLine(const double* pts, int n, double tolerance, double dis_threshold, double* line) {
  double c[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) c[k] = c[k] + pts[3 * i + k];
  for (int k = 0; k < 3; ++k) c[k] = c[k] / double(n);
  double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < n; ++i) {
    const double d[3] = {pts[3 * i] - c[0], pts[3 * i + 1] - c[1], pts[3 * i + 2] - c[2]};
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) A[r][k] = A[r][k] + d[r] * d[k];
  }
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    if (A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2] == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        // t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), cs = 1 / sqrt(t^2 + 1), sn = t cs — half of all rotations have |theta| > 2^27 (the
        // last sweeps before the off-diagonal reaches zero), where the same IEEE results come without the square roots: theta^2 >= 2^54 absorbs
        // the + 1, sqrt(fl(theta^2)) is |theta| again (binary round-to-nearest), so t = sign / (2 |theta|) <= 2^-28, t^2 + 1 rounds to 1 and
        // cs = 1, sn = t.  Where theta^2 overflows the written expression gives t = sign / inf = +-0: kept as the general case.
        const double at = std::fabs(theta);
        double t, cs, sn;
        if (at > 134217728.0 && at < 1.0e150) { t = (theta >= 0.0 ? 1.0 : -1.0) / (at + at); cs = 1.0; sn = t; }
        else { t = (theta >= 0.0 ? 1.0 : -1.0) / (at + std::sqrt(theta * theta + 1.0)); cs = 1.0 / std::sqrt(t * t + 1.0); sn = t * cs; }
        const int r = 3 - p - q;
        A[p][p] -= t * apq; A[q][q] += t * apq; A[p][q] = A[q][p] = 0.0;
        const double arp = A[r][p], arq = A[r][q];
        A[r][p] = A[p][r] = cs * arp - sn * arq;
        A[r][q] = A[q][r] = sn * arp + cs * arq;
        for (int k = 0; k < 3; ++k) { const double vp = V[k][p], vq = V[k][q]; V[k][p] = cs * vp - sn * vq; V[k][q] = sn * vp + cs * vq; }
      }
  }
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int a, int b) { return A[a][a] < A[b][b]; });
  for (int k = 0; k < 6; ++k) line[k] = 0.0;
  if (!(A[order[2]][order[2]] > tolerance * A[order[1]][order[1]])) return false;
  double dir[3] = {V[0][order[2]], V[1][order[2]], V[2][order[2]]};
  const double len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  if (len * len > 0.0) for (double& v : dir) v /= len;
  const double l[6] = {c[0], c[1], c[2], dir[0], dir[1], dir[2]};
  if (dis_threshold > 0.0)
    for (int i = 0; i < n; ++i) if (PointToLineDistance3D(pts + 3 * i, l) > dis_threshold) return false;
  for (int k = 0; k < 6; ++k) line[k] = l[k];
  return true;
}

bool WorldOk(const Velodyne& a, const Velodyne& b) {
  if (a.IsInWorldCoordinate() && b.IsInWorldCoordinate()) return true;
  fprintf(stderr, "lidar %d / %d is not in world coordinate\n", a.id, b.id);
  return false;
}
}  // namespace
// public name of the PCA line test for the line branch of the feature extractor (host/pvlm_lines.cpp)
bool FormLine3D(const double* pts, int n, double tolerance, double dis_threshold, double* line) { return FormLine(pts, n, tolerance, dis_threshold, line); }

std::vector<Point2Line> AssociatePoint2Line(const Velodyne& ref, const Velodyne& nei, const float dist_threshold, bool) {   // :478-548
  std::vector<Point2Line> out;
  if (!WorldOk(ref, nei)) return out;
  const float sq_thr = dist_threshold * dist_threshold;
  std::vector<int32_t> idx; std::vector<float> sqd;
  CornerKnn(ref, nei, dist_threshold, idx, sqd);
  for (size_t i = 0; i < nei.cornerLessSharp.size(); ++i) {
    if (!(sqd[i * kLineK + kLineK - 1] <= sq_thr)) continue;                                   // :497-498
    double pts[kLineK * 3];
    for (int j = 0; j < kLineK; ++j) {
      const PointXYZI& p = ref.cornerLessSharp[idx[i * kLineK + j]];
      pts[3 * j] = p.x; pts[3 * j + 1] = p.y; pts[3 * j + 2] = p.z;
    }
    double line[6];
    if (!FormLine(pts, kLineK, 10.0, 0.05, line)) continue;                                    // :506-509
    Vector3d a, b;
    for (int k = 0; k < 3; ++k) { a[k] = 0.1 * line[3 + k] + line[k]; b[k] = -0.1 * line[3 + k] + line[k]; }
    const PointXYZI& q = nei.cornerLessSharp[i];
    out.push_back({nei.World2Local({(double)q.x, (double)q.y, (double)q.z}), ref.World2Local(a), ref.World2Local(b)});
  }
  return out;
}

std::vector<Point2Line> AssociatePoint2LineSegmentKNN(const Velodyne& ref, const Velodyne& nei, const float dist_threshold, bool) {   // :238-317
  std::vector<Point2Line> out;
  if (!WorldOk(ref, nei) || ref.edge_segmented.empty() || nei.edge_segmented.empty()) return out;
  const float sq_thr = dist_threshold * dist_threshold;
  std::vector<int32_t> idx; std::vector<float> sqd;
  CornerKnn(ref, nei, dist_threshold, idx, sqd);
  for (size_t i = 0; i < nei.cornerLessSharp.size(); ++i) {
    if (!(sqd[i * kLineK + kLineK - 1] <= sq_thr)) continue;
    std::map<size_t, size_t> seg_count;
    for (int j = 0; j < kLineK; ++j) for (int sid : ref.point_to_segment[idx[i * kLineK + j]]) seg_count[(size_t)sid]++;
    for (const auto& kv : seg_count) {
      if (kv.second < (size_t)kLineK) continue;                                                // all five neighbours on one segment
      const Vector6d& l = ref.segment_coeffs[kv.first];                                        // LOCAL coefficients (:273-278)
      Vector3d a, b;
      for (int k = 0; k < 3; ++k) { a[k] = 0.1 * l[3 + k] + l[k]; b[k] = -0.1 * l[3 + k] + l[k]; }
      const PointXYZI& q = nei.cornerLessSharp[i];
      out.push_back({nei.World2Local({(double)q.x, (double)q.y, (double)q.z}), a, b});
    }
  }
  return out;
}

std::vector<Point2Line> AssociatePoint2LineSegment(const Velodyne& ref, const Velodyne& nei, const float dist_threshold, bool) {   // :319-383
  std::vector<Point2Line> out;
  if (!WorldOk(ref, nei) || ref.edge_segmented.empty() || nei.edge_segmented.empty()) return out;
  const std::vector<Vector6d> ref_world = TransformLines(ref.segment_coeffs, ref.GetPose());
  for (const PointXYZI& q : nei.cornerLessSharp) {
    const double p[3] = {(double)q.x, (double)q.y, (double)q.z};
    double min_distance = std::numeric_limits<double>::max();
    int seg = -1;
    for (int s = 0; s < (int)ref_world.size(); ++s) {
      const double d = PointToLineDistance3D(p, ref_world[s].data());
      if (d < min_distance) { min_distance = d; seg = s; }
    }
    if (!(min_distance <= dist_threshold)) continue;
    const Vector6d& l = ref.segment_coeffs[seg];
    Vector3d a, b;
    for (int k = 0; k < 3; ++k) { a[k] = 0.1 * l[3 + k] + l[k]; b[k] = -0.1 * l[3 + k] + l[k]; }
    out.push_back({nei.World2Local({p[0], p[1], p[2]}), a, b});
  }
  return out;
}

std::vector<Line2Line> AssociateLine2LineKNN(const Velodyne& ref, const Velodyne& nei, const float dist_threshold, bool) {   // :385-440
  std::vector<Line2Line> out;
  if (!WorldOk(ref, nei) || ref.edge_segmented.empty() || nei.edge_segmented.empty()) return out;
  const float sq_thr = dist_threshold * dist_threshold;
  const std::vector<Vector6d> nei_world = TransformLines(nei.segment_coeffs, nei.GetPose());
  const std::vector<Vector6d> ref_world = TransformLines(ref.segment_coeffs, ref.GetPose());
  std::vector<int> votes(ref.edge_segmented.size() * nei.edge_segmented.size(), 0);
  std::vector<int32_t> idx; std::vector<float> sqd;
  CornerKnn(ref, nei, dist_threshold, idx, sqd);
  for (size_t i = 0; i < nei.cornerLessSharp.size(); ++i) {
    if (!(sqd[i * kLineK + kLineK - 1] <= sq_thr)) continue;
    std::map<size_t, size_t> seg_count;
    for (int j = 0; j < kLineK; ++j) for (int sid : ref.point_to_segment[idx[i * kLineK + j]]) seg_count[(size_t)sid]++;
    for (const auto& kv : seg_count) {
      if (kv.second < (size_t)(kLineK - 2)) continue;                                          // :424
      for (int ns : nei.point_to_segment[i]) votes[(size_t)ns * ref.edge_segmented.size() + kv.first] += 1;
    }
  }
  return FindAssociations(ref, nei, ref_world, nei_world, votes);
}

// Every pair of an outer iteration in one GPU launch (pvlm_line2line_votes_batch); result[k] is what
// AssociateLine2Line(*pairs[k].first, *pairs[k].second, dist_threshold) returns.
std::vector<std::vector<Line2Line>> AssociateLine2LineBatch(const std::vector<std::pair<const Velodyne*, const Velodyne*>>& pairs,
                                                            const float dist_threshold) {
  std::vector<std::vector<Line2Line>> out(pairs.size());
  if (std::getenv("PVLM_HOST_NO_BATCH")) {   // measured variant: one launch + copies per pair, like the reference's call structure
    for (size_t k = 0; k < pairs.size(); ++k) out[k] = AssociateLine2Line(*pairs[k].first, *pairs[k].second, dist_threshold);
    return out;
  }
  std::vector<pvlm_scan*> refs, neis;
  std::vector<size_t> which;
  {
    std::vector<const Velodyne*> need;
    for (const auto& pr : pairs)
      if (pr.first->IsInWorldCoordinate() && pr.second->IsInWorldCoordinate() && !pr.first->edge_segmented.empty() && !pr.second->edge_segmented.empty()) {
        need.push_back(pr.first); need.push_back(pr.second);
      }
    Velodyne::UploadBatch(need);
  }
  for (size_t k = 0; k < pairs.size(); ++k) {
    const Velodyne& ref = *pairs[k].first; const Velodyne& nei = *pairs[k].second;
    if (!ref.IsInWorldCoordinate() || !nei.IsInWorldCoordinate()) { fprintf(stderr, "lidar %d / %d is not in world coordinate\n", ref.id, nei.id); continue; }
    if (ref.edge_segmented.empty() || nei.edge_segmented.empty()) continue;  // CheckLidarSegment
    refs.push_back(ref.DeviceScan()); neis.push_back(nei.DeviceScan()); which.push_back(k);
  }
  if (which.empty()) return out;
  Engine& e = Engine::Default();
  // the vote blocks stay on the device: what comes back is, per neighbour segment, the reference segment with the most votes and that count
  // (pvlm_line2line_best_batch — the arg-max loop of FindAssociations; round 4 copied the blocks, 70 MB at Floor size, and scanned them here)
  std::vector<int64_t> roff(which.size() + 1, 0);
  std::vector<int32_t> best_col, best_count;
  {
    StageTimer stage_timer_votes_("  (inside) line votes of all pairs on the GPU (launch + row maxima back)");
    e.Check(pvlm_line2line_best_batch(e.ctx(), (int)which.size(), refs.data(), neis.data(), dist_threshold, roff.data(), nullptr, nullptr, 0), "pvlm_line2line_best_batch");
    best_col.resize((size_t)std::max<int64_t>(roff.back(), 1)); best_count.resize(best_col.size());
    e.Check(pvlm_line2line_best_batch(e.ctx(), (int)which.size(), refs.data(), neis.data(), dist_threshold, roff.data(), best_col.data(), best_count.data(),
                                      (int64_t)best_col.size()), "pvlm_line2line_best_batch");
  }
  StageTimer stage_timer_("  (inside) FindAssociations on the row maxima (host)");
  // TransformLines(segment_coeffs, pose): once per scan of the batch, not per pair.  The scans get slots in order of first appearance (a hash of the
  // pointers: the ordered map this replaces spent 6 of the stage's 9.5 ms walking its tree at Floor size); their lines are transformed side by side.
  std::unordered_map<const Velodyne*, int> slot_of;
  slot_of.reserve(2 * which.size());
  std::vector<const Velodyne*> scan_of_slot;
  std::vector<std::pair<int, int>> slots(which.size());
  for (size_t j = 0; j < which.size(); ++j) {
    int s[2];
    const Velodyne* v[2] = {pairs[which[j]].first, pairs[which[j]].second};
    for (int q = 0; q < 2; ++q) {
      const auto ins = slot_of.emplace(v[q], (int)scan_of_slot.size());
      if (ins.second) scan_of_slot.push_back(v[q]);
      s[q] = ins.first->second;
    }
    slots[j] = {s[0], s[1]};
  }
  std::vector<std::vector<Vector6d>> world(scan_of_slot.size());
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({pvlm_thread_cap(), which.size() / 256 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
  {
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t k = next++; k < scan_of_slot.size(); k = next++) world[k] = TransformLines(scan_of_slot[k]->segment_coeffs, scan_of_slot[k]->GetPose()); };
    pvlm_run_workers(n_threads, work);
  }
  // the pairs are independent (read-only scans and row tables, one output slot each): pair-parallel
  std::atomic<size_t> next{0};
  auto work = [&]() {                                   // 64 consecutive pairs per turn: the output slots of neighbouring pairs share cache lines
    for (size_t j0 = next.fetch_add(64); j0 < which.size(); j0 = next.fetch_add(64))
      for (size_t j = j0; j < std::min(j0 + 64, which.size()); ++j) {
        const Velodyne& ref = *pairs[which[j]].first; const Velodyne& nei = *pairs[which[j]].second;
        out[which[j]] = FindAssociationsBest(ref, nei, world[(size_t)slots[j].first], world[(size_t)slots[j].second], best_col.data() + roff[j], best_count.data() + roff[j]);
      }
  };
  pvlm_run_workers(n_threads, work);
  return out;
}

// ================================================================================================
// tracks — util/Tracks.h:34-107 (UnionFind), util/Tracks.cpp:58-196 (TrackBuilder, allow_multiple_map)
// ================================================================================================
namespace {
struct UnionFind {
  std::vector<unsigned> parent, rank, size;
  void Init(unsigned n) { size.assign(n, 1); parent.resize(n); std::iota(parent.begin(), parent.end(), 0u); rank.assign(n, 0); }
  unsigned Find(unsigned i) { if (parent[i] != i) parent[i] = Find(parent[i]); return parent[i]; }
  void Union(unsigned i, unsigned j) {
    i = Find(i); j = Find(j);
    if (i == j) return;
    if (rank[i] < rank[j]) { parent[i] = j; size[j] += size[i]; }
    else { parent[j] = i; size[i] += size[j]; if (rank[i] == rank[j]) ++rank[i]; }
  }
};
}  // namespace

bool LidarLineMatch::GenerateTracks() {
  StageTimer stage_timer_("line tracks (associate + union-find)");
  std::vector<std::pair<size_t, size_t>> pairs;
  const std::vector<std::vector<int>> neighbors = FindNeighbors(lidars_, neighbor_size_);
  const bool sharded = exchange_ && exchange_->active();
  std::vector<std::pair<const Velodyne*, const Velodyne*>> todo;   // the (ref, nei) arguments of AssociateLine2Line, :68
  std::vector<size_t> todo_pair;                                   // position of todo[k] in `pairs` (sharded: this rank's pairs only)
  for (size_t i = 0; i < neighbors.size(); i++) {
    if (!lidars_[i].IsPoseValid()) continue;
    for (const int nei_id : neighbors[i]) {
      if (nei_id < 0 || nei_id >= (int)lidars_.size()) continue;
      if (!sharded || (i >= first_ && i < last_)) { todo.push_back({&lidars_[nei_id], &lidars_[i]}); todo_pair.push_back(pairs.size()); }
      pairs.push_back({i, (size_t)nei_id});
    }
  }
  const std::vector<std::vector<Line2Line>> all_ass = AssociateLine2LineBatch(todo, 0.3f);   // one launch for the whole loop
  // feature_each_pair: the (neighbour segment, reference segment) matches of every pair as a std::set orders them (sorted, unique)
  typedef std::pair<uint32_t, uint32_t> Feature;      // (scan, segment)
  std::vector<std::vector<Feature>> fpairs(pairs.size());
  for (size_t k = 0; k < all_ass.size(); ++k) {
    std::vector<Feature>& fp = fpairs[todo_pair[k]];
    for (const Line2Line& a : all_ass[k]) fp.push_back({(uint32_t)a.neighbor_line_idx, (uint32_t)a.ref_line_idx});
    std::sort(fp.begin(), fp.end()); fp.erase(std::unique(fp.begin(), fp.end()), fp.end());
  }
  if (sharded) {
    // every rank holds the matches of its own pairs: counts per pair, then the matches themselves, summed over the ranks (each entry is
    // written by exactly one rank) — the one primitive an Exchange has.  Segment ids are small integers: exact in a double.
    StageTimer stage_timer_x_("  (inside) line tracks: matches of all ranks concatenated (2 all-reduces)");
    std::vector<double> cnt(pairs.size(), 0.0);
    for (size_t k = 0; k < pairs.size(); ++k) cnt[k] = (double)fpairs[k].size();
    if (!cnt.empty()) exchange_->allreduce_sum(cnt.data(), cnt.size());
    std::vector<size_t> off(pairs.size() + 1, 0);
    for (size_t k = 0; k < pairs.size(); ++k) off[k + 1] = off[k] + (size_t)cnt[k];
    std::vector<double> flat(std::max<size_t>(2 * off.back(), 1), 0.0);
    for (size_t k = 0; k < pairs.size(); ++k)
      for (size_t m = 0; m < fpairs[k].size(); ++m) { flat[2 * (off[k] + m)] = (double)fpairs[k][m].first; flat[2 * (off[k] + m) + 1] = (double)fpairs[k][m].second; }
    exchange_->allreduce_sum(flat.data(), flat.size());
    for (size_t k = 0; k < pairs.size(); ++k) {
      fpairs[k].resize((size_t)cnt[k]);
      for (size_t m = 0; m < fpairs[k].size(); ++m) fpairs[k][m] = {(uint32_t)flat[2 * (off[k] + m)], (uint32_t)flat[2 * (off[k] + m) + 1]};
    }
  }
  // TrackBuilder(true).Build — util/Tracks.cpp:58-196 with its std::set / std::map containers replaced by sorted vectors and
  // dense tables: the same features in the same order (a set iterates in sorted order), the same unions in the same order
  StageTimer stage_timer_tb_("  (inside) TrackBuilder: union-find + filter + export (host)");
  // Feature (scan, segment) -> index in the SORTED set of all features that occur (upstream: a std::set filled from every match, then
  // numbered in iteration order).  A dense table over (scan, segment) gives the same numbering without sorting 2 x matches features and
  // without a binary search per union: mark what occurs, number the marks in table order = (scan, segment) order.
  std::vector<uint32_t> seg_base(lidars_.size() + 1, 0);
  {
    std::vector<uint32_t> seg_count(lidars_.size(), 0);
    for (size_t s = 0; s < lidars_.size(); ++s) seg_count[s] = (uint32_t)lidars_[s].edge_segmented.size();
    for (size_t i = 0; i < pairs.size(); i++)
      for (const Feature& mth : fpairs[i]) {
        seg_count[pairs[i].first] = std::max(seg_count[pairs[i].first], mth.first + 1);
        seg_count[pairs[i].second] = std::max(seg_count[pairs[i].second], mth.second + 1);
      }
    for (size_t s = 0; s < lidars_.size(); ++s) seg_base[s + 1] = seg_base[s] + seg_count[s];
  }
  std::vector<uint32_t> rank(seg_base.back(), 0);
  for (size_t i = 0; i < pairs.size(); i++)
    for (const Feature& mth : fpairs[i]) { rank[seg_base[pairs[i].first] + mth.first] = 1; rank[seg_base[pairs[i].second] + mth.second] = 1; }
  std::vector<Feature> i2f;
  for (size_t sc = 0; sc < lidars_.size(); ++sc)
    for (uint32_t cell = seg_base[sc]; cell < seg_base[sc + 1]; ++cell)
      if (rank[cell]) { rank[cell] = (uint32_t)i2f.size(); i2f.push_back({(uint32_t)sc, cell - seg_base[sc]}); }
  auto f2i = [&](const Feature& f) { return rank[seg_base[f.first] + f.second]; };
  UnionFind uf;
  uf.Init((unsigned)i2f.size());
  for (size_t i = 0; i < pairs.size(); i++)
    for (const Feature& mth : fpairs[i]) uf.Union(f2i({(uint32_t)pairs[i].first, mth.first}), f2i({(uint32_t)pairs[i].second, mth.second}));
  // Filter(min_track_length): a track must span at least min_track_length different scans
  {
    // distinct scans per root: the features come in scan order, so a root sees each of its scans in one run
    std::vector<uint32_t> scans_of(i2f.size(), 0), last_scan(i2f.size(), std::numeric_limits<uint32_t>::max());
    for (uint32_t i = 0; i < i2f.size(); i++) {
      const uint32_t root = uf.Find(i);
      if (last_scan[root] != i2f[i].first) { last_scan[root] = i2f[i].first; scans_of[root]++; }
    }
    std::vector<char> bad(i2f.size(), 0);
    for (uint32_t r = 0; r < i2f.size(); r++) bad[r] = scans_of[r] > 0 && scans_of[r] < (uint32_t)min_track_length_;
    // upstream walks the parent array once, testing each entry's CURRENT value (a root already invalidated no longer matches)
    for (unsigned& root : uf.parent)
      if (root != std::numeric_limits<uint32_t>::max() && bad[root]) { uf.size[root] = 1; root = std::numeric_limits<uint32_t>::max(); }
  }
  // ExportTracks
  std::vector<int> t2i(i2f.size(), -1);
  tracks_.clear();
  for (uint32_t i = 0; i < i2f.size(); i++) {
    const uint32_t tid = uf.parent[i];
    if (tid != std::numeric_limits<uint32_t>::max() && uf.size[tid] > 1) {
      if (t2i[tid] < 0) { t2i[tid] = (int)tracks_.size(); LineTrack t; t.id = tid; tracks_.push_back(t); }
      std::set<Feature>& fs = tracks_[(size_t)t2i[tid]].feature_pairs;
      fs.insert(fs.end(), i2f[i]);          // i2f is sorted: every insertion goes to the end
    }
  }
  for (size_t i = 0; i < tracks_.size(); i++) tracks_[i].id = (uint32_t)i;
  return true;
}


}  // namespace pvlm
