// pvlm_host_camera_lidar.cpp — part of the C++ host mirror (pvlm_host.hpp): joint_optimization/CameraLidarLineAssociate.cpp, CameraLidarOptimizer.cpp, the host Equirectangular, MVS::SelectNeighborKNN and MVS::FuseDepthImages.
// Host logic only; every residual, Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host_internal.hpp"

namespace pvlm {

// ================================================================================================
// MVS::SelectNeighborKNN — mvs/MVS.cpp:334-382
// ================================================================================================
std::vector<std::vector<NeighborInfo>> SelectNeighborKNN(const std::vector<Frame>& frames, int neighbor_size, float sq_distance_threshold) {
  std::vector<std::vector<NeighborInfo>> neighbors(frames.size());
  std::vector<std::array<float, 3>> center;
  std::vector<size_t> owner;
  for (size_t i = 0; i < frames.size(); ++i) {
    if (!frames[i].IsPoseValid()) continue;
    center.push_back({(float)frames[i].t_wc[0], (float)frames[i].t_wc[1], (float)frames[i].t_wc[2]});
    owner.push_back(i);
  }
  const int nc = (int)owner.size(), k = std::min(neighbor_size * 3, nc);
  for (size_t ref = 0; ref < frames.size(); ++ref) {
    if (!frames[ref].IsPoseValid()) continue;
    const Frame& fr = frames[ref];
    const float q[3] = {(float)fr.t_wc[0], (float)fr.t_wc[1], (float)fr.t_wc[2]};
    std::vector<std::pair<float, int>> d(nc);
    for (int j = 0; j < nc; ++j) {
      const float dx = q[0] - center[j][0], dy = q[1] - center[j][1], dz = q[2] - center[j][2];
      float sq = 0.0f; sq += dx * dx; sq += dy * dy; sq += dz * dz;
      d[j] = {sq, j};
    }
    std::stable_sort(d.begin(), d.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    for (int i = 1; i < k && (int)neighbors[ref].size() < neighbor_size; ++i) {    // i = 0: "the nearest one is always the view itself"
      if (d[i].first < sq_distance_threshold) continue;                             // too close: the baseline would be too short
      const Frame& fn = frames[owner[d[i].second]];
      NeighborInfo info;
      info.id = owner[d[i].second];
      // T_nr = T_wn^-1 T_wr:  R_nr = R_wn^T R_wr,  t_nr = R_wn^T (t_wr - t_wn)
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
          double acc = 0;
          for (int m = 0; m < 3; ++m) acc += fn.R_wc[3 * m + r] * fr.R_wc[3 * m + c];
          info.R_nr[3 * r + c] = (float)acc;
        }
        double acc = 0;
        for (int m = 0; m < 3; ++m) acc += fn.R_wc[3 * m + r] * (fr.t_wc[m] - fn.t_wc[m]);
        info.t_nr[r] = (float)acc;
      }
      neighbors[ref].push_back(info);
    }
  }
  return neighbors;
}

std::vector<Matrix3d> LidarOdometry::GetGlobalRotation() const { std::vector<Matrix3d> r; for (const Velodyne& l : lidars) r.push_back(l.GetRotation()); return r; }
std::vector<Vector3d> LidarOdometry::GetGlobalTranslation() const { std::vector<Vector3d> t; for (const Velodyne& l : lidars) t.push_back(l.GetTranslation()); return t; }


// ================================================================================================
// Equirectangular (host, scalar) — sensors/Equirectangular.h:42-182, .cpp:20-65, base/Math.h:15-29
// ================================================================================================
namespace {
inline double VectorAngle3D(const double* a, const double* b) {
  double c = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  c = c / (std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]));
  if (c >= 1.0) return 0.0;
  if (c <= -1.0) return M_PI;
  return std::acos(c);
}
inline void ProjectPointToPlane(const double* p, const double* pl, double* o) {  // normalized = true
  const double dis = std::fabs(pl[0] * p[0] + pl[1] * p[1] + pl[2] * p[2] + pl[3]);
  o[0] = p[0] - dis * pl[0]; o[1] = p[1] - dis * pl[1]; o[2] = p[2] - dis * pl[2];
  if (std::fabs(pl[0] * o[0] + pl[1] * o[1] + pl[2] * o[2] + pl[3]) > 1e-4) { o[0] = p[0] + dis * pl[0]; o[1] = p[1] + dis * pl[1]; o[2] = p[2] + dis * pl[2]; }
}
inline void FormPlane0(const double* p1, const double* p2, double* out) {  // FormPlane(p1, p2, 0).normalize() as a 4-vector
  const double p3[3] = {0, 0, 0};
  double a = ((p2[1] - p1[1]) * (p3[2] - p1[2]) - (p2[2] - p1[2]) * (p3[1] - p1[1]));
  double b = ((p2[2] - p1[2]) * (p3[0] - p1[0]) - (p2[0] - p1[0]) * (p3[2] - p1[2]));
  double c = ((p2[0] - p1[0]) * (p3[1] - p1[1]) - (p2[1] - p1[1]) * (p3[0] - p1[0]));
  double d = -(a * p1[0] + b * p1[1] + c * p1[2]);
  const double n = std::sqrt(a * a + b * b + c * c + d * d);
  if (n * n > 0.0) { a /= n; b /= n; c /= n; d /= n; }
  out[0] = a; out[1] = b; out[2] = c; out[3] = d;
}
inline Vector3d Transform4(const Matrix4d& T, const Vector3d& p) {  // (T * p.homogeneous()).hnormalized()
  double h[4];
  for (int i = 0; i < 4; ++i) h[i] = ((T[4 * i] * p[0] + T[4 * i + 1] * p[1]) + T[4 * i + 2] * p[2]) + T[4 * i + 3] * 1.0;
  return {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
}
inline Matrix4d Inverse4(const Matrix4d& A) {
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
  Matrix4d o;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o[i * 4 + j] = m[i][4 + j];
  return o;
}
}  // namespace

// ================================================================================================
// CameraLidarLineAssociate — joint_optimization/CameraLidarLineAssociate.cpp:340-475, :628-715, :754-876
// ================================================================================================
void CameraLidarLineAssociate::AssociateByAngle(const std::vector<std::array<float, 4>>& lines, const Velodyne& lidar, const Matrix4d& T_cl,
                                                const bool multiple_association, const std::vector<bool>& image_line_mask,
                                                const std::vector<bool>& lidar_line_mask) {
  // hot loop #3 (:394-426) on the GPU: votes[line][segment]
  const size_t n_seg = lidar.edge_segmented.size();
  std::vector<int> votes(lines.size() * std::max<size_t>(n_seg, 1), 0);
  Engine& e = Engine::Default();
  if (!lines.empty() && n_seg > 0)
    e.Check(pvlm_cam_lidar_votes(e.ctx(), rows, cols, &lines[0][0], (int)lines.size(), lidar.DeviceScan(), T_cl.data(), votes.data()), "pvlm_cam_lidar_votes");
  AssociateByAngleWithVotes(lines, lidar, T_cl, votes.data(), multiple_association, image_line_mask, lidar_line_mask);
}

void CameraLidarLineAssociate::AssociateByAngleWithVotes(const std::vector<std::array<float, 4>>& lines, const Velodyne& lidar, const Matrix4d& T_cl,
                                                         const int* votes, const bool multiple_association, const std::vector<bool>& image_line_mask,
                                                         const std::vector<bool>& lidar_line_mask) {
  const size_t n_seg = lidar.edge_segmented.size();
  std::vector<bool> image_mask = image_line_mask.empty() ? std::vector<bool>(lines.size(), true) : image_line_mask;
  std::vector<bool> lidar_mask = lidar_line_mask.empty() ? std::vector<bool>(n_seg, true) : lidar_line_mask;
  line_pairs.clear();
  std::vector<Vector3d> ends_cam;
  std::vector<Vector4d> lidar_plane;
  for (size_t i = 0; i < n_seg; i++) {
    const Vector3d p1 = Transform4(T_cl, lidar.end_points[2 * i]), p2 = Transform4(T_cl, lidar.end_points[2 * i + 1]);
    ends_cam.push_back(p1); ends_cam.push_back(p2);
    Vector4d pl; FormPlane0(p1.data(), p2.data(), pl.data());
    lidar_plane.push_back(pl);
  }
  const double angle_threshold = 3.0 / 180.0 * M_PI;
  Equirect eq{cols, rows};
  for (size_t li = 0; li < lines.size(); li++) {
    if (!image_mask[li]) continue;
    const std::array<float, 4>& l = lines[li];
    const double a[2] = {l[0], l[1]}, b[2] = {l[2], l[3]};
    double p1[3], p2[3], ip[4];
    eq.ImageToCam(a, 1.0, p1); eq.ImageToCam(b, 1.0, p2);
    FormPlane0(p1, p2, ip);
    const double p4[3] = {(p1[0] + p2[0]) / 2.0, (p1[1] + p2[1]) / 2.0, (p1[2] + p2[2]) / 2.0};
    const double scope = VectorAngle3D(p1, p4);
    for (size_t s = 0; s < n_seg; ++s) {   // std::map iteration order = ascending segment id
      const size_t cnt = (size_t)votes[li * n_seg + s];
      if (cnt == 0) continue;
      if (cnt < lidar.edge_segmented[s].size() / 2) continue;
      if (!lidar_mask[s]) continue;
      const double angle = PlaneAngleN(ip, lidar_plane[s].data());
      if (angle > angle_threshold) continue;
      const double mid[3] = {(ends_cam[2 * s][0] + ends_cam[2 * s + 1][0]) / 2.0, (ends_cam[2 * s][1] + ends_cam[2 * s + 1][1]) / 2.0,
                             (ends_cam[2 * s][2] + ends_cam[2 * s + 1][2]) / 2.0};
      double midp[3];
      ProjectPointToPlane(mid, ip, midp);
      if (VectorAngle3D(midp, p4) > scope) continue;
      const float angle2 = (float)VectorAngle3D(mid, midp);
      if (angle2 > angle_threshold / 2.0) continue;
      CameraLidarLinePair lp;
      lp.image_line = l; lp.lidar_line_start = ends_cam[2 * s]; lp.lidar_line_end = ends_cam[2 * s + 1];
      lp.image_line_id = (int)li; lp.lidar_line_id = (int)s; lp.angle = (float)(angle + angle2);
      line_pairs.push_back(lp);
    }
  }
  Filter(false, true);
  if (!multiple_association) UniqueLinePair(lines, ends_cam);
  const Matrix4d T_lc = Inverse4(T_cl);
  for (CameraLidarLinePair& lp : line_pairs) { lp.lidar_line_start = Transform4(T_lc, lp.lidar_line_start); lp.lidar_line_end = Transform4(T_lc, lp.lidar_line_end); }
}

void CameraLidarLineAssociate::Filter(bool filter_by_angle, bool filter_by_length) {
  (void)filter_by_angle;  // AssociateByAngle calls Filter(false, true) only
  const float min_len = 100, max_len = 2000;
  std::vector<CameraLidarLinePair> good;
  Equirect eq{cols, rows};
  for (const CameraLidarLinePair& p : line_pairs) {
    if (filter_by_length) {
      const float a[3] = {(float)p.lidar_line_start[0], (float)p.lidar_line_start[1], (float)p.lidar_line_start[2]};
      const float b[3] = {(float)p.lidar_line_end[0], (float)p.lidar_line_end[1], (float)p.lidar_line_end[2]};
      float pa[2], pb[2];
      eq.CamToImage(a, pa); eq.CamToImage(b, pb);
      const std::vector<float> seg = eq.BreakToSegments(pa, pb, 100);
      float len = 0;
      const size_t n = seg.size() / 2;
      for (size_t i = 0; i + 1 < n; i++) {
        if (std::abs(seg[2 * i] - seg[2 * (i + 1)]) > 0.8 * cols) continue;
        const float dx = seg[2 * i] - seg[2 * (i + 1)], dy = seg[2 * i + 1] - seg[2 * (i + 1) + 1];
        len += std::sqrt(dx * dx + dy * dy);
      }
      if (len < min_len || len > max_len) continue;
    }
    good.push_back(p);
  }
  line_pairs.swap(good);
}

void CameraLidarLineAssociate::UniqueLinePair(const std::vector<std::array<float, 4>>& lines, const std::vector<Vector3d>& ends) {
  struct PairScore { int idx; float score; };
  std::map<int, PairScore> i2l, l2i;
  for (const CameraLidarLinePair& pr : line_pairs) {
    const int il = pr.image_line_id, ll = pr.lidar_line_id;
    const float score = pr.angle;
    auto a = i2l.find(il); auto b = l2i.find(ll);
    const bool ha = a != i2l.end(), hb = b != l2i.end();
    if (!ha && !hb) { i2l.insert({il, {ll, score}}); l2i.insert({ll, {il, score}}); }
    else if (ha && !hb) { if (score < a->second.score) { l2i.erase(l2i.find(a->second.idx)); a->second = {ll, score}; l2i.insert({ll, {il, score}}); } }
    else if (!ha && hb) { if (score < b->second.score) { i2l.erase(i2l.find(b->second.idx)); b->second = {il, score}; i2l.insert({il, {ll, score}}); } }
    else {
      const float sa = a->second.score, sb = b->second.score;
      if (score < std::min(sa, sb)) {
        i2l.erase(b->second.idx); l2i.erase(a->second.idx); i2l.erase(a); l2i.erase(b);
        i2l.insert({il, {ll, score}}); l2i.insert({ll, {il, score}});
      } else if (score > sa && score < sb) { i2l.erase(i2l.find(b->second.idx)); l2i.erase(b); }
      else if (score < sa && score > sb) { l2i.erase(l2i.find(a->second.idx)); i2l.erase(a); }
    }
  }
  line_pairs.clear();
  for (auto& kv : i2l) {
    CameraLidarLinePair lp;
    lp.image_line = lines[kv.first]; lp.lidar_line_start = ends[2 * kv.second.idx]; lp.lidar_line_end = ends[2 * kv.second.idx + 1];
    lp.image_line_id = kv.first; lp.lidar_line_id = kv.second.idx; lp.angle = kv.second.score;
    line_pairs.push_back(lp);
  }
}


// ================================================================================================
// CameraLidarOptimizer (mapping mode) — joint_optimization/CameraLidarOptimizer.cpp:260-285, :331-548, :551-566
// ================================================================================================
static Matrix4d Mul4(const Matrix4d& A, const Matrix4d& B) {
  Matrix4d C;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j]; C[4 * i + j] = s; }
  return C;
}

std::vector<uint16_t> ProjectLidar2PanoramaDepth(const PointCloud& cloud, const int rows, const int cols, const Matrix4d& T_cl, const size_t size) {
  std::vector<float> xyz(cloud.size() * 3);
  for (size_t i = 0; i < cloud.size(); ++i) { xyz[3 * i] = cloud[i].x; xyz[3 * i + 1] = cloud[i].y; xyz[3 * i + 2] = cloud[i].z; }
  std::vector<uint16_t> img((size_t)rows * cols, 0);
  Engine& e = Engine::Default();
  e.Check(pvlm_project_lidar_depth(e.ctx(), rows, cols, (int64_t)cloud.size(), xyz.data(), T_cl.data(), (unsigned)size, img.data()), "pvlm_project_lidar_depth");
  return img;
}

std::vector<std::vector<int>> CameraLidarOptimizer::NeighborEachFrame(const int neighbor_size, const bool temporal) const {
  std::vector<std::vector<int>> out(frames.size());
  if (temporal) {                                                       // :555-567
    for (int frame_id = 0; frame_id < (int)frames.size(); frame_id++) {
      int start = std::max(0, frame_id - (neighbor_size / 2));
      const int end = std::min((int)lidars.size(), start + neighbor_size);
      start = std::max(0, end - neighbor_size);
      for (int l = start; l < end; l++) out[frame_id].push_back(l);
    }
    return out;
  }
  // :569-608 — the neighbor_size LiDAR scans whose centres are nearest to the camera centre (float32 centres, flann::L2_Simple order, equal
  // distances by position like every k-NN of this mirror), then the scans before and after the frame's own index when they are not among them
  std::vector<std::array<float, 3>> center;
  std::vector<int> owner;
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    const Vector3d& t = lidars[i].GetTranslation();
    center.push_back({float(t[0]), float(t[1]), float(t[2])});
    owner.push_back((int)i);
  }
  std::vector<std::pair<float, int>> d(center.size());
  for (int i = 0; i < (int)frames.size(); i++) {
    if (!frames[i].IsPoseValid()) continue;
    const Matrix4d T = frames[i].GetPose();
    const float q[3] = {float(T[3]), float(T[7]), float(T[11])};
    for (size_t j = 0; j < center.size(); ++j) {
      const float dx = q[0] - center[j][0], dy = q[1] - center[j][1], dz = q[2] - center[j][2];
      float s = 0.0f; s += dx * dx; s += dy * dy; s += dz * dz;
      d[j] = {s, (int)j};
    }
    const size_t k = std::min<size_t>((size_t)std::max(neighbor_size, 0), d.size());
    std::partial_sort(d.begin(), d.begin() + k, d.end());
    std::vector<int> neighbors;
    for (size_t j = 0; j < k; ++j) neighbors.push_back(owner[(size_t)d[j].second]);
    const std::set<int> have(neighbors.begin(), neighbors.end());
    if (have.count(i - 1) == 0 && i - 1 >= 0) neighbors.push_back(i - 1);
    if (have.count(i + 1) == 0 && i + 1 < (int)lidars.size()) neighbors.push_back(i + 1);
    out[(size_t)i] = neighbors;
  }
  return out;
}

CameraLidarOptimizer::LinePairs CameraLidarOptimizer::AssociateLineMulti(const int neighbor_size, const bool temporal) {
  StageTimer stage_timer_("camera-LiDAR line association");
  const std::vector<std::vector<int>> nb = NeighborEachFrame(neighbor_size, temporal);
  LinePairs all;
  // the voting loops of every (frame, LiDAR) pair in ONE launch (upstream: omp parallel for over frames, :345)
  struct Job { size_t f; int lid; Matrix4d T_cl; };
  std::vector<Job> jobs;
  std::vector<pvlm_scan*> scans;
  std::vector<const Velodyne*> scan_of_job;
  std::vector<int64_t> line_off(1, 0);
  std::vector<float> lines_flat;
  std::vector<double> T_flat;
  for (size_t f = 0; f < frames.size(); f++) {
    for (const int lid : nb[f]) {
      const Velodyne& lidar = lidars[lid];
      Matrix4d T_cl = T_cl_init;
      if (frames[f].IsPoseValid() && lidar.IsPoseValid()) T_cl = Mul4(Inverse4(frames[f].GetPose()), lidar.GetPose());
      all[{f, (size_t)lid}] = {};
      if (lidar.edge_segmented.empty() || frames[f].lines.empty()) continue;
      jobs.push_back({f, lid, T_cl});
      scan_of_job.push_back(&lidar);
      for (const auto& l : frames[f].lines) lines_flat.insert(lines_flat.end(), l.begin(), l.end());
      line_off.push_back((int64_t)lines_flat.size() / 4);
      T_flat.insert(T_flat.end(), T_cl.begin(), T_cl.end());
    }
  }
  if (jobs.empty()) return all;
  Velodyne::UploadBatch(scan_of_job);
  for (const Velodyne* v : scan_of_job) scans.push_back(v->DeviceScan());
  if (std::getenv("PVLM_HOST_NO_BATCH")) {   // measured variant: per-pair launches
    for (const Job& j : jobs) {
      CameraLidarLineAssociate associate(frames[j.f].rows, frames[j.f].cols);
      associate.AssociateByAngle(frames[j.f].lines, lidars[j.lid], j.T_cl, true);
      all[{j.f, (size_t)j.lid}] = associate.GetAssociatedPairs();
    }
    return all;
  }
  const int rows = frames[jobs[0].f].rows, cols = frames[jobs[0].f].cols;   // one image size per sequence, like AddCameraResidual assumes
  Engine& e = Engine::Default();
  // the votes come back sparse — a few per cent of the (line, segment) counters are non-zero: 43 MB of dense blocks for a Room sequence — and
  // every pair's block is spread out again in a buffer of its own size
  std::vector<int64_t> voff(jobs.size() + 1, 0);
  std::vector<int64_t> nz_index((size_t)std::max<int64_t>(4 * (line_off.back()), 1024));
  std::vector<int32_t> nz_count(nz_index.size());
  int64_t n_nz = 0;
  pvlm_status rc = pvlm_cam_lidar_votes_batch_sparse(e.ctx(), (int)jobs.size(), rows, cols, line_off.data(), lines_flat.data(), scans.data(), T_flat.data(), voff.data(),
                                                     nz_index.data(), nz_count.data(), (int64_t)nz_index.size(), &n_nz);
  if (rc == PVLM_ERR_CAPACITY && n_nz > (int64_t)nz_index.size()) {
    nz_index.resize((size_t)n_nz); nz_count.resize((size_t)n_nz);
    rc = pvlm_cam_lidar_votes_batch_sparse(e.ctx(), (int)jobs.size(), rows, cols, line_off.data(), lines_flat.data(), scans.data(), T_flat.data(), voff.data(),
                                           nz_index.data(), nz_count.data(), (int64_t)nz_index.size(), &n_nz);
  }
  e.Check(rc, "pvlm_cam_lidar_votes_batch_sparse");
  std::vector<int32_t> votes;
  int64_t k = 0;
  for (size_t j = 0; j < jobs.size(); ++j) {
    const Frame& fr = frames[jobs[j].f];
    if (fr.rows != rows || fr.cols != cols) throw std::runtime_error("AssociateLineMulti: frames of different image size");
    votes.assign((size_t)std::max<int64_t>(voff[j + 1] - voff[j], 1), 0);
    for (; k < n_nz && nz_index[(size_t)k] < voff[j + 1]; ++k) votes[(size_t)(nz_index[(size_t)k] - voff[j])] = nz_count[(size_t)k];
    CameraLidarLineAssociate associate(fr.rows, fr.cols);
    associate.AssociateByAngleWithVotes(fr.lines, lidars[jobs[j].lid], jobs[j].T_cl, votes.data(), true);
    all[{jobs[j].f, (size_t)jobs[j].lid}] = associate.GetAssociatedPairs();
  }
  return all;
}

// Calibration mode (CameraLidarOptimizer.cpp:32-87).  The two functors of this mode, Plane2Plane_Relative (base/CostFunction.h:294-348) and
// PlaneRelativeIOUResidual (:509-565), map a LiDAR point with ONE pose, P_c = R(aa_cl) P_l + t_cl.  That is the chain of Plane2Plane_Global /
// PlaneIOUResidual — P_c = R(aa_cw) (R(-aa_lw) (P_l - t_lw)) + t_cw — with the LiDAR pose at the identity, where it is exact: ceres'
// AngleAxisRotatePoint takes its first-order branch for a zero rotation and returns the point unchanged.  So the blocks are kinds 4 and 5 of
// the GPU evaluation with (aa_cw, t_cw) = (aa_cl, t_cl) free and a constant identity for the second pose; the derivative with respect to
// (aa_cl, t_cl) is the first half of the row.  Plane2Plane_Relative returns weight * angle * 180 / pi: folded into the block weight,
// i.e. (weight * 180 / pi) * angle — mathematically the same, up to 1 ulp away from upstream's left-to-right product (the twin test
// compares at 1e-6).
int CameraLidarOptimizer::Optimize(const LinePairs& line_pairs, const Matrix4d& T_cl, double* final_cost, int* successful_steps, int* residual_blocks) {
  ceres_like::Problem problem;
  ceres_like::LossFunction* loss_function = new ceres_like::HuberLoss(2.0 * M_PI / 180.0);                 // :36
  const Matrix3d R = {T_cl[0], T_cl[1], T_cl[2], T_cl[4], T_cl[5], T_cl[6], T_cl[8], T_cl[9], T_cl[10]};
  Vector3d angle_axis, t = {T_cl[3], T_cl[7], T_cl[11]};
  RotationMatrixToAngleAxis(R, &angle_axis);
  Vector3d aa_id = {0, 0, 0}, t_id = {0, 0, 0};
  if (frames.empty()) { delete loss_function; T_cl_optimized = T_cl; return 1; }
  const Equirect eq{frames[0].cols, frames[0].rows};                                                       // :41
  size_t blocks = 0;
  for (const auto& kv : line_pairs)
    for (const CameraLidarLinePair& pair : kv.second) {
      // ImageToCam(cv::Point2f, float(5.0)) -> cv::Point3f; the plane through p1, p2 and the centre in FLOAT arithmetic, then widened (:51-57)
      const float a2[2] = {pair.image_line[0], pair.image_line[1]}, b2[2] = {pair.image_line[2], pair.image_line[3]};
      float p1[3], p2[3];
      eq.ImageToCam(a2, 5.0f, p1); eq.ImageToCam(b2, 5.0f, p2);
      const float p3[3] = {0.f, 0.f, 0.f};
      const double a = ((p2[1] - p1[1]) * (p3[2] - p1[2]) - (p2[2] - p1[2]) * (p3[1] - p1[1]));
      const double b = ((p2[2] - p1[2]) * (p3[0] - p1[0]) - (p2[0] - p1[0]) * (p3[2] - p1[2]));
      const double c = ((p2[0] - p1[0]) * (p3[1] - p1[1]) - (p2[1] - p1[1]) * (p3[0] - p1[0]));
      problem.AddResidualBlock(Plane2Plane_Global::Create({a, b, c}, pair.lidar_line_end, pair.lidar_line_start, 1.0 * 180.0 / M_PI), loss_function,
                               angle_axis.data(), t.data(), aa_id.data(), t_id.data());                     // :59-60
      // PlaneRelativeIOUResidual(plane, middle, p1, p2, 2): angle = VectorAngle3D(p1, p2) / 2.f and the midpoint, both in float (:521-527);
      // VectorAngle3D<float> (base/Geometry.hpp:432-448) with the float overloads of sqrt / acos [recalled: <cmath> in scope]
      float cos_angle = (p1[0] * p2[0] + p1[1] * p2[1]) + p1[2] * p2[2];
      const float norm1 = std::sqrt((p1[0] * p1[0] + p1[1] * p1[1]) + p1[2] * p1[2]), norm2 = std::sqrt((p2[0] * p2[0] + p2[1] * p2[1]) + p2[2] * p2[2]);
      cos_angle /= (norm1 * norm2);
      const float full = cos_angle >= 1.f ? 0.f : (cos_angle <= -1.f ? (float)M_PI : std::acos(cos_angle));
      const float half_arc = full / 2.f;
      const Vector3d mid_i = {(double)((p1[0] + p2[0]) / 2.f), (double)((p1[1] + p2[1]) / 2.f), (double)((p1[2] + p2[2]) / 2.f)};
      const Vector3d mid_l = {(pair.lidar_line_start[0] + pair.lidar_line_end[0]) / 2.0, (pair.lidar_line_start[1] + pair.lidar_line_end[1]) / 2.0,
                              (pair.lidar_line_start[2] + pair.lidar_line_end[2]) / 2.0};
      problem.AddResidualBlock(PlaneIOUResidual::Create({a, b, c, 0.0}, mid_l, mid_i, (double)half_arc, 2.0), nullptr, angle_axis.data(), t.data(),
                               aa_id.data(), t_id.data());                                                   // :62-64
      blocks += 2;
    }
  if (blocks == 0) {
    delete loss_function; T_cl_optimized = T_cl;
    if (residual_blocks) *residual_blocks = 0;
    if (final_cost) *final_cost = 0.0;                  // nothing to solve: the callers read these unconditionally
    if (successful_steps) *successful_steps = 0;
    return 1;
  }
  problem.SetParameterBlockConstant(aa_id.data());
  problem.SetParameterBlockConstant(t_id.data());
  ceres_like::Solver::Options options;                                                                     // :71-77
  options.max_num_iterations = 50;
  options.linear_solver_type = ceres_like::SPARSE_SCHUR;
  options.num_threads = 10;
  ceres_like::Solver::Summary summary;
  ceres_like::Solve(options, &problem, &summary);
  Matrix3d Ro;
  AngleAxisToRotationMatrix(angle_axis, &Ro);                                                              // :83-86
  T_cl_optimized = {Ro[0], Ro[1], Ro[2], t[0], Ro[3], Ro[4], Ro[5], t[1], Ro[6], Ro[7], Ro[8], t[2], 0, 0, 0, 1};
  if (final_cost) *final_cost = summary.final_cost;
  if (successful_steps) *successful_steps = summary.num_successful_steps;
  if (residual_blocks) *residual_blocks = (int)blocks;
  return 1;
}

int CameraLidarOptimizer::Optimize(const LinePairs& line_pairs, std::vector<PointTrack>& structure, const bool refine_camera_rotation,
                                   const bool refine_camera_trans, const bool refine_lidar_rotation, const bool refine_lidar_trans,
                                   const bool refine_structure, double& cost, int& steps) {
  std::vector<Vector3d> aa_cw(frames.size(), Vector3d{0, 0, 0}), t_cw(frames.size(), Vector3d{0, 0, 0});
  std::vector<Vector3d> aa_lw(lidars.size(), Vector3d{0, 0, 0}), t_lw(lidars.size(), Vector3d{0, 0, 0});
  std::vector<bool> frame_valid(frames.size());
  for (size_t i = 0; i < frames.size(); i++) {
    frame_valid[i] = frames[i].IsPoseValid();
    if (!frame_valid[i]) continue;
    const Matrix3d& R = frames[i].R_wc;
    const Matrix3d R_cw = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    RotationMatrixToAngleAxis(R_cw, &aa_cw[i]);
    const Vector3d rt = MatVec(R_cw, frames[i].t_wc);
    t_cw[i] = {-rt[0], -rt[1], -rt[2]};
  }
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    const Matrix3d& R = lidars[i].GetRotation();
    const Matrix3d R_lw = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    RotationMatrixToAngleAxis(R_lw, &aa_lw[i]);
    const Vector3d rt = MatVec(R_lw, lidars[i].GetTranslation());
    t_lw[i] = {-rt[0], -rt[1], -rt[2]};
  }
  {   // lidars[i].Transform2LidarWorld() of every posed scan (CameraLidarOptimizer.cpp:409-417), the resident copies in one device call
    std::vector<Velodyne*> posed;
    for (Velodyne& l : lidars) if (l.IsPoseValid() && l.valid) posed.push_back(&l);
    Velodyne::TransformBatch(posed, true, config.num_threads);
  }
  ceres_like::Problem problem;
  ceres_like::LossFunction* loss1 = new ceres_like::HuberLoss(3 * M_PI / 180.0);
  const size_t n_cl = AddCameraLidarResidual(frames.empty() ? 0 : frames[0].rows, frames.empty() ? 0 : frames[0].cols, frame_valid, lidars, aa_cw, t_cw,
                                             aa_lw, t_lw, line_pairs, loss1, problem, config.camera_lidar_weight);
  if (n_cl == 0) delete loss1;
  // 3. camera-camera: reprojection of the triangulated tracks (CameraLidarOptimizer.cpp:431-432)
  if (!structure.empty()) AddCameraResidual(frames, aa_cw, t_cw, structure, problem, RESIDUAL_TYPE::ANGLE_RESIDUAL_1, config.camera_weight);
  const std::vector<std::vector<int>> neighbors = FindNeighbors(lidars, 6);
  if (config.line_to_line_residual) {
    LidarLineMatch matcher(lidars);
    matcher.SetNeighborSize(4);
    matcher.SetMinTrackLength(3);
    matcher.GenerateTracks();
    AddLidarLineToLineResidual2(neighbors, lidars, aa_lw, t_lw, problem, matcher.GetTracks(), config.point_to_line_dis_threshold, config.angle_residual,
                                config.normalize_distance);   // lidar_weight is NOT passed here (CameraLidarOptimizer.cpp:452-453)
  }
  if (config.point_to_plane_residual)
    AddLidarPointToPlaneResidual(neighbors, lidars, aa_lw, t_lw, problem, config.point_to_plane_dis_threshold, config.lidar_plane_tolerance,
                                 config.angle_residual, config.normalize_distance, config.lidar_weight);
  if (!refine_structure)                                                         // :462-466
    for (PointTrack& track : structure) problem.SetParameterBlockConstant(track.point_3d.data());
  for (size_t i = 0; i < frames.size(); i++)
    if (frame_valid[i]) {
      if (!refine_camera_rotation) problem.SetParameterBlockConstant(aa_cw[i].data());
      if (!refine_camera_trans) problem.SetParameterBlockConstant(t_cw[i].data());
    }
  for (size_t i = 0; i < lidars.size(); i++)
    if (lidars[i].IsPoseValid() && lidars[i].valid) {
      if (!refine_lidar_rotation) problem.SetParameterBlockConstant(aa_lw[i].data());
      if (!refine_lidar_trans) problem.SetParameterBlockConstant(t_lw[i].data());
    }
  if (!frames.empty()) { problem.SetParameterBlockConstant(aa_cw[0].data()); problem.SetParameterBlockConstant(t_cw[0].data()); }   // :490-491
  last_blocks_ = problem.NumResidualBlocks();
  ceres_like::Solver::Options options = SetOptionsSfM(config.num_threads);
  ceres_like::Solver::Summary summary;
  ceres_like::Solve(options, &problem, &summary);
  last_history_ = summary.cost_history;
  if (!summary.IsSolutionUsable()) return 0;
  for (size_t i = 0; i < frames.size(); i++) {
    if (!frame_valid[i]) continue;
    Matrix3d R_cw;
    AngleAxisToRotationMatrix(aa_cw[i], &R_cw);
    frames[i].R_wc = {R_cw[0], R_cw[3], R_cw[6], R_cw[1], R_cw[4], R_cw[7], R_cw[2], R_cw[5], R_cw[8]};
    const Vector3d rt = MatVec(frames[i].R_wc, t_cw[i]);
    frames[i].t_wc = {-rt[0], -rt[1], -rt[2]};
  }
  {
    std::vector<Velodyne*> posed;
    for (Velodyne& l : lidars) if (l.IsPoseValid() && l.IsInWorldCoordinate()) posed.push_back(&l);
    Velodyne::TransformBatch(posed, false, config.num_threads);
  }
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].IsPoseValid()) continue;
    Matrix3d R_lw;
    AngleAxisToRotationMatrix(aa_lw[i], &R_lw);
    const Matrix3d R_wl = {R_lw[0], R_lw[3], R_lw[6], R_lw[1], R_lw[4], R_lw[7], R_lw[2], R_lw[5], R_lw[8]};
    const Vector3d rt = MatVec(R_wl, t_lw[i]);
    lidars[i].SetPose(R_wl, {-rt[0], -rt[1], -rt[2]});
  }
  cost = summary.final_cost;
  steps = summary.num_successful_steps;
  return 1;
}

bool CameraLidarOptimizer::JointOptimize() {
  double last_cost = 0, curr_cost = 0;
  int last_step = INT32_MAX, curr_step = INT32_MAX;
  LinePairs pairs = AssociateLineMulti(neighbor_size_joint, true);
  for (int iter = 0; iter < num_iteration_joint; iter++) {
    size_t npairs = 0;
    for (auto& kv : pairs) npairs += kv.second.size();
    Optimize(pairs, structure, true, true, true, true, true, curr_cost, curr_step);
    log.push_back({curr_cost, curr_step, last_blocks_, npairs, last_history_});
    pairs.clear();
    pairs = AssociateLineMulti(neighbor_size_joint, true);
    if (std::fabs(curr_cost - last_cost) / last_cost < 0.01) break;
    if (curr_step < 5 && last_step < 5) break;
    last_cost = curr_cost;
    last_step = curr_step;
  }
  return true;
}

// ================================================================================================
// MVS::FuseDepthImages — mvs/MVS.cpp:2168-2334 (ConfToWeight :2337-2340, BGR2HSV util/Visualization.cpp:57-77)
// ================================================================================================
namespace {
struct FuseState {
  int rows, cols;
  std::vector<DepthFrame>& frames;
  std::vector<char> loaded;                       // depth_filter present (not released)
  std::vector<int> references;                    // frame_depth_filter_count
  std::vector<std::vector<uint16_t>> owner;       // `occupied`: 65535 = free
  std::vector<float> ray;                         // PreComputeI2C
  bool Has(size_t f) const { return loaded[f] != 0; }
  void Read(size_t f) {                           // ReadFrameDepth(<id>_geo|_pho.bin, frames[f], true)
    if (frames[f].depth_file.empty()) return;
    frames[f].depth_filter = frames[f].depth_file;
    loaded[f] = 1;
  }
  void Drop(size_t f) { if (--references[f] <= 0) { frames[f].depth_filter.clear(); loaded[f] = 0; } }
};
inline float WeightOfConf(float conf, float depth) { return 1.f / (std::max(1.f - conf, 0.03f) * (depth * depth)); }
inline void ToWorld(const float* p, const Matrix4d& T, float* o) {          // TranslatePoint<float, double>, base/Geometry.hpp:545-551
  for (int k = 0; k < 3; ++k) o[k] = (float)(p[0] * T[4 * k] + p[1] * T[4 * k + 1] + p[2] * T[4 * k + 2] + T[4 * k + 3]);
}
inline bool SkyBlue(const float* bgr_f) {                                    // on cv::Vec3b(color): saturate_cast = clamp(cvRound)
  unsigned char c[3];
  for (int k = 0; k < 3; ++k) { const long v = std::lrint(bgr_f[k]); c[k] = (unsigned char)std::min(255l, std::max(0l, v)); }
  const float r = c[2] / 255.f, g = c[1] / 255.f, b = c[0] / 255.f;
  const float hi = std::max(r, std::max(g, b)), lo = std::min(r, std::min(g, b));
  float h = 0, s = 0, v = 0;
  if (hi != 0) {
    const float d = hi - lo;
    if (hi == r) h = 60.f * ((g - b) / d + 6 * (g < b));
    else if (hi == g) h = 60.f * ((b - r) / d + 2);
    else h = 60.f * ((r - g) / d + 4);
    h = h / 360.f; s = d / hi; v = hi;
  }
  h *= 180.f; s *= 255.f; v *= 255.f;
  return h >= 100 && h <= 124 && s >= 43 && s <= 200 && v >= 150 && v <= 255;
}
}  // namespace

std::vector<PointXYZRGB> FuseDepthImages(int rows, int cols, std::vector<DepthFrame>& frames, const std::vector<std::vector<NeighborInfo>>& neighbors, float max_depth,
                                         float depth_diff_threshold) {
  const size_t n = frames.size(), npix = (size_t)rows * cols;
  if (neighbors.size() != n) throw std::invalid_argument("FuseDepthImages: one neighbour list per frame");
  for (const DepthFrame& f : frames)
    if ((!f.depth_filter.empty() && f.depth_filter.size() != npix) || (!f.depth_file.empty() && f.depth_file.size() != npix) || f.conf.size() != npix || f.bgr.size() != 3 * npix)
      throw std::invalid_argument("FuseDepthImages: map sizes");
  for (const auto& l : neighbors) for (const NeighborInfo& x : l) if (x.id >= n) throw std::invalid_argument("FuseDepthImages: neighbour id");
  FuseState S{rows, cols, frames, std::vector<char>(n, 0), std::vector<int>(n, 0), {}, std::vector<float>(3 * npix)};
  S.owner.assign(n, std::vector<uint16_t>(npix, UINT16_MAX));
  const Equirect eq{cols, rows};
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) { const float px[2] = {(float)x, (float)y}; eq.ImageToCam(px, 1.f, &S.ray[3 * ((size_t)y * cols + x)]); }
  std::vector<std::pair<int, int>> order;                                       // idx_connections (:2181-2189)
  for (size_t i = 0; i < n; ++i) { S.loaded[i] = !frames[i].depth_filter.empty(); S.references[i] = (int)neighbors[i].size() + 1; order.push_back({(int)i, (int)neighbors[i].size()}); }
  std::sort(order.begin(), order.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second > b.second; });
  struct Claim { size_t frame; int x, y; };
  std::vector<Claim> agreed, in_front;                                          // view_project, invalid_depth: declared outside the frame loop upstream
  std::vector<PointXYZRGB> cloud;
  for (const std::pair<int, int>& oc : order) {
    const size_t ref = (size_t)oc.first;
    const std::vector<NeighborInfo>& nb = neighbors[ref];
    if (S.Has(ref)) {                                                           // :2205-2206 — before the read loop
      for (const NeighborInfo& x : nb) if (!S.Has(x.id)) S.Read(x.id);          // :2209-2215
      DepthFrame& F = frames[ref];
      for (size_t e = 0; e < npix; ++e) {
        const float depth = F.depth_filter[e];
        if (depth <= 0 || depth >= max_depth * 0.8) continue;
        uint16_t& mine = S.owner[ref][e];
        if (mine != UINT16_MAX) continue;
        mine = (uint16_t)F.id;
        float weight = WeightOfConf(F.conf[e], depth);
        const float X0[3] = {S.ray[3 * e] * depth, S.ray[3 * e + 1] * depth, S.ray[3 * e + 2] * depth};
        float X[3], colour[3];
        ToWorld(X0, F.T_wc, X);
        for (int k = 0; k < 3; ++k) { X[k] = X[k] * weight; colour[k] = (float)F.bgr[3 * e + k] * weight; }
        for (const NeighborInfo& x : nb) {
          if (!S.Has(x.id)) continue;
          DepthFrame& N = frames[x.id];
          float X1[3], uv[2];
          for (int r = 0; r < 3; ++r) { float acc = 0; for (int c = 0; c < 3; ++c) acc += x.R_nr[3 * r + c] * X0[c]; X1[r] = acc + x.t_nr[r]; }
          eq.CamToImage(X1, uv);
          const int u = (int)std::round(uv[0]), v = (int)std::round(uv[1]);
          if (u < 0 || v < 0 || u >= cols || v >= rows) continue;
          const size_t ne = (size_t)v * cols + u;
          const float n_depth = N.depth_filter[ne];
          if (n_depth <= 0) continue;
          uint16_t& theirs = S.owner[x.id][ne];
          if (theirs != UINT16_MAX) continue;
          if (std::abs((depth - n_depth) / depth) < depth_diff_threshold) {
            agreed.push_back({x.id, u, v});
            const float w = WeightOfConf(N.conf[ne], n_depth);
            const float P[3] = {S.ray[3 * ne] * n_depth, S.ray[3 * ne + 1] * n_depth, S.ray[3 * ne + 2] * n_depth};
            float Pw[3];
            ToWorld(P, N.T_wc, Pw);
            for (int k = 0; k < 3; ++k) { X[k] += Pw[k] * w; colour[k] += (float)N.bgr[3 * ne + k] * w; }
            weight += w;
            theirs = mine;
          }
          if (std::sqrt((double)X1[0] * X1[0] + (double)X1[1] * X1[1] + (double)X1[2] * X1[2]) < n_depth) in_front.push_back({x.id, u, v});
        }
        if (agreed.size() < 2) {
          for (const Claim& c : agreed) S.owner[c.frame][(size_t)c.y * cols + c.x] = UINT16_MAX;
          mine = UINT16_MAX;
        } else {
          const float inv = 1.f / weight;
          PointXYZRGB p;
          p.x = X[0] * inv; p.y = X[1] * inv; p.z = X[2] * inv;
          for (int k = 0; k < 3; ++k) colour[k] = colour[k] * inv;
          p.r = (unsigned char)colour[2]; p.g = (unsigned char)colour[1]; p.b = (unsigned char)colour[0];
          for (const Claim& c : in_front) if (S.Has(c.frame)) frames[c.frame].depth_filter[(size_t)c.y * cols + c.x] = 0;
          if (SkyBlue(colour)) continue;                                        // :2316 — leaves both lists filled for the next pixel
          cloud.push_back(p);
        }
        in_front.clear();
        agreed.clear();
      }
    }
    S.Drop(ref);                                                                // next_image (:2322-2331)
    for (const NeighborInfo& x : nb) S.Drop(x.id);
  }
  return cloud;
}


}  // namespace pvlm
