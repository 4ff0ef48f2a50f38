// pvlm_host.cpp — implementation of the C++ host mirror (see pvlm_host.hpp for the reference
// interfaces each piece stands in for).  Host logic only (neighbour lists, vote post-processing,
// tracks, problem bookkeeping, the trust-region loop, the small linear solves); every residual,
// Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host_internal.hpp"

namespace pvlm {

std::map<std::string, double>& Stages() { static std::map<std::string, double> m; return m; }
std::map<std::string, long>& StageCallCounts() { static std::map<std::string, long> m; return m; }
void AddStageSeconds(const char* name, double seconds) { Stages()[name] += seconds; ++StageCallCounts()[name]; }
const std::map<std::string, double>& StageSeconds() { return Stages(); }
const std::map<std::string, long>& StageCalls() { return StageCallCounts(); }

// ================================================================================================
// Engine
// ================================================================================================
static int g_device = 0;
void Engine::SetDevice(int device) { g_device = device; }
Engine::Engine(int device) {
  pvlm_status st = pvlm_create(device, &ctx_);
  if (st != PVLM_OK) throw std::runtime_error("pvlm_create failed (" + std::to_string((int)st) + "): no usable MI355X / HIP device; there is no CPU fallback");
  // PVLM_HOST_RESERVE_MB: size the engine's device pool once, at start-up (hipMalloc of fresh memory costs 40-70 ms per GB
  // on this driver — paid here instead of inside the first association / solve of the process)
  if (const char* mb = std::getenv("PVLM_HOST_RESERVE_MB"))
    if (std::atof(mb) > 0) Check(pvlm_reserve(ctx_, (size_t)(std::atof(mb) * 1048576.0)), "pvlm_reserve");
  // PVLM_HOST_RESERVE_STAGING_MB: likewise the pinned staging window of the scan uploads (hipHostMalloc + first touch of its 64 MB: 25 ms inside the first
  // association of the process otherwise)
  if (const char* mb = std::getenv("PVLM_HOST_RESERVE_STAGING_MB"))
    if (std::atof(mb) > 0) Check(pvlm_reserve_staging(ctx_, (int64_t)(std::atof(mb) * 1048576.0)), "pvlm_reserve_staging");
  // PVLM_HOST_PRELOAD=1: likewise the code objects of the library's kernels (HIP loads one at the first launch of a kernel of its file: ~60 ms spread over the first
  // EstimatePose of the process otherwise)
  if (const char* on = std::getenv("PVLM_HOST_PRELOAD"))
    if (std::atoi(on) > 0) Check(pvlm_preload(ctx_), "pvlm_preload");
}
Engine::~Engine() { if (ctx_) pvlm_destroy(ctx_); }
Engine& Engine::Default() {
  static Engine e(g_device);
  return e;
}
void Engine::Check(pvlm_status st, const char* what) const {
  if (st != PVLM_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string((int)st) + "): " + pvlm_last_error(ctx_));
}

// ================================================================================================
// small dense helpers (row-major 3x3 / 4x4)
// ================================================================================================
// ceres::RotationMatrixToAngleAxis / AngleAxisToRotationMatrix ([recalled] Ceres 2.0.0 rotation.h):
// the callers hand Eigen column-major data; here R is row-major, element (r,c) = R[3r+c].
void RotationMatrixToAngleAxis(const Matrix3d& R, Vector3d* aa) {
  auto M = [&](int r, int c) { return R[3 * r + c]; };
  double q[4];
  const double trace = M(0, 0) + M(1, 1) + M(2, 2);
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (M(2, 1) - M(1, 2)) * t; q[2] = (M(0, 2) - M(2, 0)) * t; q[3] = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
    q[i + 1] = 0.5 * t; t = 0.5 / t;
    q[0] = (M(k, j) - M(j, k)) * t; q[j + 1] = (M(j, i) + M(i, j)) * t; q[k + 1] = (M(k, i) + M(i, k)) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double s = std::sqrt(s2);
    const double two_theta = 2.0 * ((q[0] < 0.0) ? std::atan2(-s, -q[0]) : std::atan2(s, q[0]));
    const double k = two_theta / s;
    *aa = {q[1] * k, q[2] * k, q[3] * k};
  } else {
    *aa = {q[1] * 2.0, q[2] * 2.0, q[3] * 2.0};
  }
}
void AngleAxisToRotationMatrix(const Vector3d& a, Matrix3d* Rout) {
  Matrix3d& R = *Rout;
  const double th2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (th2 > std::numeric_limits<double>::epsilon()) {
    const double th = std::sqrt(th2), wx = a[0] / th, wy = a[1] / th, wz = a[2] / th, c = std::cos(th), s = std::sin(th), k = 1.0 - c;
    R = {c + wx * wx * k, wx * wy * k - wz * s, wy * s + wx * wz * k, wz * s + wx * wy * k, c + wy * wy * k, -wx * s + wy * wz * k,
         -wy * s + wx * wz * k, wx * s + wy * wz * k, c + wz * wz * k};
  } else {
    R = {1, -a[2], a[1], a[2], 1, -a[0], -a[1], a[0], 1};
  }
}

// ================================================================================================
// Velodyne
// ================================================================================================
Matrix4d Velodyne::GetPose() const {
  return {R_wl_[0], R_wl_[1], R_wl_[2], t_wl_[0], R_wl_[3], R_wl_[4], R_wl_[5], t_wl_[1], R_wl_[6], R_wl_[7], R_wl_[8], t_wl_[2], 0, 0, 0, 1};
}
bool Velodyne::IsPoseValid() const {
  for (int k = 0; k < 3; ++k) if (std::isinf(t_wl_[k]) || std::isnan(t_wl_[k])) return false;
  for (int k = 0; k < 9; ++k) if (std::fabs(R_wl_[k]) > 1e-12) return true;  // !R_wl.isZero()
  return false;
}
Vector3d Velodyne::World2Local(const Vector3d& p) const {
  const Vector3d a = MatTVec(R_wl_, p), b = MatTVec(R_wl_, t_wl_);
  return {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
}
Vector3d Velodyne::Local2World(const Vector3d& p) const {
  const Vector3d a = MatVec(R_wl_, p);
  return {a[0] + t_wl_[0], a[1] + t_wl_[1], a[2] + t_wl_[2]};
}
// pcl::transformPointCloud(cloud, cloud, Matrix4d): per coordinate float(m0*x + m1*y + m2*z + m3)
static void TransformCloud(PointCloud& c, const Matrix3d& R, const Vector3d& t) {
  for (PointXYZI& p : c) {
    const double x = p.x, y = p.y, z = p.z;
    p.x = static_cast<float>(R[0] * x + R[1] * y + R[2] * z + t[0]);
    p.y = static_cast<float>(R[3] * x + R[4] * y + R[5] * z + t[1]);
    p.z = static_cast<float>(R[6] * x + R[7] * y + R[8] * z + t[2]);
  }
}
// the matrix of Transform2LidarWorld (T_wl) or Transform2Local (T_lw = [R_wl^T | -R_wl^T t_wl], sensors/Velodyne.cpp:1822-1826): host and device
// copies of a scan are moved by the SAME twelve doubles
static void TransformOf(const Matrix3d& R_wl, const Vector3d& t_wl, bool to_world, Matrix3d& R, Vector3d& t) {
  R = R_wl; t = t_wl;
  if (!to_world) {
    R = {R_wl[0], R_wl[3], R_wl[6], R_wl[1], R_wl[4], R_wl[7], R_wl[2], R_wl[5], R_wl[8]};
    const Vector3d rt = MatVec(R, t_wl);
    t = {-rt[0], -rt[1], -rt[2]};
  }
}
// the host half of Transform2LidarWorld / Transform2Local: no device call, so it may run on a worker thread
static void TransformScanClouds(bool to_world, PointCloud& surfFlat, PointCloud& surfLessFlat, PointCloud& cornerLessSharp, PointCloud& cloud,
                                std::vector<PointCloud>& edge_segmented, const Matrix3d& R_wl, const Vector3d& t_wl) {
  Matrix3d R; Vector3d t;
  TransformOf(R_wl, t_wl, to_world, R, t);
  TransformCloud(surfFlat, R, t); TransformCloud(surfLessFlat, R, t); TransformCloud(cornerLessSharp, R, t);
  TransformCloud(cloud, R, t);
  for (PointCloud& s : edge_segmented) TransformCloud(s, R, t);
}
static bool ReuploadMode() { static const bool on = std::getenv("PVLM_HOST_REUPLOAD") != nullptr; return on; }
void Velodyne::Transform2LidarWorld() {
  if (world_ || !IsPoseValid()) return;
  TransformBatch({this}, true, 1);
}
void Velodyne::Transform2Local() {
  if (!world_ || !IsPoseValid()) return;
  TransformBatch({this}, false, 1);
}
void Velodyne::TransformBatch(const std::vector<Velodyne*>& scans, bool to_world, int num_threads) {
  StageTimer stage_timer_(to_world ? "scan clouds to the world frame (host clouds scan-parallel; resident device copies re-posed in place, K26)"
                                   : "scan clouds back to the local frame (host clouds scan-parallel; resident device copies re-posed in place, K26)");
  std::vector<Velodyne*> todo;
  for (Velodyne* v : scans) if (v && v->IsPoseValid() && v->world_ != to_world) todo.push_back(v);
  if (todo.empty()) return;
  // the resident copies (calling thread: the engine's pool is not thread-safe): queued before the host workers start, collected after them
  std::vector<pvlm_scan*> resident; std::vector<double> T12;
  if (ReuploadMode()) { for (Velodyne* v : todo) v->InvalidateDevice(); }
  else
    for (Velodyne* v : todo) {
      if (!v->dev_) continue;
      Matrix3d R; Vector3d t;
      TransformOf(v->R_wl_, v->t_wl_, to_world, R, t);
      resident.push_back(v->dev_);
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T12.push_back(R[3 * r + c]); T12.push_back(t[r]); }
    }
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::max(num_threads, 1), todo.size() / 16 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t k = next++; k < todo.size(); k = next++) {
      Velodyne& v = *todo[k];
      TransformScanClouds(to_world, v.surfFlat, v.surfLessFlat, v.cornerLessSharp, v.cloud, v.edge_segmented, v.R_wl_, v.t_wl_);
      v.world_ = to_world;
    }
  };
  if (resident.empty()) { pvlm_run_workers(n_threads, work); return; }
  // the device call is host-side work too (descriptor tables, the grid plans between its two synchronisations): it runs on the calling thread
  // while the workers move the host's clouds
  std::exception_ptr failure;
  std::thread host_side([&]() { try { pvlm_run_workers(n_threads, work); } catch (...) { failure = std::current_exception(); } });
  pvlm_status st;
  {
    StageTimer stage_timer_dev_("  (inside the re-posing) pvlm_scan_transform_batch");
    Engine& e = Engine::Default();
    st = pvlm_scan_transform_batch(e.ctx(), (int)resident.size(), resident.data(), T12.data(), to_world ? 1 : 0);
  }
  host_side.join();
  if (failure) std::rethrow_exception(failure);
  Engine::Default().Check(st, "pvlm_scan_transform_batch");
}
void Velodyne::InvalidateDevice() const {
  if (dev_) { pvlm_scan_destroy(Engine::Default().ctx(), dev_); dev_ = nullptr; }
}
void Velodyne::PoseChanged() const {
  if (!dev_) return;
  if (ReuploadMode()) { InvalidateDevice(); return; }
  Engine& e = Engine::Default();
  e.Check(pvlm_scan_set_pose(e.ctx(), dev_, R_wl_.data(), t_wl_.data()), "pvlm_scan_set_pose");
}
Velodyne::~Velodyne() { if (dev_) pvlm_scan_destroy(Engine::Default().ctx(), dev_); }
Velodyne::Velodyne(const Velodyne& o)
    : id(o.id), valid(o.valid), name(o.name), cloud(o.cloud), cornerLessSharp(o.cornerLessSharp), surfFlat(o.surfFlat), surfLessFlat(o.surfLessFlat),
      N_SCANS(o.N_SCANS), horizon_scans(o.horizon_scans), cloud_scan(o.cloud_scan), cornerSharp(o.cornerSharp), cornerBeforeFilter(o.cornerBeforeFilter),
      edge_segmented(o.edge_segmented), point_to_segment(o.point_to_segment), segment_coeffs(o.segment_coeffs), end_points(o.end_points),
      R_wl_(o.R_wl_), t_wl_(o.t_wl_), world_(o.world_), layout_(o.layout_), dev_(nullptr) {}
Velodyne& Velodyne::operator=(const Velodyne& o) {
  if (this == &o) return *this;
  InvalidateDevice();
  id = o.id; valid = o.valid; name = o.name; cloud = o.cloud; cornerLessSharp = o.cornerLessSharp; surfFlat = o.surfFlat; surfLessFlat = o.surfLessFlat;
  edge_segmented = o.edge_segmented; point_to_segment = o.point_to_segment; segment_coeffs = o.segment_coeffs; end_points = o.end_points;
  N_SCANS = o.N_SCANS; horizon_scans = o.horizon_scans; cloud_scan = o.cloud_scan; cornerSharp = o.cornerSharp; cornerBeforeFilter = o.cornerBeforeFilter; layout_ = o.layout_;
  R_wl_ = o.R_wl_; t_wl_ = o.t_wl_; world_ = o.world_;
  return *this;
}
// Host-side flattening of one scan's members into the arrays a pvlm_scan_desc points at.
namespace {
struct ScanStaging {
  std::vector<float> seg_xyz;
  std::vector<int> off, ids, seg_size;
  std::vector<double> coeffs, ends;
  pvlm_scan_desc d;
  void Fill(const Velodyne& v, const Matrix3d& R_wl, const Vector3d& t_wl) {
    // the three clouds are handed over where they lie: point_stride_floats = 4 lets the upload gather x, y, z and the intensity out of the PointXYZI records
    // itself (round 5 flattened them into six scratch vectors first: 28 ms of first-touch page faults for the 1593 scans of Floor)
    static_assert(sizeof(PointXYZI) == 4 * sizeof(float), "PointXYZI is handed to pvlm_scan_upload with point_stride_floats = 4");
    off.assign(v.cornerLessSharp.size() + 1, 0); ids.clear(); seg_size.resize(v.edge_segmented.size());
    for (size_t i = 0; i < v.cornerLessSharp.size(); ++i) {
      if (i < v.point_to_segment.size()) for (int s : v.point_to_segment[i]) ids.push_back(s);
      off[i + 1] = (int)ids.size();
    }
    for (size_t s = 0; s < v.edge_segmented.size(); ++s) seg_size[s] = (int)v.edge_segmented[s].size();
    coeffs.resize(v.segment_coeffs.size() * 6); ends.assign(v.edge_segmented.size() * 6, 0.0);
    for (size_t s = 0; s < v.segment_coeffs.size(); ++s) std::memcpy(&coeffs[6 * s], v.segment_coeffs[s].data(), 48);
    for (size_t s = 0; s < v.edge_segmented.size() && 2 * s + 1 < v.end_points.size(); ++s) {
      std::memcpy(&ends[6 * s], v.end_points[2 * s].data(), 24); std::memcpy(&ends[6 * s + 3], v.end_points[2 * s + 1].data(), 24);
    }
    if (ids.empty()) ids.push_back(0);
    std::memset(&d, 0, sizeof(d));
    d.id = v.id; d.R_wl = R_wl.data(); d.t_wl = t_wl.data();
    d.point_stride_floats = 4;
    d.n_surf_flat = (int)v.surfFlat.size(); if (d.n_surf_flat) { d.surf_flat_xyz = &v.surfFlat[0].x; d.surf_flat_tag = &v.surfFlat[0].intensity; }
    d.n_surf_less_flat = (int)v.surfLessFlat.size(); if (d.n_surf_less_flat) { d.surf_less_flat_xyz = &v.surfLessFlat[0].x; d.surf_less_flat_tag = &v.surfLessFlat[0].intensity; }
    d.n_corner = (int)v.cornerLessSharp.size(); if (d.n_corner) d.corner_xyz = &v.cornerLessSharp[0].x;
    d.p2s_offsets = off.data(); d.p2s_ids = ids.data();
    d.n_segments = (int)std::min(v.edge_segmented.size(), v.segment_coeffs.size()); d.segment_size = seg_size.data();
    d.segment_coeffs = coeffs.data(); d.end_points = ends.data();
    // the segments' own point lists (edge_segmented), for the device-built line-to-line blocks
    seg_xyz.clear();
    for (int k = 0; k < d.n_segments; ++k) for (const PointXYZI& p : v.edge_segmented[(size_t)k]) { seg_xyz.push_back(p.x); seg_xyz.push_back(p.y); seg_xyz.push_back(p.z); }
    if (seg_xyz.empty()) seg_xyz.push_back(0.f);
    d.seg_points_xyz = seg_xyz.data();
  }
};
}  // namespace

pvlm_scan* Velodyne::DeviceScan() const {
  if (dev_) return dev_;
  StageTimer stage_timer_("  (inside the stages below) scan upload: host SoA staging + pvlm_scan_upload");
  ScanStaging st;
  st.Fill(*this, R_wl_, t_wl_);
  Engine& e = Engine::Default();
  e.Check(pvlm_scan_upload(e.ctx(), &st.d, &dev_), "pvlm_scan_upload");
  return dev_;
}

// Every scan of `scans` that has no device mirror yet, in ONE pvlm_scan_upload_batch (one staging copy, one slab, one
// grid build): what the adders call before they walk their (scan, neighbour) pairs — at Room scale all 454 scans are
// re-posed, hence re-uploaded, at every outer iteration of EstimatePose.
void Velodyne::UploadBatch(const std::vector<const Velodyne*>& scans) {
  std::vector<const Velodyne*> todo;
  {
    std::set<const Velodyne*> seen;
    for (const Velodyne* v : scans) if (v && !v->dev_ && seen.insert(v).second) todo.push_back(v);
  }
  if (todo.empty()) return;
  if (todo.size() == 1 || std::getenv("PVLM_HOST_NO_BATCH")) { for (const Velodyne* v : todo) v->DeviceScan(); return; }
  StageTimer stage_timer_("  (inside the stages below) scan upload: host SoA staging + pvlm_scan_upload");
  // the staging arrays outlive the call: the same scans come back at every outer iteration of EstimatePose, and filling vectors that keep
  // their capacity costs a third of filling fresh ones (no allocation, no first-touch page faults: 27 -> 9 ms for the 1593 scans of Floor)
  // (one set per calling thread: two LidarOdometry objects on two threads must not share it; released with the thread)
  static thread_local std::vector<ScanStaging> st_of_this_thread;
  std::vector<ScanStaging>& st = st_of_this_thread;      // the workers below fill the CALLING thread's set: a thread_local named inside their lambda would be their own (empty) one
  if (st.size() < todo.size()) st.resize(todo.size());
  std::vector<pvlm_scan_desc> descs(todo.size());
  {   // the flattening is per scan and independent: scan-parallel, like FindNeighbors
    StageTimer stage_timer_fill_("  (inside the scan upload) host SoA staging of the scans (Fill)");
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({pvlm_thread_cap(), todo.size() / 32 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t k = next++; k < todo.size(); k = next++) { st[k].Fill(*todo[k], todo[k]->R_wl_, todo[k]->t_wl_); descs[k] = st[k].d; } };
    pvlm_run_workers(n_threads, work);
  }
  std::vector<pvlm_scan*> out(todo.size(), nullptr);
  Engine& e = Engine::Default();
  StageTimer stage_timer_abi_("  (inside the scan upload) pvlm_scan_upload_batch");
  e.Check(pvlm_scan_upload_batch(e.ctx(), (int)todo.size(), descs.data(), out.data()), "pvlm_scan_upload_batch");
  for (size_t k = 0; k < todo.size(); ++k) todo[k]->dev_ = out[k];
}


// ---- motion compensation (sensors/Velodyne.cpp:1635-1674) ------------------------------------------------------------------------------------
void Velodyne::UndistortBatch(const std::vector<Velodyne*>& scans, const std::vector<Matrix4d>& T_we) {
  if (scans.size() != T_we.size()) throw std::invalid_argument("UndistortBatch: one end pose per scan");
  std::vector<pvlm_undistort_scan> descs;
  std::vector<std::array<double, 24>> poses;          // R_wl, t_wl, R_we, t_we per scan (the descriptors point into it)
  poses.reserve(scans.size());
  std::vector<Velodyne*> done;
  for (size_t k = 0; k < scans.size(); ++k) {
    Velodyne* v = scans[k];
    if (!v || !v->IsPoseValid()) continue;                                   // :1644-1645
    if (v->cloud.empty()) v->LoadLidar(v->name);                             // :1650-1651
    if (v->cloud.empty()) continue;
    std::array<double, 24> p{};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) { p[3 * r + c] = v->R_wl_[3 * r + c]; p[12 + 3 * r + c] = T_we[k][4 * r + c]; }
      p[9 + r] = v->t_wl_[r]; p[21 + r] = T_we[k][4 * r + 3];
    }
    poses.push_back(p);
    done.push_back(v);
  }
  if (done.empty()) return;
  descs.resize(done.size());
  for (size_t k = 0; k < done.size(); ++k)
    descs[k] = pvlm_undistort_scan{&done[k]->cloud[0].x, (int)done[k]->cloud.size(), (int)(sizeof(PointXYZI) / sizeof(float)), &poses[k][0], &poses[k][9], &poses[k][12], &poses[k][21]};
  Engine& e = Engine::Default();
  e.Check(pvlm_undistort_batch(e.ctx(), (int)descs.size(), descs.data()), "pvlm_undistort_batch");
  for (Velodyne* v : done) {                                                 // :1663-1667
    v->cloud_scan.clear(); v->cornerLessSharp.clear(); v->cornerSharp.clear(); v->surfFlat.clear(); v->surfLessFlat.clear();
    v->InvalidateDevice();
  }
}

void Velodyne::Reset() {
  cloud_scan.clear(); cornerLessSharp.clear(); cornerSharp.clear(); surfLessFlat.clear(); surfFlat.clear();
  edge_segmented.clear(); point_to_segment.clear(); segment_coeffs.clear(); end_points.clear(); cornerBeforeFilter.clear();
  layout_ = RingLayout();
  InvalidateDevice();
}

bool Velodyne::UndistortCloud(const Matrix4d& T_we) {
  if (!IsPoseValid()) return false;
  if (cloud.empty()) LoadLidar(name);
  if (cloud.empty()) return false;
  UndistortBatch({this}, {T_we});
  return true;
}

}  // namespace pvlm
