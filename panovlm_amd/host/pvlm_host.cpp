// pvlm_host.cpp — implementation of the C++ host mirror (see pvlm_host.hpp for the reference
// interfaces each piece stands in for).  Host logic only (neighbour lists, vote post-processing,
// tracks, problem bookkeeping, the trust-region loop, the small linear solves); every residual,
// Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host.hpp"
#include "../csrc/pvlm_workers.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iomanip>
#include <climits>
#include <limits>
#include <mutex>
#include <sstream>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <unordered_map>

namespace pvlm {

namespace {
std::map<std::string, double>& Stages() { static std::map<std::string, double> m; return m; }
std::map<std::string, long>& StageCallCounts() { static std::map<std::string, long> m; return m; }
struct StageTimer {
  const char* name; std::chrono::steady_clock::time_point t0;
  explicit StageTimer(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~StageTimer() { Stages()[name] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++StageCallCounts()[name]; }
};
}  // namespace
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
static inline Vector3d MatVec(const Matrix3d& R, const Vector3d& p) {
  return {(R[0] * p[0] + R[1] * p[1]) + R[2] * p[2], (R[3] * p[0] + R[4] * p[1]) + R[5] * p[2], (R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]};
}
static inline Vector3d MatTVec(const Matrix3d& R, const Vector3d& p) {
  return {(R[0] * p[0] + R[3] * p[1]) + R[6] * p[2], (R[1] * p[0] + R[4] * p[1]) + R[7] * p[2], (R[2] * p[0] + R[5] * p[1]) + R[8] * p[2]};
}

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
// pose files — util/FileIO.cpp:11-79 (ReadPoseT), :168-191 (ExportPoseT)
// ================================================================================================
bool ReadPoseT(std::string file_path, bool with_invalid, std::vector<Matrix3d>& rotation_list, std::vector<Vector3d>& trans_list,
               std::vector<std::string>& name_list) {
  std::ifstream in(file_path);
  if (!in.is_open()) { fprintf(stderr, "Fail to open %s\n", file_path.c_str()); return false; }
  while (!in.eof()) {
    Matrix3d R = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Vector3d t = {INFINITY, INFINITY, INFINITY};
    std::string str;
    std::getline(in, str);
    std::vector<std::string> sub;
    { std::stringstream ss(str); std::string tmp; while (std::getline(ss, tmp, ' ')) sub.push_back(tmp); }  // SplitString(str, ' ')
    std::string curr_name;
    bool pose_valid = true;
    if (sub.size() == 13) { curr_name = sub[0]; sub.erase(sub.begin()); }
    if (sub.size() == 12) {
      for (const std::string& s : sub)
        if (s.find("inf") != std::string::npos || s.find("nan") != std::string::npos) { pose_valid = false; break; }
      if (pose_valid) {
        double v[12];
        for (int k = 0; k < 12; ++k) { std::stringstream ss(sub[k]); ss >> v[k]; }  // str2num<double>
        R = {v[0], v[1], v[2], v[4], v[5], v[6], v[8], v[9], v[10]};
        t = {v[3], v[7], v[11]};
      }
    }
    if (pose_valid || (!pose_valid && with_invalid)) { rotation_list.push_back(R); trans_list.push_back(t); name_list.push_back(curr_name); }
    if (in.peek() == EOF) break;
  }
  return true;
}

void ExportPoseT(const std::string file_path, const std::vector<Matrix3d>& rotation_list, const std::vector<Vector3d>& trans_list,
                 const std::vector<std::string>& name_list, int precision) {
  std::ofstream out(file_path);
  if (!out.is_open()) { fprintf(stderr, "Fail to write %s\n", file_path.c_str()); return; }
  out << std::setprecision(precision);
  for (size_t i = 0; i < rotation_list.size() && i < trans_list.size(); i++) {
    if (i < name_list.size()) out << name_list[i] << " ";
    const Matrix3d& R = rotation_list[i]; const Vector3d& t = trans_list[i];
    out << R[0] << " " << R[1] << " " << R[2] << " " << t[0] << " " << R[3] << " " << R[4] << " " << R[5] << " " << t[1] << " " << R[6] << " " << R[7]
        << " " << R[8] << " " << t[2] << std::endl;
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
// the host half of Transform2LidarWorld / Transform2Local: no device call, so it may run on a worker thread
static void TransformScanClouds(Velodyne& v, bool to_world, PointCloud& surfFlat, PointCloud& surfLessFlat, PointCloud& cornerLessSharp, PointCloud& cloud,
                                std::vector<PointCloud>& edge_segmented, const Matrix3d& R_wl, const Vector3d& t_wl) {
  (void)v;
  Matrix3d R = R_wl; Vector3d t = t_wl;
  if (!to_world) {
    R = {R_wl[0], R_wl[3], R_wl[6], R_wl[1], R_wl[4], R_wl[7], R_wl[2], R_wl[5], R_wl[8]};
    const Vector3d rt = MatVec(R, t_wl);
    t = {-rt[0], -rt[1], -rt[2]};
  }
  TransformCloud(surfFlat, R, t); TransformCloud(surfLessFlat, R, t); TransformCloud(cornerLessSharp, R, t);
  TransformCloud(cloud, R, t);
  for (PointCloud& s : edge_segmented) TransformCloud(s, R, t);
}
void Velodyne::Transform2LidarWorld() {
  if (world_ || !IsPoseValid()) return;
  TransformScanClouds(*this, true, surfFlat, surfLessFlat, cornerLessSharp, cloud, edge_segmented, R_wl_, t_wl_);
  world_ = true;
  InvalidateDevice();
}
void Velodyne::Transform2Local() {
  if (!world_ || !IsPoseValid()) return;
  TransformScanClouds(*this, false, surfFlat, surfLessFlat, cornerLessSharp, cloud, edge_segmented, R_wl_, t_wl_);
  world_ = false;
  InvalidateDevice();
}
void Velodyne::TransformBatch(const std::vector<Velodyne*>& scans, bool to_world, int num_threads) {
  StageTimer stage_timer_(to_world ? "scan clouds to the world frame (host, scan-parallel)" : "scan clouds back to the local frame (host, scan-parallel)");
  std::vector<Velodyne*> todo;
  for (Velodyne* v : scans) if (v && v->IsPoseValid() && v->world_ != to_world) todo.push_back(v);
  if (todo.empty()) return;
  for (Velodyne* v : todo) v->InvalidateDevice();           // calling thread
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::max(num_threads, 1), todo.size() / 16 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t k = next++; k < todo.size(); k = next++) {
      Velodyne& v = *todo[k];
      TransformScanClouds(v, to_world, v.surfFlat, v.surfLessFlat, v.cornerLessSharp, v.cloud, v.edge_segmented, v.R_wl_, v.t_wl_);
      v.world_ = to_world;
    }
  };
  pvlm_run_workers(n_threads, work);
}
void Velodyne::InvalidateDevice() const {
  if (dev_) { pvlm_scan_destroy(Engine::Default().ctx(), dev_); dev_ = nullptr; }
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
  std::vector<float> fx, ft, lx, lt, cx, seg_xyz;
  std::vector<int> off, ids, seg_size;
  std::vector<double> coeffs, ends;
  pvlm_scan_desc d;
  void Fill(const Velodyne& v, const Matrix3d& R_wl, const Vector3d& t_wl) {
    auto flat = [](const PointCloud& c, std::vector<float>& xyz, std::vector<float>* tag) {
      xyz.resize(c.size() * 3); if (tag) tag->resize(c.size());
      for (size_t i = 0; i < c.size(); ++i) { xyz[3 * i] = c[i].x; xyz[3 * i + 1] = c[i].y; xyz[3 * i + 2] = c[i].z; if (tag) (*tag)[i] = c[i].intensity; }
    };
    flat(v.surfFlat, fx, &ft); flat(v.surfLessFlat, lx, &lt); flat(v.cornerLessSharp, cx, nullptr);
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
    d.n_surf_flat = (int)v.surfFlat.size(); d.surf_flat_xyz = fx.data(); d.surf_flat_tag = ft.data();
    d.n_surf_less_flat = (int)v.surfLessFlat.size(); d.surf_less_flat_xyz = lx.data(); d.surf_less_flat_tag = lt.data();
    d.n_corner = (int)v.cornerLessSharp.size(); d.corner_xyz = cx.data(); d.p2s_offsets = off.data(); d.p2s_ids = ids.data();
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
  // every scan's list is independent of the others: scan-parallel (at Floor size — 1593 scans, all inside the 20 m radius of the
  // synthetic room — the serial loop was 0.1 s per call, four calls per EstimatePose)
  neighbors_all.assign(lidars.size(), std::vector<int>());
  auto one = [&](size_t i) {
    std::vector<int> neighbors;
    if (lidars[i].IsPoseValid()) {
      const Vector3d& t = lidars[i].GetTranslation();
      const float q[3] = {float(t[0]), float(t[1]), float(t[2])};
      std::vector<std::pair<float, int>> d(nc);
      for (int j = 0; j < nc; ++j) {
        const float dx = q[0] - center[j][0], dy = q[1] - center[j][1], dz = q[2] - center[j][2];
        float s = 0.0f; s += dx * dx; s += dy * dy; s += dz * dz;
        d[j] = {s, j};
      }
      // one sorted list serves both searches below: nearestKSearch (its first neighbor_size entries) and radiusSearch (its
      // prefix within 20 m, also ascending) — ties in pcl's order = position
      std::sort(d.begin(), d.end());
      for (int j = 0; j < std::min(neighbor_size, nc); ++j) neighbors.push_back(d[j].second);
      if (!neighbors.empty()) neighbors.erase(neighbors.begin());  // the first one is the scan itself
      for (int& n : neighbors) n = owner[n];
      std::set<int> nset(neighbors.begin(), neighbors.end());
      int ni = (int)i - 1;
      while (ni >= 0 && !lidars[ni].IsPoseValid()) ni--;
      if (ni >= 0 && nset.count(ni) == 0) neighbors.push_back(ni);
      ni = (int)i + 1;
      while (ni < (int)lidars.size() && !lidars[ni].IsPoseValid()) ni++;
      if (ni < (int)lidars.size() && nset.count(ni) == 0) neighbors.push_back(ni);
      const float r2 = float(20.0 * 20.0);  // radiusSearch(20 m): FLANN keeps dist < r^2
      const int loop_length = 200;
      for (int j = 0; j < nc && d[j].first < r2; ++j) {
        const int n_idx = owner[d[j].second];
        int same_loop = 0;
        for (int v : nset) {
          if (std::abs(n_idx - v) <= loop_length) same_loop++;
          if (same_loop >= 2) break;
        }
        if (same_loop < 2 && nset.count(n_idx) == 0) { neighbors.push_back(n_idx); nset.insert(n_idx); }
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
  e.Check(pvlm_assoc_point2plane(e.ctx(), 1, &r, &n, plane_tolerance, dist_threshold, PVLM_POINT2PLANE_METER, 0, 1.0, &rs), "pvlm_assoc_point2plane");
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
static inline double PlaneAngle(const double* a, const double* b) {  // base/Geometry.hpp:471-485
  double c = std::fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  c = c / (std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]));
  return c >= 1.0 ? 0.0 : std::acos(c);
}

static inline double PlaneAngleN(const double* a, const double* b) {  // PlaneAngle(..., normalized = true)
  const double c = std::fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  return c >= 1.0 ? 0.0 : std::acos(c);
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
  std::map<int, Line2Line> m;
  const int nr = (int)ref.edge_segmented.size(), nn = (int)nei.edge_segmented.size();
  for (int s = 0; s < nn && nr > 0; ++s) {
    const int max_col = best_col[s], max_count = best_count[s];
    if ((size_t)max_count < nei.edge_segmented[s].size() / 2) continue;
    if (PlaneAngle(&ref_world[max_col][3], &nei_world[s][3]) * 180.0 / M_PI > 7) continue;
    const Vector6d& loc = ref.segment_coeffs[max_col];
    Line2Line a;
    a.neighbor_line_idx = s; a.ref_line_idx = max_col;
    for (int c = 0; c < 3; ++c) { a.line_point1[c] = 0.1 * loc[3 + c] + loc[c]; a.line_point2[c] = -0.1 * loc[3 + c] + loc[c]; }
    auto it = m.find(max_col);
    if (it == m.end()) m.insert({max_col, a});
    else {
      const double d1 = PointToLineDistance3D(nei_world[it->second.neighbor_line_idx].data(), ref_world[max_col].data());
      const double d2 = PointToLineDistance3D(nei_world[s].data(), ref_world[max_col].data());
      if (d2 < d1) it->second = a;
    }
  }
  std::vector<Line2Line> out;
  for (auto& kv : m) out.push_back(kv.second);
  return out;
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
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
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
  std::map<const Velodyne*, std::vector<Vector6d>> world;      // TransformLines(segment_coeffs, pose): once per scan of the batch, not per pair
  for (size_t j = 0; j < which.size(); ++j)
    for (const Velodyne* v : {pairs[which[j]].first, pairs[which[j]].second})
      if (!world.count(v)) world.emplace(v, TransformLines(v->segment_coeffs, v->GetPose()));
  // the pairs are independent (read-only scans and row tables, one output slot each): pair-parallel
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>({pvlm_thread_cap(), which.size() / 256 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t j = next++; j < which.size(); j = next++) {
      const Velodyne& ref = *pairs[which[j]].first; const Velodyne& nei = *pairs[which[j]].second;
      out[which[j]] = FindAssociationsBest(ref, nei, world.find(&ref)->second, world.find(&nei)->second, best_col.data() + roff[j], best_count.data() + roff[j]);
    }
  };
  pvlm_run_workers(n_threads, work);
  return out;
}

// ================================================================================================
// LoadLidar — sensors/Velodyne.cpp:92-172 (+ the part of pcl::io::loadPCDFile a PointXYZI cloud needs)
// ================================================================================================
namespace {
// LZF decompression (the codec of PCD "binary_compressed"): control byte < 32 = literal run of ctrl + 1 bytes; otherwise
// a back reference of length (ctrl >> 5) + 2 (length 7 reads one extension byte) at offset ((ctrl & 31) << 8 | next) + 1.
bool LzfDecompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_len) return false;
      std::memcpy(out + op, in + ip, run); ip += run; op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
      if (ip >= in_len) return false;
      const size_t off = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
      len += 2;
      if (off > op || op + len > out_len) return false;
      for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - off];   // may overlap: byte by byte
    }
  }
  return op == out_len;
}

struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };

bool ReadPcd(const std::string& path, PointCloud& cloud) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<PcdField> fields;
  size_t points = 0, width = 0, height = 1;
  std::string data_mode, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string key; ls >> key;
    if (key == "FIELDS") { std::string n; while (ls >> n) { PcdField fd; fd.name = n; fields.push_back(fd); } }
    else if (key == "SIZE") { for (PcdField& fd : fields) ls >> fd.size; }
    else if (key == "TYPE") { for (PcdField& fd : fields) ls >> fd.type; }
    else if (key == "COUNT") { for (PcdField& fd : fields) ls >> fd.count; }
    else if (key == "WIDTH") ls >> width;
    else if (key == "HEIGHT") ls >> height;
    else if (key == "POINTS") ls >> points;
    else if (key == "DATA") { ls >> data_mode; break; }
  }
  if (fields.empty() || data_mode.empty()) return false;
  if (points == 0) points = width * height;
  size_t stride = 0;
  for (PcdField& fd : fields) { fd.offset = stride; stride += (size_t)fd.size * fd.count; }
  int ix = -1, iy = -1, iz = -1, ii = -1;
  for (size_t k = 0; k < fields.size(); ++k) {
    if (fields[k].name == "x") ix = (int)k; else if (fields[k].name == "y") iy = (int)k; else if (fields[k].name == "z") iz = (int)k;
    else if (fields[k].name == "intensity") ii = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0) return false;
  for (int k : {ix, iy, iz}) if (fields[k].type != 'F' || fields[k].size != 4) return false;   // PointXYZI: float32 coordinates
  for (const PcdField& fd : fields) if (fd.size <= 0 || fd.size > 8 || fd.count <= 0 || fd.count > 4096) return false;
  // a header must not make us allocate more than the file can hold (corrupt / hostile POINTS, WIDTH x HEIGHT)
  const std::streampos body = f.tellg();
  f.seekg(0, std::ios::end);
  const size_t remaining = body < 0 ? 0 : (size_t)(f.tellg() - body);
  f.seekg(body);
  if (stride == 0) return false;
  if (data_mode == "ascii" && points > remaining) return false;                         // >= 1 byte per point
  else if (data_mode == "binary" && points > remaining / stride) return false;
  else if (data_mode == "binary_compressed") {
    if (remaining < 8 || points > 0xffffffffull / stride) return false;                 // the block length is a uint32
    if (points * stride / 100 > remaining) return false;                                // LZF expands at most 264 bytes per 3
  } else if (data_mode != "ascii" && data_mode != "binary") return false;
  cloud.assign(points, PointXYZI{0, 0, 0, 0});
  auto as_float = [](const PcdField& fd, const unsigned char* p) -> float {
    if (fd.type == 'F' && fd.size == 4) { float v; std::memcpy(&v, p, 4); return v; }
    if (fd.type == 'F' && fd.size == 8) { double v; std::memcpy(&v, p, 8); return (float)v; }
    if (fd.type == 'U' && fd.size == 1) return (float)*p;
    if (fd.type == 'U' && fd.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return (float)v; }
    if (fd.type == 'U' && fd.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return (float)v; }
    if (fd.type == 'I' && fd.size == 4) { int32_t v; std::memcpy(&v, p, 4); return (float)v; }
    return 0.f;
  };
  if (data_mode == "ascii") {
    for (size_t i = 0; i < points; ++i) {
      if (!std::getline(f, line)) return false;
      std::istringstream ls(line);
      for (size_t k = 0; k < fields.size(); ++k)
        for (int c = 0; c < fields[k].count; ++c) {
          std::string tok; ls >> tok;
          if (c > 0) continue;
          float v = (tok == "nan" || tok == "-nan" || tok == "NaN") ? NAN : (float)std::strtod(tok.c_str(), nullptr);
          if ((int)k == ix) cloud[i].x = v; else if ((int)k == iy) cloud[i].y = v; else if ((int)k == iz) cloud[i].z = v; else if ((int)k == ii) cloud[i].intensity = v;
        }
    }
    return true;
  }
  std::vector<unsigned char> raw;
  if (data_mode == "binary") {
    raw.resize(points * stride);
    f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size());
    if ((size_t)f.gcount() != raw.size()) return false;
    for (size_t i = 0; i < points; ++i) {
      const unsigned char* p = raw.data() + i * stride;
      cloud[i].x = as_float(fields[ix], p + fields[ix].offset); cloud[i].y = as_float(fields[iy], p + fields[iy].offset);
      cloud[i].z = as_float(fields[iz], p + fields[iz].offset);
      if (ii >= 0) cloud[i].intensity = as_float(fields[ii], p + fields[ii].offset);
    }
    return true;
  }
  if (data_mode == "binary_compressed") {
    uint32_t csize = 0, usize = 0;
    f.read(reinterpret_cast<char*>(&csize), 4); f.read(reinterpret_cast<char*>(&usize), 4);
    if (!f || usize != points * stride || remaining < 8 || csize > remaining - 8) return false;
    std::vector<unsigned char> comp(csize);
    f.read(reinterpret_cast<char*>(comp.data()), csize);
    if ((size_t)f.gcount() != csize) return false;
    raw.resize(usize);
    if (!LzfDecompress(comp.data(), csize, raw.data(), usize)) return false;
    // the uncompressed block is field-major (all x, all y, ...)
    size_t base = 0;
    std::vector<size_t> fbase(fields.size());
    for (size_t k = 0; k < fields.size(); ++k) { fbase[k] = base; base += (size_t)fields[k].size * fields[k].count * points; }
    for (size_t i = 0; i < points; ++i) {
      auto at = [&](int k) { return raw.data() + fbase[k] + i * (size_t)fields[k].size * fields[k].count; };
      cloud[i].x = as_float(fields[ix], at(ix)); cloud[i].y = as_float(fields[iy], at(iy)); cloud[i].z = as_float(fields[iz], at(iz));
      if (ii >= 0) cloud[i].intensity = as_float(fields[ii], at(ii));
    }
    return true;
  }
  return false;
}
}  // namespace

bool Velodyne::LoadLidar(std::string file_path) {
  if (file_path.empty()) file_path = name;
  const std::string::size_type pos = file_path.rfind('.');
  const std::string type = pos == std::string::npos ? "" : file_path.substr(pos);
  if (type != ".pcd") { fprintf(stderr, "unknown point cloud format, only .pcd is mirrored (the reference also reads .ply)\n"); return false; }
  PointCloud raw;
  if (!ReadPcd(file_path, raw)) { fprintf(stderr, "Fail to load lidar data at %s\n", file_path.c_str()); return false; }
  name = file_path;
  cloud.clear();
  for (const PointXYZI& p : raw) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;      // pcl::removeNaNFromPointCloud
    const float dis = p.x * p.x + p.y * p.y + p.z * p.z;                                  // removeClosedPointCloud(0.5), float
    if (dis < 0.5f * 0.5f) continue;
    // T_cam_lidar (:127-132): X right, Y forward, Z up  ->  X right, Y down, Z forward
    cloud.push_back({p.x, -p.z, p.y, p.intensity});
  }
  if (cloud.size() < 4000) { fprintf(stderr, "lidar %d is invalid, only %zu points in point cloud\n", id, cloud.size()); valid = false; }
  return true;
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

// ================================================================================================
// ceres-like problem / solver
// ================================================================================================
namespace ceres_like {

static const int kStride[6] = {7, 7, 9, 9, 10, 12};

bool CostFunction::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
  Engine& e = Engine::Default();
  if (kind == kReprojKind) {   // {aa_cw, t_cw, point_3d}: a one-observation reprojection set
    const int64_t off[2] = {0, 1};
    const int cam = 0;
    pvlm_baset* bs = nullptr;
    if (pvlm_ba_create(e.ctx(), 1, 1, off, &cam, row.data(), parameters[2], weight, &bs) != PVLM_OK) return false;
    bool ok = pvlm_set_poses(e.ctx(), 1, parameters[0], parameters[1]) == PVLM_OK;
    double J[9];
    ok = ok && pvlm_ba_eval(e.ctx(), bs, residuals, jacobians ? J : nullptr) == PVLM_OK;
    pvlm_ba_destroy(e.ctx(), bs);
    if (ok && jacobians)
      for (int b = 0; b < 3; ++b)
        if (jacobians[b]) for (int k = 0; k < 3; ++k) jacobians[b][k] = J[3 * b + k];
    return ok && std::isfinite(residuals[0]);
  }
  double aa[6] = {parameters[0][0], parameters[0][1], parameters[0][2], parameters[2][0], parameters[2][1], parameters[2][2]};
  double t[6] = {parameters[1][0], parameters[1][1], parameters[1][2], parameters[3][0], parameters[3][1], parameters[3][2]};
  const int64_t off[2] = {0, 1};
  const int ref = 0, nei = 1;
  pvlm_resset* rs = nullptr;
  if (pvlm_resset_upload(e.ctx(), (pvlm_functor)kind, flags, weight, 1, 1, off, &ref, &nei, row.data(), kStride[kind], &rs) != PVLM_OK) return false;
  bool ok = pvlm_set_poses(e.ctx(), 2, aa, t) == PVLM_OK;
  double J[12];
  ok = ok && pvlm_eval(e.ctx(), rs, residuals, jacobians ? J : nullptr) == PVLM_OK;
  pvlm_resset_destroy(e.ctx(), rs);
  if (ok && jacobians)
    for (int b = 0; b < 4; ++b)
      if (jacobians[b]) for (int k = 0; k < 3; ++k) jacobians[b][k] = J[3 * b + k];
  return ok && std::isfinite(residuals[0]);
}

struct Problem::Impl {
  // parameter blocks: every double[3] the caller registered, in first-seen order
  std::unordered_map<double*, int> block_id;                 // only looked up, never iterated: ids are handed out in arrival order
  std::vector<double*> blocks;
  std::vector<bool> constant;
  // poses = (aa block, t block) pairs, in first-seen order
  std::unordered_map<unsigned long long, int> pose_id;       // key: block id of aa << 32 | block id of t
  std::vector<std::pair<int, int>> poses;
  // A group = one device residual set.  Host-built groups collect consecutive AddResidualBlock
  // calls with identical (kind, flags, weight, loss); consecutive blocks with the same pose pair
  // form a segment.  dev_* are the pose ids the device set uses (for sets handed in by the
  // association kernels these are the caller's list indices), ref/nei the Problem pose ids.
  struct Group {
    int kind = 0; unsigned flags = 0; double weight = 1.0; LossFunction* loss = nullptr;
    std::vector<double> rows; std::vector<int64_t> off; std::vector<int> ref, nei, dev_ref, dev_nei;
    pvlm_resset* set = nullptr; bool external = false;
    pvlm_neq* neq = nullptr; int dev_poses = 0;
    std::vector<int> ui, uj;            // unordered dev-id pairs of the neq structure
    std::vector<int> dev_to_pose;       // dev id -> Problem pose id (-1 unused)
  };
  std::vector<Group> groups;
  // Reprojection blocks (camera pose + free 3-D point), one group per (weight, loss).  At Solve the observations
  // are sorted by point and handed to the GPU (pvlm_baset); the point blocks are eliminated there.
  struct Bundle {
    double weight = 1.0; LossFunction* loss = nullptr;
    std::vector<int> obs_pose, obs_point;      // Problem pose id / parameter-block id of the point, insertion order
    std::vector<double> obs_bearing;           // 3 per observation (as handed to Create, un-normalised)
    pvlm_baset* set = nullptr;
    std::vector<int> point_blocks;             // device point index -> parameter-block id
    std::vector<int> dev_to_pose;              // device camera id -> Problem pose id
    std::vector<int> ui, uj;                   // co-visible device camera pairs (packed layout)
  };
  std::vector<Bundle> bundles;
  std::vector<bool> is_point;                  // per parameter block
  std::vector<CostFunction*> owned_costs;
  std::set<LossFunction*> owned_losses;
  int num_blocks = 0;

  int Block(double* p) {
    auto it = block_id.find(p);
    if (it != block_id.end()) return it->second;
    const int id = (int)blocks.size();
    block_id[p] = id; blocks.push_back(p); constant.push_back(false); is_point.push_back(false);
    return id;
  }
  int Pose(double* aa, double* t) {
    const std::pair<int, int> k(Block(aa), Block(t));
    const unsigned long long key = ((unsigned long long)(unsigned)k.first << 32) | (unsigned)k.second;
    auto it = pose_id.find(key);
    if (it != pose_id.end()) return it->second;
    const int id = (int)poses.size();
    pose_id[key] = id; poses.push_back(k);
    return id;
  }
};

Problem::Problem() : impl_(new Impl()) {}
Problem::~Problem() {
  Engine& e = Engine::Default();
  for (auto& g : impl_->groups) { if (g.neq) pvlm_neq_destroy(e.ctx(), g.neq); if (g.set) pvlm_resset_destroy(e.ctx(), g.set); }
  for (auto& b : impl_->bundles) if (b.set) pvlm_ba_destroy(e.ctx(), b.set);
  for (CostFunction* c : impl_->owned_costs) delete c;
  for (LossFunction* l : impl_->owned_losses) delete l;
  delete impl_;
}
int Problem::NumResidualBlocks() const { return impl_->num_blocks; }
void Problem::RegisterPoses(std::vector<Vector3d>& aa_list, std::vector<Vector3d>& t_list) {
  for (size_t i = 0; i < aa_list.size() && i < t_list.size(); ++i) impl_->Pose(aa_list[i].data(), t_list[i].data());
}
void Problem::SetParameterBlockConstant(double* block) { impl_->constant[impl_->Block(block)] = true; }

void Problem::AddResidualBlock(CostFunction* cost, LossFunction* loss, double* aa_r, double* t_r, double* aa_n, double* t_n) {
  Impl& I = *impl_;
  const int pr = I.Pose(aa_r, t_r), pn = I.Pose(aa_n, t_n);
  if (loss) I.owned_losses.insert(loss);
  I.owned_costs.push_back(cost);
  // blocks are grouped by (functor, flags, weight, loss) regardless of the order they arrive in
  // (AddCameraLidarResidual alternates two functors); inside a group the insertion order is kept.
  int gi = -1;
  for (int k = (int)I.groups.size() - 1; k >= 0; --k) {
    const Impl::Group& c = I.groups[k];
    if (!c.external && !c.set && c.kind == cost->kind && c.flags == cost->flags && c.weight == cost->weight && c.loss == loss) { gi = k; break; }
  }
  if (gi < 0) {
    Impl::Group g; g.kind = cost->kind; g.flags = cost->flags; g.weight = cost->weight; g.loss = loss; g.off.push_back(0);
    I.groups.push_back(g);
    gi = (int)I.groups.size() - 1;
  }
  Impl::Group& g = I.groups[gi];
  if (g.ref.empty() || g.ref.back() != pr || g.nei.back() != pn) { g.ref.push_back(pr); g.nei.push_back(pn); g.off.push_back(g.off.back()); }
  g.rows.insert(g.rows.end(), cost->row.begin(), cost->row.end());
  g.off.back() += 1;
  I.num_blocks++;
}

void Problem::AddResidualRows(int kind, unsigned flags, double weight, LossFunction* loss, double* aa_r, double* t_r, double* aa_n, double* t_n,
                              const double* rows, size_t n) {
  if (n == 0) return;
  Impl& I = *impl_;
  const int pr = I.Pose(aa_r, t_r), pn = I.Pose(aa_n, t_n);
  if (loss) I.owned_losses.insert(loss);
  int gi = -1;
  for (int k = (int)I.groups.size() - 1; k >= 0; --k) {
    const Impl::Group& c = I.groups[k];
    if (!c.external && !c.set && c.kind == kind && c.flags == flags && c.weight == weight && c.loss == loss) { gi = k; break; }
  }
  if (gi < 0) {
    Impl::Group g; g.kind = kind; g.flags = flags; g.weight = weight; g.loss = loss; g.off.push_back(0);
    I.groups.push_back(g);
    gi = (int)I.groups.size() - 1;
  }
  Impl::Group& g = I.groups[gi];
  if (g.ref.empty() || g.ref.back() != pr || g.nei.back() != pn) { g.ref.push_back(pr); g.nei.push_back(pn); g.off.push_back(g.off.back()); }
  g.rows.insert(g.rows.end(), rows, rows + n * (size_t)kStride[kind]);
  g.off.back() += (int64_t)n;
  I.num_blocks += (int)n;
}

void Problem::AddResidualBlock(CostFunction* cost, LossFunction* loss, double* aa_c, double* t_c, double* point_3d) {
  Impl& I = *impl_;
  if (cost->kind != kReprojKind) throw std::runtime_error("three-block AddResidualBlock expects PanoramaReprojResidual_1Angle");
  const int pose = I.Pose(aa_c, t_c);
  const int pb = I.Block(point_3d);
  I.is_point[pb] = true;
  if (loss) I.owned_losses.insert(loss);
  I.owned_costs.push_back(cost);
  int bi = -1;
  for (int k = (int)I.bundles.size() - 1; k >= 0; --k)
    if (!I.bundles[k].set && I.bundles[k].weight == cost->weight && I.bundles[k].loss == loss) { bi = k; break; }
  if (bi < 0) { Impl::Bundle b; b.weight = cost->weight; b.loss = loss; I.bundles.push_back(b); bi = (int)I.bundles.size() - 1; }
  Impl::Bundle& b = I.bundles[bi];
  b.obs_pose.push_back(pose); b.obs_point.push_back(pb);
  b.obs_bearing.insert(b.obs_bearing.end(), cost->row.begin(), cost->row.begin() + 3);
  I.num_blocks++;
}

void Problem::AddResidualSet(pvlm_resset* set, LossFunction* loss, std::vector<Vector3d>* aa_list, std::vector<Vector3d>* t_list) {
  Impl& I = *impl_;
  if (loss) I.owned_losses.insert(loss);
  int64_t n = 0; int P = 0, kind = 0; unsigned flags = 0;
  pvlm_resset_info(set, &n, &P, &kind, &flags);
  Impl::Group g; g.kind = kind; g.flags = flags; g.loss = loss; g.set = set; g.external = true;
  g.off.resize((size_t)P + 1); g.dev_ref.resize((size_t)std::max(P, 1)); g.dev_nei.resize((size_t)std::max(P, 1));
  Engine& e = Engine::Default();
  e.Check(pvlm_resset_download(e.ctx(), set, g.off.data(), g.dev_ref.data(), g.dev_nei.data(), nullptr), "pvlm_resset_download");
  g.dev_ref.resize(P); g.dev_nei.resize(P);
  for (int p = 0; p < P; ++p) {
    g.ref.push_back(I.Pose((*aa_list)[g.dev_ref[p]].data(), (*t_list)[g.dev_ref[p]].data()));
    g.nei.push_back(I.Pose((*aa_list)[g.dev_nei[p]].data(), (*t_list)[g.dev_nei[p]].data()));
  }
  I.groups.push_back(g);
  I.num_blocks += (int)n;
}

std::string Solver::Summary::BriefReport() const {
  char b[256];
  snprintf(b, sizeof(b), "pvlm LM: blocks %d, initial cost %.6e, final cost %.6e, successful %d, unsuccessful %d, %s", num_residual_blocks,
           initial_cost, final_cost, num_successful_steps, num_unsuccessful_steps, message.c_str());
  return b;
}

namespace {

// Skyline (profile) Cholesky of a symmetric positive definite matrix given as dense row-major
// lower triangle accessor.  Pose graphs of LiDAR odometry are block-banded (temporal neighbours)
// plus a few loop closures, which the envelope captures.
struct Skyline {
  int n = 0;
  std::vector<int> first;          // first stored column of each row
  std::vector<size_t> start;       // offset of row i in val (entries first[i]..i)
  std::vector<double> val;
  double& at(int i, int j) { return val[start[i] + (size_t)(j - first[i])]; }
  void Init(const std::vector<int>& f) {
    n = (int)f.size(); first = f; start.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) start[i + 1] = start[i] + (size_t)(i - first[i] + 1);
    val.assign(start[n], 0.0);
  }
  bool Factor() {
    for (int i = 0; i < n; ++i) {
      for (int j = first[i]; j <= i; ++j) {
        double s = at(i, j);
        const int k0 = std::max(first[i], first[j]);
        for (int k = k0; k < j; ++k) s -= at(i, k) * at(j, k);
        if (j < i) at(i, j) = s / at(j, j);
        else { if (!(s > 0.0)) return false; at(i, i) = std::sqrt(s); }
      }
    }
    return true;
  }
  void Solve(std::vector<double>& b) {
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = first[i]; k < i; ++k) s -= at(i, k) * b[k]; b[i] = s / at(i, i); }
    for (int i = n - 1; i >= 0; --i) { b[i] /= at(i, i); for (int k = first[i]; k < i; ++k) b[k] -= at(i, k) * b[i]; }
  }
};

// Gauss-Newton blocks of the four-block groups at one point.  The block STRUCTURE is fixed for a whole Solve (the sorted key
// list is built once and shared); an evaluation only refills the numbers: H[36 k ...] = block keys[k] = (pose a <= pose b),
// 6x6 row-major d2/dx_a dx_b.  (Round 2 rebuilt a std::map of 36-double nodes at every evaluation: 1.6 ms per LM step at Room scale.)
using BlockKeys = std::vector<std::pair<int, int>>;
struct Assembled {
  double cost = 0;
  std::vector<double> g;                       // n_free
  std::shared_ptr<const BlockKeys> keys;
  std::vector<double> H;                       // 36 per key
};
// one iteration protocol for the flat table and for the std::map the reprojection path still uses
template <typename F> inline void ForEachBlock(const Assembled& A, F&& f) {
  if (!A.keys) return;
  for (size_t k = 0; k < A.keys->size(); ++k) f((*A.keys)[k], A.H.data() + 36 * k);
}
template <typename F> inline void ForEachBlock(const std::map<std::pair<int, int>, std::array<double, 36>>& H, F&& f) {
  for (auto& kv : H) f(kv.first, kv.second.data());
}

}  // namespace

void Solve(const Solver::Options& opt, Problem* problem, Solver::Summary* summary) {
  StageTimer stage_timer_("solve (LM)");
  Problem::Impl& I = *problem->impl();
  Engine& e = Engine::Default();
  *summary = Solver::Summary();
  summary->num_residual_blocks = I.num_blocks;
  const int NP = (int)I.poses.size();
  const Exchange* xch = (opt.exchange && opt.exchange->active()) ? opt.exchange : nullptr;
  // a rank of a sharded solve may own no residual block at all and still has to take part in every exchange
  if (!xch && (I.num_blocks == 0 || NP == 0)) { summary->message = "no residual blocks"; return; }
  if (xch) {
    // Entry into a sharded Solve is collective: a rank that cannot take part (no registered poses, reprojection blocks — which are
    // not sharded) must not leave its peers waiting in the first all-reduce.  Every rank contributes its own verdict to one
    // exchange and all of them leave together, with the same message.
    bool has_bundle = false;
    for (auto& b : I.bundles) if (!b.obs_pose.empty()) has_bundle = true;
    // sum and sum of squares of the pose counts: world * sum(NP^2) == sum(NP)^2 holds exactly when all counts are equal (Cauchy-Schwarz),
    // and every rank evaluates the same reduced numbers — a test against the local NP alone let the rank whose count equals the mean
    // pass while its peers threw, and hang in the next exchange
    double agreed[4] = {NP == 0 ? 1.0 : 0.0, has_bundle ? 1.0 : 0.0, (double)NP, (double)NP * (double)NP};
    xch->allreduce_sum(agreed, 4);
    if (agreed[0] > 0) throw std::runtime_error("sharded Solve: " + std::to_string((int)agreed[0]) + " rank(s) entered without registered poses (Problem::RegisterPoses must run on every rank)");
    if (agreed[1] > 0) throw std::runtime_error("sharded Solve: reprojection blocks are not sharded (camera terms run on one GPU)");
    if ((double)xch->world * agreed[3] != agreed[2] * agreed[2]) throw std::runtime_error("sharded Solve: the ranks registered different numbers of poses");
  }
  // concatenation of every rank's list through the one primitive an Exchange has
  auto all_concat = [&](const std::vector<double>& mine) {
    std::vector<double> cnt((size_t)xch->world, 0.0);
    cnt[(size_t)xch->rank] = (double)mine.size();
    xch->allreduce_sum(cnt.data(), cnt.size());
    size_t total = 0, off = 0;
    for (int r = 0; r < xch->world; ++r) { if (r == xch->rank) off = total; total += (size_t)cnt[(size_t)r]; }
    std::vector<double> all(std::max<size_t>(total, 1), 0.0);
    std::copy(mine.begin(), mine.end(), all.begin() + (std::ptrdiff_t)off);
    xch->allreduce_sum(all.data(), all.size());
    all.resize(total);
    return all;
  };

  // ---- device sets, ONE pose numbering and ONE normal-equation structure for all four-block groups --------------
  // Every group's segments are renumbered to the Problem's pose ids (a set that came from the association carries the caller's list
  // indices: pvlm_resset_set_pose_ids, once), so that a linearisation needs one pose table and the groups' blocks are summed on the
  // device into one packed buffer [diag NP x 36 | off U x 36 | g NP x 6 | cost] over the union (ui < uj) of their pose pairs.
  StageTimer* stage_timer_setup_ = new StageTimer("solve: residual-set upload + structures");
  std::vector<int> gui, guj;
  {
    std::set<std::pair<int, int>> up;
    for (auto& g : I.groups) {
      const int P = (int)g.ref.size();
      if (!g.set) {
        g.dev_ref = g.ref; g.dev_nei = g.nei;
        e.Check(pvlm_resset_upload(e.ctx(), (pvlm_functor)g.kind, g.flags, g.weight, g.off.back(), P, g.off.data(), g.dev_ref.data(), g.dev_nei.data(),
                                   g.rows.data(), kStride[g.kind], &g.set), "pvlm_resset_upload");
        std::vector<double>().swap(g.rows);
      } else if (g.dev_ref != g.ref || g.dev_nei != g.nei) {
        e.Check(pvlm_resset_set_pose_ids(e.ctx(), g.set, g.ref.data(), g.nei.data()), "pvlm_resset_set_pose_ids");
        g.dev_ref = g.ref; g.dev_nei = g.nei;
      }
      for (int p = 0; p < P; ++p) up.insert({std::min(g.ref[p], g.nei[p]), std::max(g.ref[p], g.nei[p])});
    }
    for (auto& u : up) { gui.push_back(u.first); guj.push_back(u.second); }
    for (auto& g : I.groups) {
      if (g.neq && g.dev_poses == NP && g.ui == gui && g.uj == guj) continue;      // a second Solve on an unchanged Problem
      if (g.neq) { pvlm_neq_destroy(e.ctx(), g.neq); g.neq = nullptr; }
      g.dev_poses = NP; g.ui = gui; g.uj = guj;
      g.dev_to_pose.resize((size_t)NP);
      for (int p = 0; p < NP; ++p) g.dev_to_pose[(size_t)p] = p;
      e.Check(pvlm_neq_create(e.ctx(), NP, (int)gui.size(), gui.data(), guj.data(), &g.neq), "pvlm_neq_create");
    }
  }
  delete stage_timer_setup_;
  // ---- reprojection sets: observations sorted by point, points resident on the GPU ------------------
  bool have_bundles = false;
  for (auto& b : I.bundles) {
    if (b.obs_pose.empty()) continue;
    have_bundles = true;
    if (b.set) continue;
    StageTimer stage_timer_bundle_("solve: reprojection set creation (host grouping by point + pvlm_ba_create)");
    std::unordered_map<int, int> pidx, cidx;      // ids are handed out in order of first appearance: the container's order plays no role
    const size_t n = b.obs_pose.size();
    std::vector<int> obs_dev_point(n);
    for (size_t i = 0; i < n; ++i) {
      auto it = pidx.find(b.obs_point[i]);
      if (it == pidx.end()) { it = pidx.insert({b.obs_point[i], (int)b.point_blocks.size()}).first; b.point_blocks.push_back(b.obs_point[i]); }
      obs_dev_point[i] = it->second;
    }
    const int M = (int)b.point_blocks.size();
    std::vector<int64_t> off((size_t)M + 1, 0);
    for (size_t i = 0; i < n; ++i) off[(size_t)obs_dev_point[i] + 1]++;
    for (int p = 0; p < M; ++p) off[(size_t)p + 1] += off[p];
    std::vector<int64_t> fill(off.begin(), off.end() - 1);
    std::vector<int> cam(n); std::vector<double> bearing(3 * n), points((size_t)M * 3);
    for (size_t i = 0; i < n; ++i) {   // stable: insertion order inside a point's track
      const size_t dst = (size_t)fill[obs_dev_point[i]]++;
      auto ic = cidx.find(b.obs_pose[i]);
      if (ic == cidx.end()) { ic = cidx.insert({b.obs_pose[i], (int)b.dev_to_pose.size()}).first; b.dev_to_pose.push_back(b.obs_pose[i]); }
      cam[dst] = ic->second;
      for (int k = 0; k < 3; ++k) bearing[3 * dst + k] = b.obs_bearing[3 * i + k];
    }
    std::vector<unsigned char> frozen((size_t)M, 0); bool any_frozen = false;
    for (int p = 0; p < M; ++p) {
      for (int k = 0; k < 3; ++k) points[(size_t)p * 3 + k] = I.blocks[b.point_blocks[p]][k];
      if (I.constant[b.point_blocks[p]]) { frozen[p] = 1; any_frozen = true; }
    }
    e.Check(pvlm_ba_create(e.ctx(), M, (int64_t)n, off.data(), cam.data(), bearing.data(), points.data(), b.weight, &b.set), "pvlm_ba_create");
    if (any_frozen) e.Check(pvlm_ba_set_constant(e.ctx(), b.set, frozen.data()), "pvlm_ba_set_constant");
    int nu = 0;
    pvlm_ba_structure(b.set, nullptr, nullptr, nullptr, &nu, nullptr, nullptr);
    b.ui.resize(nu); b.uj.resize(nu);
    pvlm_ba_structure(b.set, nullptr, nullptr, nullptr, nullptr, b.ui.data(), b.uj.data());
    std::vector<int>().swap(b.obs_pose); std::vector<int>().swap(b.obs_point); std::vector<double>().swap(b.obs_bearing);
    b.obs_pose.push_back(-1);   // keeps "non-empty" for later Solve calls on the same Problem
  }

  // ---- free-parameter layout (pose blocks; the point blocks never reach the host system) -------------
  std::vector<int> block_off(I.blocks.size(), -1);
  int n_free = 0;
  // sharded solve: the poses were registered up front (RegisterPoses), so a rank also knows poses none of ITS blocks
  // touch; a pose no rank touches is left out of the system, as if it had never been added
  std::vector<char> touched(I.blocks.size(), xch ? 0 : 1);
  std::vector<std::pair<int, int>> xkeys;          // sharded solve: sorted union of the ranks' 6x6 block keys (pose a <= pose b)
  if (xch) {
    std::vector<double> use(I.blocks.size(), 0.0), keys;
    std::set<std::pair<int, int>> mine;
    for (auto& g : I.groups)
      for (size_t p = 0; p < g.ref.size(); ++p) {
        for (int q : {g.ref[p], g.nei[p]}) { use[(size_t)I.poses[q].first] = 1.0; use[(size_t)I.poses[q].second] = 1.0; mine.insert({q, q}); }
        mine.insert({std::min(g.ref[p], g.nei[p]), std::max(g.ref[p], g.nei[p])});
      }
    xch->allreduce_sum(use.data(), use.size());
    for (size_t b = 0; b < use.size(); ++b) touched[b] = use[b] > 0.0;
    for (auto& k : mine) { keys.push_back((double)k.first); keys.push_back((double)k.second); }
    const std::vector<double> all = all_concat(keys);
    std::set<std::pair<int, int>> uni;
    for (size_t i = 0; i + 1 < all.size(); i += 2) uni.insert({(int)all[i], (int)all[i + 1]});
    xkeys.assign(uni.begin(), uni.end());
  }
  for (int p = 0; p < NP; ++p)
    for (int b : {I.poses[p].first, I.poses[p].second})
      if (!I.constant[b] && touched[b] && block_off[b] < 0) { block_off[b] = n_free; n_free += 3; }
  if (n_free == 0) { summary->message = "all parameter blocks constant"; }

  std::vector<double> x(3 * I.blocks.size());
  auto load_x = [&]() { for (size_t b = 0; b < I.blocks.size(); ++b) for (int k = 0; k < 3; ++k) x[3 * b + k] = I.blocks[b][k]; };
  auto store_x = [&](const std::vector<double>& v) { for (size_t b = 0; b < I.blocks.size(); ++b) for (int k = 0; k < 3; ++k) I.blocks[b][k] = v[3 * b + k]; };
  load_x();
  // scalar row/col index of (pose, half, k)
  auto idx = [&](int pose, int r) { const int b = r < 3 ? I.poses[pose].first : I.poses[pose].second; return block_off[b] < 0 ? -1 : block_off[b] + (r % 3); };

  // ---- fixed block structure of the four-block groups (built once per Solve) ----------------------------------------
  // Unsharded: the key list IS the packed layout — the NP diagonal blocks, then the U pair blocks — so an evaluation's table is a plain
  // copy of the buffer the GPU filled.  Sharded: the sorted union over the ranks (the exchanged buffer is then the table), filled
  // through a slot map.
  const int n_groups = (int)I.groups.size();
  const int U = (int)gui.size();
  auto keys = std::make_shared<BlockKeys>();
  std::vector<int> slot_of_packed;              // sharded: packed block (diag p | pair u) -> position in the key list
  if (xch) {
    *keys = xkeys;
    auto slot_of = [&](int a, int b) {
      const auto it = std::lower_bound(keys->begin(), keys->end(), std::make_pair(a, b));
      if (it == keys->end() || *it != std::make_pair(a, b)) throw std::runtime_error("Solve: block key missing from the structure");
      return (int)(it - keys->begin());
    };
    std::vector<char> used((size_t)NP, 0);
    for (auto& g : I.groups) for (size_t p = 0; p < g.ref.size(); ++p) { used[(size_t)g.ref[p]] = 1; used[(size_t)g.nei[p]] = 1; }
    slot_of_packed.assign((size_t)NP + (size_t)U, -1);
    for (int p = 0; p < NP; ++p) if (used[(size_t)p]) slot_of_packed[(size_t)p] = slot_of(p, p);
    for (int u = 0; u < U; ++u) slot_of_packed[(size_t)NP + (size_t)u] = slot_of(gui[(size_t)u], guj[(size_t)u]);
  } else {
    keys->reserve((size_t)NP + (size_t)U);
    for (int p = 0; p < NP; ++p) keys->push_back({p, p});
    for (int u = 0; u < U; ++u) keys->push_back({gui[(size_t)u], guj[(size_t)u]});
  }
  // pinned landing buffer of the packed normal equations; released on every exit path (Solve has several)
  struct PackedIO {
    double* packed = nullptr; size_t count = 0; pvlm_ctx* ctx;
    std::vector<double> aa, tt;
    std::vector<pvlm_neq*> neq; std::vector<const pvlm_resset*> sets; std::vector<pvlm_loss> loss; std::vector<double> loss_a;
    explicit PackedIO(pvlm_ctx* c) : ctx(c) {}
    ~PackedIO() { pvlm_synchronize(ctx); if (packed) pvlm_host_free(ctx, packed); }
  } io(e.ctx());
  if (n_groups > 0) {
    io.count = (size_t)pvlm_neq_size(I.groups[0].neq);
    void* p = nullptr;
    e.Check(pvlm_host_alloc(e.ctx(), (int64_t)(io.count * sizeof(double)), &p), "pvlm_host_alloc");
    io.packed = static_cast<double*>(p);
    for (auto& g : I.groups) {
      io.neq.push_back(g.neq); io.sets.push_back(g.set);
      io.loss.push_back(g.loss ? (pvlm_loss)g.loss->kind() : PVLM_LOSS_NONE); io.loss_a.push_back(g.loss ? g.loss->a() : 0.0);
    }
  }
  io.aa.assign((size_t)NP * 3, 0.0); io.tt.assign((size_t)NP * 3, 0.0);

  // evaluates cost (+ H, g when want_H) of the four-block groups at parameter vector v: ONE pose table, ONE submission for all groups
  // (per group: pair table, fused kernel, epilogue, gather ADDING into the shared packed buffer), one queued copy, ONE synchronisation.
  // The groups' blocks are summed in group order, as the host used to add them.
  long evaluations = 0;
  auto evaluate = [&](const std::vector<double>& v, bool want_H, Assembled& A) {
    // the first linearisation of a Solve binds the structures to the residual sets (CSR upload), sizes the per-structure buffers
    // and, once per process, loads the kernels' code objects: timed apart from the steady LM steps
    StageTimer stage_timer_eval_(evaluations++ == 0 ? "solve: first linearisation of a Solve (structures bound, buffers sized, code objects loaded)"
                                                    : "solve: GPU linearisation + block assembly");
    static const bool eval_trace = std::getenv("PVLM_HOST_EVAL_TRACE") != nullptr;   // per-call phase times on stderr (profiling tools)
    const auto tr0 = std::chrono::steady_clock::now();
    A.cost = 0; A.g.assign(n_free, 0.0); A.keys = keys; A.H.clear();
    if (n_groups > 0) {
      for (int p = 0; p < NP; ++p)
        for (int k = 0; k < 3; ++k) { io.aa[3 * (size_t)p + k] = v[3 * I.poses[p].first + k]; io.tt[3 * (size_t)p + k] = v[3 * I.poses[p].second + k]; }
      e.Check(pvlm_set_poses(e.ctx(), NP, io.aa.data(), io.tt.data()), "pvlm_set_poses");
      e.Check(pvlm_neq_accumulate_sets(e.ctx(), n_groups, io.neq.data(), io.sets.data(), io.loss.data(), io.loss_a.data(), io.packed), "pvlm_neq_accumulate_sets");
    }
    const auto tr1 = std::chrono::steady_clock::now();
    if (n_groups > 0) e.Check(pvlm_synchronize(e.ctx()), "pvlm_synchronize");
    const auto tr2 = std::chrono::steady_clock::now();
    if (n_groups > 0) {
      const double* packed = io.packed;
      A.cost = packed[io.count - 1];
      if (want_H) {
        const double* gg = packed + ((size_t)NP + (size_t)U) * 36;
        for (int p = 0; p < NP; ++p)
          for (int half = 0; half < 2; ++half) {
            const int b = half ? I.poses[p].second : I.poses[p].first;
            if (block_off[b] >= 0) for (int k = 0; k < 3; ++k) A.g[block_off[b] + k] += gg[(size_t)p * 6 + 3 * half + k];
          }
        if (!xch) A.H.assign(packed, packed + ((size_t)NP + (size_t)U) * 36);
        else {
          A.H.assign(36 * keys->size(), 0.0);
          for (size_t q = 0; q < slot_of_packed.size(); ++q)
            if (slot_of_packed[q] >= 0) std::copy(packed + 36 * q, packed + 36 * (q + 1), A.H.begin() + 36 * (std::ptrdiff_t)slot_of_packed[q]);
        }
      }
    } else if (want_H) A.H.assign(36 * keys->size(), 0.0);
    if (eval_trace) {
      const auto tr3 = std::chrono::steady_clock::now();
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      fprintf(stderr, "[eval %ld] submission %.0f us, synchronisation %.0f us, table %.0f us (%d groups, %zu keys)\n", evaluations, us(tr0, tr1), us(tr1, tr2), us(tr2, tr3),
              n_groups, keys->size());
    }
    if (xch) {
      // the one exchange of an evaluation: [cost | g | blocks in key order], summed over the ranks;
      // afterwards every rank holds the same Assembled, bit for bit
      StageTimer stage_timer_x_("solve: exchange of the normal equations");
      std::vector<double> buf(want_H ? 1 + (size_t)n_free + A.H.size() : 1, 0.0);
      buf[0] = A.cost;
      if (want_H) { std::copy(A.g.begin(), A.g.end(), buf.begin() + 1); std::copy(A.H.begin(), A.H.end(), buf.begin() + 1 + n_free); }
      xch->allreduce_sum(buf.data(), buf.size());
      A.cost = buf[0];
      if (want_H) { std::copy(buf.begin() + 1, buf.begin() + 1 + n_free, A.g.begin()); std::copy(buf.begin() + 1 + n_free, buf.end(), A.H.begin()); }
    }
  };

  // ---- reprojection sets: reduced camera system for a given trust-region radius ------------------------
  struct Reduced {
    double cost = 0, gmax_points = 0;
    std::vector<double> g_red, g_cam, Udiag;                  // n_free each
    std::map<std::pair<int, int>, std::array<double, 36>> H;  // Schur complement blocks (pose a <= pose b)
  };
  auto bundle_poses = [&](const Problem::Impl::Bundle& b, const std::vector<double>& v) {
    const int nd = (int)b.dev_to_pose.size();
    std::vector<double> aa((size_t)nd * 3), tt((size_t)nd * 3);
    for (int d = 0; d < nd; ++d) {
      const int p = b.dev_to_pose[d];
      for (int k = 0; k < 3; ++k) { aa[3 * d + k] = v[3 * I.poses[p].first + k]; tt[3 * d + k] = v[3 * I.poses[p].second + k]; }
    }
    e.Check(pvlm_set_poses(e.ctx(), nd, aa.data(), tt.data()), "pvlm_set_poses");
  };
  auto bundle_reduce = [&](const std::vector<double>& v, double radius, bool init, Reduced& R) {
    StageTimer stage_timer_br_("solve: reprojection blocks reduced on the GPU + host scatter of the camera system");
    R = Reduced(); R.g_red.assign(n_free, 0.0); R.g_cam.assign(n_free, 0.0); R.Udiag.assign(n_free, 0.0);
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      bundle_poses(b, v);
      std::vector<double> packed((size_t)pvlm_ba_packed_size(b.set), 0.0);
      e.Check(pvlm_ba_reduce(e.ctx(), b.set, b.loss ? b.loss->kind() : PVLM_LOSS_NONE, b.loss ? b.loss->a() : 0.0, init ? 1 : 0, radius,
                             opt.min_lm_diagonal, opt.max_lm_diagonal, packed.data()), "pvlm_ba_reduce");
      const int nd = (int)b.dev_to_pose.size(), nu = (int)b.ui.size();
      const double* Hd = packed.data(); const double* Ho = Hd + (size_t)nd * 36; const double* gg = Ho + (size_t)nu * 36;
      const double* Ud = gg + (size_t)nd * 6 + 1; const double* gc = Ud + (size_t)nd * 6;
      R.cost += gg[(size_t)nd * 6];
      R.gmax_points = std::max(R.gmax_points, packed.back());
      for (int d = 0; d < nd; ++d) {
        const int p = b.dev_to_pose[d];
        auto& blk = R.H[{p, p}];
        for (int k = 0; k < 36; ++k) blk[k] += Hd[(size_t)d * 36 + k];
        for (int r = 0; r < 6; ++r) {
          const int i = idx(p, r);
          if (i < 0) continue;
          R.g_red[i] += gg[(size_t)d * 6 + r]; R.g_cam[i] += gc[(size_t)d * 6 + r]; R.Udiag[i] += Ud[(size_t)d * 6 + r];
        }
      }
      for (int u = 0; u < nu; ++u) {
        const int pa = b.dev_to_pose[b.ui[u]], pb = b.dev_to_pose[b.uj[u]];
        const double* src = Ho + (size_t)u * 36;
        if (pa <= pb) { auto& blk = R.H[{pa, pb}]; for (int k = 0; k < 36; ++k) blk[k] += src[k]; }
        else { auto& blk = R.H[{pb, pa}]; for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) blk[r * 6 + c] += src[c * 6 + r]; }
      }
    }
  };
  // back-substitutes the points for the (unscaled) camera step; out3 += [model decrease, |dX|^2, |X|^2]
  auto bundle_step = [&](const std::vector<double>& step, double* out3) {
    StageTimer stage_timer_bs_("solve: point back-substitution / candidate cost (GPU)");
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      const int nd = (int)b.dev_to_pose.size();
      std::vector<double> dcam((size_t)nd * 6, 0.0);
      for (int d = 0; d < nd; ++d)
        for (int r = 0; r < 6; ++r) { const int i = idx(b.dev_to_pose[d], r); if (i >= 0) dcam[(size_t)d * 6 + r] = step[i]; }
      double o[3];
      e.Check(pvlm_ba_step(e.ctx(), b.set, b.loss ? b.loss->kind() : PVLM_LOSS_NONE, b.loss ? b.loss->a() : 0.0, dcam.data(), o), "pvlm_ba_step");
      for (int k = 0; k < 3; ++k) out3[k] += o[k];
    }
  };
  auto bundle_cost = [&](const std::vector<double>& v, bool candidate) {
    StageTimer stage_timer_bc_("solve: point back-substitution / candidate cost (GPU)");
    double c = 0.0;
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      bundle_poses(b, v);
      double ci = 0.0;
      e.Check(pvlm_ba_cost(e.ctx(), b.set, b.loss ? b.loss->kind() : PVLM_LOSS_NONE, b.loss ? b.loss->a() : 0.0, candidate ? 1 : 0, &ci), "pvlm_ba_cost");
      c += ci;
    }
    return c;
  };
  auto finish_points = [&]() {   // the refined structure goes back into the caller's point blocks
    for (auto& b : I.bundles) {
      if (!b.set) continue;
      std::vector<double> X(b.point_blocks.size() * 3);
      e.Check(pvlm_ba_get_points(e.ctx(), b.set, 0, X.data()), "pvlm_ba_get_points");
      for (size_t p = 0; p < b.point_blocks.size(); ++p) for (int k = 0; k < 3; ++k) I.blocks[b.point_blocks[p]][k] = X[3 * p + k];
    }
  };

  double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
  Assembled A;
  evaluate(x, true, A);
  Reduced R; R.g_red.assign(n_free, 0.0); R.g_cam.assign(n_free, 0.0); R.Udiag.assign(n_free, 0.0);
  bool R_valid = true;
  if (have_bundles) bundle_reduce(x, radius, true, R);
  double cost = A.cost + R.cost;
  summary->initial_cost = summary->final_cost = cost;
  summary->cost_history.push_back(cost);
  summary->num_successful_steps = 1;  // iteration 0 counts as successful in Ceres' summary ([recalled])
  summary->usable = std::isfinite(cost);
  if (!summary->usable) { summary->message = "initial cost is not finite"; return; }
  if (n_free == 0 && !have_bundles) return;

  // envelope of the camera/LiDAR system (four-block groups + Schur complement blocks)
  auto build_first = [&](const Assembled& As, const Reduced& Rs) {
    std::vector<int> first(n_free);
    for (int i = 0; i < n_free; ++i) first[i] = i;
    auto add = [&](const std::pair<int, int>& key) {
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        const int i = idx(key.first, r), j = idx(key.second, c);
        if (i < 0 || j < 0) continue;
        const int hi = std::max(i, j), lo = std::min(i, j);
        first[hi] = std::min(first[hi], lo);
      }
    };
    ForEachBlock(As, [&](const std::pair<int, int>& key, const double*) { add(key); });
    ForEachBlock(Rs.H, [&](const std::pair<int, int>& key, const double*) { add(key); });
    return first;
  };
  // diagonal of the FULL J^T J on the free pose columns (before any elimination)
  auto full_diag = [&](const Assembled& As, const Reduced& Rs) {
    std::vector<double> d(n_free, 0.0);
    ForEachBlock(As, [&](const std::pair<int, int>& key, const double* blk) {
      if (key.first != key.second) return;
      for (int r = 0; r < 6; ++r) { const int i = idx(key.first, r); if (i >= 0) d[i] += blk[r * 6 + r]; }   // += : two poses may share a parameter block
    });
    for (int i = 0; i < n_free; ++i) d[i] += Rs.Udiag[i];
    return d;
  };

  // Jacobi scaling from the initial Jacobian: 1 / (1 + sqrt(diag(J^T J)))   (Ceres jacobi_scaling)
  std::vector<double> scale(n_free, 1.0);
  {
    const std::vector<double> d0 = full_diag(A, R);
    for (int i = 0; i < n_free; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(std::max(0.0, d0[i])));
  }

  int iter = 0;
  auto gmax = [&](const Assembled& As, const Reduced& Rs) {
    double m = Rs.gmax_points;
    for (int i = 0; i < n_free; ++i) m = std::max(m, std::fabs(As.g[i] + Rs.g_cam[i]));
    return m;
  };
  if (gmax(A, R) <= opt.gradient_tolerance) { summary->message = "gradient tolerance reached"; finish_points(); return; }
  auto fill = [&](Skyline& S, const auto& H) {
    ForEachBlock(H, [&](const std::pair<int, int>& key, const double* blk) {
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        const int i = idx(key.first, r), j = idx(key.second, c);
        if (i < 0 || j < 0) continue;
        const double v = blk[r * 6 + c] * scale[i] * scale[j];
        if (key.first == key.second) { if (i >= j) S.at(i, j) += v; }   // diagonal block: lower triangle once
        else if (i >= j) S.at(i, j) += v; else S.at(j, i) += v;
      }
    });
  };
  // Large reduced systems (Room / Floor sized joint problems: thousands of unknowns) are assembled, factorised and solved
  // on the GPU (pvlm_spd_solve_blocks: blocked Cholesky kernels); small ones by the host skyline Cholesky.
  const char* gpu_min_env = std::getenv("PVLM_GPU_CHOLESKY_MIN");
  const bool gpu_chol = n_free >= (gpu_min_env ? std::atoi(gpu_min_env) : 1500);
  // v^T (D H D) v over a block list, without forming the matrix
  auto quad_form = [&](const auto& H, const std::vector<double>& v) {
    double q = 0.0;
    ForEachBlock(H, [&](const std::pair<int, int>& key, const double* blk) {
      double b = 0.0;
      for (int r = 0; r < 6; ++r) {
        const int i = idx(key.first, r);
        if (i < 0) continue;
        double row = 0.0;
        for (int c = 0; c < 6; ++c) { const int j = idx(key.second, c); if (j >= 0) row += blk[r * 6 + c] * scale[j] * v[j]; }
        b += scale[i] * v[i] * row;
      }
      q += key.first == key.second ? b : 2.0 * b;
    });
    return q;
  };
  while (iter < opt.max_num_iterations) {
    ++iter;
    if (!R_valid) { bundle_reduce(x, radius, false, R); R_valid = true; }
    // scaled system  (D (H + S_points) D + diag(clamp(diag(D J^T J D))) / radius) dy = -D g
    const std::vector<double> dfull = full_diag(A, R);
    std::vector<double> rhs(n_free), damp(n_free);
    for (int i = 0; i < n_free; ++i) {
      rhs[i] = -(A.g[i] + R.g_red[i]) * scale[i];
      const double hs = dfull[i] * scale[i] * scale[i];
      damp[i] = std::min(std::max(hs, opt.min_lm_diagonal), opt.max_lm_diagonal) / radius;
    }
    bool step_ok;
    std::vector<double> dy = rhs;
    double model_change = 0.0, dn = 0.0, xn = 0.0;
    Skyline S0;                      // four-block groups only: the Gauss-Newton model of those blocks
    if (gpu_chol) {
      StageTimer stage_timer_chol_("solve: GPU Cholesky");
      StageTimer* stage_timer_push_ = new StageTimer("  (inside the GPU Cholesky stage) host block list");
      std::vector<int> rows, cols, mirror; std::vector<double> blocks;
      auto push = [&](const auto& H) {
        ForEachBlock(H, [&](const std::pair<int, int>& key, const double* blk) {
          for (int r = 0; r < 6; ++r) { rows.push_back(idx(key.first, r)); cols.push_back(idx(key.second, r)); }
          mirror.push_back(key.first != key.second ? 1 : 0);
          blocks.insert(blocks.end(), blk, blk + 36);
        });
      };
      push(A);
      if (have_bundles) push(R.H);
      delete stage_timer_push_;
      int info = 0;
      e.Check(pvlm_spd_solve_blocks(e.ctx(), n_free, (int)mirror.size(), rows.data(), cols.data(), mirror.data(), blocks.data(), scale.data(), damp.data(),
                                    dy.data(), &info), "pvlm_spd_solve_blocks");
      step_ok = info == 0;
    } else {
      S0.Init(build_first(A, R));
      fill(S0, A);
      Skyline S = S0;
      if (have_bundles) fill(S, R.H);
      for (int i = 0; i < n_free; ++i) S.at(i, i) += damp[i];
      { StageTimer stage_timer_chol_("solve: host skyline Cholesky"); step_ok = n_free == 0 || S.Factor(); }
      if (step_ok && n_free) S.Solve(dy);
    }
    if (step_ok) {
      // model_cost_change = -(g'.dy + 1/2 dy^T H' dy) over the four-block groups ...
      double gd = 0.0, dHd = 0.0;
      for (int i = 0; i < n_free; ++i) gd += (A.g[i] * scale[i]) * dy[i];
      if (gpu_chol) {
        std::vector<double> unit(n_free);
        for (int i = 0; i < n_free; ++i) unit[i] = dy[i];
        dHd = quad_form(A, unit);
      } else {
        for (int i = 0; i < n_free; ++i) {
          double s = 0.0;
          for (int k = S0.first[i]; k < i; ++k) s += S0.at(i, k) * dy[k];
          dHd += dy[i] * (2.0 * s + S0.at(i, i) * dy[i]);
        }
      }
      model_change = -(gd + 0.5 * dHd);
      // ... plus the reprojection blocks' own model decrease after back-substituting their points
      if (have_bundles) {
        std::vector<double> step(n_free);
        for (int i = 0; i < n_free; ++i) step[i] = dy[i] * scale[i];
        double o3[3] = {0, 0, 0};
        bundle_step(step, o3);
        model_change += o3[0]; dn += o3[1]; xn += o3[2];
      }
      step_ok = model_change > 0.0 && std::isfinite(model_change);
    }
    bool accepted = false;
    if (step_ok) {
      std::vector<double> cand = x;
      for (size_t b = 0; b < I.blocks.size(); ++b)
        if (block_off[b] >= 0) for (int k = 0; k < 3; ++k) { const double d = dy[block_off[b] + k] * scale[block_off[b] + k]; cand[3 * b + k] += d; dn += d * d; xn += x[3 * b + k] * x[3 * b + k]; }
      Assembled C;
      evaluate(cand, true, C);
      const double ccost = C.cost + (have_bundles ? bundle_cost(cand, true) : 0.0);
      const double rho = (cost - ccost) / model_change;
      if (opt.minimizer_progress_to_stdout)
        printf("iter %2d cost %.8e -> %.8e  model %.3e rho %.3f radius %.3e\n", iter, cost, ccost, model_change, rho, radius);
      if (std::isfinite(ccost) && rho > opt.min_relative_decrease) {
        accepted = true;
        const double cost_change = cost - ccost;
        x = cand; A = std::move(C);
        for (auto& b : I.bundles) if (b.set) e.Check(pvlm_ba_accept(e.ctx(), b.set), "pvlm_ba_accept");
        const double f = 1.0 - std::pow(2.0 * rho - 1.0, 3);
        radius = std::min(opt.max_trust_region_radius, radius / std::max(1.0 / 3.0, f));
        decrease_factor = 2.0;
        if (have_bundles) { bundle_reduce(x, radius, false, R); R_valid = true; }   // gradient at the new point + next system
        summary->num_successful_steps++;
        const double prev = cost;
        cost = ccost;
        summary->cost_history.push_back(cost);
        if (std::fabs(cost_change) <= opt.function_tolerance * prev) { summary->message = "function tolerance reached"; break; }
        if (gmax(A, R) <= opt.gradient_tolerance) { summary->message = "gradient tolerance reached"; break; }
        if (std::sqrt(dn) <= opt.parameter_tolerance * (std::sqrt(xn) + opt.parameter_tolerance)) { summary->message = "parameter tolerance reached"; break; }
      }
    }
    if (!accepted) {
      summary->num_unsuccessful_steps++;
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      R_valid = !have_bundles;
      if (radius < opt.min_trust_region_radius) { summary->message = "trust region collapsed"; break; }
    }
  }
  if (summary->message.empty()) summary->message = "maximum number of iterations reached";
  store_x(x);
  finish_points();
  summary->final_cost = cost;
  summary->usable = std::isfinite(cost);
}

}  // namespace ceres_like

// ================================================================================================
// functor factories — base/CostFunction.h ::Create
// ================================================================================================
using ceres_like::CostFunction;
static CostFunction* MakeCost(int kind, unsigned flags, double weight, std::initializer_list<double> row) {
  CostFunction* c = new CostFunction();
  c->kind = kind; c->flags = flags; c->weight = weight; c->row.assign(row.begin(), row.end());
  return c;
}
CostFunction* Point2Plane_Meter::Create(const Vector3d& p, const Vector4d& pl, const double w) {
  return MakeCost(PVLM_POINT2PLANE_METER, 0, w, {p[0], p[1], p[2], pl[0], pl[1], pl[2], pl[3]});
}
CostFunction* Point2Plane_Angle::Create(const Vector3d& p, const Vector4d& pl, const bool normalize, const double w) {
  return MakeCost(PVLM_POINT2PLANE_ANGLE, normalize ? PVLM_FLAG_NORMALIZE_DISTANCE : 0, w, {p[0], p[1], p[2], pl[0], pl[1], pl[2], pl[3]});
}
CostFunction* Point2Line_Meter::Create(const Vector3d& p, const Vector3d& a, const Vector3d& b, const double w) {
  return MakeCost(PVLM_POINT2LINE_METER, 0, w, {p[0], p[1], p[2], a[0], a[1], a[2], b[0], b[1], b[2]});
}
CostFunction* Point2Line_Angle::Create(const Vector3d& p, const Vector3d& a, const Vector3d& b, const bool normalize, const double w) {
  return MakeCost(PVLM_POINT2LINE_ANGLE, normalize ? PVLM_FLAG_NORMALIZE_DISTANCE : 0, w, {p[0], p[1], p[2], a[0], a[1], a[2], b[0], b[1], b[2]});
}
CostFunction* Plane2Plane_Global::Create(const Vector3d& n, const Vector3d& a, const Vector3d& b, const double w) {
  return MakeCost(PVLM_PLANE2PLANE_GLOBAL, 0, 1.0, {n[0], n[1], n[2], a[0], a[1], a[2], b[0], b[1], b[2], w});
}
CostFunction* PanoramaReprojResidual_1Angle::Create(const Vector3d& pt, double w) {
  CostFunction* c = MakeCost(kReprojKind, 0, w, {pt[0], pt[1], pt[2]});
  c->num_blocks = 3;
  return c;
}
CostFunction* PlaneIOUResidual::Create(const Vector4d& pl, const Vector3d& mn, const Vector3d& mr, const double angle, const double w) {
  return MakeCost(PVLM_PLANE_IOU, 0, 1.0, {pl[0], pl[1], pl[2], pl[3], mn[0], mn[1], mn[2], mr[0], mr[1], mr[2], angle, w});
}

// ================================================================================================
// util/Optimization.cpp adders
// ================================================================================================
size_t AddLidarPointToPlaneResidual(const std::vector<std::vector<int>>& neighbors, const std::vector<Velodyne>& lidars,
                                    std::vector<Vector3d>& aa_list, std::vector<Vector3d>& t_list, ceres_like::Problem& problem,
                                    double point_to_plane_dis_threshold, double plane_tolerance, bool angle_residual, bool normalized_distance,
                                    double weight, const std::pair<size_t, size_t>* ref_range) {
  StageTimer stage_timer_("point-to-plane association");
  // util/Optimization.cpp:513-517: one loss object shared by every block of this adder
  ceres_like::LossFunction* loss = new ceres_like::HuberLoss(angle_residual ? 2 * M_PI / 180.0 : 0.2);
  std::vector<pvlm_scan*> refs, neis;
  std::vector<const Velodyne*> holders;
  const size_t i_lo = ref_range ? ref_range->first : 0, i_hi = ref_range ? std::min(ref_range->second, lidars.size()) : lidars.size();
  for (int pass = 0; pass < 2; ++pass) {       // pass 0: which scans take part (uploaded in one batch), pass 1: the pair list
    for (size_t i = i_lo; i < i_hi; i++) {
      if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;                 // :525-526
      for (int n_idx : neighbors[i]) {
        if (n_idx < 0 || n_idx == (int)i || n_idx >= (int)lidars.size()) continue;  // :531-532
        if (!lidars[n_idx].IsPoseValid()) continue;                                // :533-534
        if (!lidars[i].IsInWorldCoordinate() || !lidars[n_idx].IsInWorldCoordinate()) continue;  // CheckLidarCoordinate -> empty result
        if (pass == 0) { holders.push_back(&lidars[i]); holders.push_back(&lidars[n_idx]); }
        else { refs.push_back(lidars[i].DeviceScan()); neis.push_back(lidars[n_idx].DeviceScan()); }
      }
    }
    if (pass == 0) Velodyne::UploadBatch(holders);
  }
  // parameter blocks are looked up by lidars[i].id (:527-528,:541-542); DeviceScan() carries that id
  Engine& e = Engine::Default();
  pvlm_resset* rs = nullptr;
  e.Check(pvlm_assoc_point2plane(e.ctx(), (int)refs.size(), refs.data(), neis.data(), plane_tolerance, (float)point_to_plane_dis_threshold,
                                 angle_residual ? PVLM_POINT2PLANE_ANGLE : PVLM_POINT2PLANE_METER,
                                 normalized_distance ? PVLM_FLAG_NORMALIZE_DISTANCE : 0u, weight, &rs), "pvlm_assoc_point2plane");
  int64_t n = 0;
  pvlm_resset_info(rs, &n, nullptr, nullptr, nullptr);
  if (n == 0) { pvlm_resset_destroy(e.ctx(), rs); delete loss; return 0; }
  problem.AddResidualSet(rs, loss, &aa_list, &t_list);
  return (size_t)n;
}

size_t AddLidarPointToLineResidual(const std::vector<std::vector<int>>& neighbors, const std::vector<Velodyne>& lidars,
                                   std::vector<Vector3d>& aa_list, std::vector<Vector3d>& t_list, ceres_like::Problem& problem,
                                   double thr, bool use_segment, bool angle_residual, bool normalized_distance, double weight,
                                   const std::pair<size_t, size_t>* ref_range) {
  ceres_like::LossFunction* loss = new ceres_like::HuberLoss(angle_residual ? 2 * M_PI / 180.0 : 0.2);   // :449-453 (Huber for both variants here)
  size_t num = 0;
  const size_t i_lo = ref_range ? ref_range->first : 0, i_hi = ref_range ? std::min(ref_range->second, lidars.size()) : lidars.size();
  for (size_t i = i_lo; i < i_hi; i++) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    double* aa_r = aa_list[lidars[i].id].data(); double* t_r = t_list[lidars[i].id].data();
    for (int n_idx : neighbors[i]) {
      if (n_idx < 0 || n_idx == (int)i || n_idx >= (int)lidars.size()) continue;
      if (!lidars[n_idx].IsPoseValid()) continue;
      if (std::abs(n_idx - (int)i) > 1) continue;                                              // :475
      double* t_n = t_list[lidars[n_idx].id].data(); double* aa_n = aa_list[lidars[n_idx].id].data();
      const std::vector<Point2Line> ass = use_segment ? AssociatePoint2LineSegmentKNN(lidars[i], lidars[n_idx], (float)thr)
                                                      : AssociatePoint2Line(lidars[i], lidars[n_idx], (float)thr);
      for (const Point2Line& a : ass) {
        if (angle_residual) problem.AddResidualBlock(Point2Line_Angle::Create(a.point, a.line_point1, a.line_point2, normalized_distance, weight), loss, aa_r, t_r, aa_n, t_n);
        else problem.AddResidualBlock(Point2Line_Meter::Create(a.point, a.line_point1, a.line_point2, weight), loss, aa_r, t_r, aa_n, t_n);
        num++;
      }
    }
  }
  if (num == 0) delete loss;
  return num;
}

size_t AddLidarLineToLineResidual2(const std::vector<std::vector<int>>& neighbors, const std::vector<Velodyne>& lidars,
                                   std::vector<Vector3d>& aa_list, std::vector<Vector3d>& t_list, ceres_like::Problem& problem,
                                   const std::vector<LineTrack>& tracks, double thr, bool angle_residual, bool normalized_distance, double weight,
                                   const std::pair<size_t, size_t>* ref_range) {
  StageTimer stage_timer_("line-to-line association + blocks");
  const size_t i_lo = ref_range ? ref_range->first : 0, i_hi = ref_range ? std::min(ref_range->second, lidars.size()) : lidars.size();
  ceres_like::LossFunction* loss = new ceres_like::HuberLoss(angle_residual ? 2 * M_PI / 180.0 : 0.2);
  // only looked up, never iterated (upstream: std::map, :369-377): key = lidar id << 32 | line id
  std::unordered_map<unsigned long long, std::vector<uint32_t>> lines_to_track;
  auto line_key = [](uint32_t lidar, uint32_t line) { return ((unsigned long long)lidar << 32) | line; };
  {
    StageTimer stage_timer_l2t_("  (inside) line-to-line: lines_to_track map (host)");
    size_t n_keys = 0;
    for (const LineTrack& t : tracks) n_keys += t.feature_pairs.size();
    lines_to_track.reserve(n_keys);
    for (const LineTrack& t : tracks) for (const auto& pr : t.feature_pairs) lines_to_track[line_key(pr.first, pr.second)].push_back(t.id);
  }
  size_t num = 0;
  // all AssociateLine2Line(lidars[i], lidars[n_idx], thr) calls of the loop below (:379) in one GPU launch
  std::vector<std::pair<const Velodyne*, const Velodyne*>> todo;
  for (size_t i = i_lo; i < i_hi; i++) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    for (int n_idx : neighbors[i]) {
      if (n_idx < 0 || n_idx == (int)i || n_idx >= (int)lidars.size()) continue;
      if (!lidars[n_idx].IsPoseValid() || !lidars[n_idx].valid) continue;
      todo.push_back({&lidars[i], &lidars[n_idx]});
    }
  }
  const std::vector<std::vector<Line2Line>> all_ass = AssociateLine2LineBatch(todo, (float)thr);
  // The association + track filter decide WHICH (neighbour segment, reference segment) pairs contribute (:379-400); the
  // blocks themselves — one per point of the neighbour segment, :410-434 — are built on the GPU from the scans' segment
  // point lists (pvlm_line2line_residuals): same rows in the same order as the X::Create + AddResidualBlock calls,
  // without 600 k heap objects, a host SoA staging copy and a 180 MB upload per outer iteration at Room scale.
  std::vector<pvlm_scan*> refs, neis;
  std::vector<int> m_pair, m_nei, m_ref;
  size_t next = 0;
  StageTimer* stage_timer_filter_ = new StageTimer("  (inside) line-to-line: track filter of the matches (host)");
  {
    // which matches of a pair survive is a read-only question to the track tables: pair-parallel; the lists are then joined in pair order
    struct PairTodo { size_t i; int n_idx; };
    std::vector<PairTodo> pairs_todo;
    for (size_t i = i_lo; i < i_hi; i++) {
      if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
      for (int n_idx : neighbors[i]) {
        if (n_idx < 0 || n_idx == (int)i || n_idx >= (int)lidars.size()) continue;
        if (!lidars[n_idx].IsPoseValid() || !lidars[n_idx].valid) continue;
        pairs_todo.push_back({i, n_idx});
      }
    }
    std::vector<std::vector<std::pair<int, int>>> kept(pairs_todo.size());       // (neighbour line, reference line) per pair
    std::vector<size_t> kept_points(pairs_todo.size(), 0);
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({pvlm_thread_cap(), pairs_todo.size() / 256 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
    std::atomic<size_t> cursor{0};
    auto work = [&]() {
      for (size_t p = cursor++; p < pairs_todo.size(); p = cursor++) {
        const size_t i = pairs_todo[p].i; const int n_idx = pairs_todo[p].n_idx;
        for (const Line2Line& a : all_ass[next + p]) {
          auto it = lines_to_track.find(line_key((uint32_t)i, (uint32_t)a.ref_line_idx));
          if (it == lines_to_track.end()) continue;
          bool valid = false;
          for (uint32_t tid : it->second) if (tracks[tid].IsInside({(uint32_t)n_idx, (uint32_t)a.neighbor_line_idx})) { valid = true; break; }
          if (!valid) continue;
          const size_t pts = lidars[n_idx].edge_segmented[a.neighbor_line_idx].size();
          if (pts == 0) continue;
          kept[p].push_back({a.neighbor_line_idx, a.ref_line_idx});
          kept_points[p] += pts;
        }
      }
    };
    { StageTimer stage_timer_w_("    (inside the track filter) pair-parallel lookups"); pvlm_run_workers(n_threads, work); }
    StageTimer stage_timer_j_("    (inside the track filter) join in pair order");
    for (size_t p = 0; p < pairs_todo.size(); ++p) {
      if (kept[p].empty()) continue;
      refs.push_back(lidars[pairs_todo[p].i].DeviceScan()); neis.push_back(lidars[(size_t)pairs_todo[p].n_idx].DeviceScan());
      for (const std::pair<int, int>& m : kept[p]) { m_pair.push_back((int)refs.size() - 1); m_nei.push_back(m.first); m_ref.push_back(m.second); }
      num += kept_points[p];
    }
    next += pairs_todo.size();
  }
  delete stage_timer_filter_;
  if (num == 0) { delete loss; return 0; }
  Engine& e = Engine::Default();
  pvlm_resset* rs = nullptr;
  StageTimer stage_timer_rows_("  (inside) line-to-line: blocks built on the GPU (pvlm_line2line_residuals)");
  e.Check(pvlm_line2line_residuals(e.ctx(), (int)refs.size(), refs.data(), neis.data(), (int)m_pair.size(), m_pair.data(), m_nei.data(), m_ref.data(),
                                   angle_residual ? PVLM_POINT2LINE_ANGLE : PVLM_POINT2LINE_METER,
                                   (angle_residual && normalized_distance) ? PVLM_FLAG_NORMALIZE_DISTANCE : 0u, weight, &rs), "pvlm_line2line_residuals");
  // loss is nullptr for the angle variant (util/Optimization.cpp:417), Huber for the metric one
  if (angle_residual) { delete loss; loss = nullptr; }
  problem.AddResidualSet(rs, loss, &aa_list, &t_list);
  return num;
}

ceres_like::Solver::Options SetOptionsSfM(const int num_threads) {
  // util/Optimization.cpp:608-634: Ceres defaults (50 iterations, LM) with a *_SCHUR linear solver — the point
  // blocks are eliminated, which is what Solve does for the reprojection sets on the GPU.
  ceres_like::Solver::Options o;
  o.minimizer_progress_to_stdout = false;
  o.num_threads = num_threads;
  o.linear_solver_type = ceres_like::SPARSE_SCHUR;
  return o;
}

ceres_like::Solver::Options SetOptionsLidar(const int num_threads, const int lidar_size) {
  ceres_like::Solver::Options o;
  o.minimizer_progress_to_stdout = false;
  o.linear_solver_type = lidar_size <= 50 ? ceres_like::DENSE_SCHUR : (lidar_size <= 2000 ? ceres_like::SPARSE_SCHUR : ceres_like::ITERATIVE_SCHUR);
  o.num_threads = num_threads;
  o.max_num_iterations = 20;
  o.max_linear_solver_iterations = 100;
  return o;
}

// ================================================================================================
// Exchange factories
// ================================================================================================
Exchange MakeRcclExchange(int world, int rank, const unsigned char id[128]) {
  Engine& e = Engine::Default();
  struct State {
    pvlm_comm* comm = nullptr;
    ~State() { if (comm) pvlm_comm_destroy(Engine::Default().ctx(), comm); }
  };
  auto st = std::make_shared<State>();
  e.Check(pvlm_comm_create(e.ctx(), world, rank, id, &st->comm), "pvlm_comm_create");
  Exchange x; x.world = world; x.rank = rank;
  x.allreduce_sum = [st](double* buf, size_t count) {
    Engine& en = Engine::Default();
    en.Check(pvlm_allreduce_sum_f64_host(en.ctx(), st->comm, buf, (int64_t)count), "pvlm_allreduce_sum_f64_host");
  };
  return x;
}

std::pair<size_t, size_t> Exchange::BalancedRange(const std::vector<double>& weight, int of_rank) const {
  const size_t n = weight.size();
  const size_t rk = (size_t)(of_rank < 0 ? rank : of_rank), W = (size_t)std::max(world, 1);
  double total = 0;
  for (double w : weight) total += w;
  if (!(total > 0)) return {n * rk / W, n * (rk + 1) / W};
  auto boundary = [&](size_t r) -> size_t {
    if (r == 0) return 0;
    if (r >= W) return n;
    const double want = total * (double)r / (double)W;
    double acc = 0;
    for (size_t i = 0; i < n; ++i) { if (acc >= want) return i; acc += weight[i]; }
    return n;
  };
  return {boundary(rk), boundary(rk + 1)};
}

Exchange MakeFileExchange(int world, int rank, const std::string& dir) {
  auto seq = std::make_shared<long>(0);
  Exchange x; x.world = world; x.rank = rank;
  x.allreduce_sum = [world, rank, dir, seq](double* buf, size_t count) {
    const long s = (*seq)++;
    auto name = [&](long q, int r) { return dir + "/x" + std::to_string(q) + "_" + std::to_string(r) + ".bin"; };
    {
      const std::string tmp = name(s, rank) + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f) throw std::runtime_error("file exchange: cannot write " + tmp);
      const uint64_t n = count;
      fwrite(&n, sizeof(n), 1, f); fwrite(buf, sizeof(double), count, f);
      fclose(f);
      if (rename(tmp.c_str(), name(s, rank).c_str()) != 0) throw std::runtime_error("file exchange: rename failed");
    }
    std::vector<double> sum(count, 0.0), part(count);
    for (int r = 0; r < world; ++r) {             // rank order: the same sum, bit for bit, on every rank
      FILE* f = nullptr;
      for (int spin = 0; spin < 1200000 && !(f = fopen(name(s, r).c_str(), "rb")); ++spin) std::this_thread::sleep_for(std::chrono::microseconds(100));
      if (!f) throw std::runtime_error("file exchange: rank " + std::to_string(r) + " never arrived at exchange " + std::to_string(s));
      uint64_t n = 0;
      if (fread(&n, sizeof(n), 1, f) != 1 || n != count || fread(part.data(), sizeof(double), count, f) != count) { fclose(f); throw std::runtime_error("file exchange: size mismatch between ranks"); }
      fclose(f);
      for (size_t i = 0; i < count; ++i) sum[i] += part[i];
    }
    std::copy(sum.begin(), sum.end(), buf);
    if (s >= 2) remove(name(s - 2, rank).c_str());   // every rank has read exchange s-2 before it wrote s-1, and all of s-1 has been read here
  };
  return x;
}

// ================================================================================================
// LidarOdometry — lidar_mapping/LidarOdometry.cpp:15-187
// ================================================================================================
bool LidarOdometry::RefinePose(double& cost, int& steps, bool use_segment) {
  {
    std::vector<Velodyne*> all;
    for (Velodyne& l : lidars) if (l.IsPoseValid() && !l.IsInWorldCoordinate()) all.push_back(&l);
    Velodyne::TransformBatch(all, true, config.num_threads);
  }
  std::vector<Vector3d> aa_list(lidars.size(), Vector3d{1, 1, 1}), t_list(lidars.size(), Vector3d{1, 1, 1});
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    const Matrix3d& R = lidars[i].GetRotation();
    const Matrix3d R_lw = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    RotationMatrixToAngleAxis(R_lw, &aa_list[i]);
    const Vector3d rt = MatVec(R_lw, lidars[i].GetTranslation());
    t_list[i] = {-rt[0], -rt[1], -rt[2]};
  }
  const std::vector<std::vector<int>> neighbors_all = FindNeighbors(lidars, 6);
  ceres_like::Problem problem;
  // sharded run: this rank adds the blocks of its reference scans only; pose ids are the list indices on every rank
  const bool sharded = exchange_.active();
  // contiguous ranges of reference scans with equal association work: weight(i) = queries of i's pairs (+ the corner points of
  // the line term).  The lists are replicated, so every rank derives the same partition.
  std::vector<double> shard_weight(lidars.size(), 0.0);
  if (sharded)
    for (size_t i = 0; i < lidars.size(); i++) {
      if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
      for (int k : neighbors_all[i]) {
        if (k < 0 || k == (int)i || k >= (int)lidars.size() || !lidars[(size_t)k].valid || !lidars[(size_t)k].IsPoseValid()) continue;
        if (config.point_to_plane_residual) shard_weight[i] += (double)lidars[(size_t)k].surfFlat.size();
        if (config.line_to_line_residual && use_segment) shard_weight[i] += (double)lidars[(size_t)k].cornerLessSharp.size();
      }
    }
  const std::pair<size_t, size_t> my_range = sharded ? exchange_.BalancedRange(shard_weight) : std::pair<size_t, size_t>{0, lidars.size()};
  const std::pair<size_t, size_t>* range = sharded ? &my_range : nullptr;
  if (sharded) problem.RegisterPoses(aa_list, t_list);
  if (config.point_to_line_residual)                                                           // LidarOdometry.cpp:38-41
    AddLidarPointToLineResidual(neighbors_all, lidars, aa_list, t_list, problem, config.point_to_line_dis_threshold, use_segment,
                                config.angle_residual, config.normalize_distance, 1.0, range);
  if (config.line_to_line_residual && use_segment) {
    LidarLineMatch matcher(lidars);
    matcher.SetNeighborSize(4);
    matcher.SetMinTrackLength(3);
    if (sharded) matcher.SetShard(&exchange_, my_range.first, my_range.second);
    matcher.GenerateTracks();
    AddLidarLineToLineResidual2(neighbors_all, lidars, aa_list, t_list, problem, matcher.GetTracks(), config.point_to_line_dis_threshold,
                                config.angle_residual, config.normalize_distance, 1.0, range);
  }
  if (config.point_to_plane_residual)
    AddLidarPointToPlaneResidual(neighbors_all, lidars, aa_list, t_list, problem, config.point_to_plane_dis_threshold, config.lidar_plane_tolerance,
                                 config.angle_residual, config.normalize_distance, 1.0, range);
  double total_blocks = (double)problem.NumResidualBlocks();
  if (sharded) {
    ShardLog sl{my_range.first, my_range.second, std::vector<double>((size_t)exchange_.world, 0.0), (int)problem.NumResidualBlocks()};
    for (int r = 0; r < exchange_.world; ++r) {
      const auto rg = exchange_.BalancedRange(shard_weight, r);
      for (size_t i = rg.first; i < rg.second; ++i) sl.queries_per_rank[(size_t)r] += shard_weight[i];
    }
    shard_log.push_back(sl);
  }
  if (sharded) exchange_.allreduce_sum(&total_blocks, 1);        // the decision below must be the same on every rank
  if (total_blocks == 0) { fprintf(stderr, "no residual\n"); return false; }
  // gauge: first valid pose constant — only if it takes part in the problem (Ceres would abort otherwise)
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    problem.SetParameterBlockConstant(aa_list[i].data());
    problem.SetParameterBlockConstant(t_list[i].data());
    break;
  }
  ceres_like::Solver::Options options = SetOptionsLidar(config.num_threads, (int)lidars.size());
  if (sharded) options.exchange = &exchange_;
  ceres_like::Solver::Summary summary;
  ceres_like::Solve(options, &problem, &summary);
  {
    std::vector<Velodyne*> all;
    for (Velodyne& l : lidars) if (l.valid && l.IsPoseValid()) all.push_back(&l);
    Velodyne::TransformBatch(all, false, config.num_threads);      // with the poses the clouds were posed with: before the setters below
  }
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].valid || !lidars[i].IsPoseValid()) continue;
    Matrix3d R_lw;
    AngleAxisToRotationMatrix(aa_list[i], &R_lw);
    const Matrix3d R_wl = {R_lw[0], R_lw[3], R_lw[6], R_lw[1], R_lw[4], R_lw[7], R_lw[2], R_lw[5], R_lw[8]};
    const Vector3d rt = MatVec(R_wl, t_list[i]);
    lidars[i].SetRotation(R_wl);
    lidars[i].SetTranslation({-rt[0], -rt[1], -rt[2]});
  }
  cost = summary.final_cost;
  steps = summary.num_successful_steps;
  log.push_back({cost, steps, (int)total_blocks});
  return summary.IsSolutionUsable();
}

bool LidarOdometry::EstimatePose(const int max_iteration) {
  // lidar_mapping/LidarOdometry.cpp:131-147: features are extracted once, scan-parallel (omp there, std::thread here).
  // Scans that arrive with their feature clouds (or without raw points) are left alone; upstream's ReOrderVLP /
  // ExtractFeatures return early for those as well (sensors/Velodyne.cpp:376-377, :542-543).
  {
    StageTimer stage_timer_features_("feature extraction (range-image stages on the GPU, picks on the host)");
    // invalid scans first, on the calling thread: SetRotation / SetTranslation give the scan's device copy back to the engine's
    // context (InvalidateDevice -> pvlm_scan_destroy), whose pool is not thread-safe — never from the workers below
    for (Velodyne& l : lidars)
      if (!l.valid || !l.IsPoseValid()) { l.SetRotation({0, 0, 0, 0, 0, 0, 0, 0, 0}); l.SetTranslation({INFINITY, INFINITY, INFINITY}); }
    // the scans that still need their features: range-image stages of all of them in one GPU batch, picks on config.num_threads
    // host threads (Velodyne::ExtractFeaturesBatch).  PVLM_HOST_FEATURES=1: everything on the host, scan by scan, as upstream does.
    std::vector<Velodyne*> need;
    for (Velodyne& l : lidars) {
      if (!l.valid || !l.IsPoseValid()) continue;            // reset above
      if (!l.cloud.empty() && l.surfFlat.empty() && l.surfLessFlat.empty() && l.cornerLessSharp.empty() && !l.IsInWorldCoordinate()) need.push_back(&l);
    }
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::max(config.num_threads, 1), std::max<size_t>(need.size(), 1), (size_t)std::max(1u, std::thread::hardware_concurrency())}));
    if (!need.empty() && !std::getenv("PVLM_HOST_FEATURES")) {
      Velodyne::ExtractFeaturesBatch(need, config.max_curvature, config.intersection_angle_threshold, config.extraction_method, config.lidar_segmentation, true, (int)n_threads);
    } else if (!need.empty()) {
      std::atomic<size_t> next{0};
      std::mutex failure_lock;
      std::exception_ptr failure;          // e.g. an extraction method that is not mirrored: rethrown on the calling thread
      auto work = [&]() {
        for (size_t i = next++; i < need.size(); i = next++) {
          try {
            need[i]->ReOrderVLP();
            need[i]->ExtractFeatures(config.max_curvature, config.intersection_angle_threshold, config.extraction_method, config.lidar_segmentation);
          } catch (...) {
            std::lock_guard<std::mutex> g(failure_lock);
            if (!failure) failure = std::current_exception();
          }
        }
      };
      pvlm_run_workers(n_threads, work);
      if (failure) std::rethrow_exception(failure);
    }
  }
  {
    std::vector<Velodyne*> all;
    for (Velodyne& l : lidars) if (l.valid && l.IsPoseValid()) all.push_back(&l);
    Velodyne::TransformBatch(all, true, config.num_threads);
  }
  bool segmented = false;
  for (Velodyne& l : lidars) { segmented = !l.edge_segmented.empty(); if (segmented) break; }
  double curr_cost = 0, last_cost = 0;
  int curr_step = INT16_MAX, last_step = INT16_MAX;
  for (int iter = 0; iter < max_iteration; iter++) {
    RefinePose(curr_cost, curr_step, segmented);
    if (std::fabs(curr_cost - last_cost) / last_cost < 0.01) break;   // LidarOdometry.cpp:171-175
    if (curr_step < 5 && last_step < 5) break;                        // :176-180
    last_cost = curr_cost;
    last_step = curr_step;
  }
  return true;
}

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
template <typename T>
inline T FastAtan2(const T& y, const T& x) {
  T ax = std::abs(x), ay = std::abs(y);
  T a = std::min(ax, ay) / (std::max(ax, ay) + (T)DBL_EPSILON);
  T s = a * a;
  T r = ((-0.04432655554792128 * s + 0.1555786518463281) * s - 0.3258083974640975) * s * a + 0.9997878412794807 * a;
  if (ay > ax) r = M_PI_2 - r;
  if (x < 0) r = M_PI - r;
  if (y < 0) r = -r;
  return r;
}
struct Equirect {
  int cols, rows;
  template <typename T> void ImageToCam(const T* px, T r, T* cam) const {
    T sx = (2 * px[0] / cols - 1) * M_PI;
    T sy = (0.5 - px[1] / rows) * M_PI;
    T cy = (T)std::cos((double)sy);
    cam[0] = r * cy * (T)std::sin((double)sx);
    cam[1] = -r * (T)std::sin((double)sy);
    cam[2] = r * cy * (T)std::cos((double)sx);
  }
  template <typename T> void CamToImage(const T* cam, T* px) const {
    T lon = FastAtan2(cam[0], cam[2]);
    T lat = -FastAtan2(cam[1], (T)std::sqrt(cam[0] * cam[0] + cam[2] * cam[2]));
    px[0] = cols * (0.5 + lon / (2.0 * M_PI));
    px[1] = rows * (0.5 - lat / M_PI);
  }
  std::vector<float> BreakToSegments(const float* start, const float* end, float seg_length) const {
    float p1[3], p2[3];
    ImageToCam(start, 5.0f, p1);
    ImageToCam(end, 5.0f, p2);
    const float sl[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    const float length = std::sqrt((start[0] - end[0]) * (start[0] - end[0]) + (start[1] - end[1]) * (start[1] - end[1]));
    const int count = length / seg_length + 1;
    std::vector<float> seg = {start[0], start[1]};
    for (int i = 1; i < count; i++) {
      const float f = i * 1.f / count;
      const float p[3] = {p1[0] + f * sl[0], p1[1] + f * sl[1], p1[2] + f * sl[2]};
      float pixel[2];
      CamToImage(p, pixel);
      const float lastx = seg[seg.size() - 2];
      if (std::abs(pixel[0] - lastx) > 0.8 * cols) {
        const float gq = p1[0] / (p1[0] - p2[0]);
        const float q[3] = {p1[0] + gq * sl[0], p1[1] + gq * sl[1], p1[2] + gq * sl[2]};
        float left[2];
        CamToImage(q, left);
        left[0] = 0;
        const float right[2] = {float(cols - 1), left[1]};
        if (pixel[0] > lastx) { seg.insert(seg.end(), {left[0], left[1], right[0], right[1]}); }
        else { seg.insert(seg.end(), {right[0], right[1], left[0], left[1]}); }
      }
      seg.push_back(pixel[0]); seg.push_back(pixel[1]);
    }
    seg.push_back(end[0]); seg.push_back(end[1]);
    return seg;
  }
};
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
// AddCameraLidarResidual — util/Optimization.cpp:564-607
// ================================================================================================
size_t AddCameraLidarResidual(int rows, int cols, const std::vector<bool>& frame_pose_valid, const std::vector<Velodyne>& lidars,
                              std::vector<Vector3d>& aa_cw, std::vector<Vector3d>& t_cw, std::vector<Vector3d>& aa_lw, std::vector<Vector3d>& t_lw,
                              const std::map<std::pair<size_t, size_t>, std::vector<CameraLidarLinePair>>& line_pairs,
                              ceres_like::LossFunction* loss, ceres_like::Problem& problem, double weight) {
  size_t num = 0;
  Equirect eq{cols, rows};
  for (const auto& kv : line_pairs) {
    const size_t frame_id = kv.first.first, lidar_id = kv.first.second;
    if (!lidars[lidar_id].IsPoseValid() || !frame_pose_valid[frame_id]) continue;
    for (const CameraLidarLinePair& lp : kv.second) {
      const double a[2] = {lp.image_line[0], lp.image_line[1]}, b[2] = {lp.image_line[2], lp.image_line[3]};
      double p1[3], p2[3];
      eq.ImageToCam(a, 1.0, p1); eq.ImageToCam(b, 1.0, p2);
      // FormPlane(p1, p2, 0), NOT normalised here (the functor constructors normalise, CostFunction.h:361,459)
      const double p3[3] = {0, 0, 0};
      const double pa = ((p2[1] - p1[1]) * (p3[2] - p1[2]) - (p2[2] - p1[2]) * (p3[1] - p1[1]));
      const double pb = ((p2[2] - p1[2]) * (p3[0] - p1[0]) - (p2[0] - p1[0]) * (p3[2] - p1[2]));
      const double pc = ((p2[0] - p1[0]) * (p3[1] - p1[1]) - (p2[1] - p1[1]) * (p3[0] - p1[0]));
      const double pd = -(pa * p1[0] + pb * p1[1] + pc * p1[2]);
      // point order passed is (end, start)  (Optimization.cpp:592)
      problem.AddResidualBlock(Plane2Plane_Global::Create({pa, pb, pc}, lp.lidar_line_end, lp.lidar_line_start, lp.weight * weight), loss,
                               aa_cw[frame_id].data(), t_cw[frame_id].data(), aa_lw[lidar_id].data(), t_lw[lidar_id].data());
      // full arc angle, VectorAngle3D(p1, p2, normalized = true)  (:596)
      double c = p1[0] * p2[0] + p1[1] * p2[1] + p1[2] * p2[2];
      const double angle = c >= 1.0 ? 0.0 : (c <= -1.0 ? M_PI : std::acos(c));
      const Vector3d mid_l = {(lp.lidar_line_end[0] + lp.lidar_line_start[0]) / 2.0, (lp.lidar_line_end[1] + lp.lidar_line_start[1]) / 2.0,
                              (lp.lidar_line_end[2] + lp.lidar_line_start[2]) / 2.0};
      const Vector3d mid_i = {(p1[0] + p2[0]) / 2.0, (p1[1] + p2[1]) / 2.0, (p1[2] + p2[2]) / 2.0};
      problem.AddResidualBlock(PlaneIOUResidual::Create({pa, pb, pc, pd}, mid_l, mid_i, angle, 2.0 * weight), loss, aa_cw[frame_id].data(),
                               t_cw[frame_id].data(), aa_lw[lidar_id].data(), t_lw[lidar_id].data());
      num += 2;
    }
  }
  return num;
}

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

// ================================================================================================
// AddCameraResidual — util/Optimization.cpp:172-222 (ANGLE_RESIDUAL_1)
// ================================================================================================
size_t AddCameraResidual(const std::vector<Frame>& frames, std::vector<Vector3d>& angleAxis_cw_list, std::vector<Vector3d>& t_cw_list,
                         std::vector<PointTrack>& structure, ceres_like::Problem& problem, int residual_type, double weight) {
  StageTimer stage_timer_("reprojection blocks");
  if (residual_type != ANGLE_RESIDUAL_1)
    throw std::runtime_error("AddCameraResidual: only ANGLE_RESIDUAL_1 (the variant CameraLidarOptimizer::Optimize uses) is mirrored");
  if (frames.empty() || structure.empty()) return 0;
  const Equirect eq{frames[0].GetImageCols(), frames[0].GetImageRows()};           // :178
  ceres_like::LossFunction* loss_function = new ceres_like::HuberLoss(4.0 * M_PI / 180.0);   // :180-181
  size_t num_residual = 0;
  for (size_t i = 0; i < structure.size(); i++) {
    PointTrack& track = structure[i];
    for (const std::pair<uint32_t, uint32_t>& pair : track.feature_pairs) {
      const uint32_t frame_idx = pair.first;
      if (!frames[frame_idx].IsPoseValid()) continue;                                // :192-193
      // eq.ImageToCam(keypoint.pt) binds to ImageToCam(const cv::Point2i&): the float keypoint is converted with
      // saturate_cast<int> (= cvRound, round-half-even), then un-projected in float with r = 1 (Equirectangular.h:153-161)
      const std::array<float, 2>& kp = frames[frame_idx].keypoints[pair.second];
      const float px[2] = {(float)(int)std::lrintf(kp[0]), (float)(int)std::lrintf(kp[1])};
      float cam[3];
      eq.ImageToCam(px, 1.f, cam);
      ceres_like::CostFunction* cost_function = PanoramaReprojResidual_1Angle::Create({(double)cam[0], (double)cam[1], (double)cam[2]}, weight);
      problem.AddResidualBlock(cost_function, loss_function, angleAxis_cw_list[frame_idx].data(), t_cw_list[frame_idx].data(), track.point_3d.data());
      num_residual++;
    }
  }
  if (num_residual == 0) delete loss_function;
  return num_residual;
}

std::vector<std::vector<int>> CameraLidarOptimizer::NeighborEachFrame(const int neighbor_size, const bool temporal) const {
  std::vector<std::vector<int>> out(frames.size());
  if (!temporal) throw std::runtime_error("NeighborEachFrame: only the temporal branch (the one JointOptimize uses) is mirrored");
  for (int frame_id = 0; frame_id < (int)frames.size(); frame_id++) {
    int start = std::max(0, frame_id - (neighbor_size / 2));
    const int end = std::min((int)lidars.size(), start + neighbor_size);
    start = std::max(0, end - neighbor_size);
    for (int l = start; l < end; l++) out[frame_id].push_back(l);
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
    lidars[i].Transform2LidarWorld();
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
  for (size_t i = 0; i < lidars.size(); i++) {
    if (!lidars[i].IsPoseValid()) continue;
    if (lidars[i].IsInWorldCoordinate()) lidars[i].Transform2Local();
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
