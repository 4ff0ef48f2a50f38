// pvlm_host.hpp — C++ host side above the C ABI (include/pvlm.h), mirroring PanoVLM's own
// interfaces for the ICP hot path so that a PanoVLM maintainer (and the parity tests) can call it
// with the names, argument meaning and error behaviour of the reference:
//
//   lidar_mapping/LidarFeatureAssociate.h:22-125   Point2Line / Line2Line / Point2Plane,
//                                                  FindNeighbors, AssociatePoint2Plane, AssociateLine2Line
//   base/CostFunction.h:350-934                    Point2Plane_{Meter,Angle}, Point2Line_{Meter,Angle},
//                                                  Plane2Plane_Global, PlaneIOUResidual  (::Create)
//   util/Optimization.h:39-158                     AddLidarPointToPlaneResidual, AddLidarLineToLineResidual2,
//                                                  AddCameraLidarResidual, SetOptionsLidar
//   lidar_mapping/LidarLineMatch.h, util/Tracks.h  LidarLineMatch::GenerateTracks (line tracks)
//   lidar_mapping/LidarOdometry.h:40-81            LidarOdometry::RefinePose / EstimatePose
//   joint_optimization/CameraLidarLineAssociate.h  CameraLidarLineAssociate::AssociateByAngle
//   sensors/Velodyne.h:80-91,213-233               the Velodyne data contract + frame transforms
//   sensors/Equirectangular.h                      Equirectangular (host, scalar)
//
// Eigen / PCL / Ceres are not dependencies: small POD types stand in for Eigen::Vector3d etc. and
// the ceres:: classes the reference touches (CostFunction, LossFunction, Problem, Solver) are
// re-declared in namespace pvlm::ceres_like with the same member names.  All heavy work goes
// through libpvlm.so (HIP); nothing here computes a residual on the CPU.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pvlm.h"

namespace pvlm {

using Vector3d = std::array<double, 3>;
using Vector4d = std::array<double, 4>;
using Vector6d = std::array<double, 6>;
using Matrix3d = std::array<double, 9>;   // row-major
using Matrix4d = std::array<double, 16>;  // row-major

struct PointXYZI { float x, y, z, intensity; };  // pcl::PointXYZI payload
using PointCloud = std::vector<PointXYZI>;

// sensors/Velodyne.h:57-66
enum PointClassification { POINT_NORMAL = 0x01, POINT_LESS_SHARP = 0x02, POINT_SHARP = 0x04, POINT_FLAT = 0x08, POINT_GROUND = 0x10,
                           POINT_DISABLE = 0x20, POINT_OCCLUDED = 0x40 };
// sensors/Velodyne.h:50-55
enum ExtractionMethod { LOAM = 1, DOUBLE_EXTRACTION = 2, ADAPTIVE = 3 };

// Ring-ordered view of one scan, built by Velodyne::ReOrderVLP (private members of the reference's Velodyne,
// sensors/Velodyne.h:97-120) and the per-point working arrays of ExtractFeatures (freed by the reference at the end of
// the call; kept here when a trace is asked for, for the parity tests).
struct RingLayout {
  std::vector<float> range_image;                      // N_SCANS x horizon_scans, row-major; 0 = no return
  std::vector<std::pair<int, int>> point_idx_to_image; // cloud_scan index -> (ring, column)
  std::vector<int> image_to_point_idx;                 // (ring, column) -> cloud_scan index, -1 = empty
  std::vector<int> scanStartInd, scanEndInd;           // per ring: first + 5, last - 5 (inclusive)
};
struct ExtractionTrace {
  std::vector<float> curvature;
  std::vector<int> state, sort_ind, left_neighbor, right_neighbor;
};

// base/Config.h:111-130 — the knobs the hot path reads (defaults = config/Room.txt:67-79)
struct Config {
  bool angle_residual = true;
  bool point_to_line_residual = false;
  bool line_to_line_residual = true;
  bool point_to_plane_residual = true;
  bool normalize_distance = true;
  double point_to_line_dis_threshold = 0.3;
  double point_to_plane_dis_threshold = 1.0;
  double lidar_plane_tolerance = 0.05;
  double lidar_weight = 1.0;
  double camera_lidar_weight = 1.0;
  double camera_weight = 1.0;
  int num_threads = 25;
  int num_iteration_lidar = 7;
  // feature extraction (base/Config.h:72-76; config/Room.txt:32-35 sets max_curvature = 1000)
  int extraction_method = ADAPTIVE;
  float max_curvature = 5;
  float intersection_angle_threshold = 5;
  bool lidar_segmentation = true;
};

// Wall-clock seconds accumulated per stage of the mirrored call surface (association, track building, solve, ...)
// since the process started — what tools/room_like_odometry.py prints beside the end-to-end time.
void AddStageSeconds(const char* name, double seconds);   // wall-clock accounting of the host stages (see StageSeconds)
const std::map<std::string, double>& StageSeconds();
const std::map<std::string, long>& StageCalls();       // how many times each stage ran

// The process-wide engine (one pvlm_ctx).  Throws std::runtime_error when no GPU / library.
class Engine {
 public:
  static Engine& Default();
  static void SetDevice(int device);  // before first use
  pvlm_ctx* ctx() const { return ctx_; }
  void Check(pvlm_status st, const char* what) const;
  ~Engine();
 private:
  explicit Engine(int device);
  pvlm_ctx* ctx_ = nullptr;
};

// ---- sensors/Velodyne.h data contract ---------------------------------------------------------------
class Velodyne {
 public:
  int id = 0;
  bool valid = true;
  std::string name;                               // file name of the raw scan
  PointCloud cloud;                               // raw points as loaded (LoadLidar), camera-style axes
  PointCloud cornerLessSharp, surfFlat, surfLessFlat;
  int N_SCANS = 16;                               // sensors/Velodyne.h:72-73 (constructor arguments upstream)
  int horizon_scans = 1800;
  PointCloud cloud_scan;                          // ReOrderVLP: the scan ring by ring, intensity = ring id
  PointCloud cornerSharp;
  PointCloud cornerBeforeFilter;                  // cornerLessSharp as ExtractEdgeFeatures2 left it (sensors/Velodyne.cpp:1271)
  std::vector<PointCloud> edge_segmented;
  std::vector<std::set<int>> point_to_segment;
  std::vector<Vector6d> segment_coeffs;  // LiDAR-local (point, unit direction)
  std::vector<Vector3d> end_points;      // 2 per segment, LiDAR-local

  Velodyne() { SetRotation({0, 0, 0, 0, 0, 0, 0, 0, 0}); SetTranslation({INFINITY, INFINITY, INFINITY}); }
  // The pose setters hand the new pose to the scan's device copy (pvlm_scan_set_pose: the resident clouds stay where they are, like the host's);
  // MarkWorld gives the device copy back to the engine's context (InvalidateDevice -> pvlm_scan_destroy), whose pool is not thread-safe: call
  // them from the thread that owns the engine, never from workers.
  void SetPose(const Matrix3d& R_wl, const Vector3d& t_wl) { R_wl_ = R_wl; t_wl_ = t_wl; PoseChanged(); }
  void SetRotation(const Matrix3d& R_wl) { R_wl_ = R_wl; PoseChanged(); }
  void SetTranslation(const Vector3d& t_wl) { t_wl_ = t_wl; PoseChanged(); }
  const Matrix3d& GetRotation() const { return R_wl_; }
  const Vector3d& GetTranslation() const { return t_wl_; }
  Matrix4d GetPose() const;
  bool IsPoseValid() const;                       // sensors/Velodyne.cpp:1894-1899
  bool IsInWorldCoordinate() const { return world_; }
  void MarkWorld(bool w = true) { world_ = w; InvalidateDevice(); }  // clouds already carry world-frame floats
  Vector3d World2Local(const Vector3d& p) const;  // sensors/Velodyne.cpp:1850-1853
  Vector3d Local2World(const Vector3d& p) const;  // :1856-1859
  // LoadLidar (sensors/Velodyne.cpp:92-145): reads a .pcd file (the PCL formats: ascii, binary, binary_compressed; fields
  // x y z [intensity], float32), drops non-finite points (pcl::removeNaNFromPointCloud), drops points closer than 0.5 m
  // (removeClosedPointCloud, :148-172), swaps the axes to the camera convention (x, y, z) -> (x, -z, y) and marks the scan
  // invalid when fewer than 4000 points remain.  Returns false when the file cannot be read (the reference logs and returns).
  bool LoadLidar(std::string file_path = "");
  // sensors/Velodyne.cpp:1635-1674: motion compensation of the raw cloud — point i of n moves by i / n of the way from this scan's pose to T_we (the pose
  // at the sweep's end); the feature clouds are cleared as upstream clears them.  One scan: a batch of one; UndistortBatch: one device call for all.
  bool UndistortCloud(const Matrix4d& T_we);
  void Reset();                                   // sensors/Velodyne.cpp:1676-1720: everything but the raw cloud, the pose and the flags
  static void UndistortBatch(const std::vector<Velodyne*>& scans, const std::vector<Matrix4d>& T_we);
  // ReOrderVLP (sensors/Velodyne.cpp:371-526): firing order -> ring order, range image, ring/column of every point.
  void ReOrderVLP();
  // ExtractFeatures (sensors/Velodyne.cpp:531-760), method ADAPTIVE only (config/Room.txt:32), PLANAR BRANCH: optional
  // range-image Segmentation (:1438-1586), adaptive-window curvature (:623-657), per-sector sort (:707-723),
  // ExtractEdgeFeatures2 (:883-1000) and ExtractPlaneFeatures2 (:1098-1189, surfLessFlat through the 0.2 m voxel grid).
  // Fills cornerSharp / cornerLessSharp (intensity = index into cloud_scan) / surfFlat / surfLessFlat and, between the
  // edge and the planar picks like upstream (:752), runs EdgeToLine unless edge_to_line is false (cornerLessSharp is then
  // the reference's cornerBeforeFilter and no segments exist).  Sequential per scan (a state machine, a BFS and greedy
  // non-maximum suppression over 28.8 k points): host code, like upstream; run it under `omp parallel for` over scans
  // as lidar_mapping/LidarOdometry.cpp:131-147 does.  Throws std::invalid_argument for another method.
  void ExtractFeatures(float max_curvature = 50, float intersect_angle_threshold = 5, int method = ADAPTIVE, bool segment = true,
                       ExtractionTrace* trace = nullptr, bool edge_to_line = true);
  // EdgeToLine (sensors/Velodyne.cpp:1269-1324) with ExtractLineFeatures (sensors/LidarLineExtraction.cpp:296-389): line
  // segments grown from the edge points -> edge_segmented / segment_coeffs / end_points / point_to_segment, cornerLessSharp
  // and cornerSharp filtered down to the members of segments.  Upstream's RANSAC line fit of a fused group
  // (pcl::SACSegmentation, :150-160) is replaced by the exhaustive 2-point maximum-consensus line (host/pvlm_lines.cpp).
  // grown: the growth phase done ahead for this scan's edge points by K27 (pvlm_line_grow_batch: every (start point, neighbour pair) task of the scan grown on the
  // GPU) — the walk over the start points, the fusion and the filters run here on its segments.  Null (or a result the device gave back, status != 0): the growth
  // runs here too, segment after segment as upstream.  PVLM_EDGE_GROW=tasks: the host itself grows every task through the kernel's code (csrc/pvlm_linegrow_core.h)
  // and replays the walk — the CPU check of the task formulation (tests/test_lines_cpu.py).
  void EdgeToLine(const pvlm_line_grow_result* grown = nullptr);
  // ReOrderVLP + ExtractFeatures for MANY scans: the per-point / per-ring stages (ring and column of every return, range image,
  // Segmentation, adaptive-window curvature — sensors/Velodyne.cpp:371-526, :1438-1586, :623-657), the sector orders, the picks of
  // ExtractEdgeFeatures2 / ExtractPlaneFeatures2 and the pcl::VoxelGrid of the less-flat points (:883-1000, :1098-1189) run on the GPU
  // (pvlm_ring_extract_batch_picks; a call's scans go as a few device batches that overlap the host work of the previous one), EdgeToLine
  // and the assembly of the clouds on `num_threads` host threads.  A scan with a ring the device left undecided, and every scan of a
  // batch the device refused (non-finite coordinate, capacity), takes the host's own picks / extraction.  Every scan ends
  // up exactly as ReOrderVLP() + ExtractFeatures(...) leave it (tests/test_host_gpu.py), except that Layout().range_image and
  // .image_to_point_idx are only filled when traces are asked for (nothing downstream reads them).  Scans that ReOrderVLP /
  // ExtractFeatures would leave alone (invalid, already re-ordered, no points) are skipped; all scans of a call share N_SCANS and
  // horizon_scans of the first.  Call from the thread that owns the engine.
  static void ExtractFeaturesBatch(const std::vector<Velodyne*>& scans, float max_curvature = 50, float intersect_angle_threshold = 5, int method = ADAPTIVE,
                                   bool segment = true, bool edge_to_line = true, int num_threads = 1, std::vector<ExtractionTrace>* traces = nullptr);
  const RingLayout& Layout() const { return layout_; }
  void Transform2LidarWorld();                    // :1773-1808  (float clouds, in place)
  void Transform2Local();                         // :1810-1848
  // Every scan of the list to the world (local) frame — the loops of LidarOdometry.cpp:148-152 and :120-130 over all scans: the host's clouds are
  // transformed scan-parallel, the device copies of the resident scans IN PLACE by one pvlm_scan_transform_batch (K26: the same float arithmetic,
  // the voxel grids rebuilt from device-side bounding boxes on the way to the world frame) — a scan is uploaded once per EstimatePose, not once per
  // outer iteration.  PVLM_HOST_REUPLOAD=1: the device copies are dropped instead and uploaded again by the next association (the round-5 path;
  // tests compare the two).
  static void TransformBatch(const std::vector<Velodyne*>& scans, bool to_world, int num_threads);

  // device mirror of the clouds (uploaded lazily by the association entry points)
  pvlm_scan* DeviceScan() const;
  static void UploadBatch(const std::vector<const Velodyne*>& scans);   // all of them that are not resident yet, in one pvlm_scan_upload_batch
  void InvalidateDevice() const;
  void PoseChanged() const;                       // the device copy learns the pose the setters stored
  ~Velodyne();
  Velodyne(const Velodyne& o);
  Velodyne& operator=(const Velodyne& o);

 private:
  Matrix3d R_wl_;
  Vector3d t_wl_;
  bool world_ = false;
  RingLayout layout_;
  void Segmentation();                            // sensors/Velodyne.cpp:1438-1586
  // the second half of ExtractFeatures (:707-753, ExtractEdgeFeatures2, EdgeToLine, ExtractPlaneFeatures2) from the per-point arrays
  // per-point arrays the picks read (cloud_scan.size() entries each; the caller keeps them alive).  Window ends: left / right, or — when both are
  // null — index -+ half_window (-1 = no window).  sorted / sector_host: the sector orders of pvlm_ring_result, or null = sort every sector here.
  struct PickInputs {
    const float* curvature; const float* range; const int* left; const int* right; const int* half_window; const int* sorted; const unsigned char* sector_host;
  };
  void PickFeatures(float max_curvature, float intersect_angle_threshold, const PickInputs& in, ExtractionTrace* trace, bool edge_to_line);
  // the same from the picks the device made (pvlm_ring_extract_batch_picks, K24): the clouds are assembled ring by ring, EdgeToLine runs here
  void AssemblePicks(const pvlm_ring_result& r, ExtractionTrace* trace, bool edge_to_line, const pvlm_line_grow_result* grown = nullptr);
  mutable pvlm_scan* dev_ = nullptr;
};

// ---- lidar_mapping/LidarFeatureAssociate.h ------------------------------------------------------------
struct Point2Plane { Vector3d point; Vector4d plane_coeff; };
struct Point2Line { Vector3d point, line_point1, line_point2; };
struct Line2Line { int neighbor_line_idx, ref_line_idx; Vector3d line_point1, line_point2; };

std::vector<std::vector<int>> FindNeighbors(const std::vector<Velodyne>& lidars, const int neighbor_size);
std::vector<std::vector<int>> FindNeighborsConsecutive(const std::vector<Velodyne>& lidars, const int neighbor_size);
std::vector<Point2Plane> AssociatePoint2Plane(const Velodyne& ref_lidar, const Velodyne& nei_lidar, double plane_tolerance,
                                              const float dist_threshold = 0.7f, bool visualization = false);
std::vector<Line2Line> AssociateLine2Line(const Velodyne& ref_lidar, const Velodyne& nei_lidar, const float dist_threshold = 0.7f,
                                          bool visualization = false);
// The k-NN based variants (lidar_mapping/LidarFeatureAssociate.cpp:238-440, :478-548; used when
// config.point_to_line_residual is set — not by the Room/Floor configs).  The 5-NN search in ref.cornerLessSharp runs on
// the GPU (pvlm_knn), the per-query 5-point PCA / segment counting on the host.
std::vector<Point2Line> AssociatePoint2Line(const Velodyne& ref_lidar, const Velodyne& nei_lidar, const float dist_threshold = 0.7f,
                                            bool visualization = false);
std::vector<Point2Line> AssociatePoint2LineSegmentKNN(const Velodyne& ref_lidar, const Velodyne& nei_lidar, const float dist_threshold = 0.7f,
                                                      bool visualization = false);
std::vector<Point2Line> AssociatePoint2LineSegment(const Velodyne& ref_lidar, const Velodyne& nei_lidar, const float dist_threshold = 0.7f,
                                                   bool visualization = false);
std::vector<Line2Line> AssociateLine2LineKNN(const Velodyne& ref_lidar, const Velodyne& nei_lidar, const float dist_threshold = 0.7f,
                                             bool visualization = false);
// Extension (not in PanoVLM): the AssociateLine2Line calls of a whole outer iteration in one GPU launch;
// result[k] equals AssociateLine2Line(*pairs[k].first, *pairs[k].second, dist_threshold).
std::vector<std::vector<Line2Line>> AssociateLine2LineBatch(const std::vector<std::pair<const Velodyne*, const Velodyne*>>& pairs,
                                                            const float dist_threshold = 0.7f);
std::vector<Vector6d> TransformLines(const std::vector<Vector6d>& line_coeffs, const Matrix4d& T);
std::vector<Line2Line> FindAssociations(const Velodyne& ref_lidar, const Velodyne& nei_lidar, const std::vector<Vector6d>& ref_world,
                                        const std::vector<Vector6d>& nei_world, const std::vector<int>& line_matrix);

// ---- util/Tracks.h + lidar_mapping/LidarLineMatch.h ----------------------------------------------------
struct LineTrack {
  uint32_t id = 0;
  std::set<std::pair<uint32_t, uint32_t>> feature_pairs;  // {lidar id, line id}
  bool IsInside(const std::pair<uint32_t, uint32_t>& p) const { return feature_pairs.count(p) > 0; }
};
struct Exchange;
class LidarLineMatch {
 public:
  explicit LidarLineMatch(const std::vector<Velodyne>& lidars) : lidars_(lidars) {}
  void SetNeighborSize(int n) { neighbor_size_ = n; }
  void SetMinTrackLength(int n) { min_track_length_ = n; }
  // Sharded run: this rank associates the pairs of the scans [first, last) only — AssociateLine2Line of :68 is per pair —
  // and the matches of all ranks are concatenated through the exchange (two small all-reduces: the match counts per pair, then the matches)
  // before the union-find, which every rank runs on the complete list: the tracks are the same on every rank and equal to the one-process run's.
  void SetShard(const Exchange* exchange, size_t first, size_t last) { exchange_ = exchange; first_ = first; last_ = last; }
  bool GenerateTracks();
  const std::vector<LineTrack>& GetTracks() const { return tracks_; }
 private:
  const std::vector<Velodyne>& lidars_;
  int neighbor_size_ = 4, min_track_length_ = 3;
  std::vector<LineTrack> tracks_;
  const Exchange* exchange_ = nullptr;
  size_t first_ = 0, last_ = 0;
};

// ---- the ceres:: surface the reference touches -----------------------------------------------------------
namespace ceres_like {

class CostFunction {
 public:
  virtual ~CostFunction() {}
  // ceres::CostFunction::Evaluate: parameters = {aa_r, t_r, aa_n, t_n}; jacobians may be null, and each
  // jacobians[i] may be null.  Evaluated on the GPU (one-row residual set) — API parity, not the fast path.
  // The reprojection functor has THREE blocks {aa_cw, t_cw, point_3d} (num_blocks == 3).
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const;
  int num_blocks = 4;
  int kind = 0;
  unsigned flags = 0;
  double weight = 1.0;
  std::vector<double> row;  // the ABI row of this block
};
class LossFunction { public: virtual ~LossFunction() {} virtual pvlm_loss kind() const = 0; virtual double a() const = 0; };
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a) {}
  pvlm_loss kind() const override { return PVLM_LOSS_HUBER; }
  double a() const override { return a_; }
 private:
  double a_;
};

class Problem {
 public:
  Problem();
  ~Problem();
  // Takes ownership of cost (and of loss, shared by many blocks like in Ceres).
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* aa_r, double* t_r, double* aa_n, double* t_n);
  // Bulk form (extension): n blocks of one functor on the same four parameter blocks, rows = n x stride in the ABI row
  // layout of include/pvlm.h — what n X::Create + AddResidualBlock calls would add, without n heap objects.
  void AddResidualRows(int kind, unsigned flags, double weight, LossFunction* loss, double* aa_r, double* t_r, double* aa_n, double* t_n,
                       const double* rows, size_t n);
  // Three-block form of AddCameraResidual (util/Optimization.cpp:198-201): camera pose + a free 3-D point.
  // The point blocks live on the GPU during Solve and are eliminated there (Schur complement).
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* aa_c, double* t_c, double* point_3d);
  // Device-resident hand-off: a residual set produced by the association kernels; segment p uses
  // the parameter blocks (aa[ref_p], t[ref_p], aa[nei_p], t[nei_p]) of the given lists (ids = list index).
  void AddResidualSet(pvlm_resset* set, LossFunction* loss, std::vector<Vector3d>* aa_list, std::vector<Vector3d>* t_list);
  void SetParameterBlockConstant(double* block);
  // Registers every (angle-axis, translation) pair of the two lists as a pose, in list order, before any residual block
  // is added: pose ids then equal list indices on every rank of a sharded solve (otherwise ids follow first use).
  void RegisterPoses(std::vector<Vector3d>& aa_list, std::vector<Vector3d>& t_list);
  int NumResidualBlocks() const;
  struct Impl;
  Impl* impl() const { return impl_; }
 private:
  Problem(const Problem&) = delete;
  Impl* impl_;
};

}  // namespace ceres_like

// ---- multi-GPU: how the ranks of a sharded solve add up their normal equations ------------
// One process per GPU.  Every rank holds all scans and the full pose vector; a rank builds residual blocks only for its
// block of REFERENCE scans (the `i` loop of util/Optimization.cpp:521-560 and :345-441), evaluates them on its GPU and
// the ranks sum [cost | g | 6x6 blocks] once per evaluation.  Everything else of the LM iteration is replicated: the
// summed system is bit-identical on every rank, so every rank takes the same decisions and no parameter broadcast is needed.
// allreduce_sum: in place on a HOST buffer, must leave the same bits on every rank (RCCL's ring all-reduce does; the
// file exchange sums in rank order).
struct Exchange {
  int world = 1, rank = 0;
  std::function<void(double* buf, size_t count)> allreduce_sum;
  bool active() const { return world > 1 && (bool)allreduce_sum; }
  // block partition of n reference scans, the same rule as panovlm_amd/sharding.py
  std::pair<size_t, size_t> Range(size_t n) const { return {n * (size_t)rank / (size_t)world, n * ((size_t)rank + 1) / (size_t)world}; }
  // contiguous ranges of (nearly) equal summed weight: boundary r is the first scan whose weight prefix reaches r / world of the
  // total.  With weight(i) = sum of the query counts of i's neighbour scans this is SURVEY.md §8 E's "rebalancing by sum Nq":
  // temporal neighbour lists give every scan the same weight, loop-closure pairs make some scans heavier.  Every rank computes
  // the same boundaries from the same (replicated) weights.  All-zero weights: Range(n).
  std::pair<size_t, size_t> BalancedRange(const std::vector<double>& weight, int of_rank = -1) const;
};
// RCCL over xGMI through the C ABI (pvlm_comm_* on the engine's context): `id` = the 128 bytes rank 0 got from
// pvlm_comm_unique_id, carried to the other ranks by the launcher.  Collective: every rank calls it.
Exchange MakeRcclExchange(int world, int rank, const unsigned char id[128]);
// Ranks on one host exchanging through a directory (functional tests on a one-GPU box; not a fast path).
Exchange MakeFileExchange(int world, int rank, const std::string& dir);

namespace ceres_like {

enum LinearSolverType { DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR };
struct Solver {
  struct Options {
    const Exchange* exchange = nullptr;   // sharded solve: see Exchange (the Problem must have its poses registered with RegisterPoses)
    bool minimizer_progress_to_stdout = false;
    int num_threads = 1;
    int max_num_iterations = 50;
    int max_linear_solver_iterations = 500;
    LinearSolverType linear_solver_type = DENSE_SCHUR;
    // Ceres 2.0 trust-region defaults ([recalled])
    double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3, function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  };
  struct Summary {
    double initial_cost = 0, final_cost = 0;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    int num_residual_blocks = 0;
    bool usable = false;
    std::string message;
    std::vector<double> cost_history;
    bool IsSolutionUsable() const { return usable; }
    std::string BriefReport() const;
  };
};
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary);

}  // namespace ceres_like

// ---- base/CostFunction.h functors: X::Create(...) ------------------------------------------------------
struct Point2Plane_Meter { static ceres_like::CostFunction* Create(const Vector3d& curr_point, const Vector4d& plane, const double weight = 1); };
struct Point2Plane_Angle { static ceres_like::CostFunction* Create(const Vector3d& curr_point, const Vector4d& plane, const bool normalize, const double weight = 1); };
struct Point2Line_Meter { static ceres_like::CostFunction* Create(const Vector3d& curr_point, const Vector3d& last_point_a, const Vector3d& last_point_b, const double weight = 1); };
struct Point2Line_Angle { static ceres_like::CostFunction* Create(const Vector3d& curr_point, const Vector3d& last_point_a, const Vector3d& last_point_b, const bool normalize, const double weight = 1); };
struct Plane2Plane_Global { static ceres_like::CostFunction* Create(const Vector3d& plane_ref, const Vector3d& point_a, const Vector3d& point_b, const double w = 1.0); };
struct PlaneIOUResidual { static ceres_like::CostFunction* Create(const Vector4d& ref_plane, const Vector3d& middle_neighbor, const Vector3d& middle_ref, const double angle, const double weight = 1.0); };
// base/CostFunction.h:218-247 — AutoDiffCostFunction<.,1,3,3,3> on (angleAxis_cw, t_cw, point_3d)
struct PanoramaReprojResidual_1Angle { static ceres_like::CostFunction* Create(const Vector3d& pt, double weight = 1.0); };
constexpr int kReprojKind = 100;   // CostFunction::kind of the three-block functor (not a pvlm_functor)

// ---- util/Optimization.h ---------------------------------------------------------------------------------
size_t AddLidarPointToPlaneResidual(const std::vector<std::vector<int>>& neighbors, const std::vector<Velodyne>& lidars,
                                    std::vector<Vector3d>& angleAxis_lw_list, std::vector<Vector3d>& t_lw_list,
                                    ceres_like::Problem& problem, double point_to_plane_dis_threshold, double plane_tolerance,
                                    bool angle_residual, bool normalized_distance, double weight = 1.0,
                                    const std::pair<size_t, size_t>* ref_range = nullptr /* sharded run: reference scans [first, second) only */);
// util/Optimization.cpp:443-504 — only pairs of consecutive scans (|n_idx - i| <= 1) get point-to-line blocks
size_t AddLidarPointToLineResidual(const std::vector<std::vector<int>>& neighbors, const std::vector<Velodyne>& lidars,
                                   std::vector<Vector3d>& angleAxis_lw_list, std::vector<Vector3d>& t_lw_list, ceres_like::Problem& problem,
                                   double point_to_line_dis_threshold, bool use_segment, bool angle_residual, bool normalized_distance,
                                   double weight = 1.0, const std::pair<size_t, size_t>* ref_range = nullptr);
size_t AddLidarLineToLineResidual2(const std::vector<std::vector<int>>& neighbors, const std::vector<Velodyne>& lidars,
                                   std::vector<Vector3d>& angleAxis_lw_list, std::vector<Vector3d>& t_lw_list,
                                   ceres_like::Problem& problem, const std::vector<LineTrack>& lidar_line_tracks,
                                   double point_to_line_dis_threshold, bool angle_residual, bool normalized_distance, double weight = 1.0,
                                   const std::pair<size_t, size_t>* ref_range = nullptr);
// AddCameraLidarResidual (util/Optimization.cpp:564-607): two blocks per matched line pair —
// Plane2Plane_Global (weight line_pair.weight * weight) and PlaneIOUResidual (weight 2 * weight) — on
// (angleAxis_cw[frame], t_cw[frame], angleAxis_lw[lidar], t_lw[lidar]).  `frame_pose_valid[f]` stands for
// frames[f].IsPoseValid(); rows/cols are the panorama size (frames[0].GetImageRows/Cols).
struct CameraLidarLinePair;
size_t AddCameraLidarResidual(int rows, int cols, const std::vector<bool>& frame_pose_valid, const std::vector<Velodyne>& lidars,
                              std::vector<Vector3d>& angleAxis_cw_list, std::vector<Vector3d>& t_cw_list,
                              std::vector<Vector3d>& angleAxis_lw_list, std::vector<Vector3d>& t_lw_list,
                              const std::map<std::pair<size_t, size_t>, std::vector<CameraLidarLinePair>>& line_pairs,
                              ceres_like::LossFunction* loss_function, ceres_like::Problem& problem, double weight);
ceres_like::Solver::Options SetOptionsLidar(const int num_threads, const int lidar_size);
ceres_like::Solver::Options SetOptionsSfM(const int num_threads);                      // util/Optimization.cpp:608-634
// util/Tracks.h:109-133 — a triangulated feature track: (frame index, keypoint index) pairs + the 3-D point
struct PointTrack {
  uint32_t id = 0xffffffffu;
  std::set<std::pair<uint32_t, uint32_t>> feature_pairs;
  Vector3d point_3d{{0, 0, 0}};
};
enum RESIDUAL_TYPE { ANGLE_RESIDUAL_1 = 0, ANGLE_RESIDUAL_2 = 1, PIXEL_RESIDUAL = 2 };
struct Frame;
// AddCameraResidual (util/Optimization.cpp:172-222), ANGLE_RESIDUAL_1 branch (the one Optimize uses,
// CameraLidarOptimizer.cpp:431-432): one PanoramaReprojResidual_1Angle per (track, observation) whose frame has a
// valid pose; bearing = eq.ImageToCam(keypoint.pt) — which resolves to the cv::Point2i overload, i.e. the keypoint
// is rounded to the nearest pixel (cvRound) and un-projected in float; HuberLoss(4 deg).
size_t AddCameraResidual(const std::vector<Frame>& frames, std::vector<Vector3d>& angleAxis_cw_list, std::vector<Vector3d>& t_cw_list,
                         std::vector<PointTrack>& structure, ceres_like::Problem& problem, int residual_type, double weight);

// util/FileIO.cpp:11-79, :168-191 — pose text files: one row per pose, optional name + 12 numbers
// (R row-major interleaved with t: r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz); rows containing
// "inf"/"nan" are invalid poses (R = 0, t = inf) and are kept only when with_invalid is set.
// ExportPoseT writes with the ostream default precision (6 significant digits) like the reference;
// pass precision = 17 for a lossless file.
bool ReadPoseT(std::string file_path, bool with_invalid, std::vector<Matrix3d>& rotation_list, std::vector<Vector3d>& trans_list,
               std::vector<std::string>& name_list);
void ExportPoseT(const std::string file_path, const std::vector<Matrix3d>& rotation_list, const std::vector<Vector3d>& trans_list,
                 const std::vector<std::string>& name_list, int precision = 6);

// FormLine(points, tolerance, dis_threshold) of base/Geometry.hpp:220-260 on n x 3 doubles: false (and a zero line) when the
// points do not form a line; line = (centroid, unit direction)
bool FormLine3D(const double* pts, int n, double tolerance, double dis_threshold, double* line);

// ceres/rotation.h pieces the callers use (lidar_mapping/LidarOdometry.cpp:31,105)
void RotationMatrixToAngleAxis(const Matrix3d& R, Vector3d* angle_axis);
void AngleAxisToRotationMatrix(const Vector3d& angle_axis, Matrix3d* R);

// ---- lidar_mapping/LidarOdometry.h --------------------------------------------------------------------------
class LidarOdometry {
 public:
  LidarOdometry(const std::vector<Velodyne>& lidars, const Config& config) : lidars(lidars), config(config) {}
  // EstimatePose(max_iteration): outer loop of LidarOdometry.cpp:116-187 WITHOUT the feature extraction
  // (the scans handed in already carry their feature clouds — SURVEY.md §8 A0 / "next" row N3).
  bool EstimatePose(const int max_iteration);
  bool RefinePose(double& cost, int& steps, bool use_segment);
  // lidar_mapping/LidarOdometry.cpp:189-263 without the PCD export: every scan with a usable pose is motion-compensated with the pose that ends its sweep
  // (the next usable scan's pose through SlerpPose; the last scan continues the motion of the one before), all scans in one device call
  bool UndistortLidars(const float gap_time);
  // lidar_mapping/LidarOdometry.cpp:306-: upstream reloads the raw scans from their files before the motion compensation (EstimatePose moved the clouds to the
  // world frame and back in float, and may have dropped them); the in-memory form of that reload: scan k gets clouds[k] again
  void ReloadClouds(const std::vector<PointCloud>& clouds);
  void ResetAllLidars();                          // lidar_mapping/LidarOdometry.cpp:295-304 (the reload of an empty cloud is LoadLidar's business)
  const std::vector<Velodyne>& GetLidarData() const { return lidars; }
  std::vector<Matrix3d> GetGlobalRotation() const;
  std::vector<Vector3d> GetGlobalTranslation() const;
  // per-outer-iteration log (cost, successful steps, residual blocks) for tests
  struct IterLog { double cost; int steps; int residual_blocks; };
  std::vector<IterLog> log;
  // sharded runs, per outer iteration: this rank's range of reference scans, the association queries (sum Nq over the pairs of
  // a rank's reference scans) of EVERY rank — computed from the replicated neighbour lists — and this rank's own residual blocks
  struct ShardLog { size_t first, last; std::vector<double> queries_per_rank; int local_blocks; };
  std::vector<ShardLog> shard_log;
  // Sharded run (one process per GPU): set before EstimatePose / RefinePose on every rank.  Association and
  // residual evaluation are done for the rank's block of reference scans only; poses, neighbour lists and line tracks are
  // replicated.  residual_blocks in the log is then the sum over the ranks.
  void SetExchange(const Exchange& e) { exchange_ = e; }
 private:
  std::vector<Velodyne> lidars;
  Config config;
  Exchange exchange_;
};

// base/Geometry.hpp:572-583: the pose a fraction `ratio` of the way from pose_w1 to pose_w2 (rotation by slerp, translation of T_21 scaled)
Matrix4d SlerpPose(const Matrix4d& pose_w1, const Matrix4d& pose_w2, double ratio);

// ---- sensors/Equirectangular.h + joint_optimization/CameraLidarLineAssociate.h -------------------------------
struct CameraLidarLinePair {
  std::array<float, 4> image_line{{0, 0, 0, 0}};
  Vector3d lidar_line_start{{0, 0, 0}}, lidar_line_end{{0, 0, 0}};
  int image_line_id = -1, lidar_line_id = -1;
  float angle = -1, weight = 1;
};
class CameraLidarLineAssociate {
 public:
  CameraLidarLineAssociate(int rows, int cols) : rows(rows), cols(cols) {}
  // AssociateByAngle (CameraLidarLineAssociate.cpp:340-475): `lidar` carries LOCAL-frame corner points.
  void AssociateByAngle(const std::vector<std::array<float, 4>>& lines, const Velodyne& lidar, const Matrix4d& T_cl,
                        const bool multiple_association = true, const std::vector<bool>& image_line_mask = {},
                        const std::vector<bool>& lidar_line_mask = {});
  // Same with the vote matrix (n_lines x n_segments, from pvlm_cam_lidar_votes[_batch]) handed in — lets
  // AssociateLineMulti run the voting loops of all (frame, LiDAR) pairs in one launch.
  void AssociateByAngleWithVotes(const std::vector<std::array<float, 4>>& lines, const Velodyne& lidar, const Matrix4d& T_cl, const int* votes,
                                 const bool multiple_association = true, const std::vector<bool>& image_line_mask = {},
                                 const std::vector<bool>& lidar_line_mask = {});
  const std::vector<CameraLidarLinePair>& GetAssociatedPairs() const { return line_pairs; }
 private:
  void Filter(bool filter_by_angle, bool filter_by_length);
  void UniqueLinePair(const std::vector<std::array<float, 4>>& lines, const std::vector<Vector3d>& lidar_lines_endpoint);
  int rows, cols;
  std::vector<CameraLidarLinePair> line_pairs;
};


// util/Visualization.h:407-441 — sparse LiDAR depth image (uint16 depth * 256, row-major rows x cols) seen from the
// camera: the LiDAR-seeded depth prior of mvs/MVS.cpp:510-514.  `cloud` is in the LiDAR frame.
std::vector<uint16_t> ProjectLidar2PanoramaDepth(const PointCloud& cloud, const int rows, const int cols, const Matrix4d& T_cl, const size_t size = 3);

// ---- joint_optimization/CameraLidarOptimizer.h (mapping mode) --------------------------------------------------
// sensors/Frame.h contract as far as the path needs it: image size, pose T_wc, the detected image lines
// (image_lines_all[frame].GetLines()).
struct Frame {
  int id = 0, rows = 2880, cols = 5760;
  Matrix3d R_wc{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  Vector3d t_wc{{0, 0, 0}};
  bool pose_valid = true;
  std::vector<std::array<float, 4>> lines;
  std::vector<std::array<float, 2>> keypoints;   // GetKeyPoints()[k].pt (SIFT keypoints, pixels)
  bool IsPoseValid() const { return pose_valid; }
  int GetImageRows() const { return rows; }
  int GetImageCols() const { return cols; }
  Matrix4d GetPose() const { return {R_wc[0], R_wc[1], R_wc[2], t_wc[0], R_wc[3], R_wc[4], R_wc[5], t_wc[1], R_wc[6], R_wc[7], R_wc[8], t_wc[2], 0, 0, 0, 1}; }
};

// ---- mvs/MVS.h:45-57, mvs/MVS.cpp:334-382 — who the neighbours of a reference view are (the `nei` / R_nr / t_nr arguments of
// pvlm_mvs_*).  SelectNeighborKNN: the 3 x neighbor_size nearest camera centres (float32, as pcl::KdTreeFLANN returns them),
// the first hit skipped as "self", candidates closer than the squared distance threshold skipped, the first neighbor_size
// kept; T_nr = T_wn^-1 T_wr in double, stored as float (cv::Matx33f / cv::Vec3f).  The rigid inverse is used where
// upstream calls Eigen's general Matrix4d::inverse() — equal to ~1e-16 before the rounding to float.
struct NeighborInfo {
  size_t id = 0;
  std::array<float, 9> R_nr{};   // reference -> neighbour, row-major
  std::array<float, 3> t_nr{};
};
std::vector<std::vector<NeighborInfo>> SelectNeighborKNN(const std::vector<Frame>& frames, int neighbor_size, float sq_distance_threshold);

// ---- mvs/MVS.cpp:2168-2334 — MVS::FuseDepthImages(use_filtered_depth = true), the second half of MVS::FuseDepthMaps (:224-231; the first
// half, MergeDepthImages, is a loop over pvlm_mvs_depth_to_cloud).  The greedy confidence-weighted fusion walks frames by decreasing
// neighbour count and pixels in raster order; a pixel's fate depends on every claim made before it, so this is host code here as
// upstream.  DepthFrame holds the cv::Mat members of sensors/Frame.h the function touches.  Upstream's bookkeeping is kept: a depth map
// is released after (neighbours + 1) visits and read again from the depth FILE when needed as a neighbour (depth_file: the unfiltered
// estimate <id>_geo.bin / _pho.bin; empty = no file), a reference whose map is gone is skipped, the claim lists survive the `continue`
// of the sky-colour test.  depth_filter maps are modified (depths in front of which an accepted point lies are zeroed).
struct DepthFrame {
  int id = 0;
  std::vector<float> depth_filter;       // rows x cols, empty = cv::Mat::empty()
  std::vector<float> depth_file;         // rows x cols or empty
  std::vector<float> conf;               // rows x cols: <id>_filter.bin
  std::vector<unsigned char> bgr;        // rows x cols x 3
  Matrix4d T_wc{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
};
struct PointXYZRGB { float x, y, z; unsigned char r, g, b; };
std::vector<PointXYZRGB> FuseDepthImages(int rows, int cols, std::vector<DepthFrame>& frames, const std::vector<std::vector<NeighborInfo>>& neighbors, float max_depth,
                                         float depth_diff_threshold);

class CameraLidarOptimizer {
 public:
  using LinePairs = std::map<std::pair<size_t, size_t>, std::vector<CameraLidarLinePair>>;
  // LiDAR scans are handed in in their LOCAL frame with feature clouds + segments (ExtractLidarLines, :151-175)
  CameraLidarOptimizer(const Matrix4d& T_cl, const std::vector<Velodyne>& lidars, const std::vector<Frame>& frames, const Config& config,
                       int neighbor_size_joint = 3, int num_iteration_joint = 5)
      : T_cl_init(T_cl), lidars(lidars), frames(frames), config(config), neighbor_size_joint(neighbor_size_joint), num_iteration_joint(num_iteration_joint) {}
  // JointOptimize(true), MAPPING mode loop of CameraLidarOptimizer.cpp:260-285.  The SfM reprojection term
  // (AddCameraResidual, free 3-D points) is added when a structure has been set (SetStructure stands for
  // ReadPointTracks(points.bin), :255); with an empty structure the problem has the LiDAR terms only.
  bool JointOptimize();
  void SetStructure(const std::vector<PointTrack>& s) { structure = s; }
  const std::vector<PointTrack>& GetStructure() const { return structure; }
  std::vector<std::vector<int>> NeighborEachFrame(const int neighbor_size, const bool temporal) const;   // :551-610
  LinePairs AssociateLineMulti(const int neighbor_size, const bool temporal = true);                      // :331-384
  int Optimize(const LinePairs& line_pairs, std::vector<PointTrack>& structure, const bool refine_camera_rotation, const bool refine_camera_trans,
               const bool refine_lidar_rotation, const bool refine_lidar_trans, const bool refine_structure, double& cost, int& steps);   // :387-548
  int Optimize(const LinePairs& line_pairs, const bool refine_camera_rotation, const bool refine_camera_trans, const bool refine_lidar_rotation,
               const bool refine_lidar_trans, double& cost, int& steps) {                                  // no SfM term
    std::vector<PointTrack> none;
    return Optimize(line_pairs, none, refine_camera_rotation, refine_camera_trans, refine_lidar_rotation, refine_lidar_trans, true, cost, steps);
  }
  // Calibration mode, CameraLidarOptimizer.cpp:32-87: ONE unknown, the camera <- LiDAR transform, refined on the associated line pairs
  // with Plane2Plane_Relative (residual in DEGREES, Huber 2 deg expressed in radians as upstream has it) and PlaneRelativeIOUResidual
  // (weight 2, half of the image line's arc, no loss).  Returns 1 like upstream; the result is GetOptimizedTcl().
  int Optimize(const LinePairs& line_pairs, const Matrix4d& T_cl, double* final_cost = nullptr, int* successful_steps = nullptr, int* residual_blocks = nullptr);
  const Matrix4d& GetOptimizedTcl() const { return T_cl_optimized; }
  const std::vector<Velodyne>& GetLidars() const { return lidars; }
  const std::vector<Frame>& GetFrames() const { return frames; }
  struct IterLog { double cost; int steps; int residual_blocks; size_t line_pairs; std::vector<double> cost_history; };
  std::vector<IterLog> log;
 private:
  Matrix4d T_cl_init;
  Matrix4d T_cl_optimized = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<Velodyne> lidars;
  std::vector<Frame> frames;
  std::vector<PointTrack> structure;
  Config config;
  int neighbor_size_joint, num_iteration_joint;
  int last_blocks_ = 0;
  std::vector<double> last_history_;
};

}  // namespace pvlm
