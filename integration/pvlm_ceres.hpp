// pvlm_ceres.hpp — keep Ceres as the outer trust-region solver, evaluate PanoVLM's LiDAR residual blocks on the GPU.
//
// Reference-side binding for integration level B of INTEGRATION.md: drop-in bodies for the four adders of
// CameraLidarOptimizer::Optimize / LidarOdometry::RefinePose —
//   AddLidarPointToPlaneResidualGpu   (util/Optimization.cpp:506-562)   point-to-plane, association on the GPU
//   AddLidarLineToLineResidual2Gpu    (util/Optimization.cpp:329-441)   Point2Line rows of every track-confirmed segment pair, built on the GPU
//   AddCameraLidarResidualGpu         (util/Optimization.cpp:564-607)   Plane2Plane_Global + PlaneIOUResidual per camera<->LiDAR line pair
//   AddCameraResidualGpu              (util/Optimization.cpp:172-222)   PanoramaReprojResidual_1Angle, three parameter blocks
// all evaluated through ONE CeresBatch (one ceres::EvaluationCallback).  The first one, as the pattern,
//   1. associates every (ref, neighbour) scan pair on the GPU in one call (pvlm_assoc_point2plane — the result stays in
//      HBM as a residual set),
//   2. adds ONE thin ceres::SizedCostFunction<1,3,3,3,3> per correspondence, in the reference's order, with the
//      reference's parameter-block pointers and its shared HuberLoss — so Ceres' problem structure, robust loss and
//      cost accounting are exactly what they were,
//   3. evaluates all of them with one batched kernel launch per Ceres evaluation point through a
//      ceres::EvaluationCallback (Problem::Options::evaluation_callback, Ceres >= 2.0).
//
// Ceres itself is not in this repository's image: the file is written against the public Ceres 2.0 API and PanoVLM's own
// types, and is what a PanoVLM maintainer adds next to util/Optimization.cpp.  It contains no PanoVLM code; only
// include/pvlm.h (C ABI) is used from this repository.  CI compiles and RUNS it against an interface-only test double of
// the few Ceres classes it touches (tests/cpp/ceres_double/ceres/ceres.h — labelled as such: it pins nothing about Ceres'
// behaviour, it only proves that this file compiles and that the rows it hands out equal pvlm_eval's;
// tests/test_host_gpu.py::test_ceres_adapter_rows).
//
// The boundary is a PCIe link (55 GB/s measured on the MI355X box, bench.py -> "pcie"): the link is the roof of this mode, so the batch
// is delivered in as few bytes as carry the information, into page-locked memory, asynchronously.  Point functors (point-to-plane,
// point-to-line): 32-byte force rows [r | g] (pvlm_eval_force_host_async); the moment c = (R_rn P_n + t_rn - t_rw) x g is rebuilt
// inside Evaluate from the block's point, which the batch holds on the host (downloaded once, when the set is added).  Plane / IOU
// blocks of the camera-LiDAR term: 56-byte wrench rows [r | c | g] (pvlm_eval_wrench_host_async).  The 1 x 12 Jacobian row is
// formed from the row and its pair's 3x3 tables inside Evaluate — ~50 multiply-adds on the Ceres worker thread that asks for it —
// instead of shipping 104 bytes per block.
//
// Usage inside LidarOdometry::RefinePose (lidar_mapping/LidarOdometry.cpp:36-80):
//     pvlm::CeresBatch batch(ctx);                                  // before the ceres::Problem
//     ceres::Problem::Options po; po.evaluation_callback = &batch;
//     ceres::Problem problem(po);
//     ... pvlm::AddLidarPointToPlaneResidualGpu(batch, scans, neighbors, lidars, angleAxis_lw_list, t_lw_list, problem, ...);
//     ceres::Solve(options, &problem, &summary);                    // unchanged
#pragma once
#include <ceres/ceres.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <utility>
#include <vector>

#include "pvlm.h"

namespace pvlm {

// One batched GPU evaluation per Ceres evaluation point, shared by all the row functors below.
class CeresBatch : public ceres::EvaluationCallback {
 public:
  explicit CeresBatch(pvlm_ctx* ctx) : ctx_(ctx) {}
  ~CeresBatch() override {
    for (Set& s : sets_) { pvlm_host_free(ctx_, s.rows); pvlm_host_free(ctx_, s.tables); pvlm_resset_destroy(ctx_, s.set); }
    for (Bundle& b : bundles_) pvlm_ba_destroy(ctx_, b.set);
  }
  CeresBatch(const CeresBatch&) = delete;
  CeresBatch& operator=(const CeresBatch&) = delete;

  // The pose storage Ceres optimises in place: angleAxis_lw_list / t_lw_list (LidarOdometry.cpp:23-33), n x 3 each,
  // contiguous (eigen_vector<Eigen::Vector3d>::data()->data()).
  // This is pose array 0: the LiDAR poses, whose index is the id the association kernels put into their residual sets.
  void SetPoseStorage(int n_poses, const double* angle_axis, const double* translation) {
    if (arrays_.empty()) arrays_.push_back(PoseArray());
    arrays_[0] = PoseArray{n_poses, angle_axis, translation, 0};
    Rebase();
  }
  // A further pose array behind the ones already known (the joint problem: angleAxis_cw_list / t_cw_list after the LiDAR
  // lists, CameraLidarOptimizer.cpp:394-417).  Returns the id of its first pose in the batch's pose table.
  int AddPoseArray(int n_poses, const double* angle_axis, const double* translation) {
    arrays_.push_back(PoseArray{n_poses, angle_axis, translation, 0});
    Rebase();
    return arrays_.back().base;
  }

  // Takes ownership of a residual set produced by the association kernels; `pair_of_row[i]` = segment of block i.
  // Returns the set's index.
  int AddSet(pvlm_resset* set, std::vector<int> pair_of_row) {
    Set s; s.set = set; s.pair_of_row = std::move(pair_of_row);
    int kind = 0;
    pvlm_resset_info(set, &s.n, &s.n_pairs, &kind, nullptr);
    s.width = kind <= PVLM_POINT2LINE_ANGLE ? 4 : 7;
    if (s.width == 4 && s.n > 0) {                 // the points of the blocks: the first three entries of every row, once
      const int stride = kind <= PVLM_POINT2PLANE_ANGLE ? 7 : 9;
      std::vector<double> rows((size_t)s.n * stride);
      if (pvlm_resset_download(ctx_, set, nullptr, nullptr, nullptr, rows.data()) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
      s.points.resize((size_t)s.n * 3);
      for (int64_t i = 0; i < s.n; ++i) for (int k = 0; k < 3; ++k) s.points[(size_t)i * 3 + k] = rows[(size_t)i * stride + k];
    }
    void* p = nullptr;
    if (pvlm_host_alloc(ctx_, (int64_t)s.n * s.width * (int64_t)sizeof(double), &p) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    s.rows = static_cast<double*>(p);
    if (pvlm_host_alloc(ctx_, (int64_t)s.n_pairs * PVLM_PAIR_TABLE * (int64_t)sizeof(double), &p) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    s.tables = static_cast<double*>(p);
    sets_.push_back(std::move(s));
    return (int)sets_.size() - 1;
  }
  // [r | c(3) | g(3)] (width 7) or [r | g(3)] (width 4, point functors: the point of the block is point(set, row)) of block `row`,
  // and the table of its pair: [R_rn(9) | t_rn(3) | t_rw(3) | J_l(aa_r)(9) | M_n(9)]
  const double* row(int set, int64_t row) const { const Set& s = sets_[(size_t)set]; return s.rows + (size_t)row * s.width; }
  int row_width(int set) const { return sets_[(size_t)set].width; }
  const double* point(int set, int64_t row) const { return sets_[(size_t)set].points.data() + (size_t)row * 3; }
  const double* table(int set, int64_t row) const { const Set& s = sets_[(size_t)set]; return s.tables + (size_t)s.pair_of_row[(size_t)row] * PVLM_PAIR_TABLE; }
  pvlm_resset* set(int set) const { return sets_[(size_t)set].set; }
  int num_sets() const { return (int)sets_.size(); }

  // Takes ownership of a reprojection set (pvlm_ba_create: observations grouped by point, camera ids = ids in THIS batch's
  // pose table).  point_blocks[p] = the three doubles Ceres optimises for point p (PointTrack::point_3d.data()).
  int AddBundle(pvlm_baset* set, std::vector<const double*> point_blocks) {
    Bundle b; b.set = set; b.points = std::move(point_blocks);
    int n_points = 0;
    pvlm_ba_structure(set, &n_points, &b.n_obs, nullptr, nullptr, nullptr, nullptr);
    if ((size_t)n_points != b.points.size()) throw std::runtime_error("CeresBatch::AddBundle: one parameter block per point of the set");
    b.r.assign((size_t)b.n_obs, 0.0); b.J.assign((size_t)b.n_obs * 9, 0.0); b.X.assign(b.points.size() * 3, 0.0);
    bundles_.push_back(std::move(b));
    return (int)bundles_.size() - 1;
  }
  // residual and [d/d angleAxis_cw (3) | d/d t_cw (3) | d/d X (3)] of observation `obs` (the order of pvlm_ba_create)
  double reproj_residual(int bundle, int64_t obs) const { return bundles_[(size_t)bundle].r[(size_t)obs]; }
  const double* reproj_row(int bundle, int64_t obs) const { return bundles_[(size_t)bundle].J.data() + (size_t)obs * 9; }
  pvlm_baset* bundle(int b) const { return bundles_[(size_t)b].set; }

  // ceres::EvaluationCallback — called once before the residual blocks are evaluated at a (possibly new) point.  The
  // wrench rows serve cost-only and Jacobian evaluations alike, so a repeated call at the same point costs nothing.
  void PrepareForEvaluation(bool /*evaluate_jacobians*/, bool new_evaluation_point) override {
    if (!new_evaluation_point && valid_) return;
    const double* aa = nullptr; const double* t = nullptr; int n = 0;
    if (arrays_.size() == 1) { aa = arrays_[0].aa; t = arrays_[0].t; n = arrays_[0].n; }     // Ceres' own storage, no copy
    else {
      for (const PoseArray& a : arrays_) {
        std::copy(a.aa, a.aa + 3 * (size_t)a.n, aa_all_.begin() + 3 * (size_t)a.base);
        std::copy(a.t, a.t + 3 * (size_t)a.n, t_all_.begin() + 3 * (size_t)a.base);
      }
      aa = aa_all_.data(); t = t_all_.data(); n = (int)(aa_all_.size() / 3);
    }
    if (pvlm_set_poses(ctx_, n, aa, t) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    for (Set& s : sets_)   // ONE kernel per slice of a set, its copy queued right behind it; nothing waits until the end
      if ((s.width == 4 ? pvlm_eval_force_host_async(ctx_, s.set, s.rows, s.tables) : pvlm_eval_wrench_host_async(ctx_, s.set, s.rows, s.tables)) != PVLM_OK)
        throw std::runtime_error(pvlm_last_error(ctx_));
    if (pvlm_synchronize(ctx_) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    for (Bundle& b : bundles_) {   // the points are Ceres parameter blocks too: their current values travel with every evaluation
      for (size_t p = 0; p < b.points.size(); ++p) for (int k = 0; k < 3; ++k) b.X[3 * p + k] = b.points[p][k];
      if (pvlm_ba_set_points(ctx_, b.set, b.X.data()) != PVLM_OK || pvlm_ba_eval(ctx_, b.set, b.r.data(), b.J.data()) != PVLM_OK)
        throw std::runtime_error(pvlm_last_error(ctx_));
    }
    valid_ = true;
  }

 private:
  struct Set { pvlm_resset* set = nullptr; int64_t n = 0; int n_pairs = 0; int width = 7; double* rows = nullptr; double* tables = nullptr; std::vector<int> pair_of_row;
               std::vector<double> points; };
  struct Bundle { pvlm_baset* set = nullptr; int64_t n_obs = 0; std::vector<const double*> points; std::vector<double> r, J, X; };
  struct PoseArray { int n = 0; const double* aa = nullptr; const double* t = nullptr; int base = 0; };
  void Rebase() {
    int base = 0;
    for (PoseArray& a : arrays_) { a.base = base; base += a.n; }
    aa_all_.assign(3 * (size_t)base, 0.0); t_all_.assign(3 * (size_t)base, 0.0);
    valid_ = false;
  }
  pvlm_ctx* ctx_;
  std::vector<PoseArray> arrays_;
  std::vector<double> aa_all_, t_all_;
  std::vector<Set> sets_;
  std::vector<Bundle> bundles_;
  bool valid_ = false;
};

// Row i of a batched evaluation as a Ceres cost function: the same signature Ceres sees from
// AutoDiffCostFunction<Point2Plane_Angle,1,3,3,3,3> (base/CostFunction.h:721-727).  Evaluate only reads the batch,
// so Ceres' worker threads (num_threads = 25, config/Room.txt:24) may call it concurrently.
class CeresRow : public ceres::SizedCostFunction<1, 3, 3, 3, 3> {
 public:
  CeresRow(const CeresBatch* batch, int set, int64_t row) : batch_(batch), set_(set), row_(row) {}
  bool Evaluate(double const* const*, double* residuals, double** jacobians) const override {
    const double* w = batch_->row(set_, row_);                 // [r | c | g], or [r | g] for the point functors
    residuals[0] = w[0];
    if (jacobians) {
      const double* T = batch_->table(set_, row_);
      const double* R = T; const double* Jl = T + 15; const double* Mn = T + 24;
      double moment[3];
      const double* c = w + 1; const double* g = w + 4;
      if (batch_->row_width(set_) == 4) {
        // c = (P_r - t_rw) x g,  P_r = R_rn P_n + t_rn  (csrc/pvlm_functors.h eval_wrench, the same operations in the same order)
        g = w + 1;
        const double* P = batch_->point(set_, row_); const double* trn = T + 9; const double* trw = T + 12;
        const double m[3] = {R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + trn[0] - trw[0], R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + trn[1] - trw[1],
                             R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + trn[2] - trw[2]};
        moment[0] = m[1] * g[2] - m[2] * g[1]; moment[1] = m[2] * g[0] - m[0] * g[2]; moment[2] = m[0] * g[1] - m[1] * g[0];
        c = moment;
      }
      if (jacobians[0]) for (int k = 0; k < 3; ++k) jacobians[0][k] = c[0] * Jl[k] + c[1] * Jl[3 + k] + c[2] * Jl[6 + k];     // d/d angleAxis_rw
      if (jacobians[1]) for (int k = 0; k < 3; ++k) jacobians[1][k] = g[k];                                                    // d/d t_rw
      if (jacobians[2]) for (int k = 0; k < 3; ++k) jacobians[2][k] = c[0] * Mn[k] + c[1] * Mn[3 + k] + c[2] * Mn[6 + k];     // d/d angleAxis_nw
      if (jacobians[3]) for (int k = 0; k < 3; ++k) jacobians[3][k] = -(g[0] * R[k] + g[1] * R[3 + k] + g[2] * R[6 + k]);     // d/d t_nw
    }
    return std::isfinite(residuals[0]);
  }

 private:
  const CeresBatch* batch_; int set_; int64_t row_;
};

// Body for AddLidarPointToPlaneResidual (util/Optimization.cpp:506-562).  `scans[i]` is the device copy of lidars[i]
// (pvlm_scan_upload of its world-frame surfFlat / surfLessFlat clouds, pose and id); LidarT only needs IsPoseValid(),
// valid and id, i.e. PanoVLM's Velodyne.  PoseList = eigen_vector<Eigen::Vector3d>.
template <typename LidarT, typename PoseList>
size_t AddLidarPointToPlaneResidualGpu(CeresBatch& batch, pvlm_ctx* ctx, const std::vector<pvlm_scan*>& scans,
                                       const std::vector<std::vector<int>>& neighbors, const std::vector<LidarT>& lidars,
                                       PoseList& angleAxis_lw_list, PoseList& t_lw_list, ceres::Problem& problem,
                                       double point_to_plane_dis_threshold, double plane_tolerance, bool angle_residual,
                                       bool normalized_distance, double weight) {
  ceres::LossFunction* loss_function = new ceres::HuberLoss(angle_residual ? 2 * M_PI / 180.0 : 0.2);   // :513-517
  std::vector<pvlm_scan*> ref, nei;
  for (size_t i = 0; i < lidars.size(); i++) {                                   // the (i, n_idx) loop nest of :521-535
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    for (int n_idx : neighbors[i]) {
      if (n_idx < 0 || n_idx == (int)i || n_idx >= (int)lidars.size()) continue;
      if (!lidars[n_idx].IsPoseValid()) continue;
      ref.push_back(scans[i]); nei.push_back(scans[n_idx]);
    }
  }
  pvlm_resset* set = nullptr;
  if (pvlm_assoc_point2plane(ctx, (int)ref.size(), ref.data(), nei.data(), plane_tolerance, (float)point_to_plane_dis_threshold,
                             angle_residual ? PVLM_POINT2PLANE_ANGLE : PVLM_POINT2PLANE_METER,
                             normalized_distance ? PVLM_FLAG_NORMALIZE_DISTANCE : 0u, weight, &set) != PVLM_OK)
    throw std::runtime_error(pvlm_last_error(ctx));
  int64_t n = 0; int n_pairs = 0;
  pvlm_resset_info(set, &n, &n_pairs, nullptr, nullptr);
  if (n == 0) { pvlm_resset_destroy(ctx, set); delete loss_function; return 0; }
  std::vector<int64_t> offsets((size_t)n_pairs + 1);
  std::vector<int> pair_ref((size_t)n_pairs), pair_nei((size_t)n_pairs);       // = lidars[.].id of each segment
  pvlm_resset_download(ctx, set, offsets.data(), pair_ref.data(), pair_nei.data(), nullptr);
  std::vector<int> pair_of_row((size_t)n);
  for (int p = 0; p < n_pairs; ++p) for (int64_t row = offsets[p]; row < offsets[p + 1]; ++row) pair_of_row[(size_t)row] = p;
  const int set_id = batch.AddSet(set, std::move(pair_of_row));
  batch.SetPoseStorage((int)angleAxis_lw_list.size(), angleAxis_lw_list.data()->data(), t_lw_list.data()->data());
  for (int p = 0; p < n_pairs; ++p)                                              // same order as the reference's push_back order
    for (int64_t row = offsets[p]; row < offsets[p + 1]; ++row)
      problem.AddResidualBlock(new CeresRow(&batch, set_id, row), loss_function, angleAxis_lw_list[pair_ref[p]].data(),
                               t_lw_list[pair_ref[p]].data(), angleAxis_lw_list[pair_nei[p]].data(), t_lw_list[pair_nei[p]].data());
  return (size_t)n;
}


// Adds one CeresRow per block of `set` (taken over by the batch), in the set's order, on the parameter blocks
// (aa[ref], t[ref], aa[nei], t[nei]) of the segment table — ids below `split` index the first list pair, ids from `split` on the
// second (minus split).  Shared by the adapters below.
template <typename PoseListA, typename PoseListB>
inline size_t AddRowsOfSet(CeresBatch& batch, pvlm_ctx* ctx, pvlm_resset* set, ceres::LossFunction* loss, ceres::Problem& problem,
                           PoseListA& aa_a, PoseListA& t_a, PoseListB& aa_b, PoseListB& t_b, int split) {
  int64_t n = 0; int n_pairs = 0;
  pvlm_resset_info(set, &n, &n_pairs, nullptr, nullptr);
  std::vector<int64_t> offsets((size_t)n_pairs + 1);
  std::vector<int> pair_ref((size_t)n_pairs), pair_nei((size_t)n_pairs);
  if (pvlm_resset_download(ctx, set, offsets.data(), pair_ref.data(), pair_nei.data(), nullptr) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx));
  std::vector<int> pair_of_row((size_t)n);
  for (int p = 0; p < n_pairs; ++p) for (int64_t row = offsets[p]; row < offsets[p + 1]; ++row) pair_of_row[(size_t)row] = p;
  const int set_id = batch.AddSet(set, std::move(pair_of_row));
  auto aa_of = [&](int id) { return id < split ? aa_a[(size_t)id].data() : aa_b[(size_t)(id - split)].data(); };
  auto t_of = [&](int id) { return id < split ? t_a[(size_t)id].data() : t_b[(size_t)(id - split)].data(); };
  for (int p = 0; p < n_pairs; ++p)
    for (int64_t row = offsets[p]; row < offsets[p + 1]; ++row)
      problem.AddResidualBlock(new CeresRow(&batch, set_id, row), loss, aa_of(pair_ref[p]), t_of(pair_ref[p]), aa_of(pair_nei[p]), t_of(pair_nei[p]));
  return (size_t)n;
}

// Body for AddLidarLineToLineResidual2 (util/Optimization.cpp:329-441).  The association and the track filter stay what they
// are upstream (:379-400: AssociateLine2Line + lines_to_track / LineTrack::IsInside decide WHICH (reference segment, neighbour
// segment) pairs contribute); what moves to the GPU is the block building of :404-434 — one Point2Line block per POINT of the
// neighbour segment, 600 k heap-allocated cost functions per outer iteration at Room scale — through pvlm_line2line_residuals,
// which forms the same rows in the same order from the scans' resident segment point lists, and their evaluation.
//   associate(i, n_idx) -> the (ref_line_idx, neighbor_line_idx) list of AssociateLine2Line(lidars[i], lidars[n_idx], thr)
//   LidarT: IsPoseValid(), valid, id, edge_segmented[k].size();  TrackT: id, feature_pairs, IsInside()   (PanoVLM's Velodyne / LineTrack)
// Loss: nullptr for the angle variant (:417), HuberLoss(0.2) for the metric one (:337-340, :430) — as upstream.
template <typename LidarT, typename PoseList, typename TrackT, typename AssociateFn>
size_t AddLidarLineToLineResidual2Gpu(CeresBatch& batch, pvlm_ctx* ctx, const std::vector<pvlm_scan*>& scans, const std::vector<std::vector<int>>& neighbors,
                                      const std::vector<LidarT>& lidars, PoseList& angleAxis_lw_list, PoseList& t_lw_list, ceres::Problem& problem,
                                      const std::vector<TrackT>& lidar_line_tracks, AssociateFn associate, bool angle_residual, bool normalized_distance,
                                      double weight) {
  std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> lines_to_track;                 // :343-349
  for (const TrackT& track : lidar_line_tracks) for (const auto& pr : track.feature_pairs) lines_to_track[pr].push_back(track.id);
  std::vector<pvlm_scan*> ref, nei;
  std::vector<int> match_pair, match_nei_seg, match_ref_seg;
  size_t num_residual = 0;
  for (size_t i = 0; i < lidars.size(); i++) {                                                   // :353-400
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    for (int n_idx : neighbors[i]) {
      if (n_idx < 0 || n_idx == (int)i || n_idx >= (int)lidars.size()) continue;
      if (!lidars[n_idx].IsPoseValid() || !lidars[n_idx].valid) continue;
      bool pair_open = false;
      for (const auto& ass : associate(i, (size_t)n_idx)) {                                      // (ref_line_idx, neighbor_line_idx)
        auto it = lines_to_track.find({(uint32_t)i, (uint32_t)ass.first});
        if (it == lines_to_track.end()) continue;
        bool valid = false;
        for (uint32_t track_id : it->second) if (lidar_line_tracks[track_id].IsInside({(uint32_t)n_idx, (uint32_t)ass.second})) { valid = true; break; }
        if (!valid) continue;
        const size_t pts = lidars[n_idx].edge_segmented[(size_t)ass.second].size();
        if (pts == 0) continue;
        if (!pair_open) { ref.push_back(scans[i]); nei.push_back(scans[(size_t)n_idx]); pair_open = true; }
        match_pair.push_back((int)ref.size() - 1); match_nei_seg.push_back(ass.second); match_ref_seg.push_back(ass.first);
        num_residual += pts;
      }
    }
  }
  if (num_residual == 0) return 0;
  pvlm_resset* set = nullptr;
  if (pvlm_line2line_residuals(ctx, (int)ref.size(), ref.data(), nei.data(), (int)match_pair.size(), match_pair.data(), match_nei_seg.data(), match_ref_seg.data(),
                               angle_residual ? PVLM_POINT2LINE_ANGLE : PVLM_POINT2LINE_METER,
                               (angle_residual && normalized_distance) ? PVLM_FLAG_NORMALIZE_DISTANCE : 0u, weight, &set) != PVLM_OK)
    throw std::runtime_error(pvlm_last_error(ctx));
  ceres::LossFunction* loss_function = angle_residual ? nullptr : new ceres::HuberLoss(0.2);
  batch.SetPoseStorage((int)angleAxis_lw_list.size(), angleAxis_lw_list.data()->data(), t_lw_list.data()->data());
  return AddRowsOfSet(batch, ctx, set, loss_function, problem, angleAxis_lw_list, t_lw_list, angleAxis_lw_list, t_lw_list, (int)angleAxis_lw_list.size());
}

// Body for AddCameraLidarResidual (util/Optimization.cpp:564-607): per matched (image line, LiDAR segment) pair one
// Plane2Plane_Global block (weight line_pair.weight * weight) and one PlaneIOUResidual block (weight 2 * weight) on
// (angleAxis_cw[frame], t_cw[frame], angleAxis_lw[lidar], t_lw[lidar]) with the caller's loss.  The few lines that turn a line pair
// into the two functors' constructor arguments (:583-598: ImageToCam of the end points, FormPlane, VectorAngle3D, the two
// mid points) stay PanoVLM code — `make_rows(line_pair, weight, p2p, iou)` fills
//     p2p[10] = [plane.head(3) | lidar_line_end | lidar_line_start | line_pair.weight * weight]
//     iou[12] = [plane (4) | (end + start) / 2 | (p1 + p2) / 2 | angle | 2 * weight]
// (the argument order of Plane2Plane_Global::Create / PlaneIOUResidual::Create).  `cam_pose_base` = what
// batch.AddPoseArray(angleAxis_cw_list ...) returned; LiDAR poses are pose array 0.
template <typename FrameT, typename LidarT, typename PoseList, typename LinePairMap, typename MakeRowsFn>
size_t AddCameraLidarResidualGpu(CeresBatch& batch, pvlm_ctx* ctx, const std::vector<FrameT>& frames, const std::vector<LidarT>& lidars,
                                 PoseList& angleAxis_cw_list, PoseList& t_cw_list, PoseList& angleAxis_lw_list, PoseList& t_lw_list,
                                 const LinePairMap& line_pairs, ceres::LossFunction* loss_function, ceres::Problem& problem, double weight,
                                 int cam_pose_base, MakeRowsFn make_rows) {
  std::vector<double> rows_p2p, rows_iou;
  std::vector<int64_t> offsets{0};
  std::vector<int> pair_cam, pair_lidar;
  for (auto it = line_pairs.begin(); it != line_pairs.end(); it++) {                               // :572-581
    const size_t frame_id = it->first.first, lidar_id = it->first.second;
    if (!lidars[lidar_id].IsPoseValid() || !frames[frame_id].IsPoseValid()) continue;
    for (const auto& line_pair : it->second) {
      double p2p[10], iou[12];
      make_rows(line_pair, weight, p2p, iou);
      rows_p2p.insert(rows_p2p.end(), p2p, p2p + 10); rows_iou.insert(rows_iou.end(), iou, iou + 12);
    }
    if ((int64_t)(rows_p2p.size() / 10) == offsets.back()) continue;
    offsets.push_back((int64_t)(rows_p2p.size() / 10));
    pair_cam.push_back(cam_pose_base + (int)frame_id); pair_lidar.push_back((int)lidar_id);
  }
  const int64_t n = offsets.back();
  if (n == 0) return 0;
  pvlm_resset *set_p2p = nullptr, *set_iou = nullptr;
  const int n_pairs = (int)pair_cam.size();
  if (pvlm_resset_upload(ctx, PVLM_PLANE2PLANE_GLOBAL, 0u, 1.0, n, n_pairs, offsets.data(), pair_cam.data(), pair_lidar.data(), rows_p2p.data(), 10, &set_p2p) != PVLM_OK ||
      pvlm_resset_upload(ctx, PVLM_PLANE_IOU, 0u, 1.0, n, n_pairs, offsets.data(), pair_cam.data(), pair_lidar.data(), rows_iou.data(), 12, &set_iou) != PVLM_OK)
    throw std::runtime_error(pvlm_last_error(ctx));
  std::vector<int> pair_of_row((size_t)n);
  for (int p = 0; p < n_pairs; ++p) for (int64_t row = offsets[p]; row < offsets[p + 1]; ++row) pair_of_row[(size_t)row] = p;
  const int id_p2p = batch.AddSet(set_p2p, pair_of_row), id_iou = batch.AddSet(set_iou, pair_of_row);
  for (int p = 0; p < n_pairs; ++p) {
    const size_t f = (size_t)(pair_cam[p] - cam_pose_base), l = (size_t)pair_lidar[p];
    for (int64_t row = offsets[p]; row < offsets[p + 1]; ++row) {                                 // interleaved like upstream: plane-to-plane, then IOU
      problem.AddResidualBlock(new CeresRow(&batch, id_p2p, row), loss_function, angleAxis_cw_list[f].data(), t_cw_list[f].data(),
                               angleAxis_lw_list[l].data(), t_lw_list[l].data());
      problem.AddResidualBlock(new CeresRow(&batch, id_iou, row), loss_function, angleAxis_cw_list[f].data(), t_cw_list[f].data(),
                               angleAxis_lw_list[l].data(), t_lw_list[l].data());
    }
  }
  return (size_t)(2 * n);
}

// One observation of a batched reprojection evaluation as a Ceres cost function: the signature Ceres sees from
// AutoDiffCostFunction<PanoramaReprojResidual_1Angle, 1, 3, 3, 3> (base/CostFunction.h:218-247).
class CeresReprojRow : public ceres::SizedCostFunction<1, 3, 3, 3> {
 public:
  CeresReprojRow(const CeresBatch* batch, int bundle, int64_t obs) : batch_(batch), bundle_(bundle), obs_(obs) {}
  bool Evaluate(double const* const*, double* residuals, double** jacobians) const override {
    residuals[0] = batch_->reproj_residual(bundle_, obs_);
    if (jacobians) {
      const double* J = batch_->reproj_row(bundle_, obs_);
      for (int b = 0; b < 3; ++b) if (jacobians[b]) for (int k = 0; k < 3; ++k) jacobians[b][k] = J[3 * b + k];
    }
    return std::isfinite(residuals[0]);
  }

 private:
  const CeresBatch* batch_; int bundle_; int64_t obs_;
};

// Body for AddCameraResidual (util/Optimization.cpp:172-222), ANGLE_RESIDUAL_1 — the variant CameraLidarOptimizer::Optimize uses
// (joint_optimization/CameraLidarOptimizer.cpp:431-432): one PanoramaReprojResidual_1Angle per (track, observation) whose frame
// has a valid pose, on (angleAxis_cw[frame], t_cw[frame], track.point_3d), HuberLoss(4 deg).
//   bearing(frame_idx, keypoint_idx, out[3]) = eq.ImageToCam(frames[frame_idx].GetKeyPoints()[keypoint_idx].pt)   (:201, PanoVLM code)
//   TrackT: feature_pairs (frame index, keypoint index), point_3d   (PanoVLM's PointTrack)
template <typename FrameT, typename PoseList, typename TrackT, typename BearingFn>
size_t AddCameraResidualGpu(CeresBatch& batch, pvlm_ctx* ctx, const std::vector<FrameT>& frames, PoseList& angleAxis_cw_list, PoseList& t_cw_list,
                            std::vector<TrackT>& structure, ceres::Problem& problem, double weight, int cam_pose_base, BearingFn bearing) {
  ceres::LossFunction* loss_function = new ceres::HuberLoss(4.0 * M_PI / 180.0);                  // :179-181
  std::vector<int64_t> point_offsets{0};
  std::vector<int> cam_ids, obs_frame, obs_point;
  std::vector<double> bearings, points;
  std::vector<const double*> point_blocks;
  for (size_t i = 0; i < structure.size(); i++) {                                                // :186-219
    TrackT& track = structure[i];
    for (const auto& pair : track.feature_pairs) {
      const uint32_t frame_idx = pair.first;
      if (!frames[frame_idx].IsPoseValid()) continue;
      double b[3];
      bearing((size_t)frame_idx, (size_t)pair.second, b);
      bearings.insert(bearings.end(), b, b + 3);
      cam_ids.push_back(cam_pose_base + (int)frame_idx); obs_frame.push_back((int)frame_idx); obs_point.push_back((int)i);
    }
    point_offsets.push_back((int64_t)cam_ids.size());
    for (int k = 0; k < 3; ++k) points.push_back(track.point_3d.data()[k]);
    point_blocks.push_back(track.point_3d.data());
  }
  const int64_t n_obs = (int64_t)cam_ids.size();
  if (n_obs == 0) { delete loss_function; return 0; }
  pvlm_baset* set = nullptr;
  if (pvlm_ba_create(ctx, (int)structure.size(), n_obs, point_offsets.data(), cam_ids.data(), bearings.data(), points.data(), weight, &set) != PVLM_OK)
    throw std::runtime_error(pvlm_last_error(ctx));
  const int bundle = batch.AddBundle(set, std::move(point_blocks));
  for (int64_t o = 0; o < n_obs; ++o)
    problem.AddResidualBlock(new CeresReprojRow(&batch, bundle, o), loss_function, angleAxis_cw_list[(size_t)obs_frame[(size_t)o]].data(),
                             t_cw_list[(size_t)obs_frame[(size_t)o]].data(), structure[(size_t)obs_point[(size_t)o]].point_3d.data());
  return (size_t)n_obs;
}

}  // namespace pvlm
