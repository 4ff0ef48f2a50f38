// pvlm_ceres.hpp — keep Ceres as the outer trust-region solver, evaluate PanoVLM's LiDAR residual blocks on the GPU.
//
// Reference-side binding for integration level B of INTEGRATION.md: a drop-in body for
// AddLidarPointToPlaneResidual (util/Optimization.cpp:506-562) that
//   1. associates every (ref, neighbour) scan pair on the GPU in one call (pvlm_assoc_point2plane — the result stays in
//      HBM as a residual set),
//   2. adds ONE thin ceres::SizedCostFunction<1,3,3,3,3> per correspondence, in the reference's order, with the
//      reference's parameter-block pointers and its shared HuberLoss — so Ceres' problem structure, robust loss and
//      cost accounting are exactly what they were,
//   3. evaluates all of them with one batched kernel launch per Ceres evaluation point through a
//      ceres::EvaluationCallback (Problem::Options::evaluation_callback, Ceres >= 2.0).
//
// NOT compiled in this repository's image (no Ceres / Eigen / PCL there): it is written against the public Ceres 2.0
// API and PanoVLM's own types, and is the file a PanoVLM maintainer adds next to util/Optimization.cpp.  It contains
// no PanoVLM code.  Only include/pvlm.h (C ABI) is used from this repository.
//
// Usage inside LidarOdometry::RefinePose (lidar_mapping/LidarOdometry.cpp:36-80):
//     pvlm::CeresBatch batch(ctx);                                  // before the ceres::Problem
//     ceres::Problem::Options po; po.evaluation_callback = &batch;
//     ceres::Problem problem(po);
//     ... pvlm::AddLidarPointToPlaneResidualGpu(batch, scans, neighbors, lidars, angleAxis_lw_list, t_lw_list, problem, ...);
//     ceres::Solve(options, &problem, &summary);                    // unchanged
#pragma once
#include <ceres/ceres.h>

#include <cmath>
#include <stdexcept>
#include <vector>

#include "pvlm.h"

namespace pvlm {

// One batched GPU evaluation per Ceres evaluation point, shared by all the row functors below.
class CeresBatch : public ceres::EvaluationCallback {
 public:
  explicit CeresBatch(pvlm_ctx* ctx) : ctx_(ctx) {}
  ~CeresBatch() override { for (Set& s : sets_) pvlm_resset_destroy(ctx_, s.set); }

  // The pose storage Ceres optimises in place: angleAxis_lw_list / t_lw_list (LidarOdometry.cpp:23-33), n x 3 each,
  // contiguous (eigen_vector<Eigen::Vector3d>::data()->data()).
  void SetPoseStorage(int n_poses, const double* angle_axis, const double* translation) { n_ = n_poses; aa_ = angle_axis; t_ = translation; }

  // Takes ownership of a residual set produced by the association kernels; returns its index.
  int AddSet(pvlm_resset* set) {
    Set s; s.set = set;
    pvlm_resset_info(set, &s.n, nullptr, nullptr, nullptr);
    s.r.resize((size_t)s.n); s.J.resize((size_t)s.n * 12);
    sets_.push_back(std::move(s));
    return (int)sets_.size() - 1;
  }
  const double* residual(int set, int64_t row) const { return &sets_[set].r[(size_t)row]; }
  const double* jacobian(int set, int64_t row) const { return &sets_[set].J[(size_t)row * 12]; }

  // ceres::EvaluationCallback — called once before the residual blocks are evaluated at a (possibly new) point.
  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    if (!new_evaluation_point && (have_jacobians_ || !evaluate_jacobians)) return;
    if (pvlm_set_poses(ctx_, n_, aa_, t_) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    for (Set& s : sets_)   // ONE kernel launch per set for all of its residual blocks (K5, materialise mode)
      if (pvlm_eval(ctx_, s.set, s.r.data(), evaluate_jacobians ? s.J.data() : nullptr) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    have_jacobians_ = evaluate_jacobians;
  }

 private:
  struct Set { pvlm_resset* set = nullptr; int64_t n = 0; std::vector<double> r, J; };
  pvlm_ctx* ctx_;
  int n_ = 0; const double* aa_ = nullptr; const double* t_ = nullptr;
  std::vector<Set> sets_;
  bool have_jacobians_ = false;
};

// Row i of a batched evaluation as a Ceres cost function: the same signature Ceres sees from
// AutoDiffCostFunction<Point2Plane_Angle,1,3,3,3,3> (base/CostFunction.h:721-727).  Evaluate only reads the batch,
// so Ceres' worker threads (num_threads = 25, config/Room.txt:24) may call it concurrently.
class CeresRow : public ceres::SizedCostFunction<1, 3, 3, 3, 3> {
 public:
  CeresRow(const CeresBatch* batch, int set, int64_t row) : batch_(batch), set_(set), row_(row) {}
  bool Evaluate(double const* const*, double* residuals, double** jacobians) const override {
    residuals[0] = *batch_->residual(set_, row_);
    if (jacobians) {
      const double* J = batch_->jacobian(set_, row_);   // [d/daa_r | d/dt_r | d/daa_n | d/dt_n]
      for (int b = 0; b < 4; ++b)
        if (jacobians[b]) { jacobians[b][0] = J[3 * b]; jacobians[b][1] = J[3 * b + 1]; jacobians[b][2] = J[3 * b + 2]; }
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
  const int set_id = batch.AddSet(set);
  batch.SetPoseStorage((int)angleAxis_lw_list.size(), angleAxis_lw_list.data()->data(), t_lw_list.data()->data());
  for (int p = 0; p < n_pairs; ++p)                                              // same order as the reference's push_back order
    for (int64_t row = offsets[p]; row < offsets[p + 1]; ++row)
      problem.AddResidualBlock(new CeresRow(&batch, set_id, row), loss_function, angleAxis_lw_list[pair_ref[p]].data(),
                               t_lw_list[pair_ref[p]].data(), angleAxis_lw_list[pair_nei[p]].data(), t_lw_list[pair_nei[p]].data());
  return (size_t)n;
}

}  // namespace pvlm
