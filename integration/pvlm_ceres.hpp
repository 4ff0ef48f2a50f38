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
// Ceres itself is not in this repository's image: the file is written against the public Ceres 2.0 API and PanoVLM's own
// types, and is what a PanoVLM maintainer adds next to util/Optimization.cpp.  It contains no PanoVLM code; only
// include/pvlm.h (C ABI) is used from this repository.  CI compiles and RUNS it against an interface-only test double of
// the few Ceres classes it touches (tests/cpp/ceres_double/ceres/ceres.h — labelled as such: it pins nothing about Ceres'
// behaviour, it only proves that this file compiles and that the rows it hands out equal pvlm_eval's;
// tests/test_host_gpu.py::test_ceres_adapter_rows).
//
// The boundary is a PCIe link (30 GB/s on the MI355X box): the batch is delivered as 56-byte wrench rows [r | c | g]
// (pvlm_eval_wrench_host_async) into page-locked memory, asynchronously, and the 1 x 12 Jacobian row is formed from the
// row and its pair's 3x3 tables inside Evaluate — 36 multiply-adds on the Ceres worker thread that asks for it —
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

#include <cmath>
#include <stdexcept>
#include <vector>

#include "pvlm.h"

namespace pvlm {

// One batched GPU evaluation per Ceres evaluation point, shared by all the row functors below.
class CeresBatch : public ceres::EvaluationCallback {
 public:
  explicit CeresBatch(pvlm_ctx* ctx) : ctx_(ctx) {}
  ~CeresBatch() override {
    for (Set& s : sets_) { pvlm_host_free(ctx_, s.rows); pvlm_host_free(ctx_, s.tables); pvlm_resset_destroy(ctx_, s.set); }
  }
  CeresBatch(const CeresBatch&) = delete;
  CeresBatch& operator=(const CeresBatch&) = delete;

  // The pose storage Ceres optimises in place: angleAxis_lw_list / t_lw_list (LidarOdometry.cpp:23-33), n x 3 each,
  // contiguous (eigen_vector<Eigen::Vector3d>::data()->data()).
  void SetPoseStorage(int n_poses, const double* angle_axis, const double* translation) { n_ = n_poses; aa_ = angle_axis; t_ = translation; }

  // Takes ownership of a residual set produced by the association kernels; `pair_of_row[i]` = segment of block i.
  // Returns the set's index.
  int AddSet(pvlm_resset* set, std::vector<int> pair_of_row) {
    Set s; s.set = set; s.pair_of_row = std::move(pair_of_row);
    pvlm_resset_info(set, &s.n, &s.n_pairs, nullptr, nullptr);
    void* p = nullptr;
    if (pvlm_host_alloc(ctx_, (int64_t)s.n * 7 * (int64_t)sizeof(double), &p) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    s.rows = static_cast<double*>(p);
    if (pvlm_host_alloc(ctx_, (int64_t)s.n_pairs * PVLM_PAIR_TABLE * (int64_t)sizeof(double), &p) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    s.tables = static_cast<double*>(p);
    sets_.push_back(std::move(s));
    return (int)sets_.size() - 1;
  }
  // [r | c(3) | g(3)] of block `row`, and the table of its pair: [R_rn(9) | t_rn(3) | t_rw(3) | J_l(aa_r)(9) | M_n(9)]
  const double* row(int set, int64_t row) const { return sets_[(size_t)set].rows + (size_t)row * 7; }
  const double* table(int set, int64_t row) const { const Set& s = sets_[(size_t)set]; return s.tables + (size_t)s.pair_of_row[(size_t)row] * PVLM_PAIR_TABLE; }

  // ceres::EvaluationCallback — called once before the residual blocks are evaluated at a (possibly new) point.  The
  // wrench rows serve cost-only and Jacobian evaluations alike, so a repeated call at the same point costs nothing.
  void PrepareForEvaluation(bool /*evaluate_jacobians*/, bool new_evaluation_point) override {
    if (!new_evaluation_point && valid_) return;
    if (pvlm_set_poses(ctx_, n_, aa_, t_) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    for (Set& s : sets_)   // ONE kernel per slice of a set, its copy queued right behind it; nothing waits until the end
      if (pvlm_eval_wrench_host_async(ctx_, s.set, s.rows, s.tables) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    if (pvlm_synchronize(ctx_) != PVLM_OK) throw std::runtime_error(pvlm_last_error(ctx_));
    valid_ = true;
  }

 private:
  struct Set { pvlm_resset* set = nullptr; int64_t n = 0; int n_pairs = 0; double* rows = nullptr; double* tables = nullptr; std::vector<int> pair_of_row; };
  pvlm_ctx* ctx_;
  int n_ = 0; const double* aa_ = nullptr; const double* t_ = nullptr;
  std::vector<Set> sets_;
  bool valid_ = false;
};

// Row i of a batched evaluation as a Ceres cost function: the same signature Ceres sees from
// AutoDiffCostFunction<Point2Plane_Angle,1,3,3,3,3> (base/CostFunction.h:721-727).  Evaluate only reads the batch,
// so Ceres' worker threads (num_threads = 25, config/Room.txt:24) may call it concurrently.
class CeresRow : public ceres::SizedCostFunction<1, 3, 3, 3, 3> {
 public:
  CeresRow(const CeresBatch* batch, int set, int64_t row) : batch_(batch), set_(set), row_(row) {}
  bool Evaluate(double const* const*, double* residuals, double** jacobians) const override {
    const double* w = batch_->row(set_, row_);                 // [r | c | g]
    residuals[0] = w[0];
    if (jacobians) {
      const double* T = batch_->table(set_, row_);
      const double* c = w + 1; const double* g = w + 4;
      const double* R = T; const double* Jl = T + 15; const double* Mn = T + 24;
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

}  // namespace pvlm
