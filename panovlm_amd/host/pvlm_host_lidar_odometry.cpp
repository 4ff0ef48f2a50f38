// pvlm_host_lidar_odometry.cpp — part of the C++ host mirror (pvlm_host.hpp): lidar_mapping/LidarOdometry.cpp:15-187 (RefinePose, EstimatePose).
// Host logic only; every residual, Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host_internal.hpp"

namespace pvlm {

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
    StageTimer stage_timer_features_("feature extraction (range image, picks and voxel grid on the GPU, EdgeToLine on the host)");
    // invalid scans first, on the calling thread: SetRotation / SetTranslation give the scan's device copy back to the engine's
    // context (InvalidateDevice -> pvlm_scan_destroy), whose pool is not thread-safe — never from the workers below
    for (Velodyne& l : lidars)
      if (!l.valid || !l.IsPoseValid()) { l.SetRotation({0, 0, 0, 0, 0, 0, 0, 0, 0}); l.SetTranslation({INFINITY, INFINITY, INFINITY}); }
    // the scans that still need their features: range-image stages, picks and voxel grid of all of them in GPU batches, EdgeToLine on config.num_threads
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


// ---- base/Geometry.hpp:572-583, lidar_mapping/LidarOdometry.cpp:189-263 -----------------------------------------------------------------------
namespace {
struct Rigid { double R[9], t[3]; };
Rigid RigidOf(const Matrix4d& T) { Rigid o; for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) o.R[3 * r + c] = T[4 * r + c]; o.t[r] = T[4 * r + 3]; } return o; }
Matrix4d MatrixOf(const Rigid& p) { return {p.R[0], p.R[1], p.R[2], p.t[0], p.R[3], p.R[4], p.R[5], p.t[1], p.R[6], p.R[7], p.R[8], p.t[2], 0, 0, 0, 1}; }
Rigid InverseOf(const Rigid& p) {       // a pose's inverse as (R^T, -R^T t); upstream's Matrix4d::inverse() of the same matrix agrees to the last bits
  Rigid o;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o.R[3 * r + c] = p.R[3 * c + r];
  for (int r = 0; r < 3; ++r) o.t[r] = -((o.R[3 * r] * p.t[0] + o.R[3 * r + 1] * p.t[1]) + o.R[3 * r + 2] * p.t[2]);
  return o;
}
Rigid Compose(const Rigid& a, const Rigid& b) {
  Rigid o;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o.R[3 * r + c] = (a.R[3 * r] * b.R[c] + a.R[3 * r + 1] * b.R[3 + c]) + a.R[3 * r + 2] * b.R[6 + c];
    o.t[r] = ((a.R[3 * r] * b.t[0] + a.R[3 * r + 1] * b.t[1]) + a.R[3 * r + 2] * b.t[2]) + a.t[r];
  }
  return o;
}
}  // namespace

Matrix4d SlerpPose(const Matrix4d& pose_w1, const Matrix4d& pose_w2, double ratio) {
  const Rigid w1 = RigidOf(pose_w1);
  const Rigid T_21 = Compose(InverseOf(RigidOf(pose_w2)), w1);
  const pvlm_undistort::Quat q_s1 = pvlm_undistort::slerp_at(pvlm_undistort::slerp_prepare(pvlm_undistort::quat_of_matrix(T_21.R)), ratio);
  Rigid T_s1;
  pvlm_undistort::matrix_of_quat(q_s1, T_s1.R);
  for (int k = 0; k < 3; ++k) T_s1.t[k] = T_21.t[k] * ratio;
  return MatrixOf(Compose(w1, InverseOf(T_s1)));
}

void LidarOdometry::ReloadClouds(const std::vector<PointCloud>& clouds) {
  if (clouds.size() != lidars.size()) throw std::invalid_argument("ReloadClouds: one cloud per scan");
  for (size_t k = 0; k < lidars.size(); ++k) lidars[k].cloud = clouds[k];
}

void LidarOdometry::ResetAllLidars() {
  for (Velodyne& lidar : lidars) {
    lidar.Reset();
    if (lidar.cloud.empty() && lidar.IsPoseValid()) lidar.LoadLidar(lidar.name);
  }
}

bool LidarOdometry::UndistortLidars(const float gap_time) {
  StageTimer stage_timer_("motion compensation of the sweeps (UndistortLidars)");
  const double lidar_duration = 0.1;
  const int n = (int)lidars.size();
  std::vector<Velodyne*> scans;
  std::vector<Matrix4d> ends;
  for (int i = 0; i < n; i++) {
    // the pose of the sweep's last point: the next scan's pose — the next one that has a pose — interpolated back to the end of this sweep (:206-241;
    // the conditions are upstream's as written: a neighbour is passed over only when it has neither a pose nor the valid flag, the backward search
    // tests this scan's flag, and idx <= 0 gives up)
    Matrix4d pose;
    if (!lidars[i].IsPoseValid() || !lidars[i].valid) continue;
    if (i < n - 1) {
      int idx = i + 1;
      while (idx < n && !lidars[idx].IsPoseValid() && !lidars[idx].valid) idx++;
      if (idx >= n) continue;
      pose = SlerpPose(lidars[i].GetPose(), lidars[idx].GetPose(), lidar_duration / ((idx - i) * (lidar_duration + gap_time)));
    } else {
      int idx = i - 1;
      while (idx >= 0 && !lidars[idx].IsPoseValid() && !lidars[i].valid) idx--;
      if (idx <= 0) continue;
      // one scan period before this scan, mirrored to one period after it: the three poses are taken to move alike
      pose = SlerpPose(lidars[idx].GetPose(), lidars[i].GetPose(), 1.0 - lidar_duration / ((idx - i) * (lidar_duration + gap_time)));
      const Rigid cur = RigidOf(lidars[i].GetPose());
      pose = MatrixOf(Compose(cur, Compose(InverseOf(cur), RigidOf(pose))));
    }
    scans.push_back(&lidars[i]);
    ends.push_back(pose);
  }
  Velodyne::UndistortBatch(scans, ends);
  return true;
}

}  // namespace pvlm
