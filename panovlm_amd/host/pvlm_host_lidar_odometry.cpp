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


}  // namespace pvlm
