// pvlm_host_optimization.cpp — part of the C++ host mirror (pvlm_host.hpp): the adders of util/Optimization.cpp: AddLidarPointToPlaneResidual, AddLidarLineToLineResidual2, AddLidarPointToLineResidual (:329-562), AddCameraLidarResidual (:564-607), AddCameraResidual (:172-222).
// Host logic only; every residual, Jacobian, distance and vote is produced by libpvlm.so on the GPU.
#include "pvlm_host_internal.hpp"

namespace pvlm {

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
  // lines_to_track (upstream: std::map keyed by (lidar, line), :369-377) is only looked up, never iterated: here a table with one row per segment —
  // scan s owns the rows seg_base[s] .. seg_base[s + 1] — listing the tracks the segment belongs to.  "Some track of the reference segment holds the
  // neighbour's segment" (:387-395, LineTrack::IsInside: a std::set walk per candidate) is then "the two rows share a track id".
  std::vector<size_t> seg_base(lidars.size() + 1, 0);
  for (size_t q = 0; q < lidars.size(); ++q) seg_base[q + 1] = seg_base[q] + lidars[q].edge_segmented.size();
  std::vector<uint32_t> row_start(seg_base.back() + 2, 0), row_tracks;
  auto row_of = [&](uint32_t lidar, uint32_t line) -> long long {
    if ((size_t)lidar >= lidars.size() || (size_t)line >= lidars[lidar].edge_segmented.size()) return -1;
    return (long long)(seg_base[lidar] + line);
  };
  {
    StageTimer stage_timer_l2t_("  (inside) line-to-line: lines_to_track map (host)");
    for (const LineTrack& t : tracks) for (const auto& pr : t.feature_pairs) { const long long r = row_of(pr.first, pr.second); if (r >= 0) ++row_start[(size_t)r + 2]; }
    for (size_t r = 2; r < row_start.size(); ++r) row_start[r] += row_start[r - 1];            // row_start[r + 1] = first entry of row r, advanced while filling
    row_tracks.resize(row_start.back());
    for (const LineTrack& t : tracks) for (const auto& pr : t.feature_pairs) { const long long r = row_of(pr.first, pr.second); if (r >= 0) row_tracks[row_start[(size_t)r + 1]++] = t.id; }
  }
  auto share_a_track = [&](long long a, long long b) {
    if (a < 0 || b < 0) return false;
    for (uint32_t x = row_start[(size_t)a]; x < row_start[(size_t)a + 1]; ++x)
      for (uint32_t y = row_start[(size_t)b]; y < row_start[(size_t)b + 1]; ++y)
        if (row_tracks[x] == row_tracks[y]) return true;
    return false;
  };
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
    // a worker takes 64 consecutive pairs at a time and writes a pair's results once: neighbouring entries of `kept` / `kept_points` share cache
    // lines, and with the pairs dealt out one by one every match was a write to a line seven other workers were writing too (17 ms per call at Floor
    // size whatever the thread count)
    auto work = [&]() {
      std::vector<std::pair<int, int>> mine;
      for (size_t p0 = cursor.fetch_add(64); p0 < pairs_todo.size(); p0 = cursor.fetch_add(64)) {
        for (size_t p = p0; p < std::min(p0 + 64, pairs_todo.size()); ++p) {
          const size_t i = pairs_todo[p].i; const int n_idx = pairs_todo[p].n_idx;
          mine.clear();
          size_t points = 0;
          for (const Line2Line& a : all_ass[next + p]) {
            if (!share_a_track(row_of((uint32_t)i, (uint32_t)a.ref_line_idx), row_of((uint32_t)n_idx, (uint32_t)a.neighbor_line_idx))) continue;
            const size_t pts = lidars[n_idx].edge_segmented[a.neighbor_line_idx].size();
            if (pts == 0) continue;
            mine.push_back({a.neighbor_line_idx, a.ref_line_idx});
            points += pts;
          }
          if (!mine.empty()) { kept[p] = mine; kept_points[p] = points; }
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


}  // namespace pvlm
