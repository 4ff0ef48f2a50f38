// Line branch of the LiDAR feature extraction — the segments AssociateLine2Line consumes:
//   Velodyne::EdgeToLine        sensors/Velodyne.cpp:1269-1324
//   ExtractLineFeatures         sensors/LidarLineExtraction.cpp:296-389  (seed = a point + two of its four nearest neighbours,
//                               grown at both ends by ExpandLine :10-70 while the enlarged set stays a line)
//   FuseLineSegments            sensors/LidarLineExtraction.cpp:177-250  (+ FindNeighbors :72-110, FuseLines :113-175)
//   FilterLineByScan / ByLength sensors/LidarLineExtraction.cpp:254-294
// Host code, like upstream and like the planar branch next door (pvlm_features.cpp): a few hundred edge points per scan,
// every step a dependency chain; scans are processed in parallel by LidarOdometry::EstimatePose.
//
// Own design, same results as the statement-by-statement restatement the test oracle holds (tests/test_lines_cpu.py compares
// every output array): the 5 nearest neighbours of every edge point are computed once (upstream queries a kd-tree with a
// point of the cloud every time), segment membership is a sorted vector + a stamp array instead of std::set<int>, groups
// of segments are found by a plain reachability search.
//
// ONE REDEFINITION, shared with the oracle and documented in DESIGN.md: upstream's FuseLines fits a fused group with
// pcl::SACSegmentation RANSAC (2-point samples from a generator inside PCL, 0.02 m, no refinement).  Here the fit is the
// limit RANSAC approximates — the exhaustive 2-point maximum-consensus line — which is deterministic, contains at least
// as many inliers as any RANSAC draw, and obeys the same acceptance rule (more than 4 inliers).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "pvlm_host.hpp"
#include "../csrc/pvlm_linegrow_core.h"

namespace pvlm {
namespace {

struct NeighbourTable {            // k nearest neighbours of every point of a cloud, ascending (float d2, index); k = min(5, n)
  int k = 0;
  std::vector<int> idx;
  std::vector<float> sqd;
};

NeighbourTable BuildNeighbourTable(const float* xyz, int stride, int n) {
  NeighbourTable t;
  t.k = std::min(5, n);
  t.idx.resize((size_t)n * t.k); t.sqd.resize((size_t)n * t.k);
  // the k smallest (d2, index) pairs in ascending order, kept in a k-entry list while the points stream by in index order (what a partial sort of all n
  // pairs returns: equal distances by index, and a later index never displaces an equal earlier one)
  for (int q = 0; q < n; ++q) {
    const float qx = xyz[(size_t)q * stride], qy = xyz[(size_t)q * stride + 1], qz = xyz[(size_t)q * stride + 2];
    float bd[5]; int bi[5]; int have = 0;
    for (int i = 0; i < n; ++i) {
      const float dx = qx - xyz[(size_t)i * stride], dy = qy - xyz[(size_t)i * stride + 1], dz = qz - xyz[(size_t)i * stride + 2];
      float s = 0.0f;
      s += dx * dx; s += dy * dy; s += dz * dz;                       // flann::L2_Simple order
      if (have == t.k && !(s < bd[have - 1])) continue;
      int at = have < t.k ? have++ : have - 1;
      while (at > 0 && s < bd[at - 1]) { bd[at] = bd[at - 1]; bi[at] = bi[at - 1]; --at; }
      bd[at] = s; bi[at] = i;
    }
    for (int j = 0; j < t.k; ++j) { t.idx[(size_t)q * t.k + j] = bi[j]; t.sqd[(size_t)q * t.k + j] = bd[j]; }
  }
  return t;
}

inline double Gap2(const double* a, const double* b) {        // the argument of Gap's square root
  const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
  return x * x + (y * y + z * z);
}
inline double Gap(const double* a, const double* b) { return std::sqrt(Gap2(a, b)); }
inline double PointToLine(const double* p, const double* l) {        // base/Geometry.hpp:198-211
  const double k = (l[3] * (p[0] - l[0]) + l[4] * (p[1] - l[1]) + l[5] * (p[2] - l[2])) / (l[3] * l[3] + l[4] * l[4] + l[5] * l[5]);
  const double q[3] = {k * l[3] + l[0], k * l[4] + l[1], k * l[5] + l[2]};
  return std::sqrt((q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]) + (q[2] - p[2]) * (q[2] - p[2]));
}
inline double DirectionAngle(const double* a, const double* b) {      // PlaneAngle, base/Geometry.hpp:471-485
  double c = std::fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  c = c / (std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]));
  return c >= 1.0 ? 0.0 : std::acos(c);
}
inline bool AllZero(const Vector6d& l) { for (double v : l) if (v != 0.0) return false; return true; }

// the two points of `ids` (in the given order) that are farthest apart; first pair among equals (FurthestPoints, Eigen flavour)
// (upstream compares the distances, g > length: the square root is monotone, so a pair whose SQUARED distance does not exceed the one behind the current
// maximum cannot exceed it either — the root is only taken for the others, and the comparison of the roots still decides)
void Extremes(const std::vector<double>& P, const std::vector<int>& ids, int* a, int* b, double* length) {
  *a = *b = -1; *length = -1;
  double longest2 = -1;                                       // squared distance of the pair that holds *length
  for (size_t i = 0; i + 1 < ids.size(); ++i)
    for (size_t j = i + 1; j < ids.size(); ++j) {
      const double g2 = Gap2(&P[3 * (size_t)ids[i]], &P[3 * (size_t)ids[j]]);
      if (!(g2 > longest2) && g2 == g2) continue;
      const double g = std::sqrt(g2);
      if (g > *length) { *a = (int)i; *b = (int)j; *length = g; longest2 = g2; }
    }
}
// FurthestPoints on a cloud (float squared distances, one square root): base/Geometry.hpp:619-645
void ExtremesOfCloud(const PointCloud& c, int* a, int* b, double* length) {
  *a = *b = -1; *length = -1;
  for (size_t i = 0; i + 1 < c.size(); ++i)
    for (size_t j = i + 1; j < c.size(); ++j) {
      const float dx = c[i].x - c[j].x, dy = c[i].y - c[j].y, dz = c[i].z - c[j].z;
      const double g = dx * dx + dy * dy + dz * dz;
      if (g > *length) { *a = (int)i; *b = (int)j; *length = g; }
    }
  *length = std::sqrt(*length);
}

struct Grower {
  const std::vector<double>& P;      // edge points as doubles, 3 per point
  const NeighbourTable& nn;
  std::vector<int> stamp;            // stamp[id] == epoch  <=>  id is a member of the segment being grown
  int epoch = 0;
  std::vector<double> buf;
  std::vector<int> order_buf;        // Expand's working list (1 400 calls per scan: no allocation each)

  Grower(const std::vector<double>& p, const NeighbourTable& t) : P(p), nn(t), stamp(p.size() / 3, 0) {}

  Vector6d Fit(const std::vector<int>& order, double tolerance, double dis_threshold = 0) {
    buf.resize(order.size() * 3);
    for (size_t i = 0; i < order.size(); ++i) for (int k = 0; k < 3; ++k) buf[3 * i + k] = P[3 * (size_t)order[i] + k];
    Vector6d l{};
    FormLine3D(buf.data(), (int)order.size(), tolerance, dis_threshold, l.data());
    return l;
  }

  // ExpandLine: tries the (up to four) nearest neighbours of edge point `start`; members: ascending ids, updated in place
  bool Expand(int start, std::vector<int>& members) {
    bool grown = false;
    std::vector<int>& order = order_buf;                  // the order the scatter sums see: members ascending, then accepted points as they come
    order.assign(members.begin(), members.end());
    int a, b; double length;
    Extremes(P, order, &a, &b, &length);
    // upstream fits the current members first (tolerance 3) and uses that line only in the long-line branch below; the fit is made when that
    // branch first asks for it — until a candidate is accepted, `order` without its last entry is still the member list the fit would have seen
    Vector6d line{};
    bool line_known = false;
    for (int j = 1; j < nn.k; ++j) {
      const int cand = nn.idx[(size_t)start * nn.k + j];
      if (stamp[(size_t)cand] == epoch) continue;
      if (nn.sqd[(size_t)start * nn.k + j] > (length / 2) * (length / 2)) break;
      order.push_back(cand);
      double reach2 = -1;                                  // max of the distances = root of the max of the squares (monotone, correctly rounded)
      for (int id : order) reach2 = std::max(reach2, Gap2(&P[3 * (size_t)id], &P[3 * (size_t)cand]));
      double reach = reach2 < 0 ? -1 : std::sqrt(reach2);
      reach = std::max(reach, length);
      Vector6d next;
      if (reach < 2) {
        next = Fit(order, 5.0, 0.07);                     // short lines: every point within 7 cm
        if (AllZero(next)) { order.pop_back(); continue; }
      } else {
        if (!line_known) { order.pop_back(); line = Fit(order, 3.0); order.push_back(cand); line_known = true; }
        next = Fit(order, 20.0);                          // long lines: much straighter, and the direction may not turn by more than 1 degree
        const double turn = DirectionAngle(&next[3], &line[3]) * 180.0 / M_PI;
        if (AllZero(next) || turn > 1) { order.pop_back(); continue; }
      }
      grown = true;
      stamp[(size_t)cand] = epoch;
      members.insert(std::upper_bound(members.begin(), members.end(), cand), cand);
      length = reach;
      line = next; line_known = true;
    }
    return grown;
  }
};

// exhaustive 2-point maximum consensus (see the header): indices of the inliers, ascending
std::vector<int> Consensus(const PointCloud& c, double threshold) {
  const int n = (int)c.size();
  std::vector<double> q((size_t)n * 3);
  for (int i = 0; i < n; ++i) { q[3 * (size_t)i] = c[(size_t)i].x; q[3 * (size_t)i + 1] = c[(size_t)i].y; q[3 * (size_t)i + 2] = c[(size_t)i].z; }
  const double t2 = threshold * threshold;
  // inliers of the line through points i and j; `beat`: the count to exceed — the walk stops (returning 0) as soon as the points left cannot lift the count
  // above it: only a count that exceeds the best so far is ever used
  auto count = [&](int i, int j, std::vector<int>* out, int beat) {
    const double* o = &q[3 * (size_t)i];
    const double d[3] = {q[3 * (size_t)j] - o[0], q[3 * (size_t)j + 1] - o[1], q[3 * (size_t)j + 2] - o[2]};
    const double len2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    if (!(len2 > 0.0)) return 0;
    const double bound = t2 * len2;
    int m = 0;
    for (int k = 0; k < n; ++k) {
      if (m + (n - k) <= beat) return 0;
      const double x = q[3 * (size_t)k] - o[0], y = q[3 * (size_t)k + 1] - o[1], z = q[3 * (size_t)k + 2] - o[2];
      const double cx = y * d[2] - z * d[1], cy = z * d[0] - x * d[2], cz = x * d[1] - y * d[0];
      if ((cx * cx + cy * cy) + cz * cz < bound) { ++m; if (out) out->push_back(k); }
    }
    return m;
  };
  int best = 0, bi = -1, bj = -1;
  for (int i = 0; i + 1 < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const int m = count(i, j, nullptr, best);
      if (m > best) { best = m; bi = i; bj = j; }
    }
  std::vector<int> in;
  if (bi >= 0) count(bi, bj, &in, -1);
  return in;
}

struct Segments {
  std::vector<PointCloud> points;
  std::vector<Vector6d> coeffs;
};

void Fuse(Segments& S) {
  const int n = (int)S.points.size();
  if (n == 0) return;
  const double angle_threshold = 3.0 / 180.0 * M_PI;
  std::vector<std::vector<int>> ids((size_t)n);
  for (int s = 0; s < n; ++s) {
    for (const PointXYZI& p : S.points[(size_t)s]) ids[(size_t)s].push_back(int(p.intensity));
    std::sort(ids[(size_t)s].begin(), ids[(size_t)s].end());
    ids[(size_t)s].erase(std::unique(ids[(size_t)s].begin(), ids[(size_t)s].end()), ids[(size_t)s].end());
  }
  std::vector<float> centers((size_t)n * 3);
  for (int s = 0; s < n; ++s) for (int k = 0; k < 3; ++k) centers[3 * (size_t)s + k] = (float)S.coeffs[(size_t)s][(size_t)k];
  const NeighbourTable nn = BuildNeighbourTable(centers.data(), 3, n);
  std::vector<std::vector<int>> link((size_t)n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < nn.k; ++j) {
      if (nn.sqd[(size_t)i * nn.k + j] > 1) break;
      const int o = nn.idx[(size_t)i * nn.k + j];
      if (PointToLine(S.coeffs[(size_t)i].data(), S.coeffs[(size_t)o].data()) > 0.2) continue;
      if (PointToLine(S.coeffs[(size_t)o].data(), S.coeffs[(size_t)i].data()) > 0.2) continue;
      if (DirectionAngle(&S.coeffs[(size_t)i][3], &S.coeffs[(size_t)o][3]) > angle_threshold) continue;
      size_t shared = 0;
      for (size_t a = 0, b = 0; a < ids[(size_t)i].size() && b < ids[(size_t)o].size();) {
        if (ids[(size_t)i][a] < ids[(size_t)o][b]) ++a;
        else if (ids[(size_t)i][a] > ids[(size_t)o][b]) ++b;
        else { ++shared; ++a; ++b; }
      }
      if (shared <= 2) continue;
      link[(size_t)i].push_back(o);
    }
  Segments out;
  std::vector<char> fused((size_t)n, 0), seen;
  std::vector<int> todo, group;
  for (int s = 0; s < n; ++s) {
    if (fused[(size_t)s]) continue;
    group.clear();
    if (link[(size_t)s].empty()) group.push_back(s);                   // a line that is not even its own neighbour (NaN coefficients) stays alone, un-marked
    else {
      seen.assign((size_t)n, 0);
      todo.assign(1, s);
      seen[(size_t)s] = 1;
      while (!todo.empty()) {                                          // everything reachable from s along the (directed) links
        const int v = todo.back(); todo.pop_back();
        for (int w : link[(size_t)v]) if (!seen[(size_t)w]) { seen[(size_t)w] = 1; todo.push_back(w); }
      }
      for (int v = 0; v < n; ++v) if (seen[(size_t)v]) { group.push_back(v); fused[(size_t)v] = 1; }
    }
    PointCloud cloud; Vector6d coeff{};
    if (group.size() == 1) { cloud = S.points[(size_t)group[0]]; coeff = S.coeffs[(size_t)group[0]]; }
    else {
      PointCloud all;
      std::vector<int> have;
      for (int v : group)
        for (const PointXYZI& p : S.points[(size_t)v]) {
          const int id = int(p.intensity);
          if (std::find(have.begin(), have.end(), id) != have.end()) continue;
          have.push_back(id);
          all.push_back(p);
        }
      const std::vector<int> in = Consensus(all, 0.02);
      if (in.size() > 4) {
        std::vector<double> pts(in.size() * 3);
        for (size_t k = 0; k < in.size(); ++k) {
          const PointXYZI& p = all[(size_t)in[k]];
          cloud.push_back(p);
          pts[3 * k] = p.x; pts[3 * k + 1] = p.y; pts[3 * k + 2] = p.z;
        }
        FormLine3D(pts.data(), (int)in.size(), 5.0, 0.0, coeff.data());   // may refuse: the coefficients are then zero, the segment stays (as upstream)
      }
    }
    if (!cloud.empty()) { out.points.push_back(std::move(cloud)); out.coeffs.push_back(coeff); }
  }
  S = std::move(out);
}

}  // namespace

// plain-array lists of a task grown on the host through the kernel's code
struct HostLists {
  int mm[pvlm_linegrow::kMaxMembers + 1], oo[pvlm_linegrow::kMaxMembers + 1];
  int& m(int k) { return mm[k]; }
  int& o(int k) { return oo[k]; }
};

void Velodyne::EdgeToLine(const pvlm_line_grow_result* grown) {
  edge_segmented.clear(); segment_coeffs.clear(); end_points.clear(); point_to_segment.clear();
  cornerBeforeFilter = cornerLessSharp;
  const PointCloud& E = cornerBeforeFilter;
  const int n = (int)E.size();
  Segments S;
  if (grown && (grown->status != 0 || grown->n_points != n)) grown = nullptr;      // the device handed the scan back: grown here
  static const bool host_tasks = [] { const char* v = std::getenv("PVLM_EDGE_GROW"); return v && std::strcmp(v, "tasks") == 0; }();
  // the segments of every (start point, neighbour pair) task, in task order: from K27, or (PVLM_EDGE_GROW=tasks) grown here by the kernel's code
  pvlm_line_grow_result host_result{};
  std::vector<int> h_task, h_off, h_members; std::vector<double> h_coeffs;
  if (n > 0 && !grown && host_tasks) {
    namespace lg = pvlm_linegrow;
    const int k = std::min(lg::kK, n);
    std::vector<int> idx((size_t)n * lg::kK, -1); std::vector<float> sqd((size_t)n * lg::kK, 0.f);
    for (int q = 0; q < n; ++q) lg::neighbours_of(&E[0].x, 4, n, q, k, &idx[(size_t)q * lg::kK], &sqd[(size_t)q * lg::kK]);
    const lg::Cloud C{&E[0].x, 4, n, k, idx.data(), sqd.data()};
    static const lg::Turn turn = lg::turn_thresholds();
    h_off.push_back(0);
    bool ok = true;
    for (int t = 0; t < n * lg::kCombos && ok; ++t) {
      int a, b; lg::combo(t % lg::kCombos, &a, &b);
      HostLists w; int count = 0; double coeff[6];
      const int st = lg::grow_task(C, turn, t / lg::kCombos, a, b, w, &count, coeff);
      if (st == lg::kOverflow || st == lg::kUndecided) ok = false;
      if (st != lg::kSegment) continue;
      h_task.push_back(t);
      h_members.insert(h_members.end(), w.mm, w.mm + count);
      h_off.push_back((int)h_members.size());
      h_coeffs.insert(h_coeffs.end(), coeff, coeff + 6);
    }
    if (ok) {
      host_result.status = 0; host_result.n_points = n; host_result.n_segments = (int)h_task.size();
      host_result.seg_task = h_task.data(); host_result.seg_offset = h_off.data(); host_result.members = h_members.data(); host_result.coeffs = h_coeffs.data();
      grown = &host_result;
    }
  }
  if (n > 0 && grown) {
    // upstream's walk over the start points (LidarLineExtraction.cpp:300-389) on finished segments: a point an earlier segment took is skipped with all its tasks
    std::vector<char> visited((size_t)n, 0);
    int q = 0;
    for (int i = 0; i < n; ++i) {
      while (q < grown->n_segments && grown->seg_task[q] < i * pvlm_linegrow::kCombos) ++q;
      if (visited[(size_t)i]) continue;
      visited[(size_t)i] = 1;
      for (; q < grown->n_segments && grown->seg_task[q] < (i + 1) * pvlm_linegrow::kCombos; ++q) {
        PointCloud cloud;
        for (int k = grown->seg_offset[q]; k < grown->seg_offset[q + 1]; ++k) { const int id = grown->members[k]; visited[(size_t)id] = 1; cloud.push_back(E[(size_t)id]); }
        S.points.push_back(std::move(cloud));
        Vector6d c; for (int k = 0; k < 6; ++k) c[(size_t)k] = grown->coeffs[6 * (size_t)q + (size_t)k];
        S.coeffs.push_back(c);
      }
    }
  } else if (n > 0) {
    std::vector<double> P((size_t)n * 3);
    for (int i = 0; i < n; ++i) { P[3 * (size_t)i] = E[(size_t)i].x; P[3 * (size_t)i + 1] = E[(size_t)i].y; P[3 * (size_t)i + 2] = E[(size_t)i].z; }
    const NeighbourTable nn = BuildNeighbourTable(&E[0].x, 4, n);
    Grower grow(P, nn);
    std::vector<char> visited((size_t)n, 0);
    std::vector<int> members, seed(3);
    for (int i = 0; i < n; ++i) {
      if (visited[(size_t)i]) continue;
      visited[(size_t)i] = 1;
      for (int a = 1; a + 1 < nn.k; ++a)
        for (int b = a + 1; b < nn.k; ++b) {
          seed = {i, nn.idx[(size_t)i * nn.k + a], nn.idx[(size_t)i * nn.k + b]};
          if (AllZero(grow.Fit(seed, 5.0))) continue;
          members = seed;
          std::sort(members.begin(), members.end());
          members.erase(std::unique(members.begin(), members.end()), members.end());
          ++grow.epoch;
          for (int id : members) grow.stamp[(size_t)id] = grow.epoch;
          int e0, e1; double length;
          Extremes(P, seed, &e0, &e1, &length);
          int end0 = seed[(size_t)e0], end1 = seed[(size_t)e1];
          for (bool g0 = true, g1 = true; g0 || g1;) {
            g0 = grow.Expand(end0, members);
            g1 = grow.Expand(end1, members);
            Extremes(P, members, &e0, &e1, &length);
            end0 = members[(size_t)e0]; end1 = members[(size_t)e1];
          }
          if (members.size() >= 5) {
            PointCloud cloud;
            for (int id : members) { visited[(size_t)id] = 1; cloud.push_back(E[(size_t)id]); }
            S.points.push_back(std::move(cloud));
            S.coeffs.push_back(grow.Fit(members, 1.0));
          }
        }
    }
  }
  if (n > 0) {
    Fuse(S);
    // FilterLineByScan: the points of a line come from >= 3 rings and from at least half as many rings as it has points
    Segments kept;
    for (size_t s = 0; s < S.points.size(); ++s) {
      std::vector<int> rings;
      for (const PointXYZI& p : S.points[s]) rings.push_back(layout_.point_idx_to_image[(size_t)static_cast<int>(p.intensity)].first);
      std::sort(rings.begin(), rings.end());
      const size_t distinct = (size_t)(std::unique(rings.begin(), rings.end()) - rings.begin());
      if (distinct >= S.points[s].size() / 2 && distinct >= 3) { kept.points.push_back(S.points[s]); kept.coeffs.push_back(S.coeffs[s]); }
    }
    // FilterLineByLength: longer than 30 cm
    S = Segments();
    const float min_length = 0.3;
    for (size_t s = 0; s < kept.points.size(); ++s) {
      int a, b; double length;
      ExtremesOfCloud(kept.points[s], &a, &b, &length);
      if (length > min_length) { S.points.push_back(kept.points[s]); S.coeffs.push_back(kept.coeffs[s]); }
    }
  }
  edge_segmented = std::move(S.points);
  segment_coeffs = std::move(S.coeffs);
  // end points: the two farthest points of a segment projected on its line (ProjectPoint2Line3D, base/Geometry.hpp:180-191)
  for (size_t s = 0; s < edge_segmented.size(); ++s) {
    int a, b; double length;
    ExtremesOfCloud(edge_segmented[s], &a, &b, &length);
    const double* l = segment_coeffs[s].data();
    for (int e : {a, b}) {
      const PointXYZI& p = edge_segmented[s][(size_t)e];
      const double k = (l[3] * ((double)p.x - l[0]) + l[4] * ((double)p.y - l[1]) + l[5] * ((double)p.z - l[2])) / (l[3] * l[3] + l[4] * l[4] + l[5] * l[5]);
      end_points.push_back({k * l[3] + l[0], k * l[4] + l[1], k * l[5] + l[2]});
    }
  }
  // cornerLessSharp = the members of the segments, each once, in segment order; point_to_segment = every segment a point belongs to
  cornerLessSharp.clear();
  std::vector<int> slot(cloud_scan.size(), -1);
  for (size_t s = 0; s < edge_segmented.size(); ++s)
    for (const PointXYZI& p : edge_segmented[s]) {
      const int id = int(p.intensity);
      if (slot[(size_t)id] >= 0) point_to_segment[(size_t)slot[(size_t)id]].insert((int)s);
      else {
        slot[(size_t)id] = (int)cornerLessSharp.size();
        cornerLessSharp.push_back(p);
        point_to_segment.push_back(std::set<int>{(int)s});
      }
    }
  // cornerSharp keeps only the points that survived
  PointCloud sharp;
  for (const PointXYZI& p : cornerSharp) if (slot[(size_t)int(p.intensity)] >= 0) sharp.push_back(p);
  cornerSharp.swap(sharp);
  InvalidateDevice();
}

}  // namespace pvlm
