// Growth of ONE line segment of the LiDAR line extraction:
//   ExtractLineFeatures   sensors/LidarLineExtraction.cpp:296-389   seed = an edge point + two of its four nearest neighbours (FormLine, tolerance 5), grown at both
//                                                                    ends while the enlarged set stays a line
//   ExpandLine            sensors/LidarLineExtraction.cpp:10-70
//   FormLine              base/Geometry.hpp:220-260                  scatter matrix + eigen decomposition (cyclic Jacobi here and in the test oracle: DESIGN.md §4)
//
// Upstream walks the edge points in index order and skips a point that an earlier segment has taken (`visited`); the growth of the segment of (point i, neighbours
// a, b) itself never looks at `visited` — it is a pure function of the edge cloud.  So every (i, a, b) is an independent TASK: K27 (csrc/pvlm_linegrow.hip) runs one
// task per lane for all scans of a batch, and the host replays upstream's walk over the finished tasks (host/pvlm_lines.cpp).  The arithmetic is the host mirror's,
// statement for statement (host/pvlm_lines.cpp: Grower; host/pvlm_host_lidar_associate.cpp: FormLine), in double, without contraction; the one libm call of the
// growth — acos in the 1-degree turn test — is replaced by a comparison of the cosine against thresholds the HOST derives from its own acos (turn_thresholds below),
// so that the device takes the decision the host's libm would take; a cosine inside an undecided band (none with glibc) sends the scan back to the host.
// Host/device: tests/test_lines_cpu.py drives the same functions through the host mirror (PVLM_EDGE_GROW=tasks) against the oracle without a GPU.
#pragma once
#include <cmath>

#ifndef PVLM_HD
#if defined(__HIPCC__)
#define PVLM_HD __host__ __device__
#else
#define PVLM_HD
#endif
#endif

namespace pvlm_linegrow {

constexpr int kK = 5;                  // neighbours kept per edge point (the point itself included): pcl nearestKSearch(5) of upstream
constexpr int kCombos = 6;             // (a, b), 1 <= a < b <= 4, in upstream's loop order
constexpr int kMaxMembers = 64;        // a segment that would exceed it is not grown here: status kOverflow, the scan goes to the host path

enum Status { kNone = 0, kSegment = 1, kOverflow = 2, kUndecided = 3 };

struct Turn { double sure_true, sure_false; };     // turn > 1 degree  <=>  cosine <= sure_true;   not  <=>  cosine > sure_false;   between: undecided

// edge cloud of one scan: xyz with a stride (pcl::PointXYZI records: 4), its neighbour table (k = min(5, n) entries per point, ascending (d2, index))
struct Cloud {
  const float* xyz; int stride; int n; int k;
  const int* nn_idx; const float* nn_sqd;       // n x kK, the first k of every row are valid
  PVLM_HD double x(int id, int c) const { return (double)xyz[(size_t)id * (size_t)stride + (size_t)c]; }
};

PVLM_HD inline void combo(int c, int* a, int* b) {
  const int A[kCombos] = {1, 1, 1, 2, 2, 3}, B[kCombos] = {2, 3, 4, 3, 4, 4};
  *a = A[c]; *b = B[c];
}

// the k smallest (d2, index) of point q among all points of the cloud, index order among equals (BuildNeighbourTable, host/pvlm_lines.cpp; flann::L2_Simple sums)
PVLM_HD inline void neighbours_of(const float* xyz, int stride, int n, int q, int k, int* out_idx, float* out_sqd) {
  const float qx = xyz[(size_t)q * stride], qy = xyz[(size_t)q * stride + 1], qz = xyz[(size_t)q * stride + 2];
  float bd[kK]; int bi[kK]; int have = 0;
  for (int j = 0; j < kK; ++j) { bd[j] = 0.f; bi[j] = -1; }
  for (int i = 0; i < n; ++i) {
    const float dx = qx - xyz[(size_t)i * stride], dy = qy - xyz[(size_t)i * stride + 1], dz = qz - xyz[(size_t)i * stride + 2];
    float s = 0.0f;
    s += dx * dx; s += dy * dy; s += dz * dz;
    if (have == k && !(s < bd[have - 1])) continue;
    int at = have < k ? have++ : have - 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = kK - 1; j > 0; --j) if (j == at && s < bd[j - 1]) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --at; }     // the while loop of the host, on registers
    bd[at] = s; bi[at] = i;
  }
  for (int j = 0; j < kK; ++j) { out_idx[j] = bi[j]; out_sqd[j] = bd[j]; }
}

PVLM_HD inline double gap2(const Cloud& C, int a, int b) {
  const double x = C.x(a, 0) - C.x(b, 0), y = C.x(a, 1) - C.x(b, 1), z = C.x(a, 2) - C.x(b, 2);
  return x * x + (y * y + z * z);
}

// FormLine on the points ids(0 .. m-1) in that order.  Returns false (line = 0) when the set is not a line.
template <class Ids>
PVLM_HD inline bool form_line(const Cloud& C, const Ids& ids, int m, double tolerance, double dis_threshold, double* line) {
  double c[3] = {0, 0, 0};
  for (int i = 0; i < m; ++i) { const int id = ids(i); for (int k = 0; k < 3; ++k) c[k] = c[k] + C.x(id, k); }
  for (int k = 0; k < 3; ++k) c[k] = c[k] / double(m);
  double A00 = 0, A01 = 0, A02 = 0, A10 = 0, A11 = 0, A12 = 0, A20 = 0, A21 = 0, A22 = 0;
  for (int i = 0; i < m; ++i) {
    const int id = ids(i);
    const double d0 = C.x(id, 0) - c[0], d1 = C.x(id, 1) - c[1], d2 = C.x(id, 2) - c[2];
    A00 = A00 + d0 * d0; A01 = A01 + d0 * d1; A02 = A02 + d0 * d2;
    A10 = A10 + d1 * d0; A11 = A11 + d1 * d1; A12 = A12 + d1 * d2;
    A20 = A20 + d2 * d0; A21 = A21 + d2 * d1; A22 = A22 + d2 * d2;
  }
  double A[3][3] = {{A00, A01, A02}, {A10, A11, A12}, {A20, A21, A22}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    if (A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2] == 0.0) break;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;            // (0,1), (0,2), (1,2)
      const double apq = A[p][q];
      if (apq == 0.0) continue;
      const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
      const double at = fabs(theta);
      double t, cs, sn;
      if (at > 134217728.0 && at < 1.0e150) { t = (theta >= 0.0 ? 1.0 : -1.0) / (at + at); cs = 1.0; sn = t; }
      else { t = (theta >= 0.0 ? 1.0 : -1.0) / (at + sqrt(theta * theta + 1.0)); cs = 1.0 / sqrt(t * t + 1.0); sn = t * cs; }
      const int r = 3 - p - q;
      A[p][p] -= t * apq; A[q][q] += t * apq; A[p][q] = A[q][p] = 0.0;
      const double arp = A[r][p], arq = A[r][q];
      A[r][p] = A[p][r] = cs * arp - sn * arq;
      A[r][q] = A[q][r] = sn * arp + cs * arq;
      for (int k = 0; k < 3; ++k) { const double vp = V[k][p], vq = V[k][q]; V[k][p] = cs * vp - sn * vq; V[k][q] = sn * vp + cs * vq; }
    }
  }
  // std::sort of {0, 1, 2} by A[a][a] < A[b][b]: libstdc++'s insertion sort of three elements, comparison for comparison
  int o0 = 0, o1 = 1, o2 = 2;
  const double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
  auto ev = [&](int k) { return k == 0 ? e0 : (k == 1 ? e1 : e2); };
  if (ev(o1) < ev(o0)) { const int t = o0; o0 = o1; o1 = t; }
  if (ev(o2) < ev(o0)) { const int t = o2; o2 = o1; o1 = o0; o0 = t; }
  else if (ev(o2) < ev(o1)) { const int t = o2; o2 = o1; o1 = t; }
  for (int k = 0; k < 6; ++k) line[k] = 0.0;
  if (!(ev(o2) > tolerance * ev(o1))) return false;
  double dir[3];
  for (int k = 0; k < 3; ++k) dir[k] = o2 == 0 ? V[k][0] : (o2 == 1 ? V[k][1] : V[k][2]);
  const double len = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  if (len * len > 0.0) for (int k = 0; k < 3; ++k) dir[k] /= len;
  const double l[6] = {c[0], c[1], c[2], dir[0], dir[1], dir[2]};
  if (dis_threshold > 0.0)
    for (int i = 0; i < m; ++i) {
      const int id = ids(i);
      const double p0 = C.x(id, 0), p1 = C.x(id, 1), p2 = C.x(id, 2);
      const double k = (l[3] * (p0 - l[0]) + l[4] * (p1 - l[1]) + l[5] * (p2 - l[2])) / (l[3] * l[3] + l[4] * l[4] + l[5] * l[5]);      // PointToLineDistance3D
      const double q0 = k * l[3] + l[0], q1 = k * l[4] + l[1], q2 = k * l[5] + l[2];
      if (sqrt((q0 - p0) * (q0 - p0) + (q1 - p1) * (q1 - p1) + (q2 - p2) * (q2 - p2)) > dis_threshold) return false;
    }
  for (int k = 0; k < 6; ++k) line[k] = l[k];
  return true;
}

PVLM_HD inline bool all_zero(const double* l) { for (int k = 0; k < 6; ++k) if (l[k] != 0.0) return false; return true; }

// FurthestPoints over ids(0 .. m-1): positions (a, b) of the first pair at the largest distance, and that distance (Extremes, host/pvlm_lines.cpp)
template <class Ids>
PVLM_HD inline void extremes(const Cloud& C, const Ids& ids, int m, int* a, int* b, double* length) {
  *a = *b = -1; *length = -1;
  double longest2 = -1;
  for (int i = 0; i + 1 < m; ++i) {
    const int ii = ids(i);
    const double xi = C.x(ii, 0), yi = C.x(ii, 1), zi = C.x(ii, 2);
    for (int j = i + 1; j < m; ++j) {
      const int jj = ids(j);
      const double x = xi - C.x(jj, 0), y = yi - C.x(jj, 1), z = zi - C.x(jj, 2);
      const double g2 = x * x + (y * y + z * z);
      if (!(g2 > longest2) && g2 == g2) continue;
      const double g = sqrt(g2);
      if (g > *length) { *a = i; *b = j; *length = g; longest2 = g2; }
    }
  }
}

// Workspace of a task: two id lists, `members` (ascending) and `order` (the members as they were when an expansion began, then the accepted candidates as they
// came), each of capacity kMaxMembers + 1.  W gives m(k) / o(k) as references (the device strides them over the lanes of the launch, the host uses plain arrays).
template <class W>
struct MembersView { W* w; PVLM_HD int operator()(int k) const { return w->m(k); } };
template <class W>
struct OrderView { W* w; PVLM_HD int operator()(int k) const { return w->o(k); } };
struct Seed3 { int v[3]; PVLM_HD int operator()(int k) const { return v[k]; } };

// ExpandLine at end point `start`.  n_members is updated; returns 1 grown, 0 not, < 0: -kOverflow / -kUndecided.
// members_length: the extremes' distance of the CURRENT member list when the caller has it (>= -1 computed, < -1: not known) — ExpandLine's first statement takes the
// extremes of a copy of the members, a pure function of that list: taken over instead of recomputed (O(m^2) per call) whenever the list has not changed since.
template <class W>
PVLM_HD inline int expand(const Cloud& C, const Turn& turn, int start, W& w, int* n_members, double members_length) {
  int grown = 0;
  int nm = *n_members, no = nm;
  for (int k = 0; k < nm; ++k) w.o(k) = w.m(k);
  int ea, eb; double length = members_length;
  if (members_length < -1.5) extremes(C, OrderView<W>{&w}, no, &ea, &eb, &length);
  double line[6] = {0, 0, 0, 0, 0, 0};
  bool line_known = false;
  for (int j = 1; j < C.k; ++j) {
    const int cand = C.nn_idx[(size_t)start * kK + j];
    bool member = false;
    for (int k = 0; k < nm; ++k) member = member || (w.m(k) == cand);
    if (member) continue;
    if ((double)C.nn_sqd[(size_t)start * kK + j] > (length / 2) * (length / 2)) break;
    if (no >= kMaxMembers) return -(int)kOverflow;
    w.o(no++) = cand;
    double reach2 = -1;
    {
      const double cx = C.x(cand, 0), cy = C.x(cand, 1), cz = C.x(cand, 2);
      for (int k = 0; k < no; ++k) {
        const int id = w.o(k);
        const double x = C.x(id, 0) - cx, y = C.x(id, 1) - cy, z = C.x(id, 2) - cz;
        const double g2 = x * x + (y * y + z * z);
        reach2 = reach2 < g2 ? g2 : reach2;                             // std::max(reach2, g2)
      }
    }
    double reach = reach2 < 0 ? -1 : sqrt(reach2);
    reach = reach < length ? length : reach;                            // std::max(reach, length)
    double next[6];
    if (reach < 2) {
      form_line(C, OrderView<W>{&w}, no, 5.0, 0.07, next);
      if (all_zero(next)) { --no; continue; }
    } else {
      if (!line_known) { form_line(C, OrderView<W>{&w}, no - 1, 3.0, 0.0, line); line_known = true; }
      form_line(C, OrderView<W>{&w}, no, 20.0, 0.0, next);
      // DirectionAngle(next, line) * 180 / pi > 1, decided on the cosine (see the header)
      double c = fabs(next[3] * line[3] + next[4] * line[4] + next[5] * line[5]);
      c = c / (sqrt(next[3] * next[3] + next[4] * next[4] + next[5] * next[5]) * sqrt(line[3] * line[3] + line[4] * line[4] + line[5] * line[5]));
      bool turned;
      if (c != c || c >= 1.0) turned = false;                           // NaN: acos(NaN) > 1 is false;  c >= 1: angle 0
      else if (c <= turn.sure_true) turned = true;
      else if (c > turn.sure_false) turned = false;
      else return -(int)kUndecided;
      if (all_zero(next) || turned) { --no; continue; }
    }
    grown = 1;
    // members.insert(upper_bound(cand))
    int at = nm;
    while (at > 0 && w.m(at - 1) > cand) { w.m(at) = w.m(at - 1); --at; }
    w.m(at) = cand; ++nm;
    length = reach;
    for (int k = 0; k < 6; ++k) line[k] = next[k];
    line_known = true;
  }
  *n_members = nm;
  return grown;
}

// One task: the segment of (edge point i, its neighbours a and b).  Returns kNone (seed refused, or fewer than 5 members), kSegment (members in w.m(0 .. *count-1),
// ascending; coeff = FormLine(members, 1.0), zeros when that fit refuses — kept, as upstream keeps it), kOverflow, kUndecided.
template <class W>
PVLM_HD inline int grow_task(const Cloud& C, const Turn& turn, int i, int a, int b, W& w, int* count, double* coeff) {
  *count = 0;
  if (a >= C.k || b >= C.k) return kNone;
  Seed3 seed{{i, C.nn_idx[(size_t)i * kK + a], C.nn_idx[(size_t)i * kK + b]}};
  double l[6];
  form_line(C, seed, 3, 5.0, 0.0, l);
  if (all_zero(l)) return kNone;
  // members = sorted unique seed
  int s0 = seed.v[0], s1 = seed.v[1], s2 = seed.v[2];
  if (s1 < s0) { const int t = s0; s0 = s1; s1 = t; }
  if (s2 < s1) { const int t = s1; s1 = s2; s2 = t; }
  if (s1 < s0) { const int t = s0; s0 = s1; s1 = t; }
  int nm = 0;
  w.m(nm++) = s0;
  if (s1 != s0) w.m(nm++) = s1;
  if (s2 != s1) w.m(nm++) = s2;
  int e0, e1; double length;
  extremes(C, seed, 3, &e0, &e1, &length);
  if (e0 < 0) return kUndecided;                                        // only with non-finite coordinates: left to the host
  int end0 = seed.v[e0], end1 = seed.v[e1];
  double known = -2.0;                                                  // extremes' distance of the member list as it stands (-2: not computed for this list)
  for (bool g0 = true, g1 = true; g0 || g1;) {
    int r = expand(C, turn, end0, w, &nm, known);
    if (r < 0) return -r;
    g0 = r != 0;
    if (g0) known = -2.0;
    r = expand(C, turn, end1, w, &nm, known);
    if (r < 0) return -r;
    g1 = r != 0;
    if (g1) known = -2.0;
    if (known < -1.5) {                                                 // the list changed since its extremes were taken (always so in the first round)
      extremes(C, MembersView<W>{&w}, nm, &e0, &e1, &length);
      if (e0 < 0) return kUndecided;
      end0 = w.m(e0); end1 = w.m(e1);
      known = length;
    }
  }
  if (nm < 5) return kNone;
  form_line(C, MembersView<W>{&w}, nm, 1.0, 0.0, coeff);
  *count = nm;
  return kSegment;
}

// (host only) The thresholds of the 1-degree turn test from THIS process's acos: the largest cosine whose angle still exceeds 1 degree (bisection over the doubles, the host's
// expression acos(c) * 180.0 / M_PI > 1), checked for monotonicity over 4096 neighbouring doubles on either side; where the libm is not monotone there the band
// between the first `false` and the last `true` is reported as undecided.
inline Turn turn_thresholds() {
  auto turned = [](double c) { volatile double a = std::acos(c); return a * 180.0 / M_PI > 1; };
  double lo = 0.9, hi = 1.0;                                             // turned(lo), !turned(hi)
  for (int it = 0; it < 200; ++it) {
    const double mid = lo + (hi - lo) / 2;
    if (mid <= lo || mid >= hi) break;
    if (turned(mid)) lo = mid; else hi = mid;
  }
  double last_true = lo, first_false = hi, c = lo;
  for (int k = 0; k < 4096; ++k) c = std::nextafter(c, 0.0);
  for (int k = 0; k < 8192; ++k, c = std::nextafter(c, 2.0)) {
    if (turned(c)) { if (c > last_true) last_true = c; }
    else if (c < first_false) first_false = c;
  }
  Turn t;
  if (first_false > last_true) { t.sure_true = last_true; t.sure_false = last_true; }
  else { t.sure_true = std::nextafter(first_false, 0.0); t.sure_false = last_true; }
  return t;
}

}  // namespace pvlm_linegrow
