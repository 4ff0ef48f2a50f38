// Host-compiled check of the per-element device bodies in panovlm_amd/csrc/pvlm_ring_core.h (K16 ring / azimuth certificates,
// K17 column state machine, K19 certified "joined" test, K22 curvature): the same functions the kernels of csrc/pvlm_ring.hip
// call, driven serially through the same stages as pvlm_ring_extract_batch (listed points decided by this host's libm, scans
// replayed with exact azimuths when the +z crossing stays undecided, serial scatter / union-find / compaction in place of the
// atomics), so that every array can be compared with oracle/features.hpp on a machine without a GPU
// (tests/test_ring_core_cpu.py).  TEST INFRASTRUCTURE ONLY — libpvlm.so has no host path.
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#define PVLM_HD inline
#include "../../panovlm_amd/csrc/pvlm_ring_core.h"

using namespace pvlm_ring;

namespace {
// the workgroup of columns_block, serial: a phase is a loop over the threads
struct SerialExec {
  int T;
  int threads() const { return T; }
  template <class F> void phase(F&& f) { for (int t = 0; t < T; ++t) f(t); }
};
}  // namespace

extern "C" {

// xyzi: n x 4.  ulps_override > 0 replaces kUlps' role of "how far the libm may be off" only through force_list (see below).
// stats: [0] listed points, [1] undecided edges, [2] replayed, [3] n_reordered, [4] n_kept.
// force: bit 0 = list every point, bit 1 = every azimuth exact before the state machine (all decisions from the host libm).
// threads > 0: the workgroup form of the column machine (columns_block) with that many threads; 0: the one-lane loop (columns_scan)
int chk_ring(const float* xyzi, int n, int rings, int horizon, int segment, int force, int threads,
             float* cloud_reordered, int* rc_reordered, float* range_image, int* img2pt_reordered, int* ring_count,
             float* cloud_kept, int* rc_kept, int* img2pt_kept, int* ring_count2, float* curvature, int* half_window, float* range, long long* stats) {
  const int cells = rings * horizon;
  for (int k = 0; k < 5; ++k) stats[k] = 0;
  std::fill(range_image, range_image + cells, 0.f);
  std::fill(img2pt_reordered, img2pt_reordered + cells, -1);
  std::fill(img2pt_kept, img2pt_kept + cells, -1);
  std::fill(ring_count, ring_count + kMaxRings, 0);
  std::fill(ring_count2, ring_count2 + kMaxRings, 0);
  if (n <= 0) return 0;
  RingScan sc{0, 0, n, 0, 0, 0.0};
  sc.start_ori = ori_of_atan2(std::atan2(xyzi[0], xyzi[2]));
  // natural order (rec) for the one-lane loop; chunk-transposed (rec_t, colpos_t) for the workgroup form — both kept in step
  const int T = threads > 0 ? threads : 1, L = chunk_of(n, T);
  std::vector<PointRec> rec(n), rec_t((size_t)chunk_slots(n, T));
  std::vector<signed char> ring(n);
  auto put = [&](int i, float az, int r, bool exact) { rec[i] = make_rec(az, r, exact); rec_t[(size_t)chunk_slot(i, L, T)] = rec[i]; ring[i] = (signed char)r; };
  auto make_exact = [&](int i) {
    const float* p = xyzi + 4 * (size_t)i;
    const float q = -p[1] / std::sqrt(p[0] * p[0] + p[2] * p[2]);
    put(i, std::atan2(p[0], p[2]), q == q ? ring_of_atan(std::atan(q), rings) : -1, true);
  };
  // K16
  for (int i = 0; i < n; ++i) {
    float f; int r;
    const bool list = classify_point(sc, rings, horizon, xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], &f, &r);
    put(i, f, r, false);
    if (list || (force & 1)) { make_exact(i); ++stats[0]; }
  }
  // K17 (+ replays: the points of an undecided crossing test from the host libm, as pvlm_ring_extract_batch does)
  std::vector<int> colpos(2 * (size_t)n);
  std::vector<int> cnt(kMaxRings, 0);
  if (force & 2) { for (int i = 0; i < n; ++i) make_exact(i); stats[2] = 1; }
  for (int round = 0;; ++round) {
    std::fill(cnt.begin(), cnt.end(), 0);
    int last = -1;
    int stuck;
    if (threads > 0) {
      SerialExec ex{threads};
      std::vector<int> scratch(kColumnsScratch * (size_t)threads);
      std::vector<int> colpos_t(2 * (size_t)chunk_slots(n, T));
      stuck = columns_block(ex, sc, rings, horizon, rec_t.data(), colpos_t.data(), cnt.data(), &last, scratch.data());
      if (stuck < 0) for (int i = 0; i < n; ++i) { const size_t a = (size_t)chunk_slot(i, L, T); colpos[2 * i] = colpos_t[2 * a]; colpos[2 * i + 1] = colpos_t[2 * a + 1]; }
    } else {
      stuck = columns_scan(sc, rings, horizon, rec.data(), colpos.data(), [&](int r) -> int& { return cnt[r]; }, &last);
    }
    if (stuck < 0) break;
    if (round > 64) return -2;
    ++stats[2];
    if (last >= 0) make_exact(last);
    for (int i = stuck; i < std::min(n, stuck + rings + 1); ++i) make_exact(i);
  }
  for (int r = 0; r < rings; ++r) ring_count[r] = cnt[r];
  // K18
  std::vector<int> begin(kMaxRings + 1, 0);
  for (int r = 0; r < kMaxRings; ++r) begin[r + 1] = begin[r] + ring_count[r];
  const int n_re = begin[kMaxRings];
  std::vector<int> winner(cells, -1), source(n_re);
  for (int i = 0; i < n; ++i) {
    if (colpos[2 * i] < 0) continue;
    const int r = ring[i], dst = begin[r] + colpos[2 * i + 1];
    cloud_reordered[4 * dst] = xyzi[4 * i]; cloud_reordered[4 * dst + 1] = xyzi[4 * i + 1]; cloud_reordered[4 * dst + 2] = xyzi[4 * i + 2]; cloud_reordered[4 * dst + 3] = (float)r;
    rc_reordered[2 * dst] = r; rc_reordered[2 * dst + 1] = colpos[2 * i];
    source[dst] = i;
    winner[r * horizon + colpos[2 * i]] = std::max(winner[r * horizon + colpos[2 * i]], i);
  }
  for (int i = 0; i < n; ++i) {
    if (colpos[2 * i] < 0) continue;
    const int cell = ring[i] * horizon + colpos[2 * i];
    if (winner[cell] != i) continue;
    const float x = xyzi[4 * i], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2];
    range_image[cell] = std::sqrt(x * x + y * y + z * z);
    img2pt_reordered[cell] = begin[ring[i]] + colpos[2 * i + 1];
  }
  stats[3] = n_re;
  // K19 / K20
  std::vector<int> parent(cells), size(cells, 0);
  std::vector<uint64_t> row_mask(cells, 0);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int a) { while (parent[a] != a) a = parent[a]; return a; };
  if (segment) {
    const float alpha_x = 0.2 / 180.0 * M_PI, alpha_y = 2.0 / 180.0 * M_PI, theta = 20.0 / 180.0 * M_PI;
    const float sin_x = std::sin(alpha_x), cos_x = std::cos(alpha_x), sin_y = std::sin(alpha_y), cos_y = std::cos(alpha_y);
    auto joined = [&](int a, int b, float s, float c) {
      float y = 0, x = 0;
      int j = joined_certified(range_image[a], range_image[b], s, c, theta, &y, &x);
      if (j < 0) { ++stats[1]; j = std::atan2(y, x) > theta ? 1 : 0; }
      return j > 0;
    };
    for (int cell = 0; cell < cells; ++cell) {
      const int r = cell / horizon, c = cell - r * horizon;
      const int right = r * horizon + (c + 1 == horizon ? 0 : c + 1);
      auto unite = [&](int a, int b) { a = find(a); b = find(b); if (a != b) parent[std::max(a, b)] = std::min(a, b); };
      if (right != cell && joined(cell, right, sin_x, cos_x)) unite(cell, right);
      if (r + 1 < rings && joined(cell, cell + horizon, sin_y, cos_y)) unite(cell, cell + horizon);
    }
    for (int cell = 0; cell < cells; ++cell) {
      const int root = find(cell);
      ++size[root];
      if (cell != root) row_mask[root] |= 1ull << (cell / horizon);
    }
  }
  // K21
  int n_kept = 0;
  for (int i = 0; i < n_re; ++i) {
    const int cell = rc_reordered[2 * i] * horizon + rc_reordered[2 * i + 1];
    if (segment) { const int root = find(cell); if (!keep_component(size[root], __builtin_popcountll(row_mask[root]))) continue; }
    for (int k = 0; k < 4; ++k) cloud_kept[4 * n_kept + k] = cloud_reordered[4 * i + k];
    rc_kept[2 * n_kept] = rc_reordered[2 * i]; rc_kept[2 * n_kept + 1] = rc_reordered[2 * i + 1];
    range[n_kept] = range_image[cell];
    img2pt_kept[cell] = n_kept;
    ++ring_count2[rc_reordered[2 * i]];
    ++n_kept;
  }
  stats[4] = n_kept;
  // K22
  std::vector<int> begin2(kMaxRings + 1, 0);
  for (int r = 0; r < kMaxRings; ++r) begin2[r + 1] = begin2[r] + ring_count2[r];
  const Point* P = reinterpret_cast<const Point*>(cloud_kept);
  for (int i = 0; i < n_kept; ++i) {
    const int r = (int)P[i].w;
    curvature_point(P, range, n_kept, begin2[r] + 5, begin2[r + 1] - 6, i, curvature + i, half_window + i);
  }
  return 0;
}

}  // extern "C"
