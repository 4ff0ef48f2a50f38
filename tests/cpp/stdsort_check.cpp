// Host-compiled check of csrc/pvlm_stdsort.h: the restated introsort against the toolchain's own std::sort, element for element, on inputs
// where the order of equal keys is the whole question.  Driven by tests/test_stdsort_cpu.py through ctypes.
#define PVLM_HD inline
#define PVLM_STDSORT_STATS
#include "../../panovlm_amd/csrc/pvlm_stdsort.h"

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

extern "C" {

// indices 0..n-1 sorted by float keys (the sector sort of the picks); returns 0 when both orders agree, 1 + first differing position otherwise
int chk_sort_by_float(const float* key, int n, int* order_out) {
  std::vector<int> a((size_t)n), b((size_t)n);
  std::iota(a.begin(), a.end(), 0); b = a;
  std::sort(a.begin(), a.end(), [key](int x, int y) { return key[x] < key[y]; });
  const bool sane = pvlm_stdsort::sort(b.data(), n, [key](int x, int y) { return key[x] < key[y]; });
  if (order_out) std::copy(b.begin(), b.end(), order_out);
  if (!sane) return -1;
  for (int i = 0; i < n; ++i) if (a[(size_t)i] != b[(size_t)i]) return 1 + i;
  return 0;
}

// (cell, point) pairs sorted by cell only (pcl::VoxelGrid's std::sort of cloud_point_index_idx)
struct Pair { unsigned cell, point; };
int chk_sort_pairs(const unsigned* cell, int n) {
  std::vector<Pair> a((size_t)n), b;
  for (int i = 0; i < n; ++i) a[(size_t)i] = Pair{cell[i], (unsigned)i};
  b = a;
  std::sort(a.begin(), a.end(), [](const Pair& x, const Pair& y) { return x.cell < y.cell; });
  const bool sane = pvlm_stdsort::sort(b.data(), n, [](const Pair& x, const Pair& y) { return x.cell < y.cell; });
  if (!sane) return -1;
  for (int i = 0; i < n; ++i) if (a[(size_t)i].point != b[(size_t)i].point) return 1 + i;
  return 0;
}

// the level-by-level form the device runs (lanes = a plain loop here), float keys as above; also -2 when it disagrees with the serial restatement
int chk_sort_by_levels(const float* key, int n) {
  if (n >= 8192) return 0;
  std::vector<int> a((size_t)n), b((size_t)n);
  std::iota(a.begin(), a.end(), 0); b = a;
  std::sort(a.begin(), a.end(), [key](int x, int y) { return key[x] < key[y]; });
  std::vector<unsigned> queue(2 * pvlm_stdsort::kWaveQueue), cuts((size_t)(n + 31) / 32 + 1);
  int ctr[3];
  auto lanes = [](int count, auto&& body) { for (int r = 0; r < count; ++r) body(r); };
  const bool sane = pvlm_stdsort::sort_by_levels(b.data(), n, [key](int x, int y) { return key[x] < key[y]; }, queue.data(), cuts.data(), ctr, lanes);
  if (!sane) return -1;
  for (int i = 0; i < n; ++i) if (a[(size_t)i] != b[(size_t)i]) return 1 + i;
  return 0;
}

long long chk_heap_sorted_ranges() { return pvlm_stdsort::heap_sorted_ranges; }

// M. D. McIlroy, "A killer adversary for quicksort" (1999): a comparator that fixes the keys while std::sort runs so that every pivot is among the
// smallest of its range — the concrete keys it leaves drive this library's quicksort to its depth limit.  out: n keys (a permutation of 0..n-1).
void chk_killer_keys(int n, int* out) {
  std::vector<int> val((size_t)n, n), idx((size_t)n);          // n = "gas"
  std::iota(idx.begin(), idx.end(), 0);
  int solid = 0, candidate = 0;
  auto freeze = [&](int x) { val[(size_t)x] = solid++; };
  std::sort(idx.begin(), idx.end(), [&](int x, int y) {
    if (val[(size_t)x] == n && val[(size_t)y] == n) freeze(x == candidate ? x : y);
    if (val[(size_t)x] == n) candidate = x; else if (val[(size_t)y] == n) candidate = y;
    return val[(size_t)x] < val[(size_t)y];
  });
  for (int i = 0; i < n; ++i) out[i] = val[(size_t)i] == n ? solid++ : val[(size_t)i];
}

}  // extern "C"
