// Host-compiled check of csrc/pvlm_linegrow_core.h (the task of K27): upstream's walk over the start points (sensors/LidarLineExtraction.cpp:300-389), every segment
// grown by the kernel's own code on the CPU — what pvlm_line_grow_batch has to return for the same edge cloud, bit for bit (tests/test_linegrow_gpu.py).  The core
// itself is held against the oracle's segment-after-segment growth by tests/test_lines_cpu.py (PVLM_EDGE_GROW=tasks).
// TEST INFRASTRUCTURE ONLY.  Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../panovlm_amd/csrc/pvlm_linegrow_core.h"

namespace lg = pvlm_linegrow;
struct HostLists { int mm[lg::kMaxMembers + 1], oo[lg::kMaxMembers + 1]; int& m(int k) { return mm[k]; } int& o(int k) { return oo[k]; } };

extern "C" {

// xyz: n x stride floats.  Call with null outputs to size: sizes[0] = segments, sizes[1] = members.  Returns the scan's status (0, 2 = overflow, 3 = undecided).
int chk_line_walk(const float* xyz, int n, int stride, long long* sizes, int* seg_task, int* seg_offset, int* members, double* coeffs) {
  sizes[0] = sizes[1] = 0;
  if (n <= 0) return 0;
  const int k = std::min(lg::kK, n);
  std::vector<int> idx((size_t)n * lg::kK, -1); std::vector<float> sqd((size_t)n * lg::kK, 0.f);
  for (int q = 0; q < n; ++q) lg::neighbours_of(xyz, stride, n, q, k, &idx[(size_t)q * lg::kK], &sqd[(size_t)q * lg::kK]);
  const lg::Cloud C{xyz, stride, n, k, idx.data(), sqd.data()};
  const lg::Turn turn = lg::turn_thresholds();
  std::vector<char> visited((size_t)n, 0);
  long long n_seg = 0, n_mem = 0;
  if (seg_offset) seg_offset[0] = 0;
  for (int i = 0; i < n; ++i) {
    if (visited[(size_t)i]) continue;
    visited[(size_t)i] = 1;
    HostLists w[lg::kCombos]; int count[lg::kCombos], st[lg::kCombos]; double coeff[lg::kCombos][6];
    for (int c = 0; c < lg::kCombos; ++c) {
      int a, b; lg::combo(c, &a, &b);
      st[c] = lg::grow_task(C, turn, i, a, b, w[c], &count[c], coeff[c]);
      if (st[c] == lg::kOverflow || st[c] == lg::kUndecided) return st[c];
    }
    for (int c = 0; c < lg::kCombos; ++c) {
      if (st[c] != lg::kSegment) continue;
      for (int q = 0; q < count[c]; ++q) visited[(size_t)w[c].mm[q]] = 1;
      if (seg_task) {
        seg_task[n_seg] = i * lg::kCombos + c;
        std::memcpy(members + n_mem, w[c].mm, (size_t)count[c] * sizeof(int));
        seg_offset[n_seg + 1] = (int)(n_mem + count[c]);
        std::memcpy(coeffs + 6 * n_seg, coeff[c], 6 * sizeof(double));
      }
      ++n_seg; n_mem += count[c];
    }
  }
  sizes[0] = n_seg; sizes[1] = n_mem;
  return 0;
}

}  // extern "C"
