// The permutation std::sort leaves, computed where std::sort is not available (device code).
//
// The reference sorts with keys that tie: each sector of a ring by curvature (sensors/Velodyne.cpp:896, :1110), the (voxel, point) pairs of
// pcl::VoxelGrid by voxel.  std::sort is not stable, so the order of equal keys — and with it which of two equally flat points is picked, and
// the order a voxel's float sum is taken in — is whatever the library's algorithm leaves.  A reference build uses libstdc++; its std::sort is
// the classical introsort and is restated here from its published description (D. Musser, "Introspective sorting and selection algorithms",
// 1997, as implemented in the SGI STL and kept by libstdc++ since):
//   introsort loop   while a range is longer than 16: median of (first + 1, middle, last - 1) moved to `first`, unguarded Hoare partition of
//                    (first + 1, last) around it, right part first (recursion), left part next (iteration); after 2 floor(log2 n) levels the
//                    range is heap-sorted instead (make_heap + sort_heap with the sift-to-leaf-then-push-up adjust_heap)
//   final pass       insertion sort of the first 16 elements, unguarded insertion of the rest
// This is test-pinned, not trusted: tests/cpp/stdsort_check.cpp runs it against the toolchain's own std::sort on tie-heavy, structured and
// depth-limit-forcing inputs (tests/test_stdsort_cpu.py), and stdsort_selfcheck() (pvlm_ring.hip) repeats a short version of that
// comparison against the std::sort the library itself was built with before the device is allowed to order a sector with ties.
// Three forms of the same permutation: sort() — the algorithm as written, serial; sort_by_levels() — its partition steps regrouped by recursion
// level, still host-checkable; sort_wave() — the device form (one wave), whose results the GPU tests compare with the real std::sort's
// (tests/test_ring_gpu.py: every sector order, every voxel sum; test_device_sort_equals_std_sort: random tie-heavy keys).
//
// `less(x, y)` compares two ELEMENTS (values of the array), like the comparator handed to std::sort.  The ranges a correct strict weak order
// keeps the unguarded loops in are not assumed: every loop is bounded by the range and the function returns false if a bound stopped it
// (NaN keys can do that); the caller then falls back to the real std::sort on the host.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

#ifndef PVLM_HD
#define PVLM_HD __host__ __device__ inline
#endif

namespace pvlm_stdsort {

template <class T, class Less>
PVLM_HD void adjust_heap(T* a, int hole, int len, T value, Less less) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(a[child], a[child - 1])) --child;
    a[hole] = a[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[hole] = a[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;                       // push_heap
  while (hole > top && less(a[parent], value)) {
    a[hole] = a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[hole] = value;
}

#ifdef PVLM_STDSORT_STATS
static long long heap_sorted_ranges = 0;    // host-compiled checks only: how often the depth limit was reached
#endif
template <class T, class Less>
PVLM_HD void heap_sort(T* a, int n, Less less) {   // partial_sort(first, last, last): make_heap, then sort_heap
#ifdef PVLM_STDSORT_STATS
  ++heap_sorted_ranges;
#endif
  if (n >= 2) {
    for (int parent = (n - 2) / 2;; --parent) {
      const T v = a[parent];
      adjust_heap(a, parent, n, v, less);
      if (parent == 0) break;
    }
  }
  for (int last = n; last > 1;) {
    --last;
    const T v = a[last];
    a[last] = a[0];
    adjust_heap(a, 0, last, v, less);
  }
}

// one step of the introsort loop on [first, last), last - first > 16: the median of (first + 1, middle, last - 1) goes to `first`, then the unguarded
// partition of [first + 1, last) around it.  Returns the cut, or -1 when a loop bound stopped a scan (inconsistent comparator).
template <class T, class Less>
PVLM_HD int partition_step(T* a, int first, int last, Less less) {
  const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
  int pick;
  if (less(a[ia], a[ib])) pick = less(a[ib], a[ic]) ? ib : (less(a[ia], a[ic]) ? ic : ia);
  else pick = less(a[ia], a[ic]) ? ia : (less(a[ib], a[ic]) ? ic : ib);
  { const T t = a[first]; a[first] = a[pick]; a[pick] = t; }
  int lo = first + 1, hi = last;
  for (;;) {
    while (lo < last && less(a[lo], a[first])) ++lo;
    if (lo >= last) return -1;
    --hi;
    while (hi > first && less(a[first], a[hi])) --hi;
    if (hi <= first && less(a[first], a[hi])) return -1;
    if (!(lo < hi)) return lo;
    { const T t = a[lo]; a[lo] = a[hi]; a[hi] = t; }
    ++lo;
  }
}

// sorts a[0..n) exactly as std::sort(a, a + n, less) of libstdc++ does; false = a loop bound was hit (inconsistent comparator), result unspecified
template <class T, class Less>
PVLM_HD bool sort(T* a, int n, Less less) {
  if (n <= 0) return true;
  bool sane = true;
  // ---- introsort loop; the recursion on the right part is a stack of (first, last, depth) — at most one entry per level
  int stack_first[64], stack_last[64], stack_depth[64];
  int sp = 0;
  int lg = 0;
  for (int m = n; m > 1; m >>= 1) ++lg;
  stack_first[0] = 0; stack_last[0] = n; stack_depth[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    const int first = stack_first[sp];
    int last = stack_last[sp], depth = stack_depth[sp];
    while (last - first > 16) {
      if (depth == 0) { heap_sort(a + first, last - first, less); break; }
      --depth;
      const int cut = partition_step(a, first, last, less);
      if (cut < 0 || sp >= 64) return false;
      stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth; ++sp;     // introsort_loop(cut, last, depth)
      last = cut;
    }
  }
  // ---- final insertion sort
  const int guarded = n > 16 ? 16 : n;
  for (int i = 1; i < guarded; ++i) {
    const T v = a[i];
    if (less(v, a[0])) {
      for (int k = i; k > 0; --k) a[k] = a[k - 1];
      a[0] = v;
    } else {
      int k = i;
      while (k > 0 && less(v, a[k - 1])) { a[k] = a[k - 1]; --k; }
      a[k] = v;
    }
  }
  for (int i = guarded; i < n; ++i) {
    const T v = a[i];
    int k = i;
    while (k > 0 && less(v, a[k - 1])) { a[k] = a[k - 1]; --k; }
    if (k == 0) sane = false;                               // the unguarded loop would have left the range
    a[k] = v;
  }
  return sane;
}

// The same permutation with the ranges of one recursion level handled side by side (`lanes` at a time: the lanes of a wave on the device, a plain loop
// in the host-compiled check).  The introsort loop is a tree of partition steps on disjoint ranges — a step reads and writes its own range only, and
// its result does not depend on when its siblings run — and the final insertion pass never moves an element across a cut: everything left of a cut is
// <= the pivot <= everything right of it, so the unguarded scan of std::sort stops at the cut too, and the pass falls apart into one per leaf range.
// Requires a strict weak order (the device gives it integer keys); with an inconsistent comparator only the serial form above detects trouble.
//   queue : 2 x kWaveQueue words (ranges of this level / of the next: first | last << 13 | depth << 26, n < 8192)
//   cuts  : (n + 31) / 32 words, bit i = position i starts a leaf range
//   ctr   : 3 ints (queue lengths, "a loop bound was hit")
constexpr int kWaveQueue = 256;                       // ranges longer than 16 alive in one level: <= n / 17
template <class T, class Less, class Lanes>
PVLM_HD bool sort_by_levels(T* a, int n, Less less, unsigned* queue, unsigned* cuts, int* ctr, Lanes&& lanes) {
  // lanes(count, body): runs body(r) for r = 0 .. count - 1, side by side where it can, and returns when all are done (a barrier on the device)
  int lg = 0;
  for (int m = n; m > 1; m >>= 1) ++lg;
  lanes((n + 31) / 32, [&](int w) { cuts[w] = w == 0 ? 1u : 0u; });
  lanes(1, [&](int) { ctr[0] = n > 16 ? 1 : 0; ctr[1] = 0; ctr[2] = 0; queue[0] = 0u | ((unsigned)n << 13) | ((unsigned)(2 * lg) << 26); });
  for (int cur = 0;; cur ^= 1) {
    const int count = ctr[cur];
    if (count == 0) break;
    unsigned* q_in = queue + cur * kWaveQueue;
    unsigned* q_out = queue + (cur ^ 1) * kWaveQueue;
    lanes(count, [&](int r) {
      const unsigned e = q_in[r];
      const int first = (int)(e & 8191u), last = (int)((e >> 13) & 8191u), depth = (int)(e >> 26);
      if (depth == 0) { heap_sort(a + first, last - first, less); return; }
      const int cut = partition_step(a, first, last, less);
      if (cut < 0) { ctr[2] = 1; return; }
#if defined(__HIP_DEVICE_COMPILE__)
      atomicOr(&cuts[cut >> 5], 1u << (cut & 31));
      if (cut - first > 16) q_out[atomicAdd(&ctr[cur ^ 1], 1)] = (unsigned)first | ((unsigned)cut << 13) | ((unsigned)(depth - 1) << 26);
      if (last - cut > 16) q_out[atomicAdd(&ctr[cur ^ 1], 1)] = (unsigned)cut | ((unsigned)last << 13) | ((unsigned)(depth - 1) << 26);
#else
      cuts[cut >> 5] |= 1u << (cut & 31);
      if (cut - first > 16) q_out[ctr[cur ^ 1]++] = (unsigned)first | ((unsigned)cut << 13) | ((unsigned)(depth - 1) << 26);
      if (last - cut > 16) q_out[ctr[cur ^ 1]++] = (unsigned)cut | ((unsigned)last << 13) | ((unsigned)(depth - 1) << 26);
#endif
    });
    lanes(1, [&](int) { ctr[cur] = 0; });
  }
  if (ctr[2]) return false;
  // final insertion pass, leaf by leaf
  lanes(n, [&](int p) {
    if (!((cuts[p >> 5] >> (p & 31)) & 1u)) return;
    int end = p + 1;
    while (end < n && !((cuts[end >> 5] >> (end & 31)) & 1u)) ++end;
    for (int i = p + 1; i < end; ++i) {
      const T v = a[i];
      int k = i;
      while (k > p && less(v, a[k - 1])) { a[k] = a[k - 1]; --k; }
      a[k] = v;
    }
  });
  return true;
}

#if defined(__HIPCC__)
// ---- the device form: one wave per array (workgroup = 64 lanes), the array and the scratch in LDS ---------------------------------------------
// `lanes` of sort_by_levels for a workgroup that is one wave: item r goes to lane r mod 64; the barrier orders the LDS traffic of one level before the next
struct WaveLanes {
  int lane;
  template <class Body> __device__ void operator()(int count, Body&& body) const { for (int r = lane; r < count; r += 64) body(r); __syncthreads(); }
};

__device__ inline unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ inline unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// partition_step with the whole wave on one range.  The unguarded partition swaps, for k = 0, 1, ..., the k-th element from the left that is not less
// than the pivot (L_k) with the k-th from the right that is not greater (R_k), as long as L_k < R_k: its scans only ever read positions neither side has
// passed, whose content is still the original, so both lists can be read off the untouched range; an element equal to the pivot is in both lists but
// can be swapped from one side only (L_j < R_j = q = L_k < R_k would need j < k and k < j).  With k* the number of swaps the left scan ends on
// min(L_k*, R_(k*-1)) — the next original stop, or the stop element the last swap put in its way — and that is the cut.
//   pos: 2 x (last - first) unsigned shorts of LDS.  All 64 lanes call; returns the cut (-1: no stop on the left, inconsistent comparator).
template <class T, class Less>
__device__ inline int partition_wave(T* a, int first, int last, Less less, unsigned short* pos, int lane) {
  if (lane == 0) {
    const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
    int pick;
    if (less(a[ia], a[ib])) pick = less(a[ib], a[ic]) ? ib : (less(a[ia], a[ic]) ? ic : ia);
    else pick = less(a[ia], a[ic]) ? ia : (less(a[ib], a[ic]) ? ic : ib);
    const T t = a[first]; a[first] = a[pick]; a[pick] = t;
  }
  __syncthreads();
  const T pivot = a[first];
  unsigned short* left = pos;
  unsigned short* right = pos + (last - first);
  int n_left = 0, n_right = 0;
  for (int base = first + 1; base < last; base += 64) {
    const int i = base + lane;
    bool nl = false, ng = false;
    if (i < last) { const T x = a[i]; nl = !less(x, pivot); ng = !less(pivot, x); }
    const unsigned long long ml = wave_ballot(nl), mg = wave_ballot(ng);
    if (nl) left[n_left + __popcll(ml & lanes_below(lane))] = (unsigned short)i;
    if (ng) right[n_right + __popcll(mg & lanes_below(lane))] = (unsigned short)i;        // ascending; R_k = right[n_right - 1 - k]
    n_left += __popcll(ml); n_right += __popcll(mg);
  }
  __syncthreads();
  const int pairs = n_left < n_right ? n_left : n_right;
  int swaps = 0;                                                // L_k rises, R_k falls: L_k < R_k holds for k < k* and for no k beyond
  for (int base = 0; base < pairs; base += 64) {
    const int k = base + lane;
    const unsigned long long m = wave_ballot(k < pairs && left[k] < right[n_right - 1 - k]);
    swaps += __popcll(m);
    if (m != ~0ull) break;
  }
  int cut = 0x7FFFFFFF;
  if (swaps < n_left) cut = left[swaps];
  if (swaps > 0 && (int)right[n_right - swaps] < cut) cut = right[n_right - swaps];
  for (int k = lane; k < swaps; k += 64) { const int x = left[k], y = right[n_right - 1 - k]; const T t = a[x]; a[x] = a[y]; a[y] = t; }
  __syncthreads();
  return cut == 0x7FFFFFFF ? -1 : cut;
}

// std::sort(a, a + n, less) by one wave: ranges longer than kWide are partitioned by the whole wave one after the other, the shorter ones side by side
// on the lanes (sort_by_levels' loop), the final insertion pass runs leaf by leaf on the lanes.  n < 8192; all 64 lanes call with the same arguments.
//   queue: 2 x kWaveQueue + kWideStack words;  cuts: (n + 31) / 32 + 1 words;  ctr: 4 ints;  pos: 2 x n unsigned shorts
// measured on the ring batch of 454 scans (K23 ties + K24, ms): kWide 192 / 128 / 96 / 64 / 48 / 32 / 24 / 16 -> 4.64 / 4.20 / 4.08 / 3.87 / 3.71 / 3.61 / 3.68 / 3.75 —
// a partition by the whole wave costs a few LDS round trips whatever the range's length, one by a single lane a round trip per element
#ifndef PVLM_STDSORT_WIDE
#define PVLM_STDSORT_WIDE 32
#endif
constexpr int kWide = PVLM_STDSORT_WIDE, kWideStack = 64;
template <class T, class Less>
__device__ inline bool sort_wave(T* a, int n, Less less, unsigned* queue, unsigned* cuts, int* ctr, unsigned short* pos, int lane) {
  int lg = 0;
  for (int m = n; m > 1; m >>= 1) ++lg;
  for (int w = lane; w < (n + 31) / 32 + 1; w += 64) cuts[w] = w == 0 ? 1u : 0u;
  if (lane == 0) { ctr[0] = 0; ctr[1] = 0; ctr[2] = 0; }
  __syncthreads();
  unsigned* wide = queue + 2 * kWaveQueue;
  int n_wide = 0;                                               // wave-uniform
  auto pack = [](int first, int last, int depth) { return (unsigned)first | ((unsigned)last << 13) | ((unsigned)depth << 26); };
  auto push = [&](int first, int last, int depth) {             // called by every lane with the same values
    if (last - first <= 16) return;
    if (last - first > kWide && n_wide < kWideStack) { if (lane == 0) wide[n_wide] = pack(first, last, depth); ++n_wide; }
    else if (lane == 0) queue[ctr[0]++] = pack(first, last, depth);
  };
  push(0, n, 2 * lg);
  __syncthreads();
  while (n_wide > 0) {
    --n_wide;
    const unsigned e = wide[n_wide];
    const int first = (int)(e & 8191u), last = (int)((e >> 13) & 8191u), depth = (int)(e >> 26);
    if (depth == 0) { if (lane == 0) heap_sort(a + first, last - first, less); __syncthreads(); continue; }
    const int cut = partition_wave(a, first, last, less, pos, lane);
    if (cut < 0) return false;
    if (lane == 0) cuts[cut >> 5] |= 1u << (cut & 31);
    __syncthreads();
    push(first, cut, depth - 1);
    push(cut, last, depth - 1);
    __syncthreads();
  }
  const WaveLanes lanes{lane};
  for (int cur = 0;; cur ^= 1) {
    const int count = ctr[cur];
    if (count == 0) break;
    unsigned* q_in = queue + cur * kWaveQueue;
    unsigned* q_out = queue + (cur ^ 1) * kWaveQueue;
    lanes(count, [&](int r) {
      const unsigned e = q_in[r];
      const int first = (int)(e & 8191u), last = (int)((e >> 13) & 8191u), depth = (int)(e >> 26);
      if (depth == 0) { heap_sort(a + first, last - first, less); return; }
      const int cut = partition_step(a, first, last, less);
      if (cut < 0) { ctr[2] = 1; return; }
      atomicOr(&cuts[cut >> 5], 1u << (cut & 31));
      if (cut - first > 16) q_out[atomicAdd(&ctr[cur ^ 1], 1)] = pack(first, cut, depth - 1);
      if (last - cut > 16) q_out[atomicAdd(&ctr[cur ^ 1], 1)] = pack(cut, last, depth - 1);
    });
    if (lane == 0) ctr[cur] = 0;
    __syncthreads();
  }
  if (ctr[2]) return false;
  // final insertion pass: the leaf starts are gathered first, then every lane takes whole leaves
  int n_leaves = 0;
  for (int base = 0; base < n; base += 64) {
    const int p = base + lane;
    const bool start = p < n && ((cuts[p >> 5] >> (p & 31)) & 1u);
    const unsigned long long m = wave_ballot(start);
    if (start) pos[n_leaves + __popcll(m & lanes_below(lane))] = (unsigned short)p;
    n_leaves += __popcll(m);
  }
  __syncthreads();
  for (int r = lane; r < n_leaves; r += 64) {
    const int p = pos[r], end = r + 1 < n_leaves ? (int)pos[r + 1] : n;
    for (int i = p + 1; i < end; ++i) {
      const T v = a[i];
      int k = i;
      while (k > p && less(v, a[k - 1])) { a[k] = a[k - 1]; --k; }
      a[k] = v;
    }
  }
  __syncthreads();
  return true;
}
#endif

}  // namespace pvlm_stdsort
