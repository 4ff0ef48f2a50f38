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
// depth-limit-forcing inputs (tests/test_stdsort_cpu.py), and pvlm_stdsort_selfcheck() (pvlm_ring.hip) repeats a short version of that
// comparison against the std::sort the library itself was built with before the device is allowed to order a sector with ties.
//
// `less(x, y)` compares two ELEMENTS (values of the array), like the comparator handed to std::sort.  The ranges a correct strict weak order
// keeps the unguarded loops in are not assumed: every loop is bounded by the range and the function returns false if a bound stopped it
// (NaN keys can do that); the caller then falls back to the real std::sort on the host.
#pragma once

#ifndef PVLM_HD
#define PVLM_HD __host__ __device__
#endif

namespace pvlm_stdsort {

template <class T, class Less>
PVLM_HD inline void adjust_heap(T* a, int hole, int len, T value, Less less) {
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
PVLM_HD inline void heap_sort(T* a, int n, Less less) {   // partial_sort(first, last, last): make_heap, then sort_heap
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

// sorts a[0..n) exactly as std::sort(a, a + n, less) of libstdc++ does; false = a loop bound was hit (inconsistent comparator), result unspecified
template <class T, class Less>
PVLM_HD inline bool sort(T* a, int n, Less less) {
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
      // median of three to `first`
      const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
      int pick;
      if (less(a[ia], a[ib])) pick = less(a[ib], a[ic]) ? ib : (less(a[ia], a[ic]) ? ic : ia);
      else pick = less(a[ia], a[ic]) ? ia : (less(a[ib], a[ic]) ? ic : ib);
      { const T t = a[first]; a[first] = a[pick]; a[pick] = t; }
      // unguarded partition of [first + 1, last) around a[first]
      int lo = first + 1, hi = last;
      for (;;) {
        while (lo < last && less(a[lo], a[first])) ++lo;
        if (lo >= last) { sane = false; break; }
        --hi;
        while (hi > first && less(a[first], a[hi])) --hi;
        if (hi <= first && less(a[first], a[hi])) { sane = false; break; }
        if (!(lo < hi)) break;
        { const T t = a[lo]; a[lo] = a[hi]; a[hi] = t; }
        ++lo;
      }
      if (!sane) return false;
      const int cut = lo;
      if (sp >= 64) return false;
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

}  // namespace pvlm_stdsort
