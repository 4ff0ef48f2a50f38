// Per-query bodies of the association kernels (K2 exact k-NN in the voxel grid, K3 plane fit + collinearity test):
// the device side of AssociatePoint2Plane (lidar_mapping/LidarFeatureAssociate.cpp:550-630).  Host/device so that
// tests/cpp/assoc_core_check.cpp can drive the very same functions serially on a machine without a GPU
// (tests/test_assoc_core_cpu.py: search == brute force, fits == the oracle's decisions).  libpvlm.so has no host path.
//
// Arithmetic contract: distances are float32 in flann::L2_Simple order, every accept / reject decision is fp64 in the
// reference's operation order; the including translation unit is compiled with -ffp-contract=off.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef PVLM_HD
#define PVLM_HD __host__ __device__ __forceinline__
#define PVLM_ASSOC_DEVICE 1
#endif

#ifndef PVLM_ASSOC_STATS_CANDIDATE   // counting hooks of the host-compiled check (tests/cpp/assoc_core_check.cpp)
#define PVLM_ASSOC_STATS_CANDIDATE() ((void)0)
#define PVLM_ASSOC_STATS_ROW() ((void)0)
#endif
#ifndef PVLM_ASSOC_STATS_ITER
#define PVLM_ASSOC_STATS_ITER(r, dz, dy, part) ((void)0)
#define PVLM_ASSOC_STATS_RUN(len, pass) ((void)0)
#endif

namespace pvlm_assoc {

#define PVLM_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define PVLM_CELL_BIAS (1 << 20)

struct Point4 { float x, y, z, w; };   // (x, y, z, original index as float bits): one 16-byte load per candidate

struct CloudView {
  const Point4* sorted;
  const unsigned long long* keys;
  const int* cell_start;   // hash: start of the slot's cell; dense: prefix offsets, ncells + 1 entries
  const int* cell_count;
  const float* xyz;  // original order, interleaved
  const float* tag;
  const Point4* pt4; // original order: (x, y, z, tag) — clouds uploaded with tags
  int n, mask;
  int dense, nx, ny, nz;   // dense != 0: cells addressed directly as (iz*ny + iy)*nx + ix, no hashing
  int xf;                  // dense tables: cells are xf times finer along x (nx counts the FINE cells of a row); hashed tables: 1
  float ox, oy, oz, h, inv_h;
};

PVLM_HD unsigned long long mix64(unsigned long long x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}
PVLM_HD unsigned long long cell_key(int ix, int iy, int iz) {
  return (unsigned long long)((ix + PVLM_CELL_BIAS) & 0x1FFFFF) | ((unsigned long long)((iy + PVLM_CELL_BIAS) & 0x1FFFFF) << 21) |
         ((unsigned long long)((iz + PVLM_CELL_BIAS) & 0x1FFFFF) << 42);
}
PVLM_HD int cell_of(float x, float o, float inv_h) {
  float c = floorf((x - o) * inv_h);
  c = fminf(fmaxf(c, -1000000.f), 1000000.f);
  return (int)c;
}
// square root for the pruning reach only (never for a distance that is compared or returned): the raw hardware estimate,
// the caller adds a relative margin; sqrtf() expands to 20 instructions of denormal scaling and last-bit correction
#if defined(PVLM_ASSOC_DEVICE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float sqrt_reach(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
PVLM_HD float sqrt_reach(float x) { return sqrtf(x); }
#endif
// min and max of an int over the ACTIVE lanes of the wave, the same value in every lane (host: the value itself).  A shuffle butterfly
// when the whole wave is active — inactive lanes do not forward partial results, so with a partial mask the butterfly is wrong —
// otherwise one v_readlane per active lane.  Called once per query wave, before any lane has left the search.
#if defined(PVLM_ASSOC_DEVICE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void wave_minmax_i(int v, int* mn, int* mx) {
  const unsigned long long act = __ballot(1);
  int lo = v, hi = v;
  if (act == ~0ull) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
  } else {
    lo = 0x7fffffff; hi = -0x7fffffff - 1;
    for (unsigned long long m = act; m; m &= m - 1) { const int x = __builtin_amdgcn_readlane(v, __builtin_ctzll(m)); lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
  }
  *mn = __builtin_amdgcn_readfirstlane(lo); *mx = __builtin_amdgcn_readfirstlane(hi);
}
#else
PVLM_HD void wave_minmax_i(int v, int* mn, int* mx) { *mn = v; *mx = v; }
#endif
PVLM_HD unsigned f2u(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
PVLM_HD float u2f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

// Table loads of the search.  The pointers come out of a descriptor in memory, so the compiler cannot prove their address space and emits
// FLAT loads through 64-bit VGPR pointers (aperture check, vmcnt + lgkmcnt, three VALU instructions of address arithmetic per load).
// On the device they are stated to be global memory addressed as wave-uniform base + 32-bit BYTE offset: `global_load ... v_off, s[base]`
// — no pointer pairs in VGPRs (a cloud is at most 2^28 points: 16 B x index fits 32 bits).
#if defined(PVLM_ASSOC_DEVICE) && defined(__HIP_DEVICE_COMPILE__) && !defined(PVLM_K2_FLAT_LOADS)
typedef float pvlm_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Point4 load_point(const Point4* base, int j) {
  const pvlm_f4 v = *(const __attribute__((address_space(1))) pvlm_f4*)((const __attribute__((address_space(1))) char*)(const char*)base + ((unsigned)j << 4));
  Point4 p; p.x = v.x; p.y = v.y; p.z = v.z; p.w = v.w;
  return p;
}
__device__ __forceinline__ int load_int(const int* base, int i) {
  return *(const __attribute__((address_space(1))) int*)((const __attribute__((address_space(1))) char*)(const char*)base + ((unsigned)i << 2));
}
__device__ __forceinline__ float load_float(const float* base, int i) {
  return *(const __attribute__((address_space(1))) float*)((const __attribute__((address_space(1))) char*)(const char*)base + ((unsigned)i << 2));
}
#else
PVLM_HD Point4 load_point(const Point4* base, int j) { return base[j]; }
PVLM_HD int load_int(const int* base, int i) { return base[i]; }
PVLM_HD float load_float(const float* base, int i) { return base[i]; }
#endif

// ---- sorted top-K of (distance, index) ---------------------------------------------------------------------------------
// A squared distance is a non-negative float, whose bit pattern orders like its value, so (float bits << 32 | index) is
// ONE 64-bit key ordered exactly like the lexicographic (distance, index) pair the reference's sorted k-NN result
// implies.  Round 2 inserted with a compare-and-swap chain on u64 (v_cmp_lt_u64 + four v_cndmask per slot, 45
// instructions whenever ANY lane of the wave inserts).  Now the key lives in a register pair as the bit pattern of a
// POSITIVE NORMAL DOUBLE — positive doubles order like their bit patterns too — and an insertion into the sorted list is
//     key'[k] = min(key[k], max(key[k-1], c)),   key'[0] = min(key[0], c)
// i.e. 2K - 1 v_min_f64 / v_max_f64, branch-free, whatever the other lanes do.  KEY_BIAS lifts the exponent field by one
// so that a distance of exactly 0 (key high word 0) is a normal number and no denormal ever reaches the min / max units.
#define PVLM_KEY_BIAS 0x00100000u
#define PVLM_KEY_INF_HI (0x7F800000u + PVLM_KEY_BIAS)      // (+inf, index -1): an empty slot
#define PVLM_KEY_REJECT_HI (0x7F800001u + PVLM_KEY_BIAS)   // above every slot: never enters the list

#if defined(PVLM_ASSOC_DEVICE) && defined(__HIP_DEVICE_COMPILE__)
typedef double topk_key;
__device__ __forceinline__ topk_key key_make(unsigned hi, unsigned lo) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ unsigned key_hi(topk_key k) { return (unsigned)__double2hiint(k); }
__device__ __forceinline__ unsigned key_lo(topk_key k) { return (unsigned)__double2loint(k); }
// inline asm: fmin() / fmax() would be preceded by a canonicalising v_max_f64 x, x of operands the compiler cannot prove quiet
__device__ __forceinline__ topk_key key_min(topk_key a, topk_key b) { topk_key r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ topk_key key_max(topk_key a, topk_key b) { topk_key r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
typedef unsigned long long topk_key;
PVLM_HD topk_key key_make(unsigned hi, unsigned lo) { return ((unsigned long long)hi << 32) | lo; }
PVLM_HD unsigned key_hi(topk_key k) { return (unsigned)(k >> 32); }
PVLM_HD unsigned key_lo(topk_key k) { return (unsigned)(k & 0xFFFFFFFFull); }
PVLM_HD topk_key key_min(topk_key a, topk_key b) { return a < b ? a : b; }
PVLM_HD topk_key key_max(topk_key a, topk_key b) { return a < b ? b : a; }
#endif

// The distance threshold lives in the list itself: an empty slot holds the key (threshold bits + 1, index 0) — the smallest key of a
// distance ABOVE the threshold — so a candidate beyond the threshold can never displace it (equal at best: no change) and the
// candidate loop needs no threshold test of its own (round 3: v_cmp_le + v_cndmask per candidate).  Empty slots read back as
// (index -1, distance +inf).
template <int K>
struct TopK {
  topk_key key[K];
  unsigned empty_hi;                                                              // high word of an empty slot
  PVLM_HD void init(float thr2) {
    empty_hi = (thr2 < 3.0e38f ? f2u(thr2) + 1u : 0x7F800000u) + PVLM_KEY_BIAS;      // thr2 = +inf (or NaN): nothing is beyond it
#pragma unroll
    for (int k = 0; k < K; ++k) key[k] = key_make(empty_hi, 0u);
  }
  // candidate (d2, idx), d2 a non-negative float (finite or +inf; the clouds are finite)
  PVLM_HD void push(float d2, int idx) {
    const topk_key c = key_make(f2u(d2) + PVLM_KEY_BIAS, (unsigned)idx);
#ifdef PVLM_K2_NONET   // timing experiment only (wrong results): what the insertion network costs
    key[0] = key_min(key[0], c);
#else
#pragma unroll
    for (int k = K - 1; k > 0; --k) key[k] = key_min(key[k], key_max(key[k - 1], c));
    key[0] = key_min(key[0], c);
#endif
  }
  PVLM_HD bool full() const { return key_hi(key[K - 1]) < empty_hi; }
  PVLM_HD bool would_enter(float d2, int idx) const { return key_make(f2u(d2) + PVLM_KEY_BIAS, (unsigned)idx) < key[K - 1]; }
  // the pruning bound: the k-th distance, or (just above) the threshold while the list is not full
  PVLM_HD float kth_or_threshold() const { return u2f(key_hi(key[K - 1]) - PVLM_KEY_BIAS); }
  PVLM_HD float dist(int k) const { const unsigned h = key_hi(key[k]); return h < empty_hi ? u2f(h - PVLM_KEY_BIAS) : u2f(0x7F800000u); }
  PVLM_HD int index(int k) const { return key_hi(key[k]) < empty_hi ? (int)key_lo(key[k]) : -1; }
};

// ---- K2: exact k-NN of one query ------------------------------------------------------------------------------------------
// Chebyshev shells of cells around the query's cell, as in round 2, but every (z, y) row of a shell is first tested against
// the CURRENT k-th distance (or dist_threshold while the list is not full): a row whose nearest possible point is farther is
// skipped, and the x-run of a surviving row is clipped to the cells the remaining budget can reach.  A skipped cell only
// holds points with d2 > the k-th distance, which can never enter the list (a tie enters only at EQUAL distance), so the
// result is still the exact, tie-broken k-NN; the slack covers the float rounding of the points' cell assignment.
// Dense tables are `xf` times finer along x than along y / z: the cells of a (z, y) row stay one contiguous run of the sorted
// array, so the finer x resolution costs no extra row — it only lets the clip cut a run to (almost) exactly the interval the
// budget reaches.  Shells, gaps and the termination bound work on the coarse cells; visit() receives FINE x indices.
template <int K, class Visit>
PVLM_HD void knn_rows(const CloudView& cv, float qx, float qy, float qz, float max_dist, float thr2, TopK<K>& tk, Visit&& visit) {
  tk.init(thr2);
  if (cv.n <= 0) return;
  const int xf = cv.xf;
  const float ux = (qx - cv.ox) * cv.inv_h;                                   // x of the query in coarse cells
  const int cxf = cell_of(qx, cv.ox, cv.inv_h * (float)xf);                   // its fine cell, as the grid build computes it
  const int cx = cxf >= 0 ? cxf / xf : -((-cxf + xf - 1) / xf);               // coarse cell = floor(fine / xf)
  const int cy = cell_of(qy, cv.oy, cv.inv_h), cz = cell_of(qz, cv.oz, cv.inv_h);
  const float fx = ux - (float)cx, fy = (qy - cv.oy) * cv.inv_h - (float)cy, fz = (qz - cv.oz) * cv.inv_h - (float)cz;
  const float lo_min = fminf(fminf(fx, fy), fz), hi_min = fminf(fminf(1.f - fx, 1.f - fy), 1.f - fz);
  const float inside = fminf(lo_min, hi_min);  // distance (in cells) from q to the nearest face of its own cell
  const float slack = 1e-3f * cv.h + 2e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz) + 1.f);
  const int rmax = (int)ceilf(max_dist * 1.0001f * cv.inv_h);
  // Dense tables: the (dz, dy) loops only run over rows that exist in SOME lane's table range (a line- or plane-shaped cloud has a table
  // a few cells thick: with the loops over the full shell a query with fewer than K points in reach walked O(rmax^3) empty rows), and the
  // search stops once the searched block covers the whole table.  The bounds come from the wave's min / max cell (reduced once, at
  // entry), so that the loop counters stay scalar: per-lane bounds cost K2 5 % in divergent loop control.
  const int ncx = cv.dense ? cv.nx / xf : 0;
  int cz_min = 0, cz_max = 0, cy_min = 0, cy_max = 0;        // over the wave: the clipped loop bounds below stay scalar
  if (cv.dense) { wave_minmax_i(cz, &cz_min, &cz_max); wave_minmax_i(cy, &cy_min, &cy_max); }
  // Pruning arithmetic (never a distance that is compared or returned): fused multiply-adds, and the per-row bound as two FMAs off
  // per-query constants — gap(d) h - slack = |d| h + (d > 0 ? -f h - slack : (f - 1) h - slack).  The budget (k-th distance, or the
  // threshold while the list is not full) only changes when a row was scanned: it is refreshed there, not in every row test
  // (round 3: 15 VALU instructions per skipped row, 45 row tests per query).
  const float ypos = fmaf(-fy, cv.h, -slack), yneg = fmaf(fy - 1.f, cv.h, -slack), zpos = fmaf(-fz, cv.h, -slack), zneg = fmaf(fz - 1.f, cv.h, -slack);
  float budget = tk.kth_or_threshold() * 1.00001f;
  for (int r = 0; r <= rmax; ++r) {
    int dz_lo = -r, dz_hi = r, dy_lo = -r, dy_hi = r;
    if (cv.dense) {                                            // rows of SOME lane's table range; the per-lane range test below remains
      dz_lo = -cz_max > dz_lo ? -cz_max : dz_lo; dz_hi = cv.nz - 1 - cz_min < dz_hi ? cv.nz - 1 - cz_min : dz_hi;
      dy_lo = -cy_max > dy_lo ? -cy_max : dy_lo; dy_hi = cv.ny - 1 - cy_min < dy_hi ? cv.ny - 1 - cy_min : dy_hi;
    }
    for (int dz = dz_lo; dz <= dz_hi; ++dz) {
      const int z = cz + dz;
      if (cv.dense && (z < 0 || z >= cv.nz)) continue;
      const float gzm = dz == 0 ? 0.f : fmaxf(fmaf((float)(dz > 0 ? dz : -dz), cv.h, dz > 0 ? zpos : zneg), 0.f);   // gap to the row's slab minus the slack
      const float gz2 = gzm * gzm;
      for (int dy = dy_lo; dy <= dy_hi; ++dy) {
        const int y = cy + dy;
        if (cv.dense && (y < 0 || y >= cv.ny)) continue;
        const float gym = dy == 0 ? 0.f : fmaxf(fmaf((float)(dy > 0 ? dy : -dy), cv.h, dy > 0 ? ypos : yneg), 0.f);
        const float lb = fmaf(gym, gym, gz2);                         // <= d2 of every point of the row
        if (lb > budget) continue;
        const float reach = fmaf(sqrt_reach(budget - lb), 1.0001f, slack) * cv.inv_h;  // (coarse) cells the budget still reaches along x
        const int xa = (int)floorf((ux - reach) * (float)xf), xb = (int)floorf((ux + reach) * (float)xf);   // fine cells
        const bool face = (dz == -r || dz == r || dy == -r || dy == r);
        // a face row of the shell is one x-run; an interior row only owns its two end cells (one call site for both:
        // the candidate loop is inlined once)
        for (int part = 0; part < 2; ++part) {
          int x0, x1;   // fine cells
          if (face) {
            if (part) break;
            x0 = (cx - r) * xf; x1 = (cx + r + 1) * xf - 1;
          } else {
            x0 = (part ? cx + r : cx - r) * xf; x1 = x0 + xf - 1;
          }
          x0 = x0 > xa ? x0 : xa; x1 = x1 < xb ? x1 : xb;
#ifdef PVLM_K2_SHELL_BUDGET   // statistics experiment (tools/assoc_lockstep.py): the budget of a shell fixed when the shell is entered
          if (x0 <= x1) { PVLM_ASSOC_STATS_ROW(); PVLM_ASSOC_STATS_ITER(r, dz, dy, part); visit(z, y, x0, x1); }
#else
          if (x0 <= x1) { PVLM_ASSOC_STATS_ROW(); PVLM_ASSOC_STATS_ITER(r, dz, dy, part); visit(z, y, x0, x1); budget = tk.kth_or_threshold() * 1.00001f; }
#endif
        }
      }
    }
#ifdef PVLM_K2_SHELL_BUDGET
    budget = tk.kth_or_threshold() * 1.00001f;
#endif
    // every unsearched point lies outside the (2r+1)^3 block: farther than (inside + r) cells
    const float bound = (inside + (float)r) * cv.h - slack;
    if (tk.full() && bound > 0.f && tk.kth_or_threshold() < bound * bound) break;
    if ((float)r * cv.h >= max_dist * 1.0001f) break;  // everything within max_dist has been visited
    if (cv.dense && cx - r <= 0 && cx + r >= ncx - 1 && cy - r <= 0 && cy + r >= cv.ny - 1 && cz - r <= 0 && cz + r >= cv.nz - 1) break;   // ... or the whole table
  }
}

// Candidates of one run [b, e) of the sorted array.  The loads of PVLM_K2_BATCH consecutive candidates are issued together before
// the first of them goes through the insertion network (the compiler schedules a one-candidate loop as load - wait - use: the wave
// then sits out one memory latency per candidate; a run is 3.3 candidates long on average, so four loads in flight cover most runs
// with ONE latency).  Slots past the end of the run re-read its last candidate and are turned into +inf keys.
#ifndef PVLM_K2_ANYSKIP
#define PVLM_K2_ANYSKIP 1
#endif
#ifndef PVLM_K2_BATCH
#define PVLM_K2_BATCH 1   // measured (profiles/r4_assoc_variants.txt): 1 / 2 / 4 / 8 -> 1327 / 1367 / 1466 / 2436 us per dispatch: the padded slots cost more network work than the latency they hide
#endif
template <int K>
PVLM_HD void knn_scan_run(const CloudView& cv, int b, int e, float qx, float qy, float qz, float thr2, TopK<K>& tk) {
  (void)thr2;
  if (b >= e) return;
  const Point4* s = cv.sorted;
#if PVLM_K2_BATCH == 1 && defined(PVLM_K2_PREFETCH)
  // software pipeline: record j + 1 is requested before record j goes through the insertion network (the array has one spare record behind
  // its last point, so the request needs no bound test — its result is simply not used after the last iteration)
  Point4 cur = load_point(s, b);
  for (int j = b; j < e; ++j) {
    const Point4 nxt = load_point(s, j + 1);
    const float ddx = qx - cur.x, ddy = qy - cur.y, ddz = qz - cur.z;
    float d2 = 0.0f;
    d2 += ddx * ddx; d2 += ddy * ddy; d2 += ddz * ddz;  // flann::L2_Simple order
    tk.push(d2, (int)f2u(cur.w));
    PVLM_ASSOC_STATS_CANDIDATE();
    cur = nxt;
  }
  PVLM_ASSOC_STATS_RUN(e - b, 0);
#else
  const int last = e - 1;
  int stat_pass = 0; (void)stat_pass;
  for (int j = b; j < e; j += PVLM_K2_BATCH) {
    Point4 p[PVLM_K2_BATCH];
#pragma unroll
    for (int k = 0; k < PVLM_K2_BATCH; ++k) p[k] = load_point(s, k == 0 ? j : (j + k < last ? j + k : last));
#pragma unroll
    for (int k = 0; k < PVLM_K2_BATCH; ++k) {
      const float ddx = qx - p[k].x, ddy = qy - p[k].y, ddz = qz - p[k].z;
      float d2 = 0.0f;
      d2 += ddx * ddx; d2 += ddy * ddy; d2 += ddz * ddz;  // flann::L2_Simple order
      if (k > 0) d2 = j + k < e ? d2 : u2f(0x7F800000u);  // past the end: never enters the list
#ifdef PVLM_ASSOC_STATS_COUNT_PASS
      if ((k == 0 || j + k < e) && tk.would_enter(d2, (int)f2u(p[k].w))) ++stat_pass;
#endif
#if PVLM_K2_ANYSKIP && defined(PVLM_ASSOC_DEVICE) && defined(__HIP_DEVICE_COMPILE__)
      // the 19-operation insertion only when the candidate beats the last key of SOME lane of the wave (one compare + a scalar branch;
      // measured, profiles/r5_assoc_variants.txt: k_knn_pairs 1164 -> 1115 us on voxel targets, 5522 -> 4795 us on raw targets)
      if (__builtin_amdgcn_ballot_w64(tk.would_enter(d2, (int)f2u(p[k].w))) == 0ull) continue;
#endif
      tk.push(d2, (int)f2u(p[k].w));
      if (k == 0 || j + k < e) PVLM_ASSOC_STATS_CANDIDATE();
    }
  }
  PVLM_ASSOC_STATS_RUN(e - b, stat_pass);
#endif
}

template <int K, bool DENSE>
PVLM_HD void knn_search_grid(const CloudView& cv, float qx, float qy, float qz, float max_dist, float thr2, TopK<K>& tk) {
  knn_rows<K>(cv, qx, qy, qz, max_dist, thr2, tk, [&](int z, int y, int x0, int x1) {
    if (DENSE) {
      // the cells of one (z, y) row are contiguous in the sorted array: the whole x-run is one range
      x0 = x0 > 0 ? x0 : 0; x1 = x1 < cv.nx - 1 ? x1 : cv.nx - 1;
      if (x0 > x1) return;
      const int row = (z * cv.ny + y) * cv.nx;
      knn_scan_run<K>(cv, load_int(cv.cell_start, row + x0), load_int(cv.cell_start, row + x1 + 1), qx, qy, qz, thr2, tk);
    } else {
      for (int x = x0; x <= x1; ++x) {
        const unsigned long long key = cell_key(x, y, z);
        int s = (int)(mix64(key) & (unsigned long long)cv.mask);
        unsigned long long kk;
        while ((kk = cv.keys[s]) != key && kk != PVLM_EMPTY_KEY) s = (s + 1) & cv.mask;
        if (kk != key) continue;
        const int b = load_int(cv.cell_start, s);
        knn_scan_run<K>(cv, b, b + load_int(cv.cell_count, s), qx, qy, qz, thr2, tk);
      }
    }
  });
}

template <int K>
PVLM_HD void knn_search(const CloudView& cv, float qx, float qy, float qz, float max_dist, float thr2, TopK<K>& tk) {
  if (cv.dense) knn_search_grid<K, true>(cv, qx, qy, qz, max_dist, thr2, tk);
  else knn_search_grid<K, false>(cv, qx, qy, qz, max_dist, thr2, tk);
}

// ---- K3: fp64 fits (sequential-sum Householder / Jacobi arithmetic the parity tests pin, fully unrolled, static indexing) ----
struct Fit10 {
  // Householder QR with column pivoting on the 10x3 system A n = -1 (Eigen ColPivHouseholderQR
  // restated, base/Geometry.hpp:345-373), then the tolerance test.  c0,c1,c2 = columns (destroyed).
  static PVLM_HD void swap_cols(double* a, double* b) {
#pragma unroll
    for (int i = 0; i < 10; ++i) { const double t = a[i]; a[i] = b[i]; b[i] = t; }
  }
  template <int KK>
  static PVLM_HD void householder(double* ck, double* cj1, double* cj2, double& tau) {
    double tailSq = 0.0;
#pragma unroll
    for (int i = KK + 1; i < 10; ++i) tailSq += ck[i] * ck[i];
    const double c0 = ck[KK];
    double beta;
    if (tailSq <= DBL_MIN) {
      tau = 0.0; beta = c0;
#pragma unroll
      for (int i = KK + 1; i < 10; ++i) ck[i] = 0.0;
    } else {
      beta = sqrt(c0 * c0 + tailSq);
      if (c0 >= 0.0) beta = -beta;
      const double den = c0 - beta;
#pragma unroll
      for (int i = KK + 1; i < 10; ++i) ck[i] = ck[i] / den;
      tau = (beta - c0) / beta;
    }
    ck[KK] = beta;
    if (tau != 0.0) {
      if (cj1) {
        double tmp = 0.0;
#pragma unroll
        for (int i = KK + 1; i < 10; ++i) tmp += ck[i] * cj1[i];
        tmp += cj1[KK];
        cj1[KK] -= tau * tmp;
#pragma unroll
        for (int i = KK + 1; i < 10; ++i) cj1[i] -= tau * ck[i] * tmp;
      }
      if (cj2) {
        double tmp = 0.0;
#pragma unroll
        for (int i = KK + 1; i < 10; ++i) tmp += ck[i] * cj2[i];
        tmp += cj2[KK];
        cj2[KK] -= tau * tmp;
#pragma unroll
        for (int i = KK + 1; i < 10; ++i) cj2[i] -= tau * ck[i] * tmp;
      }
    }
  }
  template <int KK>
  static PVLM_HD void downdate(const double* cj, double& nU, double& nD, double thr) {
    if (nU != 0.0) {
      double temp = fabs(cj[KK]) / nU;
      temp = (1.0 + temp) * (1.0 - temp);
      temp = temp < 0.0 ? 0.0 : temp;
      const double ratio = nU / nD;
      const double temp2 = temp * ratio * ratio;
      if (temp2 <= thr) {
        double s = 0.0;
#pragma unroll
        for (int i = KK + 1; i < 10; ++i) s += cj[i] * cj[i];
        nD = sqrt(s);
        nU = nD;
      } else {
        nU *= sqrt(temp);
      }
    }
  }
  template <int KK>
  static PVLM_HD void apply_rhs(const double* ck, double tau, double* b) {
    if (tau != 0.0) {
      double tmp = 0.0;
#pragma unroll
      for (int i = KK + 1; i < 10; ++i) tmp += ck[i] * b[i];
      tmp += b[KK];
      b[KK] -= tau * tmp;
#pragma unroll
      for (int i = KK + 1; i < 10; ++i) b[i] -= tau * ck[i] * tmp;
    }
  }

  // pts: 10 x 3 (px[10], py[10], pz[10]).  Returns plane_ok; plane = (n, d).
  static PVLM_HD bool form_plane(const double* px, const double* py, const double* pz, double tol, double* plane) {
    double c0[10], c1[10], c2[10], x[3];
#pragma unroll
    for (int i = 0; i < 10; ++i) { c0[i] = px[i]; c1[i] = py[i]; c2[i] = pz[i]; }
    form_plane_solve(c0, c1, c2, x);
    return form_plane_accept(x, px, py, pz, tol, plane);
  }
  // the two halves of form_plane for callers short of registers: the solve works IN PLACE on the three coordinate arrays (destroyed), the accept test
  // wants the points again (the fast kernel's fall-back gathers them a second time instead of keeping 30 doubles alive across the factorisation)
  static PVLM_HD void form_plane_solve(double* c0, double* c1, double* c2, double* x) {
    double b[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = -1.0;
    const double eps = DBL_EPSILON;
    double nU0, nU1, nU2, nD0, nD1, nD2;
    {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int i = 0; i < 10; ++i) { s0 += c0[i] * c0[i]; s1 += c1[i] * c1[i]; s2 += c2[i] * c2[i]; }
      nD0 = nU0 = sqrt(s0); nD1 = nU1 = sqrt(s1); nD2 = nU2 = sqrt(s2);
    }
    double maxn = nU0; if (nU1 > maxn) maxn = nU1; if (nU2 > maxn) maxn = nU2;
    const double threshold_helper = (maxn * eps) * (maxn * eps) / 10.0;
    const double ndt = sqrt(eps);
    int nonzero = 3;
    int p0 = 0, p1 = 1, p2 = 2;
    double tau0, tau1, tau2;
    // ---- k = 0
    {
      int big = 0; double bigv = nU0;
      if (nU1 > bigv) { bigv = nU1; big = 1; }
      if (nU2 > bigv) { bigv = nU2; big = 2; }
      if (nonzero == 3 && bigv * bigv < threshold_helper * 10.0) nonzero = 0;
      if (big == 1) { swap_cols(c0, c1); double t = nU0; nU0 = nU1; nU1 = t; t = nD0; nD0 = nD1; nD1 = t; int q = p0; p0 = p1; p1 = q; }
      else if (big == 2) { swap_cols(c0, c2); double t = nU0; nU0 = nU2; nU2 = t; t = nD0; nD0 = nD2; nD2 = t; int q = p0; p0 = p2; p2 = q; }
      householder<0>(c0, c1, c2, tau0);
      downdate<0>(c1, nU1, nD1, ndt);
      downdate<0>(c2, nU2, nD2, ndt);
    }
    // ---- k = 1
    {
      int big = 1; double bigv = nU1;
      if (nU2 > bigv) { bigv = nU2; big = 2; }
      if (nonzero == 3 && bigv * bigv < threshold_helper * 9.0) nonzero = 1;
      if (big == 2) { swap_cols(c1, c2); double t = nU1; nU1 = nU2; nU2 = t; t = nD1; nD1 = nD2; nD2 = t; int q = p1; p1 = p2; p2 = q; }
      householder<1>(c1, c2, nullptr, tau1);
      downdate<1>(c2, nU2, nD2, ndt);
    }
    // ---- k = 2
    {
      const double bigv = nU2;
      if (nonzero == 3 && bigv * bigv < threshold_helper * 8.0) nonzero = 2;
      householder<2>(c2, nullptr, nullptr, tau2);
    }
    if (nonzero > 0) apply_rhs<0>(c0, tau0, b);
    if (nonzero > 1) apply_rhs<1>(c1, tau1, b);
    if (nonzero > 2) apply_rhs<2>(c2, tau2, b);
    // back substitution on the leading nonzero x nonzero triangle: R = [c0[0] c1[0] c2[0]; 0 c1[1] c2[1]; 0 0 c2[2]]
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    if (nonzero > 2) y2 = b[2] / c2[2];
    if (nonzero > 1) { double s = b[1]; if (nonzero > 2) s -= c2[1] * y2; y1 = s / c1[1]; }
    if (nonzero > 0) { double s = b[0]; if (nonzero > 1) s -= c1[0] * y1; if (nonzero > 2) s -= c2[0] * y2; y0 = s / c0[0]; }
    x[0] = 0.0; x[1] = 0.0; x[2] = 0.0;
    if (nonzero > 0) { if (p0 == 0) x[0] = y0; else if (p0 == 1) x[1] = y0; else x[2] = y0; }
    if (nonzero > 1) { if (p1 == 0) x[0] = y1; else if (p1 == 1) x[1] = y1; else x[2] = y1; }
    if (nonzero > 2) { if (p2 == 0) x[0] = y2; else if (p2 == 1) x[1] = y2; else x[2] = y2; }
  }
  static PVLM_HD bool form_plane_accept(const double* xs, const double* px, const double* py, const double* pz, double tol, double* plane) {
    double x[3] = {xs[0], xs[1], xs[2]};
    const double len = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const double d = 1.0 / len;
    if (len * len > 0.0) { x[0] /= len; x[1] /= len; x[2] /= len; }
    bool ok = true;
    if (tol > 0) {
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const double dist = fabs((x[0] * px[i] + x[1] * py[i]) + x[2] * pz[i] + d);
        if (dist > tol) ok = false;
      }
    }
    plane[0] = ok ? x[0] : 0.0; plane[1] = ok ? x[1] : 0.0; plane[2] = ok ? x[2] : 0.0; plane[3] = ok ? d : 0.0;
    return ok;
  }

  // ---- certified fast fit ---------------------------------------------------------------------------------------------------------------
  // form_plane() above restates Eigen's pivoted Householder QR operation by operation (~1 300 unfused fp64 instructions, 40 live doubles): it
  // is what makes a record bit-identical to the reference's.  What the reference DECIDES with it is coarser: accept iff every point's distance
  // |n.p_i + d| <= tol.  form_plane_fast() solves the same least-squares problem min ||A x + 1|| through the 3x3 normal equations (fused
  // multiply-adds, Cramer + one refinement step), bounds how far ITS solution x and the QR's x_qr can lie apart, and answers only when the
  // accept / reject decision is the same for every solution inside that bound:
  //    1  accept (plane filled: unit normal, d), 0 reject, -1 undecided — the caller then runs form_plane(), whose answer stands.
  //
  // The bound E >= ||x - x_qr|| = E_fast + E_qr, both distances to x*, the exact minimiser for the given doubles (u = 2^-53, tr = trace A^T A,
  // P = sqrt(tr) >= ||A||_2 >= max ||p_i||, r_i = x.p_i + 1):
  //   * E_fast.  For any y, ||y - x*|| <= ||A^T (A y + 1)|| / lambda_min(A^T A).  rho = sum_i p_i r_i is evaluated with fused chains: each r_i
  //     rounds by at most 3 u (||x|| ||p_i|| + 1) (Higham, Accuracy and Stability, ch. 3) and the ten-term sums by another g10 |p_ij| |r_i| each, so
  //     ||rho_computed - rho|| <= 40 u P (P ||x|| + 1).                                  E_fast = (||rho|| + 40 u P (P ||x|| + 1)) / lambda_min.
  //   * E_qr.  Column-pivoted Householder least squares is backward stable (ibid. Theorem 20.3): x_qr minimises ||(A + dA) y + 1 + db|| with
  //     ||dA_j|| <= G ||a_j||, ||db|| <= G sqrt(10), G = c 30 u for a small integer c (c = 16 taken: G = 480 u), hence ||dA||_2 <= G P and, with
  //     ||A||_2 >= P / sqrt(3), a relative perturbation eps <= sqrt(3) G < 1024 u of A and of the right-hand side.  Wedin's theorem (ibid. Theorem
  //     20.1), k = kappa_2(A) <= P / sqrt(lambda_min), k eps <= 1e-2:
  //         ||x_qr - x*|| <= k eps / (1 - k eps) (2 ||x*|| + (k + 1) ||r*|| / ||A||_2),   ||r*|| <= ||(r_i)|| (x* minimises the residual).
  //   * lambda_min(A^T A) >= det / e2 for a symmetric positive definite 3x3 matrix, e2 = the sum of its principal 2x2 minors (e2 >= lambda_max
  //     lambda_mid).  The computed cofactors give det and e2 with a relative error below 10 u tr^3 / det, and the sums M differ from A^T A by at
  //     most g10 tr in norm; the routine refuses (-1) unless tr^3 <= 5e13 det — both effects are then below 6e-2 — and uses HALF of det / e2.
  //   Distances: f(y) = max_i |y.p_i + 1| / ||y||.  With e = E / ||x|| (<= 2.5e-7 or the routine refuses: the stored record stays within 5e-7 relative of the QR's):  |f(x_qr) - f(x)| <= (4/3) e (P + f(x)), and the two evaluations
  //   themselves round by less than 16 u (P + 1/||x||) together (the reference normalises x by three divisions and adds d = 1 / len: 8 u (P + d);
  //   the fused chain here: 3 u).
  // tests/test_assoc_core_cpu.py drives this against form_plane() on 2 10^6 neighbourhoods (near and far, random, bisected onto the threshold,
  // degenerate) and against a __float128 solve: no decided case may differ, ||x - x_qr|| must stay below E and each distance below its share.
  // The same routine in STREAMING form, for a caller that does not keep the ten points (the fast kernel gathers them twice instead — 60 registers less, twice the
  // resident waves): add() every point, solve(), residual() every point again in the same order, decide().  collinear() answers the reference's collinearity test
  // (is_line) from the same moments through line_screen(): the scatter matrix S = M - s s^T / 10 derived from the sums differs from the one the reference
  // accumulates about the centroid by less than 8 u tr entrywise (the cancellation is against tr, the sums' size), handed to the screen as `slack`; -1 = undecided.
  struct FastFit {
    double m00 = 0.0, m01 = 0.0, m02 = 0.0, m11 = 0.0, m12 = 0.0, m22 = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
    double x0, x1, x2, tr, e2, inv, nn;
    double fmx = 0.0, R2 = 0.0, h0 = 0.0, h1 = 0.0, h2 = 0.0;
    PVLM_HD void add(double x, double y, double z) {
      m00 = fma(x, x, m00); m01 = fma(x, y, m01); m02 = fma(x, z, m02);
      m11 = fma(y, y, m11); m12 = fma(y, z, m12); m22 = fma(z, z, m22);
      s0 += x; s1 += y; s2 += z;
    }
    PVLM_HD int collinear(double tol) const {
      const double U = 1.1102230246251565e-16;
      const double a00 = fma(-0.1 * s0, s0, m00), a01 = fma(-0.1 * s0, s1, m01), a02 = fma(-0.1 * s0, s2, m02);
      const double a11 = fma(-0.1 * s1, s1, m11), a12 = fma(-0.1 * s1, s2, m12), a22 = fma(-0.1 * s2, s2, m22);
      return line_screen(a00, a01, a02, a11, a12, a22, tol, nullptr, (8.0 * U) * ((m00 + m11) + m22));
    }
    PVLM_HD bool solve(double tol) {
      const double c00 = fma(m11, m22, -(m12 * m12)), c01 = fma(m12, m02, -(m01 * m22)), c02 = fma(m01, m12, -(m11 * m02));
      const double c11 = fma(m00, m22, -(m02 * m02)), c12 = fma(m01, m02, -(m00 * m12)), c22 = fma(m00, m11, -(m01 * m01));
      const double det = fma(m00, c00, fma(m01, c01, m02 * c02));
      tr = (m00 + m11) + m22; e2 = (c00 + c11) + c22;
      if (!(tol > 0.0 && det > 0.0 && e2 > 0.0 && tr < 1e100 && (tr * tr) * tr <= 5e13 * det)) return false;   // also NaN / inf
      inv = 1.0 / det;
      x0 = -(fma(c00, s0, fma(c01, s1, c02 * s2))) * inv; x1 = -(fma(c01, s0, fma(c11, s1, c12 * s2))) * inv; x2 = -(fma(c02, s0, fma(c12, s1, c22 * s2))) * inv;
      {   // one refinement step on the computed system
        const double r0 = fma(m00, x0, fma(m01, x1, fma(m02, x2, s0))), r1 = fma(m01, x0, fma(m11, x1, fma(m12, x2, s1))), r2 = fma(m02, x0, fma(m12, x1, fma(m22, x2, s2)));
        x0 = fma(-(fma(c00, r0, fma(c01, r1, c02 * r2))), inv, x0); x1 = fma(-(fma(c01, r0, fma(c11, r1, c12 * r2))), inv, x1); x2 = fma(-(fma(c02, r0, fma(c12, r1, c22 * r2))), inv, x2);
      }
      nn = fma(x0, x0, fma(x1, x1, x2 * x2));
      return nn > 1e-200 && nn < 1e200;
    }
    PVLM_HD void residual(double x, double y, double z) {      // r_i, their largest magnitude, their sum of squares and rho = A^T r
      const double r = fma(x0, x, fma(x1, y, fma(x2, z, 1.0)));
      fmx = fmax(fmx, fabs(r)); R2 = fma(r, r, R2);
      h0 = fma(x, r, h0); h1 = fma(y, r, h1); h2 = fma(z, r, h2);
    }
    PVLM_HD int decide(double tol, double* plane, double* diag = nullptr) const {
      const double U = 1.1102230246251565e-16;
      const double len = sqrt(nn), rn = 1.0 / len, P = sqrt(tr);
      const double L = (2.0 * e2) * inv;                                        // >= 1 / lambda_min
      const double k = P * sqrt(L), keps = k * (1024.0 * U);
      if (!(keps <= 1e-2)) return -1;
      const double E_fast = (sqrt(fma(h0, h0, fma(h1, h1, h2 * h2))) + (40.0 * U) * P * fma(P, len, 1.0)) * L;
      const double E_qr = (1.0102 * keps) * fma(2.02, len, (k + 1.0) * sqrt(3.0 * R2 / tr) * 1.0001);
      const double E = E_fast + E_qr;
      const double e = E * rn;
      if (!(e <= 2.5e-7)) return -1;                                            // the accepted RECORD (x / ||x||, 1 / ||x||) is then within 5e-7 relative of the QR's (the bar is 1e-6); also keeps ||x*|| <= 1.01 ||x|| as E_qr assumed
      const double f = fmx * rn;
      const double B = (4.0 / 3.0) * e * (P + f) + (16.0 * U) * (P + rn);
      if (diag) { diag[0] = x0; diag[1] = x1; diag[2] = x2; diag[3] = E; diag[4] = f; diag[5] = B; diag[6] = E_fast; diag[7] = E_qr; }
      if (f + B <= tol) { plane[0] = x0 * rn; plane[1] = x1 * rn; plane[2] = x2 * rn; plane[3] = rn; return 1; }
      if (f - B > tol) { plane[0] = 0.0; plane[1] = 0.0; plane[2] = 0.0; plane[3] = 0.0; return 0; }
      return -1;
    }
  };
  static PVLM_HD int form_plane_fast(const double* px, const double* py, const double* pz, double tol, double* plane, double* diag = nullptr) {
    FastFit F;
#pragma unroll
    for (int i = 0; i < 10; ++i) F.add(px[i], py[i], pz[i]);
    if (!F.solve(tol)) return -1;
#pragma unroll
    for (int i = 0; i < 10; ++i) F.residual(px[i], py[i], pz[i]);
    return F.decide(tol, plane, diag);
  }
  // the collinearity decision from the moments alone (streaming form): 1 line, 0 no line, -1 undecided — the array form for the CPU tests
  static PVLM_HD int is_line_fast(const double* px, const double* py, const double* pz, double tol) {
    FastFit F;
#pragma unroll
    for (int i = 0; i < 10; ++i) F.add(px[i], py[i], pz[i]);
    return F.collinear(tol);
  }

  // Closed-form screen of the collinearity decision.  The eigenvalues of the symmetric 3x3 scatter matrix by the trigonometric
  // solution of its characteristic cubic (Smith 1961): q = tr/3, p^2 = |A - qI|_F^2 / 6, r = det(A - qI) / (2 p^3),
  // phi = acos(r) / 3 in [0, pi/3],  l3 = q + 2 p cos(phi),  l1 = q + 2 p cos(phi + 2 pi/3),  l2 = 3 q - l1 - l3.
  // acos by the degree-7 minimax form sqrt(1 - x) P(x) (Abramowitz & Stegun 4.4.46, |error| <= 2e-8), cos / sin of phi by
  // their Taylor polynomials up to x^12 / x^13 (remainder < 3e-11 on [0, pi/3]); the rounding of r is amplified by acos near
  // |r| = 1 to at most sqrt(2 * 1e-15) / 3 = 1.5e-8 in phi.  Every eigenvalue is therefore within 2 p (3e-8) <= 6e-8 l3 of
  // the exact one (measured against LAPACK over 10^6 matrices in tests/test_assoc_core_cpu.py: <= 2e-8 l3), while the converged
  // Jacobi loop of the reference restatement is within 1e-13.  The screen answers only when l3 - tol l2 clears a guard of
  // 1e-5 (l3 + tol |l2|) — 160 times its error — and returns -1 otherwise: then, and only then, the exact loop runs.
  // `slack`: an absolute bound on how far the eigenvalues of the matrix handed in may lie from those of the matrix the reference forms (0 when it IS that matrix;
  // the streaming fit below hands in the scatter matrix derived from the raw second moments): added to the guard on both sides.
  static PVLM_HD int line_screen(double a00, double a01, double a02, double a11, double a12, double a22, double tol, double* eig = nullptr, double slack = 0.0) {
    const double q = ((a00 + a11) + a22) * (1.0 / 3.0);
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p1 = (a01 * a01 + a02 * a02) + a12 * a12;
    const double p2 = ((b00 * b00 + b11 * b11) + b22 * b22) + 2.0 * p1;
    if (!(p2 > 1e-24 * (q * q) && p2 > 1e-280 && p2 < 1e280)) return -1;   // (nearly) a multiple of the identity, or not finite
    const double p = sqrt(p2 * (1.0 / 6.0));
    const double det = (b00 * (b11 * b22 - a12 * a12) - a01 * (a01 * b22 - a12 * a02)) + a02 * (a01 * a12 - b11 * a02);
    double r = det / (2.0 * ((p * p) * p));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double x = fabs(r);
    double P = -0.0012624911;
    P = P * x + 0.0066700901; P = P * x - 0.0170881256; P = P * x + 0.0308918810; P = P * x - 0.0501743046;
    P = P * x + 0.0889789874; P = P * x - 0.2145988016; P = P * x + 1.5707963050;
    double ac = sqrt(1.0 - x) * P;                                  // acos(|r|)
    if (r < 0.0) ac = 3.14159265358979323846 - ac;
    const double phi = ac * (1.0 / 3.0), z = phi * phi;
    double c = 1.0 / 479001600.0;
    c = c * z - 1.0 / 3628800.0; c = c * z + 1.0 / 40320.0; c = c * z - 1.0 / 720.0; c = c * z + 1.0 / 24.0; c = c * z - 0.5; c = c * z + 1.0;
    double sn = 1.0 / 6227020800.0;
    sn = sn * z - 1.0 / 39916800.0; sn = sn * z + 1.0 / 362880.0; sn = sn * z - 1.0 / 5040.0; sn = sn * z + 1.0 / 120.0; sn = sn * z - 1.0 / 6.0; sn = sn * z + 1.0;
    sn = sn * phi;
    const double l3 = q + 2.0 * p * c;
    const double l1 = q + 2.0 * p * (-0.5 * c - 0.86602540378443864676 * sn);
    const double l2 = (3.0 * q - l1) - l3;
    if (eig) { eig[0] = l1; eig[1] = l2; eig[2] = l3; }
    const double margin = l3 - tol * l2, guard = 1e-5 * (fabs(l3) + fabs(tol * l2)) + slack * (1.0 + fabs(tol));
    if (margin > guard) return 1;
    if (margin < -guard) return 0;
    return -1;
  }

  // FormLine(points, 3.0) is non-zero  <=>  largest eigenvalue > tol * middle eigenvalue of the
  // scatter matrix (base/Geometry.hpp:220-260); cyclic Jacobi, fixed sweep order (0,1),(0,2),(1,2) — the oracle's
  // eig_sym3_jacobi, which sweeps until the off-diagonal part is EXACTLY zero (8-10 sweeps of three rotations with two
  // divisions and two square roots each: the larger half of round 2's K3).  Only the DECISION w2 > tol * w1 of the
  // converged sweep is needed: line_screen() above gives it for all but a ~1e-5 band around the threshold; inside the band
  // the loop below runs, itself with a certified exit:
  //
  // Certified early exit.  Before each sweep the
  // current diagonal d (sorted) and the off-diagonal Frobenius norm E = sqrt(2 (a01^2 + a02^2 + a12^2)) bracket the exact
  // eigenvalues of the current matrix: |lambda_i - d_i| <= E (Weyl).  The remaining sweeps are orthogonal similarity
  // transforms carried out in fp64, so what the converged loop returns differs from lambda_i by no more than its
  // accumulated rounding, < 1e-13 |lambda|_max for at most 36 rotations; `guard` below is 1e-11 of the largest diagonal
  // entry.  When the brackets already decide the comparison the loop stops — the answer is the one the full loop would
  // give — otherwise it sweeps on, down to the oracle's own termination.
  static PVLM_HD bool is_line(const double* px, const double* py, const double* pz, double tol, int* sweeps_done = nullptr, bool screen = true) {
    double cx = 0.0, cy = 0.0, cz = 0.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) { cx = cx + px[i]; cy = cy + py[i]; cz = cz + pz[i]; }
    cx = cx / 10.0; cy = cy / 10.0; cz = cz / 10.0;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const double dx = px[i] - cx, dy = py[i] - cy, dz = pz[i] - cz;
      a00 = a00 + dx * dx; a01 = a01 + dx * dy; a02 = a02 + dx * dz;
      a11 = a11 + dy * dy; a12 = a12 + dy * dz; a22 = a22 + dz * dz;
    }
    // only the upper triangle is accumulated: S[r][c] += d[r]*d[c] is symmetric bit for bit
    if (screen) {
      const int fast = line_screen(a00, a01, a02, a11, a12, a22, tol);
      if (fast >= 0) { if (sweeps_done) *sweeps_done = 0; return fast != 0; }
    }
    for (int sweep = 0; sweep < 12; ++sweep) {
      const double off = a01 * a01 + a02 * a02 + a12 * a12;
      if (off == 0.0) break;
      if (sweeps_done) *sweeps_done = sweep;
      {
        double w0 = a00, w1 = a11, w2 = a22, t;
        if (w0 > w1) { t = w0; w0 = w1; w1 = t; }
        if (w1 > w2) { t = w1; w1 = w2; w2 = t; }
        if (w0 > w1) { t = w0; w0 = w1; w1 = t; }
        const double amax = fmax(fabs(w2), fabs(w0));
        const double E = sqrt(2.0 * off) * (1.0 + 1e-9) + 1e-11 * amax;   // Weyl radius + rounding of the remaining sweeps
        // (w2 - E) > tol (w1 + E)  =>  certainly a line;   (w2 + E) < tol (w1 - E)  =>  certainly not
        const double guard = 1e-11 * amax * (1.0 + fabs(tol));
        if (E == E && amax < 1e300) {    // finite
          if ((w2 - E) - tol * (w1 + E) > guard) return true;
          if (tol * (w1 - E) - (w2 + E) > guard) return false;
        }
      }
      // (p,q) = (0,1), r = 2
      if (a01 != 0.0) {
        const double theta = (a11 - a00) / (2.0 * a01);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const double app = a00, aqq = a11, apq = a01;
        a00 = app - t * apq; a11 = aqq + t * apq; a01 = 0.0;
        const double arp = a02, arq = a12;
        a02 = c * arp - s * arq; a12 = s * arp + c * arq;
      }
      // (p,q) = (0,2), r = 1
      if (a02 != 0.0) {
        const double theta = (a22 - a00) / (2.0 * a02);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const double app = a00, aqq = a22, apq = a02;
        a00 = app - t * apq; a22 = aqq + t * apq; a02 = 0.0;
        const double arp = a01, arq = a12;
        a01 = c * arp - s * arq; a12 = s * arp + c * arq;
      }
      // (p,q) = (1,2), r = 0
      if (a12 != 0.0) {
        const double theta = (a22 - a11) / (2.0 * a12);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const double app = a11, aqq = a22, apq = a12;
        a11 = app - t * apq; a22 = aqq + t * apq; a12 = 0.0;
        const double arp = a01, arq = a02;
        a01 = c * arp - s * arq; a02 = s * arp + c * arq;
      }
      if (sweeps_done) *sweeps_done = sweep + 1;
    }
    // ascending sort of (a00, a11, a22)
    double w0 = a00, w1 = a11, w2 = a22, t;
    if (w0 > w1) { t = w0; w0 = w1; w1 = t; }
    if (w1 > w2) { t = w1; w1 = w2; w2 = t; }
    if (w0 > w1) { t = w0; w0 = w1; w1 = t; }
    return w2 > tol * w1;
  }
};

// World2Local: R_wl^T p - R_wl^T t  (sensors/Velodyne.cpp:1850-1853), R row-major
PVLM_HD void world2local(const double* R, const double* t, double x, double y, double z, double* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a = (R[i] * x + R[3 + i] * y) + R[6 + i] * z;
    const double b = (R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2];
    o[i] = a - b;
  }
}

// the same with the translation term R_wl^T t (identical for every point of a cloud) taken once: Rt[i] = (R[i] t0 + R[3+i] t1) + R[6+i] t2
PVLM_HD void world2local_rt(const double* R, const double* t, double* Rt) {
#pragma unroll
  for (int i = 0; i < 3; ++i) Rt[i] = (R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2];
}
PVLM_HD void world2local_pt(const double* R, const double* Rt, double x, double y, double z, double* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = ((R[i] * x + R[3 + i] * y) + R[6 + i] * z) - Rt[i];
}

}  // namespace pvlm_assoc
