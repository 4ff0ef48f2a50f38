// K16-K23 — the range-image stages of the LiDAR feature extractor for a BATCH of raw scans:
//   Velodyne::ReOrderVLP        sensors/Velodyne.cpp:371-526    firing order -> ring order, range image, (ring, column) of every point
//   Velodyne::Segmentation      sensors/Velodyne.cpp:1438-1586  range-image labelling, small components removed
//   adaptive-window curvature   sensors/Velodyne.cpp:623-657    (ExtractFeatures, method ADAPTIVE)
// One launch per stage covers every scan of the batch (454 scans of Room, 1593 of Floor); everything between the raw
// points and the per-point curvature stays in HBM.  The sort-dependent picks (ExtractEdgeFeatures2 / ExtractPlaneFeatures2,
// pcl::VoxelGrid) stay on the host (host/pvlm_features.cpp) and read the arrays this file downloads.
// Compiled with -ffp-contract=off: every float expression below is the reference's, operation for operation.
//
// libm decisions.  Three decisions of the reference go through FLOAT libm calls of the host it was built on
// (`using namespace std`, sensors/Velodyne.cpp:7): atan (elevation -> ring), atan2 (azimuth -> column, the +z crossing)
// and atan2 again (the segmentation angle).  Their last bit belongs to that host's libm, not to IEEE-754.  The kernels
// therefore compute the TRUE value in fp64, take the interval of floats within kUlps of it — every libm result lies
// inside (glibc documents <= 2 ulp for atanf / atan2f) — and carry the DECISION (ring, column, joined / not joined) through
// the reference's own arithmetic for the interval: when both ends agree the decision is certified, whatever the libm.
// The few points or edges whose interval straddles a decision boundary (~1e-4 of the points, ~0 edges) are listed, and the
// entry point asks the host's own libm for exactly those (std::atan2 / std::atan on the same floats, what a reference
// build on this machine would call) before the dependent stage runs.  Comparisons between two azimuths (the +z-crossing
// test, :447-461) are interval comparisons; a scan whose crossing cannot be certified gets exact azimuths for all its
// points from the host and is replayed.  No decision is ever taken from an uncertified device value.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "pvlm_internal.h"
#include "pvlm_workers.h"
#include "pvlm_ring_core.h"
#include "pvlm_ring_picks.h"

using namespace pvlm_ring;

namespace {

struct PtBlock { int scan, first; };   // 256 points of one scan

// ---- K16: per raw point — ring and azimuth with their certificates -----------------------------------------------------
// rec = (the float nearest to the true atan2(x, z), the ring every float within kUlps of the true atan() gives), or the point is listed;
// likewise the column (for both states of the +z crossing).  Stored in the chunk-transposed order K17 walks (chunk_slot).
__global__ __launch_bounds__(256) void k_ring_classify(const RingScan* __restrict__ scans, const PtBlock* __restrict__ blocks, int rings, int horizon,
                                                       const float4* __restrict__ raw, PointRec* __restrict__ rec, int* __restrict__ n_listed,
                                                       int* __restrict__ listed) {
  const PtBlock b = blocks[blockIdx.x];
  const RingScan sc = scans[b.scan];
  const int i = b.first + threadIdx.x;
  if (i >= sc.n) return;
  const float4 p = raw[sc.pt0 + i];
  float f; int r;
  const bool list = classify_point(sc, rings, horizon, p.x, p.y, p.z, &f, &r);
  rec[sc.slot0 + chunk_slot(i, chunk_of(sc.n, kColumnThreads), kColumnThreads)] = make_rec(f, r, false);
  if (list) listed[atomicAdd(n_listed, 1)] = (int)(sc.pt0 + i);
}

struct PointPatch { long long slot; float az; int ring; };
__global__ void k_ring_patch(int n, const PointPatch* __restrict__ patch, PointRec* __restrict__ rec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const PointPatch p = patch[k];
  rec[p.slot] = make_rec(p.az, p.ring, true);
}

// ---- K17: the column state machine of :431-507, one workgroup per scan ----------------------------------------------------------
// columns_block (pvlm_ring_core.h): the loop's five carried scalars decompose into block-wide prefix scans (offset functions,
// last accepted point, per-ring counts); with one lane per scan — the loop replayed as written — a Room batch took 48 ms.
// colpos[i] = (column or -1, position of the point inside its ring).  status: (-1, .) = done, (i, last) = the +z crossing could not be
// certified at point i against the azimuth of point `last`.
struct BlockExec {
  __device__ int threads() const { return (int)blockDim.x; }
  template <class F> __device__ void phase(F&& f) { f((int)threadIdx.x); __syncthreads(); }
};
__global__ __launch_bounds__(1024) void k_ring_columns(const RingScan* __restrict__ scans, const int* __restrict__ todo, int rings, int horizon,
                                                       const PointRec* __restrict__ rec, int2* __restrict__ colpos, int* __restrict__ ring_count,
                                                       int2* __restrict__ status) {
  __shared__ int scratch[kColumnsScratch * 1024];     // 88 KB: one workgroup per CU
  const int s = todo ? todo[blockIdx.x] : (int)blockIdx.x;
  const RingScan sc = scans[s];
  BlockExec ex;
  int last = -1;
  const int stuck = columns_block(ex, sc, rings, horizon, rec + sc.slot0, reinterpret_cast<int*>(colpos + sc.slot0), ring_count + (size_t)s * kMaxRings, &last, scratch);
  if (threadIdx.x == 0) status[s] = make_int2(stuck, last);
}

// ---- K18: ring-ordered cloud, (ring, column) of every point, range image ---------------------------------------------------
// Several returns can land in one cell; the reference's loops leave the LAST one (in cloud order) in range_image and in
// image_to_point_idx (:497-499, :513-517): atomicMax over the raw index picks it, the second pass writes it.
__global__ __launch_bounds__(256) void k_ring_scatter(const RingScan* __restrict__ scans, const PtBlock* __restrict__ blocks, int horizon,
                                                      const float4* __restrict__ raw, const PointRec* __restrict__ rec, const int2* __restrict__ colpos,
                                                      const int* __restrict__ ring_count, float4* __restrict__ cloud_scan, int* __restrict__ source,
                                                      int2* __restrict__ rc, int* __restrict__ winner) {
  __shared__ int begin[kMaxRings];
  const PtBlock b = blocks[blockIdx.x];
  const RingScan sc = scans[b.scan];
  if (threadIdx.x == 0) { int run = 0; for (int r = 0; r < kMaxRings; ++r) { begin[r] = run; run += ring_count[(size_t)b.scan * kMaxRings + r]; } }
  __syncthreads();
  const int i = b.first + threadIdx.x;
  if (i >= sc.n) return;
  const long long at = sc.slot0 + chunk_slot(i, chunk_of(sc.n, kColumnThreads), kColumnThreads);
  const int2 cp = colpos[at];
  if (cp.x < 0) return;
  const int r = rec_ring(rec[at]);
  const int dst = begin[r] + cp.y;
  const float4 p = raw[sc.pt0 + i];
  cloud_scan[sc.pt0 + dst] = make_float4(p.x, p.y, p.z, (float)r);
  source[sc.pt0 + dst] = i;
  rc[sc.pt0 + dst] = make_int2(r, cp.x);
  atomicMax(&winner[sc.cell0 + (long long)r * horizon + cp.x], i);
}
__global__ __launch_bounds__(256) void k_ring_cells(const RingScan* __restrict__ scans, const PtBlock* __restrict__ blocks, int horizon,
                                                    const float4* __restrict__ raw, const PointRec* __restrict__ rec, const int2* __restrict__ colpos,
                                                    const int* __restrict__ ring_count, const int* __restrict__ winner, float* __restrict__ range_image,
                                                    int* __restrict__ image_to_point) {
  __shared__ int begin[kMaxRings];
  const PtBlock b = blocks[blockIdx.x];
  const RingScan sc = scans[b.scan];
  if (threadIdx.x == 0) { int run = 0; for (int r = 0; r < kMaxRings; ++r) { begin[r] = run; run += ring_count[(size_t)b.scan * kMaxRings + r]; } }
  __syncthreads();
  const int i = b.first + threadIdx.x;
  if (i >= sc.n) return;
  const long long at = sc.slot0 + chunk_slot(i, chunk_of(sc.n, kColumnThreads), kColumnThreads);
  const int2 cp = colpos[at];
  if (cp.x < 0) return;
  const int r = rec_ring(rec[at]);
  const long long cell = sc.cell0 + (long long)r * horizon + cp.x;
  if (winner[cell] != i) return;
  const float4 p = raw[sc.pt0 + i];
  range_image[cell] = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
  image_to_point[cell] = begin[r] + cp.y;
}

// ---- K19: the "joined" relation between 4-neighbours of the range image (:1500-1512), certified ------------------------------
struct EdgeQuery { long long cell; int bit; float y, x; };   // undecided edge: the host evaluates atan2f(y, x) > theta
__global__ __launch_bounds__(256) void k_seg_edges(const RingScan* __restrict__ scans, int rings, int horizon, const float* __restrict__ range_image,
                                                   float sin_x, float cos_x, float sin_y, float cos_y, float theta, unsigned char* __restrict__ edges,
                                                   int* __restrict__ n_queries, EdgeQuery* __restrict__ queries, int query_cap) {
  const RingScan sc = scans[blockIdx.y];
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= rings * horizon) return;
  const int r = cell / horizon, c = cell - r * horizon;
  const float* R = range_image + sc.cell0;
  const float here = R[cell];
  unsigned char bits = 0;
  auto test = [&](int other, int bit, float s, float cs) {
    float y = 0.f, x = 0.f;
    const int j = joined_certified(here, R[other], s, cs, theta, &y, &x);
    if (j > 0) bits |= (unsigned char)bit;
    else if (j < 0) {
      const int k = atomicAdd(n_queries, 1);
      if (k < query_cap) queries[k] = EdgeQuery{sc.cell0 + cell, bit, y, x};
    }
  };
  const int right = r * horizon + (c + 1 == horizon ? 0 : c + 1);
  if (right != cell) test(right, 1, sin_x, cos_x);
  if (r + 1 < rings) test(cell + horizon, 2, sin_y, cos_y);
  edges[sc.cell0 + cell] = bits;
}
struct EdgePatch { long long cell; int bit; };
__global__ void k_seg_patch(int n, const EdgePatch* __restrict__ patch, unsigned char* __restrict__ edges) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) atomicOr((unsigned int*)(edges + (patch[k].cell & ~3ll)), (unsigned)patch[k].bit << (8 * (int)(patch[k].cell & 3)));
}

// ---- K20: connected components of the joined relation = the BFS labels of :1463-1530 ------------------------------------------
// The horizontal part of the relation is a set of RUNS along a ring: k_seg_init finds the start of every cell's run with one
// max-scan per row and makes it the cell's parent (a tree of depth 1), so that the lock-free union-find in HBM only has the
// vertical edges and the wrap-around edge of each row left (hook the larger root under the smaller by atomicMin: the root of a
// component is its smallest cell = the seed the reference's raster-order BFS starts from).  Unioning the horizontal edges cell by
// cell instead built chains as long as a row before any find could shorten them (5.3 ms per Room batch).  Then per component:
// size and the rows holding a cell other than the seed (lineCountFlag marks pushed cells only; the seed is popped, never pushed,
// :1473-1476, :1518), with one atomic per run of equal roots inside a wave instead of one per cell.
__device__ inline int uf_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int uf_find(int* parent, int x) {
  int r = x;
  for (int q = uf_load(parent + r); q != r;) {
    const int g = uf_load(parent + q);
    if (g != q) __hip_atomic_store(parent + r, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving: an ancestor stays an ancestor
    r = q; q = g;
  }
  return r;
}
__device__ inline void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(parent + a, b);
    if (old == a) return;
    a = old;
  }
}
// grid (rings, scans): parent = start of the cell's run inside the row (runs do not wrap here), the per-cell accumulators cleared
__global__ __launch_bounds__(256) void k_seg_init(const RingScan* __restrict__ scans, int horizon, const unsigned char* __restrict__ edges, int* __restrict__ parent,
                                                  int* __restrict__ comp_size, unsigned long long* __restrict__ row_mask, int* __restrict__ image_to_point2) {
  __shared__ int part[256];
  __shared__ int carry_s;
  const RingScan sc = scans[blockIdx.y];
  const int r = blockIdx.x, t = threadIdx.x;
  const long long row0 = sc.cell0 + (long long)r * horizon;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < horizon; base += 256) {
    const int c = base + t;
    // a run starts at column 0 and wherever the cell on the left is not joined to this one
    int v = (c < horizon && (c == 0 || !(edges[row0 + c - 1] & 1))) ? c : 0;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int x = t >= off ? part[t - off] : 0;
      __syncthreads();
      part[t] = max(part[t], x);
      __syncthreads();
    }
    const int start = max(part[t], carry_s);
    if (c < horizon) {
      parent[row0 + c] = r * horizon + start; comp_size[row0 + c] = 0; row_mask[row0 + c] = 0ull; image_to_point2[row0 + c] = -1;
    }
    __syncthreads();
    if (t == 255) carry_s = start;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_seg_union(const RingScan* __restrict__ scans, int rings, int horizon, const unsigned char* __restrict__ edges,
                                                   int* __restrict__ parent) {
  const RingScan sc = scans[blockIdx.y];
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= rings * horizon) return;
  const unsigned char bits = edges[sc.cell0 + cell];
  if (!bits) return;
  const int r = cell / horizon, c = cell - r * horizon;
  int* P = parent + sc.cell0;
  if ((bits & 1) && c + 1 == horizon) uf_union(P, cell, r * horizon);      // the one horizontal edge the runs do not cover: last column -> first
  if (bits & 2) uf_union(P, cell, cell + horizon);
}
__global__ __launch_bounds__(256) void k_seg_stats(const RingScan* __restrict__ scans, int rings, int horizon, int* __restrict__ parent,
                                                   int* __restrict__ root_of, int* __restrict__ comp_size, unsigned long long* __restrict__ row_mask) {
  const RingScan sc = scans[blockIdx.y];
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const bool live = cell < rings * horizon;
  int root = -1 - (int)threadIdx.x, row = 0;          // dead lanes: distinct negative values, never equal to a neighbour's root
  if (live) { root = uf_find(parent + sc.cell0, cell); root_of[sc.cell0 + cell] = root; row = cell / horizon; }
  // consecutive lanes are consecutive cells: one atomicAdd per run of equal roots, one atomicOr per (run, row) of non-seed cells
  const int lane = threadIdx.x & 63;
  const int prev_root = __shfl_up(root, 1, 64), prev_row = __shfl_up(row, 1, 64);
  const bool head = lane == 0 || prev_root != root;
  const unsigned long long heads = __ballot(head);
  if (live && head) {
    const unsigned long long later = lane == 63 ? 0ull : heads >> (lane + 1);
    const int len = later ? __builtin_ctzll(later) + 1 : 64 - lane;
    atomicAdd(comp_size + sc.cell0 + root, len);
  }
  const bool counts = live && cell != root;
  const int prev_counts = __shfl_up(counts ? 1 : 0, 1, 64);
  if (counts && (head || prev_row != row || !prev_counts)) atomicOr(row_mask + sc.cell0 + root, 1ull << row);
}

// ---- K21: drop the points of rejected components, ring order kept (:1547-1580); one workgroup per scan -----------------------
// segment == 0: nothing is dropped (the curvature stage reads the same arrays either way).
__global__ __launch_bounds__(1024) void k_seg_compact(const RingScan* __restrict__ scans, int rings, int horizon, int segment, const int* __restrict__ ring_count,
                                                      const float4* __restrict__ cloud_scan, const int* __restrict__ source, const int2* __restrict__ rc,
                                                      const float* __restrict__ range_image, const int* __restrict__ root_of, const int* __restrict__ comp_size,
                                                      const unsigned long long* __restrict__ row_mask, float4* __restrict__ cloud2, int* __restrict__ source2,
                                                      int* __restrict__ ring_col2, float* __restrict__ range2, int* __restrict__ image_to_point2,
                                                      int* __restrict__ ring_count2, int* __restrict__ counts) {
  __shared__ int part[1024];
  __shared__ int carry_s;
  __shared__ int kept_ring[kMaxRings];
  const int s = blockIdx.x, t = threadIdx.x;
  const RingScan sc = scans[s];
  int n = 0;
  for (int r = 0; r < rings; ++r) n += ring_count[(size_t)s * kMaxRings + r];
  if (t == 0) carry_s = 0;
  if (t < kMaxRings) kept_ring[t] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 4096) {
    int keep[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + 4 * t + k;
      keep[k] = 0;
      if (i < n) {
        keep[k] = 1;
        if (segment) {
          const int2 q = rc[sc.pt0 + i];
          const long long root = sc.cell0 + root_of[sc.cell0 + (long long)q.x * horizon + q.y];
          const int size = comp_size[root];
          keep[k] = keep_component(size, __popcll(row_mask[root])) ? 1 : 0;
        }
      }
      sum += keep[k];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int x = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += x;
      __syncthreads();
    }
    int run = carry_s + part[t] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + 4 * t + k;
      if (i < n && keep[k]) {
        const float4 p = cloud_scan[sc.pt0 + i];
        const int2 q = rc[sc.pt0 + i];
        const long long cell = sc.cell0 + (long long)q.x * horizon + q.y;
        cloud2[sc.pt0 + run] = p;
        source2[sc.pt0 + run] = source[sc.pt0 + i];
        ring_col2[sc.pt0 + run] = (q.x << 16) | q.y;
        range2[sc.pt0 + run] = range_image[cell];
        atomicMax(image_to_point2 + cell, run);                     // several points of one cell: the last one stays (:1556)
        atomicAdd(&kept_ring[q.x], 1);
        ++run;
      }
    }
    __syncthreads();
    if (t == 1023) carry_s += part[1023];
    __syncthreads();
  }
  if (t < kMaxRings) ring_count2[(size_t)s * kMaxRings + t] = t < rings ? kept_ring[t] : 0;
  if (t == 0) { counts[2 * s] = n; counts[2 * s + 1] = carry_s; }
}

// ---- K22: adaptive-window curvature (:623-657), one thread per kept point ------------------------------------------------------
// Kept as upstream, including the right-hand walk guarded by the LEFT index and the window test that looks at the left end
// twice; where upstream would read past the end of the cloud (undefined behaviour) the walk stops and the point has no curvature.
__global__ __launch_bounds__(256) void k_curvature(const RingScan* __restrict__ scans, const PtBlock* __restrict__ blocks, const int* __restrict__ ring_count2,
                                                   const int* __restrict__ counts, const float4* __restrict__ cloud2, const float* __restrict__ range2,
                                                   float* __restrict__ curvature, int* __restrict__ half_window, int* __restrict__ order) {
  __shared__ int begin[kMaxRings + 1];
  const PtBlock b = blocks[blockIdx.x];
  const RingScan sc = scans[b.scan];
  const int n = counts[2 * b.scan + 1];
  if (b.first >= n) return;
  if (threadIdx.x == 0) { int run = 0; for (int r = 0; r < kMaxRings; ++r) { begin[r] = run; run += ring_count2[(size_t)b.scan * kMaxRings + r]; } begin[kMaxRings] = run; }
  __syncthreads();
  const int i = b.first + threadIdx.x;
  if (i >= n) return;
  const Point* P = reinterpret_cast<const Point*>(cloud2 + sc.pt0);
  const int ring = (int)P[i].w;
  float curv; int half;
  curvature_point(P, range2 + sc.pt0, n, begin[ring] + 5, begin[ring + 1] - 6, i, &curv, &half);
  curvature[sc.pt0 + i] = curv; half_window[sc.pt0 + i] = half;
  order[sc.pt0 + i] = i;            // outside the sectors (and in sectors left to the host) the pick order is the index order
}

// ---- K23: the order the picks visit a sector in (:707-723 cut a ring into six sectors; ExtractEdgeFeatures2 :896 and ExtractPlaneFeatures2 :1110
// walk each sector's points by curvature: std::sort of the index range with `curvature[a] < curvature[b]`).  One workgroup per sector: the keys
// (order-preserving integer image of the float, point index) go through a bitonic network in LDS.  With distinct curvatures every correct sort
// returns this permutation.  Where two curvatures of a sector are equal, the order of the equal elements is what libstdc++'s introsort leaves: the
// first lane then runs that algorithm (pvlm_stdsort.h, pinned against the real std::sort by tests/test_stdsort_cpu.py and by
// stdsort_selfcheck() below at the first call) on the sector's keys in LDS.  Left to the host (flag 1, index order): a sector with a NaN,
// with more than kSectorMax points, or every tied sector when the self-check failed (`ties_on_device` == 0).
constexpr int kSectorMax = 2048, kTieSmall = 512;
__device__ inline unsigned curvature_key(float c) {
  unsigned b = __float_as_uint(c);
  if (c == 0.f) b = 0u;                                         // -0 == +0
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(256) void k_sector_sort(const RingScan* __restrict__ scans, int rings, const int* __restrict__ ring_count2, const int* __restrict__ counts,
                                                     const float* __restrict__ curvature, int ties_on_device, int* __restrict__ order,
                                                     unsigned char* __restrict__ sector_host, int* __restrict__ tie_count, int4* __restrict__ tie_list, int tie_cap) {
  __shared__ unsigned long long key[kSectorMax];
  __shared__ int flagged;
  const int s = blockIdx.y, ring = blockIdx.x / 6, j = blockIdx.x % 6, t = threadIdx.x;
  const RingScan sc = scans[s];
  const int n = counts[2 * s + 1];
  int begin = 0;
  for (int r = 0; r < ring; ++r) begin += ring_count2[(size_t)s * kMaxRings + r];
  const int lo = begin + 5, hi = begin + ring_count2[(size_t)s * kMaxRings + ring] - 6, span = hi - lo;   // scanStartInd, scanEndInd (:520-522)
  unsigned char* out_flag = sector_host + ((size_t)s * rings + ring) * 6 + j;
  if (n == 0 || span < 6) { if (t == 0) *out_flag = 0; return; }                                          // the host never sorts such a ring
  const int sp = lo + span * j / 6, ep = lo + span * (j + 1) / 6 - 1, m = ep - sp + 1;
  if (m < 2) { if (t == 0) *out_flag = 0; return; }
  if (m > kSectorMax) { if (t == 0) *out_flag = 1; return; }
  int P = 2;
  while (P < m) P <<= 1;
  if (t == 0) flagged = 0;
  __syncthreads();
  const float* c = curvature + sc.pt0;
  for (int k = t; k < P; k += 256) {
    unsigned long long v = ~0ull;
    if (k < m) {
      const float x = c[sp + k];
      if (x != x) flagged = 2;                                                                            // NaN: `<` is no order at all
      v = ((unsigned long long)curvature_key(x) << 32) | (unsigned)(sp + k);
    }
    key[k] = v;
  }
  __syncthreads();
  if (flagged == 2) { if (t == 0) *out_flag = 1; return; }
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int k = t; k < (P >> 1); k += 256) {
        const int a = 2 * k - (k & (stride - 1)), b = a + stride;          // the k-th compare-exchange pair of this stage
        const bool up = (a & size) == 0;
        const unsigned long long x = key[a], y = key[b];
        if ((x > y) == up) { key[a] = y; key[b] = x; }
      }
      __syncthreads();
    }
  for (int k = t; k + 1 < m; k += 256) if ((key[k] >> 32) == (key[k + 1] >> 32)) flagged = 1;
  __syncthreads();
  if (!flagged) {
    for (int k = t; k < m; k += 256) order[sc.pt0 + sp + k] = (int)(unsigned)key[k];
    if (t == 0) *out_flag = 0;
    return;
  }
  // equal curvatures: the sector goes on the list of k_sector_ties (index order until then); without the self-check's blessing, to the host
  if (t == 0) {
    if (!ties_on_device) *out_flag = 1;
    else { *out_flag = 0; const int at = atomicAdd(tie_count + (m > kTieSmall ? 1 : 0), 1); tie_list[(m > kTieSmall ? tie_cap : 0) + at] = make_int4(s, sp, m, (int)(out_flag - sector_host)); }
  }
}

// std::sort's own steps (pvlm_stdsort.h) on the sectors k_sector_sort listed: one wave per sector, sorting packed (key, index) words in
// LDS by key alone — the comparator std::sort was given looks at the curvature only, which is what makes the order of equal keys the algorithm's —
// by the whole wave (pvlm_stdsort::sort_wave: long ranges partitioned by all lanes, short ones side by side on the lanes).
template <int CAP>
__global__ __launch_bounds__(64) void k_sector_ties(const RingScan* __restrict__ scans, const int4* __restrict__ list, const int* __restrict__ count,
                                                    const float* __restrict__ curvature, int* __restrict__ order, unsigned char* __restrict__ sector_host) {
  __shared__ unsigned long long e[CAP];
  __shared__ unsigned queue[2 * pvlm_stdsort::kWaveQueue + pvlm_stdsort::kWideStack], cuts[(CAP + 31) / 32 + 1];
  __shared__ unsigned short pos[2 * CAP];
  __shared__ int ctr[4];
  for (int item = blockIdx.x; item < *count; item += gridDim.x) {
    const int4 it = list[item];
    const int sp = it.y, m = it.z, t = threadIdx.x;
    const long long pt0 = scans[it.x].pt0;
    const float* c = curvature + pt0;
    for (int k = t; k < m; k += 64) e[k] = ((unsigned long long)curvature_key(c[sp + k]) << 32) | (unsigned)(sp + k);
    __syncthreads();
    const bool sane = pvlm_stdsort::sort_wave(e, m, [](unsigned long long x, unsigned long long y) { return (unsigned)(x >> 32) < (unsigned)(y >> 32); }, queue, cuts, ctr,
                                              pos, t);
    if (sane) for (int k = t; k < m; k += 64) order[pt0 + sp + k] = (int)(unsigned)e[k];
    else if (t == 0) sector_host[it.w] = 1;                       // cannot happen with an order-preserving integer key; the host's std::sort then
    __syncthreads();
  }
}

// pvlm_ring_debug_sort: sort_wave on caller-given integer keys (tests)
__global__ __launch_bounds__(64) void k_debug_sort(const unsigned* __restrict__ keys, int n, int* __restrict__ order, int* __restrict__ sane_out) {
  __shared__ unsigned long long e[4096];
  __shared__ unsigned queue[2 * pvlm_stdsort::kWaveQueue + pvlm_stdsort::kWideStack], cuts[4096 / 32 + 1];
  __shared__ unsigned short pos[2 * 4096];
  __shared__ int ctr[4];
  const int t = threadIdx.x;
  for (int k = t; k < n; k += 64) e[k] = ((unsigned long long)keys[k] << 32) | (unsigned)k;
  __syncthreads();
  const bool sane = pvlm_stdsort::sort_wave(e, n, [](unsigned long long x, unsigned long long y) { return (unsigned)(x >> 32) < (unsigned)(y >> 32); }, queue, cuts, ctr, pos, t);
  for (int k = t; k < n; k += 64) order[k] = (int)(unsigned)e[k];
  if (t == 0) *sane_out = sane ? 1 : 0;
}

// the restated introsort against the std::sort this library was built with, on tie-heavy keys: the condition for ordering tied sectors on the device
static bool stdsort_selfcheck() {
  static const bool ok = [] {
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int trial = 0; trial < 64; ++trial) {
      const int n = 17 + (int)(next() % 1500);
      const int levels = 1 + (int)(next() % (trial % 2 ? 8 : 200));
      std::vector<float> key((size_t)n);
      for (float& k : key) k = (float)(next() % (unsigned)levels) * 0.125f - 1.f;
      std::vector<int> a((size_t)n), b((size_t)n);
      for (int i = 0; i < n; ++i) a[(size_t)i] = b[(size_t)i] = i;
      const float* kp = key.data();
      std::sort(a.begin(), a.end(), [kp](int x, int y) { return kp[x] < kp[y]; });
      if (!pvlm_stdsort::sort(b.data(), n, [kp](int x, int y) { return kp[x] < kp[y]; }) || a != b) return false;
    }
    return true;
  }();
  return ok;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------
struct pvlm_ring_batch {
  pvlm_ctx* ctx = nullptr;
  int n_scans = 0, rings = 0, horizon = 0, segment = 0;
  long long total_points = 0, total_cells = 0, total_slots = 0;
  std::vector<RingScan> scans;
  std::vector<int> counts;            // 2 per scan: reordered, kept
  std::vector<int> ring_count, ring_count2;   // kMaxRings per scan
  std::vector<int> resolved_points, resolved_edges, replayed;
  // device (pool)
  RingScan* d_scans = nullptr;
  float4* d_cloud_scan = nullptr; int2* d_rc = nullptr; float* d_range_image = nullptr; int* d_image_to_point = nullptr;
  float4* d_cloud2 = nullptr; int* d_image_to_point2 = nullptr;
  // pinned host results: kept state, 6 arrays of total_points + 6 sector flags per ring
  char* h_results = nullptr; size_t results_bytes = 0;
  const int* h_source = nullptr; const int* h_ring_col = nullptr; const float* h_curvature = nullptr; const int* h_half = nullptr; const float* h_range = nullptr;
  const int* h_order = nullptr; const unsigned char* h_sector_host = nullptr;
  bool lazy_arrays = false;                       // picks on the device: curvature / window / range / order / sector flags came down only for the scans the host has to pick
  std::vector<unsigned char> has_arrays;          // per scan
  // K24 (picks != 0): states, per-ring pick lists, voxel centroids
  int picks = 0; float max_curvature = 0, angle_threshold = 0;
  const unsigned char* h_state = nullptr; const int* h_corner = nullptr; const int* h_flat = nullptr; const int* h_voxel_span = nullptr; const float* h_voxels = nullptr;
  const unsigned char* h_ring_host = nullptr; int n_voxels = 0;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

extern "C" pvlm_status pvlm_ring_batch_destroy(pvlm_ctx* ctx, pvlm_ring_batch* b);

// scratch of one run, back to the pool on every exit path
struct RingScratch {
  pvlm_ctx* ctx; std::vector<void*> p;
  ~RingScratch() { for (void* q : p) pvlm_i_free(ctx, q); }
  template <typename T> pvlm_status get(T** d, size_t count) { const pvlm_status st = pvlm_i_alloc(ctx, d, count); if (!st) p.push_back(*d); return st; }
};

static pvlm_status ring_run(pvlm_ctx* ctx, pvlm_ring_batch* B, int n_scans, const pvlm_raw_scan* raw_scans, int n_rings, int horizon, int segment, long long total,
                            int picks, float max_curvature, float angle_threshold) {
  B->picks = picks ? 1 : 0; B->max_curvature = max_curvature; B->angle_threshold = angle_threshold;
  B->ctx = ctx; B->n_scans = n_scans; B->rings = n_rings; B->horizon = horizon; B->segment = (segment & 1) ? 1 : 0;
  B->lazy_arrays = picks != 0 && (segment & 2) == 0;       // bit 1 of `segment`: always deliver the per-point arrays (parity tests, traces)
  B->has_arrays.assign((size_t)n_scans, B->lazy_arrays ? 0 : 1);
  const int cells = n_rings * horizon;
  B->total_points = total; B->total_cells = (long long)n_scans * cells;
  B->scans.resize((size_t)n_scans); B->counts.assign((size_t)n_scans * 2, 0);
  B->ring_count.assign((size_t)n_scans * kMaxRings, 0); B->ring_count2.assign((size_t)n_scans * kMaxRings, 0);
  B->resolved_points.assign((size_t)n_scans, 0); B->resolved_edges.assign((size_t)n_scans, 0); B->replayed.assign((size_t)n_scans, 0);
  std::vector<PtBlock> blocks;
  {
    long long pt0 = 0;
    for (int s = 0; s < n_scans; ++s) {
      RingScan& sc = B->scans[(size_t)s];
      sc.pt0 = pt0; sc.cell0 = (long long)s * cells; sc.n = raw_scans[s].n; sc.pad = 0; sc.start_ori = 0;
      sc.slot0 = B->total_slots; B->total_slots += chunk_slots(sc.n, kColumnThreads);
      if (sc.n > 0) sc.start_ori = ori_of_atan2(std::atan2(raw_scans[s].xyzi[0], raw_scans[s].xyzi[2]));   // float overload, :397-399
      for (int f = 0; f < sc.n; f += 256) blocks.push_back(PtBlock{s, f});
      pt0 += sc.n;
    }
  }
  if (n_scans == 0 || total == 0) return PVLM_OK;
  // ---- device memory: everything from the context's pool; the scratch goes back at the end of the call
  RingScratch tmp{ctx, {}};
  const size_t NP = (size_t)total, NC = (size_t)B->total_cells, NS = (size_t)B->total_slots;
  float4* d_raw = nullptr; PointRec* d_rec = nullptr; int* d_listed = nullptr; int* d_counter = nullptr;
  int2* d_colpos = nullptr; int* d_ring_count = nullptr; int2* d_status = nullptr; PtBlock* d_blocks = nullptr; int* d_source = nullptr; int* d_winner = nullptr;
  unsigned char* d_edges = nullptr; EdgeQuery* d_queries = nullptr; int* d_parent = nullptr; int* d_root = nullptr; int* d_comp_size = nullptr;
  unsigned long long* d_row_mask = nullptr; int* d_source2 = nullptr; int* d_ring_col2 = nullptr; float* d_range2 = nullptr; int* d_ring_count2 = nullptr;
  int* d_counts = nullptr; float* d_curv = nullptr; int* d_half = nullptr; int* d_order = nullptr; unsigned char* d_sector = nullptr; int4* d_ties = nullptr;
  const size_t n_sectors = (size_t)n_scans * n_rings * 6;
  const int query_cap = 1 << 16;
  pvlm_status st = PVLM_OK;
#define RING_GET(ptr, count) if (!st) st = tmp.get(&ptr, count)
#define RING_KEEP(ptr, count) if (!st) st = pvlm_i_alloc(ctx, &ptr, count)
  RING_KEEP(B->d_scans, (size_t)n_scans); RING_KEEP(B->d_cloud_scan, NP); RING_KEEP(B->d_rc, NP); RING_KEEP(B->d_range_image, NC);
  RING_KEEP(B->d_image_to_point, NC); RING_KEEP(B->d_cloud2, NP); RING_KEEP(B->d_image_to_point2, NC);
  RING_GET(d_raw, NP); RING_GET(d_rec, NS); RING_GET(d_listed, NP); RING_GET(d_counter, 8);
  RING_GET(d_colpos, NS); RING_GET(d_ring_count, (size_t)n_scans * kMaxRings); RING_GET(d_status, (size_t)n_scans); RING_GET(d_blocks, blocks.size());
  RING_GET(d_source, NP); RING_GET(d_winner, NC); RING_GET(d_edges, NC + 4); RING_GET(d_queries, (size_t)query_cap); RING_GET(d_parent, NC);
  RING_GET(d_root, NC); RING_GET(d_comp_size, NC); RING_GET(d_row_mask, NC); RING_GET(d_source2, NP); RING_GET(d_ring_col2, NP); RING_GET(d_range2, NP);
  RING_GET(d_ring_count2, (size_t)n_scans * kMaxRings); RING_GET(d_counts, (size_t)n_scans * 2); RING_GET(d_curv, NP); RING_GET(d_half, NP);
  RING_GET(d_order, NP); RING_GET(d_sector, n_sectors); RING_GET(d_ties, 2 * n_sectors);
  // K24: one centroid per four points is the batch's budget (a Room scan needs one per ten); rings beyond it go to the host like any undecided ring
  const size_t n_ring_slots = (size_t)n_scans * n_rings, voxel_cap = NP / 4 + 64;
  unsigned char* d_state = nullptr; int* d_corner = nullptr; int* d_flat = nullptr; int2* d_vspan = nullptr; float4* d_voxels = nullptr; unsigned char* d_ring_host = nullptr;
  if (picks) {
    RING_GET(d_state, NP); RING_GET(d_corner, n_ring_slots * (1 + kCornerSlots)); RING_GET(d_flat, n_ring_slots * (1 + kFlatSlots)); RING_GET(d_vspan, n_ring_slots);
    RING_GET(d_voxels, voxel_cap); RING_GET(d_ring_host, n_ring_slots);
  }
#undef RING_GET
#undef RING_KEEP
  if (st) return (st);
  pvlm_i_trace("ring: device memory");
  // ---- pinned buffer: raw points on the way up (16 B / point), the six result arrays (24 B / point) and the sector flags on the way down
  // with K24: + states (1 B / point), the per-ring lists and the centroids (16 B each, voxel_cap of them at most)
  const size_t picks_fixed = picks ? NP + n_ring_slots * ((1 + kCornerSlots) * 4 + (1 + kFlatSlots) * 4 + 8 + 1) + 64 : 0;
  const size_t pinned = NP * 24 + n_sectors + 256 + picks_fixed + (picks ? voxel_cap * 16 : 0);
  int fit = -1;                                    // the smallest pooled buffer that is large enough
  for (int k = 0; k < ctx->ring_pool; ++k) if (ctx->ring_bytes[k] >= pinned && (fit < 0 || ctx->ring_bytes[k] < ctx->ring_bytes[fit])) fit = k;
  if (fit >= 0) {
    B->h_results = (char*)ctx->h_ring[fit]; B->results_bytes = ctx->ring_bytes[fit];
    --ctx->ring_pool;
    ctx->h_ring[fit] = ctx->h_ring[ctx->ring_pool]; ctx->ring_bytes[fit] = ctx->ring_bytes[ctx->ring_pool];
  } else if (hipHostMalloc((void**)&B->h_results, pinned, hipHostMallocDefault) == hipSuccess) {
    B->results_bytes = pinned;
  } else { B->h_results = nullptr; PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: %zu bytes of pinned memory unavailable", pinned); return (PVLM_ERR_NOMEM); }
  pvlm_i_trace("ring: pinned buffer");
  hipStream_t S = ctx->stream;
  hipEvent_t ev[9];
  for (hipEvent_t& e : ev) if (hipEventCreate(&e) != hipSuccess) { PVLM_SET_ERR(ctx, "hipEventCreate failed"); return (PVLM_ERR_HIP); }
  struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int k = 0; k < 9; ++k) (void)hipEventDestroy(e[k]); } } evg{ev};
  {   // staging copy, scan-parallel (a scan's points are contiguous when stride_floats == 4)
    float4* h = (float4*)B->h_results;
    std::atomic<int> next{0};
    std::atomic<long long> bad{-1};
    auto work = [&]() {
      for (int s = next++; s < n_scans; s = next++) {
        const pvlm_raw_scan& r = raw_scans[s];
        float4* d = h + B->scans[(size_t)s].pt0;
        if (r.stride_floats == 4) std::memcpy(d, r.xyzi, (size_t)r.n * 16);
        else for (int i = 0; i < r.n; ++i) { const float* p = r.xyzi + (size_t)i * r.stride_floats; d[i] = make_float4(p[0], p[1], p[2], p[3]); }
        for (int i = 0; i < r.n; ++i)
          if (!std::isfinite(d[i].x) || !std::isfinite(d[i].y) || !std::isfinite(d[i].z)) { bad = B->scans[(size_t)s].pt0 + i; break; }
      }
    };
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({pvlm_thread_cap(), (size_t)n_scans / 32 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
    pvlm_run_workers(n_threads, work);
    if (bad >= 0) { PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: non-finite coordinate (point %lld of the batch)", (long long)bad); return PVLM_ERR_REFUSED; }
  }
  pvlm_i_trace("ring: staging copy");
  (void)hipEventRecord(ev[0], S);
  PVLM_HIP(ctx, hipMemcpyAsync(d_raw, B->h_results, NP * 16, hipMemcpyHostToDevice, S));
  PVLM_HIP(ctx, hipMemcpyAsync(B->d_scans, B->scans.data(), (size_t)n_scans * sizeof(RingScan), hipMemcpyHostToDevice, S));
  PVLM_HIP(ctx, hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(PtBlock), hipMemcpyHostToDevice, S));
  PVLM_HIP(ctx, hipMemsetAsync(d_counter, 0, 8 * sizeof(int), S));
  PVLM_HIP(ctx, hipMemsetAsync(d_winner, 0xFF, NC * sizeof(int), S));
  PVLM_HIP(ctx, hipMemsetAsync(B->d_range_image, 0, NC * sizeof(float), S));
  PVLM_HIP(ctx, hipMemsetAsync(B->d_image_to_point, 0xFF, NC * sizeof(int), S));
  (void)hipEventRecord(ev[1], S);
  // ---- K16 + the host's libm for the listed points
  hipLaunchKernelGGL(k_ring_classify, dim3((unsigned)blocks.size()), dim3(256), 0, S, B->d_scans, d_blocks, n_rings, horizon, d_raw, d_rec, d_counter, d_listed);
  int h_counter[4] = {0, 0, 0, 0};
  PVLM_HIP(ctx, hipMemcpyAsync(h_counter, d_counter, sizeof(int), hipMemcpyDeviceToHost, S));
  PVLM_HIP(ctx, hipStreamSynchronize(S));      // also: the staging buffer is free again
  auto scan_of = [&](long long at) { int lo = 0, hi = n_scans - 1; while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (B->scans[(size_t)mid].pt0 <= at) lo = mid; else hi = mid - 1; } return lo; };
  auto exact_of = [&](long long at, PointPatch* pp) {
    const int s = scan_of(at);
    const pvlm_raw_scan& r = raw_scans[s];
    const float* p = r.xyzi + (size_t)(at - B->scans[(size_t)s].pt0) * r.stride_floats;
    const float q = -p[1] / std::sqrt(p[0] * p[0] + p[2] * p[2]);
    const RingScan& sc = B->scans[(size_t)s];
    pp->slot = sc.slot0 + chunk_slot((int)(at - sc.pt0), chunk_of(sc.n, kColumnThreads), kColumnThreads);
    pp->az = std::atan2(p[0], p[2]); pp->ring = q == q ? ring_of_atan(std::atan(q), n_rings) : -1;
    return s;
  };
  std::vector<PointPatch> patches;
  PointPatch* d_patch = nullptr;
  auto send_patches = [&]() -> pvlm_status {
    if (patches.empty()) return PVLM_OK;
    pvlm_i_free(ctx, d_patch); d_patch = nullptr;
    pvlm_status s2 = pvlm_i_alloc(ctx, &d_patch, patches.size());
    if (s2) return s2;
    PVLM_HIP(ctx, hipMemcpyAsync(d_patch, patches.data(), patches.size() * sizeof(PointPatch), hipMemcpyHostToDevice, S));
    hipLaunchKernelGGL(k_ring_patch, dim3((unsigned)((patches.size() + 255) / 256)), dim3(256), 0, S, (int)patches.size(), d_patch, d_rec);
    PVLM_HIP(ctx, hipStreamSynchronize(S));     // `patches` is pageable
    return PVLM_OK;
  };
  if (h_counter[0] > 0) {
    std::vector<int> listed((size_t)h_counter[0]);
    PVLM_HIP(ctx, hipMemcpy(listed.data(), d_listed, listed.size() * sizeof(int), hipMemcpyDeviceToHost));
    patches.resize(listed.size());
    for (size_t k = 0; k < listed.size(); ++k) B->resolved_points[(size_t)exact_of(listed[k], &patches[k])]++;
    if ((st = send_patches())) { pvlm_i_free(ctx, d_patch); return (st); }
  }
  pvlm_i_trace("ring: upload + K16 + listed points");
  (void)hipEventRecord(ev[2], S);
  // ---- K17; a scan whose +z crossing stays undecided gets the host's azimuths for the points of that comparison and is replayed
  hipLaunchKernelGGL(k_ring_columns, dim3((unsigned)n_scans), dim3(1024), 0, S, B->d_scans, (const int*)nullptr, n_rings, horizon, d_rec, d_colpos,
                     d_ring_count, d_status);
  std::vector<int2> status((size_t)n_scans, make_int2(-1, -1));
  PVLM_HIP(ctx, hipMemcpyAsync(status.data(), d_status, (size_t)n_scans * sizeof(int2), hipMemcpyDeviceToHost, S));
  PVLM_HIP(ctx, hipStreamSynchronize(S));
  for (int round = 0;; ++round) {
    std::vector<int> again;
    for (int s = 0; s < n_scans; ++s) if (status[(size_t)s].x >= 0) again.push_back(s);
    if (again.empty()) break;
    patches.clear();
    for (int s : again) {
      const RingScan& sc = B->scans[(size_t)s];
      B->replayed[(size_t)s]++;
      const int at = status[(size_t)s].x, last = status[(size_t)s].y;
      // after 64 rounds (never seen): the whole scan from the host libm, which cannot be undecided
      const int lo = round < 64 ? at : 0, hi = round < 64 ? std::min(sc.n, at + n_rings + 1) : sc.n;
      PointPatch pp;
      if (round < 64 && last >= 0) { exact_of(sc.pt0 + last, &pp); patches.push_back(pp); }
      for (int i = lo; i < hi; ++i) { exact_of(sc.pt0 + i, &pp); patches.push_back(pp); }
    }
    if ((st = send_patches())) { pvlm_i_free(ctx, d_patch); return (st); }
    int* d_again = nullptr;
    if ((st = pvlm_i_alloc(ctx, &d_again, again.size()))) { pvlm_i_free(ctx, d_patch); return (st); }
    hipError_t e = hipMemcpyAsync(d_again, again.data(), again.size() * sizeof(int), hipMemcpyHostToDevice, S);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_ring_columns, dim3((unsigned)again.size()), dim3(1024), 0, S, B->d_scans, (const int*)d_again, n_rings, horizon, d_rec, d_colpos,
                         d_ring_count, d_status);
      e = hipMemcpyAsync(status.data(), d_status, (size_t)n_scans * sizeof(int2), hipMemcpyDeviceToHost, S);
      if (e == hipSuccess) e = hipStreamSynchronize(S);
    }
    pvlm_i_free(ctx, d_again);
    if (e != hipSuccess) { pvlm_i_free(ctx, d_patch); PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: replay failed: %s", hipGetErrorString(e)); return (PVLM_ERR_HIP); }
  }
  pvlm_i_free(ctx, d_patch); d_patch = nullptr;
  pvlm_i_trace("ring: K17");
  (void)hipEventRecord(ev[3], S);
  // ---- K18
  hipLaunchKernelGGL(k_ring_scatter, dim3((unsigned)blocks.size()), dim3(256), 0, S, B->d_scans, d_blocks, horizon, d_raw, d_rec, d_colpos, d_ring_count, B->d_cloud_scan, d_source,
                     B->d_rc, d_winner);
  hipLaunchKernelGGL(k_ring_cells, dim3((unsigned)blocks.size()), dim3(256), 0, S, B->d_scans, d_blocks, horizon, d_raw, d_rec, d_colpos, d_ring_count, d_winner, B->d_range_image,
                     B->d_image_to_point);
  (void)hipEventRecord(ev[4], S);
  // ---- K19 / K20: segmentation
  const dim3 cell_grid((unsigned)((cells + 255) / 256), (unsigned)n_scans);
  if (B->segment) {
    // the constants of :1459-1462 and their sin / cos (:1510), float libm of the host
    const float alpha_x = 0.2 / 180.0 * M_PI, alpha_y = 2.0 / 180.0 * M_PI, theta = 20.0 / 180.0 * M_PI;
    const float sin_x = std::sin(alpha_x), cos_x = std::cos(alpha_x), sin_y = std::sin(alpha_y), cos_y = std::cos(alpha_y);
    hipLaunchKernelGGL(k_seg_edges, cell_grid, dim3(256), 0, S, B->d_scans, n_rings, horizon, B->d_range_image, sin_x, cos_x, sin_y, cos_y, theta, d_edges, d_counter + 1, d_queries, query_cap);
    PVLM_HIP(ctx, hipMemcpyAsync(h_counter + 1, d_counter + 1, sizeof(int), hipMemcpyDeviceToHost, S));
    PVLM_HIP(ctx, hipStreamSynchronize(S));
    if (h_counter[1] > query_cap) { PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: %d undecided segmentation edges (capacity %d)", h_counter[1], query_cap); return (PVLM_ERR_REFUSED); }
    if (h_counter[1] > 0) {
      std::vector<EdgeQuery> q((size_t)h_counter[1]);
      PVLM_HIP(ctx, hipMemcpy(q.data(), d_queries, q.size() * sizeof(EdgeQuery), hipMemcpyDeviceToHost));
      std::vector<EdgePatch> ep;
      for (const EdgeQuery& e : q) {
        B->resolved_edges[(size_t)(e.cell / cells)]++;
        if (std::atan2(e.y, e.x) > theta) ep.push_back(EdgePatch{e.cell, e.bit});
      }
      if (!ep.empty()) {
        EdgePatch* d_ep = nullptr;
        if ((st = pvlm_i_alloc(ctx, &d_ep, ep.size()))) return (st);
        hipError_t e = hipMemcpyAsync(d_ep, ep.data(), ep.size() * sizeof(EdgePatch), hipMemcpyHostToDevice, S);
        if (e == hipSuccess) { hipLaunchKernelGGL(k_seg_patch, dim3((unsigned)((ep.size() + 255) / 256)), dim3(256), 0, S, (int)ep.size(), d_ep, d_edges); e = hipStreamSynchronize(S); }
        pvlm_i_free(ctx, d_ep);
        if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: edge patch failed: %s", hipGetErrorString(e)); return (PVLM_ERR_HIP); }
      }
    }
    (void)hipEventRecord(ev[5], S);
    hipLaunchKernelGGL(k_seg_init, dim3((unsigned)n_rings, (unsigned)n_scans), dim3(256), 0, S, B->d_scans, horizon, d_edges, d_parent, d_comp_size, d_row_mask, B->d_image_to_point2);
    hipLaunchKernelGGL(k_seg_union, cell_grid, dim3(256), 0, S, B->d_scans, n_rings, horizon, d_edges, d_parent);
    hipLaunchKernelGGL(k_seg_stats, cell_grid, dim3(256), 0, S, B->d_scans, n_rings, horizon, d_parent, d_root, d_comp_size, d_row_mask);
  } else {
    PVLM_HIP(ctx, hipMemsetAsync(B->d_image_to_point2, 0xFF, NC * sizeof(int), S));
    (void)hipEventRecord(ev[5], S);
  }
  (void)hipEventRecord(ev[6], S);
  // ---- K21 / K22
  hipLaunchKernelGGL(k_seg_compact, dim3((unsigned)n_scans), dim3(1024), 0, S, B->d_scans, n_rings, horizon, B->segment, d_ring_count, B->d_cloud_scan, d_source, B->d_rc,
                     B->d_range_image, d_root, d_comp_size, d_row_mask, B->d_cloud2, d_source2, d_ring_col2, d_range2, B->d_image_to_point2, d_ring_count2, d_counts);
  hipLaunchKernelGGL(k_curvature, dim3((unsigned)blocks.size()), dim3(256), 0, S, B->d_scans, d_blocks, d_ring_count2, d_counts, B->d_cloud2, d_range2, d_curv, d_half, d_order);
  PVLM_HIP(ctx, hipGetLastError());       // a rejected launch must fail the batch: the arrays below would be copied down uninitialised (the caller's host path takes over)
  // the five per-point arrays are final here: they go down the link on a second stream while K23 / K24 run (20 of the 27 B per point)
  char* h = B->h_results;
  hipStream_t S2 = S;
  if (!ctx->aux_stream && hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->aux_stream = nullptr; }
  if (ctx->aux_stream && hipEventRecord(ev[8], S) == hipSuccess && hipStreamWaitEvent(ctx->aux_stream, ev[8], 0) == hipSuccess) S2 = ctx->aux_stream;
  struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{S2};        // before the scratch goes back to the pool, on every exit path
  PVLM_HIP(ctx, hipMemcpyAsync(h, d_source2, NP * 4, hipMemcpyDeviceToHost, S2));
  PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 4, d_ring_col2, NP * 4, hipMemcpyDeviceToHost, S2));
  if (!B->lazy_arrays) {
    PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 8, d_curv, NP * 4, hipMemcpyDeviceToHost, S2));
    PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 12, d_half, NP * 4, hipMemcpyDeviceToHost, S2));
    PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 16, d_range2, NP * 4, hipMemcpyDeviceToHost, S2));
  }
  hipLaunchKernelGGL(k_sector_sort, dim3((unsigned)(n_rings * 6), (unsigned)n_scans), dim3(256), 0, S, B->d_scans, n_rings, d_ring_count2, d_counts, d_curv, stdsort_selfcheck() ? 1 : 0, d_order, d_sector, d_counter + 2, d_ties, (int)n_sectors);
  {
    const unsigned waves = 256u * 16u;                                     // persistent: every wave walks the list with a grid stride
    hipLaunchKernelGGL(k_sector_ties<kTieSmall>, dim3(waves), dim3(64), 0, S, B->d_scans, d_ties, d_counter + 2, d_curv, d_order, d_sector);
    hipLaunchKernelGGL(k_sector_ties<kSectorMax>, dim3(waves / 4), dim3(64), 0, S, B->d_scans, d_ties + n_sectors, d_counter + 3, d_curv, d_order, d_sector);
    PVLM_HIP(ctx, hipGetLastError());     // K23: sector orders / flags of a launch that did not run are pool garbage
  }
  int ring_cap = 0;
  if (picks) {
    ring_cap = std::min(horizon, 4096);                       // a ring holds one point per column; longer rings are refused by the kernel (-> host)
    PickArrays A;
    A.scans = B->d_scans; A.ring_count2 = d_ring_count2; A.counts = d_counts; A.cloud2 = B->d_cloud2; A.range2 = d_range2; A.curvature = d_curv; A.half_window = d_half;
    A.order = d_order; A.sector_host = d_sector; A.state = d_state; A.corner = d_corner; A.flat = d_flat; A.voxel_span = d_vspan; A.voxels = d_voxels;
    A.voxel_counter = d_counter + 4; A.voxel_cap = (int)voxel_cap; A.ring_host = d_ring_host;
    const size_t lds = pick_lds_bytes(ring_cap);
    hipLaunchKernelGGL(k_ring_picks, dim3((unsigned)n_rings, (unsigned)n_scans), dim3(64), lds, S, A, n_rings, ring_cap, max_curvature, angle_threshold);
    PVLM_HIP(ctx, hipGetLastError());     // K24 (dynamic LDS up to ~56 KB): ring_host / pick lists of a rejected launch must not reach AssemblePicks
  }
  (void)hipEventRecord(ev[7], S);
  // ---- results
  B->h_source = (const int*)h; B->h_ring_col = (const int*)(h + NP * 4); B->h_curvature = (const float*)(h + NP * 8); B->h_half = (const int*)(h + NP * 12);
  B->h_range = (const float*)(h + NP * 16); B->h_order = (const int*)(h + NP * 20); B->h_sector_host = (const unsigned char*)(h + NP * 24);
  if (!B->lazy_arrays) {
    PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 20, d_order, NP * 4, hipMemcpyDeviceToHost, S));
    PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 24, d_sector, n_sectors, hipMemcpyDeviceToHost, S));
  }
  char* hp = h + NP * 24 + ((n_sectors + 63) / 64) * 64;       // K24's results behind the sector flags
  int h_voxel_count = 0;
  if (picks) {
    char* q = hp;
    B->h_state = (const unsigned char*)q; PVLM_HIP(ctx, hipMemcpyAsync(q, d_state, NP, hipMemcpyDeviceToHost, S)); q += (NP + 63) / 64 * 64;
    B->h_corner = (const int*)q; PVLM_HIP(ctx, hipMemcpyAsync(q, d_corner, n_ring_slots * (1 + kCornerSlots) * 4, hipMemcpyDeviceToHost, S)); q += n_ring_slots * (1 + kCornerSlots) * 4;
    B->h_flat = (const int*)q; PVLM_HIP(ctx, hipMemcpyAsync(q, d_flat, n_ring_slots * (1 + kFlatSlots) * 4, hipMemcpyDeviceToHost, S)); q += n_ring_slots * (1 + kFlatSlots) * 4;
    B->h_voxel_span = (const int*)q; PVLM_HIP(ctx, hipMemcpyAsync(q, d_vspan, n_ring_slots * 8, hipMemcpyDeviceToHost, S)); q += n_ring_slots * 8;
    B->h_ring_host = (const unsigned char*)q; PVLM_HIP(ctx, hipMemcpyAsync(q, d_ring_host, n_ring_slots, hipMemcpyDeviceToHost, S)); q += (n_ring_slots + 63) / 64 * 64;
    B->h_voxels = (const float*)q;
    PVLM_HIP(ctx, hipMemcpyAsync(&h_voxel_count, d_counter + 4, sizeof(int), hipMemcpyDeviceToHost, S));
  }
  PVLM_HIP(ctx, hipMemcpyAsync(B->counts.data(), d_counts, (size_t)n_scans * 2 * sizeof(int), hipMemcpyDeviceToHost, S));
  PVLM_HIP(ctx, hipMemcpyAsync(B->ring_count.data(), d_ring_count, (size_t)n_scans * kMaxRings * sizeof(int), hipMemcpyDeviceToHost, S));
  PVLM_HIP(ctx, hipMemcpyAsync(B->ring_count2.data(), d_ring_count2, (size_t)n_scans * kMaxRings * sizeof(int), hipMemcpyDeviceToHost, S));
  if (picks) {                                                  // the centroids: as many as the rings reserved (rings that overflowed the budget are flagged)
    PVLM_HIP(ctx, hipStreamSynchronize(S));
    B->n_voxels = (int)std::min<size_t>((size_t)std::max(h_voxel_count, 0), voxel_cap);
    if (B->n_voxels > 0) PVLM_HIP(ctx, hipMemcpyAsync(const_cast<float*>(B->h_voxels), d_voxels, (size_t)B->n_voxels * 16, hipMemcpyDeviceToHost, S));
    if (B->lazy_arrays) {
      // the ring flags are on the host (the synchronisation above): the five per-point arrays — 16 of the 27 B per point of round 5's download — come down only
      // for the scans with a ring K24 left to the host (incidence angle at the libm threshold, a sector with ties beyond the kernel's bounds: 0-6 % of the scans)
      for (int sc = 0; sc < n_scans; ++sc) {
        bool undecided = false;
        for (int q = 0; q < n_rings && !undecided; ++q) undecided = B->h_ring_host[(size_t)sc * n_rings + q] != 0;
        if (!undecided) continue;
        const size_t p0 = (size_t)B->scans[(size_t)sc].pt0, np = (size_t)B->scans[(size_t)sc].n;
        if (np) {
          PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 8 + p0 * 4, d_curv + p0, np * 4, hipMemcpyDeviceToHost, S));
          PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 12 + p0 * 4, d_half + p0, np * 4, hipMemcpyDeviceToHost, S));
          PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 16 + p0 * 4, d_range2 + p0, np * 4, hipMemcpyDeviceToHost, S));
          PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 20 + p0 * 4, d_order + p0, np * 4, hipMemcpyDeviceToHost, S));
        }
        PVLM_HIP(ctx, hipMemcpyAsync(h + NP * 24 + (size_t)sc * n_rings * 6, d_sector + (size_t)sc * n_rings * 6, (size_t)n_rings * 6, hipMemcpyDeviceToHost, S));
        B->has_arrays[(size_t)sc] = 1;
      }
    }
  }
  PVLM_HIP(ctx, hipStreamSynchronize(S2));
  (void)hipEventRecord(ev[8], S);
  PVLM_HIP(ctx, hipStreamSynchronize(S));
  pvlm_i_trace("ring: segmentation, curvature, download");
  for (int k = 0; k < 8; ++k) { float ms = 0; if (hipEventElapsedTime(&ms, ev[k], ev[k + 1]) == hipSuccess) B->ms[k] = ms; }
  return PVLM_OK;
}

extern "C" {

pvlm_status pvlm_ring_batch_destroy(pvlm_ctx* ctx, pvlm_ring_batch* b) {
  if (!b) return PVLM_OK;
  if (!ctx) ctx = b->ctx;
  if (ctx) {
    (void)pvlm_i_bind(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    pvlm_i_free(ctx, b->d_scans); pvlm_i_free(ctx, b->d_cloud_scan); pvlm_i_free(ctx, b->d_rc); pvlm_i_free(ctx, b->d_range_image);
    pvlm_i_free(ctx, b->d_image_to_point); pvlm_i_free(ctx, b->d_cloud2); pvlm_i_free(ctx, b->d_image_to_point2);
  }
  if (b->h_results) {
    void* gone = b->h_results;                             // the buffer stays with the context for the next batch; a full pool lets its smallest one go
    if (ctx) {
      size_t bytes = b->results_bytes;
      if (ctx->ring_pool < pvlm_ctx::kRingPool) { ctx->h_ring[ctx->ring_pool] = gone; ctx->ring_bytes[ctx->ring_pool] = bytes; ++ctx->ring_pool; gone = nullptr; }
      else {
        int least = 0;
        for (int k = 1; k < ctx->ring_pool; ++k) if (ctx->ring_bytes[k] < ctx->ring_bytes[least]) least = k;
        if (ctx->ring_bytes[least] < bytes) { std::swap(ctx->h_ring[least], gone); ctx->ring_bytes[least] = bytes; }
      }
    }
    if (gone) (void)hipHostFree(gone);
  }
  delete b;
  return PVLM_OK;
}

static pvlm_status ring_extract(pvlm_ctx* ctx, int n_scans, const pvlm_raw_scan* raw_scans, int n_rings, int horizon, int segment, int picks, float max_curvature,
                                float angle_threshold, pvlm_ring_batch** out);
pvlm_status pvlm_ring_extract_batch(pvlm_ctx* ctx, int n_scans, const pvlm_raw_scan* raw_scans, int n_rings, int horizon, int segment, pvlm_ring_batch** out) {
  return ring_extract(ctx, n_scans, raw_scans, n_rings, horizon, segment, 0, 0.f, 0.f, out);
}
pvlm_status pvlm_ring_extract_batch_picks(pvlm_ctx* ctx, int n_scans, const pvlm_raw_scan* raw_scans, int n_rings, int horizon, int segment, float max_curvature,
                                          float intersect_angle_threshold, pvlm_ring_batch** out) {
  return ring_extract(ctx, n_scans, raw_scans, n_rings, horizon, segment, 1, max_curvature, intersect_angle_threshold, out);
}
static pvlm_status ring_extract(pvlm_ctx* ctx, int n_scans, const pvlm_raw_scan* raw_scans, int n_rings, int horizon, int segment, int picks, float max_curvature,
                                float angle_threshold, pvlm_ring_batch** out) {
  if (!ctx || !out || n_scans < 0 || (n_scans > 0 && !raw_scans)) return PVLM_ERR_ARG;
  *out = nullptr;
  if ((n_rings != 16 && n_rings != 32 && n_rings != 64) || horizon <= 0 || horizon > 65535) {
    PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: %d rings x %d columns is not a supported range image (16 / 32 / 64 rings, 1..65535 columns)", n_rings, horizon);
    return PVLM_ERR_ARG;
  }
  long long total = 0;
  for (int s = 0; s < n_scans; ++s) {
    if (raw_scans[s].n < 0 || (raw_scans[s].n > 0 && !raw_scans[s].xyzi) || raw_scans[s].stride_floats < 4) { PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: bad descriptor (scan %d)", s); return PVLM_ERR_ARG; }
    total += raw_scans[s].n;
  }
  if (total >= (1ll << 31) || (long long)n_scans * n_rings * horizon >= (1ll << 31)) { PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: batch too large (split it)"); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (ctx->capturing) { PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch inside a graph capture"); return PVLM_ERR_STATE; }
  pvlm_ring_batch* B = new (std::nothrow) pvlm_ring_batch();
  if (!B) return PVLM_ERR_NOMEM;
  pvlm_status st = PVLM_ERR_HIP;
  try {
    st = ring_run(ctx, B, n_scans, raw_scans, n_rings, horizon, segment, total, picks, max_curvature, angle_threshold);
  } catch (const std::bad_alloc&) {
    PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: out of host memory");
    st = PVLM_ERR_NOMEM;
  } catch (...) {
    PVLM_SET_ERR(ctx, "pvlm_ring_extract_batch: unexpected host exception");
    st = PVLM_ERR_HIP;
  }
  if (st) { (void)hipStreamSynchronize(ctx->stream); pvlm_ring_batch_destroy(ctx, B); return st; }
  *out = B;
  return PVLM_OK;
}

pvlm_status pvlm_ring_batch_scan(const pvlm_ring_batch* b, int scan, pvlm_ring_result* r) {
  if (!b || !r || scan < 0 || scan >= b->n_scans) return PVLM_ERR_ARG;
  const RingScan& sc = b->scans[(size_t)scan];
  std::memset(r, 0, sizeof(*r));
  r->n_raw = sc.n;
  r->n_reordered = b->counts[(size_t)scan * 2]; r->n_kept = b->counts[(size_t)scan * 2 + 1];
  r->resolved_points = b->resolved_points[(size_t)scan]; r->resolved_edges = b->resolved_edges[(size_t)scan]; r->replayed = b->replayed[(size_t)scan];
  r->ring_count_reordered = b->ring_count.data() + (size_t)scan * kMaxRings;
  r->ring_count = b->ring_count2.data() + (size_t)scan * kMaxRings;
  if (b->h_source) {
    r->source = b->h_source + sc.pt0; r->ring_col = b->h_ring_col + sc.pt0;
    if (b->has_arrays[(size_t)scan]) {
      r->curvature = b->h_curvature + sc.pt0; r->half_window = b->h_half + sc.pt0; r->range = b->h_range + sc.pt0;
      r->sorted = b->h_order + sc.pt0; r->sector_host = b->h_sector_host + (size_t)scan * b->rings * 6;
    }
  }
  if (b->picks && b->h_state) {
    const size_t slot0 = (size_t)scan * b->rings;
    r->picks = 1; r->max_curvature = b->max_curvature; r->intersect_angle_threshold = b->angle_threshold;
    r->state = b->h_state + sc.pt0; r->corner = b->h_corner + slot0 * (1 + kCornerSlots); r->flat = b->h_flat + slot0 * (1 + kFlatSlots);
    r->voxel_span = b->h_voxel_span + slot0 * 2; r->voxels = b->h_voxels; r->ring_host = b->h_ring_host + slot0;
  }
  return PVLM_OK;
}

pvlm_status pvlm_ring_debug_sort(pvlm_ctx* ctx, const unsigned* keys, int n, int* order) {
  if (!ctx || !keys || !order || n < 1 || n > 4096) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  unsigned* d_keys = nullptr; int* d_order = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_keys, (size_t)n);
  if (!st) st = pvlm_i_alloc(ctx, &d_order, (size_t)n + 1);
  if (st) { pvlm_i_free(ctx, d_keys); return st; }
  hipError_t e = hipMemcpyAsync(d_keys, keys, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
  int sane = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3(64), 0, ctx->stream, d_keys, n, d_order, d_order + n);
    e = hipMemcpyAsync(order, d_order, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&sane, d_order + n, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  pvlm_i_free(ctx, d_keys); pvlm_i_free(ctx, d_order);
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_ring_debug_sort: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  if (!sane) { PVLM_SET_ERR(ctx, "pvlm_ring_debug_sort: the sort refused the keys"); return PVLM_ERR_STATE; }
  return PVLM_OK;
}

pvlm_status pvlm_ring_batch_timing(const pvlm_ring_batch* b, double* ms8) {
  if (!b || !ms8) return PVLM_ERR_ARG;
  for (int k = 0; k < 8; ++k) ms8[k] = b->ms[k];
  return PVLM_OK;
}

pvlm_status pvlm_ring_batch_fetch(pvlm_ctx* ctx, const pvlm_ring_batch* b, int scan, int state, float* cloud_xyzi, int* ring_col_pairs, float* range_image,
                                  int* image_to_point) {
  if (!ctx || !b || scan < 0 || scan >= b->n_scans || (state != 0 && state != 1)) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const RingScan& sc = b->scans[(size_t)scan];
  const int n = b->counts[(size_t)scan * 2 + state];
  const size_t cells = (size_t)b->rings * b->horizon;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (cloud_xyzi && n > 0) PVLM_HIP(ctx, hipMemcpy(cloud_xyzi, (state ? b->d_cloud2 : b->d_cloud_scan) + sc.pt0, (size_t)n * 16, hipMemcpyDeviceToHost));
  if (ring_col_pairs && n > 0) {
    if (state == 0) PVLM_HIP(ctx, hipMemcpy(ring_col_pairs, b->d_rc + sc.pt0, (size_t)n * 8, hipMemcpyDeviceToHost));
    else for (int i = 0; i < n; ++i) { const int v = b->h_ring_col[sc.pt0 + i]; ring_col_pairs[2 * i] = v >> 16; ring_col_pairs[2 * i + 1] = v & 0xFFFF; }
  }
  if (range_image) PVLM_HIP(ctx, hipMemcpy(range_image, b->d_range_image + sc.cell0, cells * 4, hipMemcpyDeviceToHost));
  if (image_to_point) PVLM_HIP(ctx, hipMemcpy(image_to_point, (state ? b->d_image_to_point2 : b->d_image_to_point) + sc.cell0, cells * 4, hipMemcpyDeviceToHost));
  return PVLM_OK;
}

}  // extern "C"

// pvlm_preload: HIP loads the code object of a translation unit at the first launch of one of its kernels (15 ms for the larger ones) — an empty launch from here
// moves that out of the first call that needs this file's kernels
__global__ void k_preload_ring() {}
void pvlm_i_preload_ring(hipStream_t s) { hipLaunchKernelGGL(k_preload_ring, dim3(1), dim3(1), 0, s); }
