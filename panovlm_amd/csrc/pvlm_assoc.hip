// K1 voxel-hash build, K2 exact k-NN, K3 plane fit + accept tests, ordered compaction:
// the device side of AssociatePoint2Plane (lidar_mapping/LidarFeatureAssociate.cpp:550-630).
// Compiled with -ffp-contract=off: distances (float32, FLANN L2_Simple order) and every
// accept/reject decision (fp64) must match a non-FMA x86-64 build of the reference bit for bit.
//
// Data layout: a cloud is uploaded once per scan; its points are counting-sorted into cells of
// a spatial hash (open addressing, 64-bit cell keys) as float4 (x, y, z, original index) so that
// one 16-byte load fetches a candidate.  A query walks Chebyshev shells of cells around its own
// cell and keeps a register-resident sorted top-k ordered by (distance, index); it stops as soon as
// the k-th distance is provably inside the searched block, or the shell radius exceeds
// dist_threshold.  Results are therefore the EXACT k nearest neighbours (ties by index).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <new>
#include <thread>

#include "pvlm_internal.h"
#include "pvlm_workers.h"
#include "pvlm_assoc_core.h"

using namespace pvlm_assoc;

#define EMPTY_KEY PVLM_EMPTY_KEY

// one descriptor per ordered scan pair of an association batch
struct PairDesc {
  CloudView ref;
  const float* q_xyz;
  const float* q_tag;
  int nq;
  double Rr[9], tr[3], Rn[9], tn[3];  // R_wl / t_wl of ref and nei
  long long tmp_base;                 // first row of this pair in the batch temp arrays
  int chunk_base;                     // first chunk (PVLM_K3_CHUNK queries) of this pair in the batch's chain
  long long dst_row;                  // first row of the pair's segment in the batch's column block (a multiple of 16; the segment has room for nq rows)
};

// ---- K1 ---------------------------------------------------------------------------------------
// exclusive scan of count[T] -> start[T] in three small launches (tile sums, scan of the tile sums by
// one block, per-tile scan + offset): T reaches 4 M cells for dense grids, a single block walking it
// serially cost 220 us per cloud.
#define SCAN_TILE 2048
__global__ __launch_bounds__(256) void k_scan_tile_sums(int T, const int* __restrict__ count, int* __restrict__ tile_sum) {
  __shared__ int red[256];
  const int base = blockIdx.x * SCAN_TILE;
  int s = 0;
  for (int i = threadIdx.x; i < SCAN_TILE; i += 256) if (base + i < T) s += count[base + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(1024) void k_scan_small(int n, int* __restrict__ v) {  // in-place exclusive scan, n <= 1024 * per
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = t * per, hi = min(n, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += v[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int x = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = lo; i < hi; ++i) { const int c = v[i]; v[i] = run; run += c; }
}
__global__ __launch_bounds__(256) void k_scan_tiles(int T, const int* __restrict__ count, const int* __restrict__ tile_off, int* __restrict__ start) {
  __shared__ int part[256];
  const int base = blockIdx.x * SCAN_TILE, t = threadIdx.x;
  const int per = SCAN_TILE / 256;
  int loc[SCAN_TILE / 256];
  int s = 0;
#pragma unroll
  for (int k = 0; k < per; ++k) { const int i = base + t * per + k; loc[k] = i < T ? count[i] : 0; s += loc[k]; }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int x = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  int run = tile_off[blockIdx.x] + part[t] - s;
#pragma unroll
  for (int k = 0; k < per; ++k) { const int i = base + t * per + k; if (i < T) start[i] = run; run += loc[k]; }
}

// ---- K1, batched: the grids of every cloud of an upload batch in three launches -------------------------------------
// Hashed table: open addressing on 64-bit cell keys; dense table: linear cell index (iz*ny + iy)*nx + ix.  Points take
// their slot inside the cell by atomic cursor: the order inside a cell varies from run to run, the k-NN result does not
// (top-k is ordered by (distance, original index)).
struct GridDesc {
  const float* xyz; int n; int dense, nx, ny, nz; int T;
  float ox, oy, oz, inv_h, inv_hx;   // inv_hx: dense tables are xf times finer along x (nx counts the fine cells)
  unsigned long long* keys; int* count; int* start; int* cursor; int* slot; float4* sorted;
  const float* tag; float4* pt4;     // pt4 (clouds with tags): (x, y, z, tag) per point in ORIGINAL order — what K3 gathers by neighbour index
};
struct GridBlock { int cloud, first; };   // 256 points of one cloud

__global__ __launch_bounds__(256) void k_grid_count(const GridDesc* __restrict__ desc, const GridBlock* __restrict__ blocks) {
  const GridBlock b = blocks[blockIdx.x];
  const GridDesc& d = desc[b.cloud];
  const int i = b.first + threadIdx.x;
  if (i >= d.n) return;
  const float x = d.xyz[3 * i], y = d.xyz[3 * i + 1], z = d.xyz[3 * i + 2];
  if (d.dense) {
    const int ix = min(max(cell_of(x, d.ox, d.inv_hx), 0), d.nx - 1), iy = min(max(cell_of(y, d.oy, d.inv_h), 0), d.ny - 1),
              iz = min(max(cell_of(z, d.oz, d.inv_h), 0), d.nz - 1);
    const int c = (iz * d.ny + iy) * d.nx + ix;
    d.slot[i] = c;
    atomicAdd(&d.count[c], 1);
  } else {
    const unsigned long long key = cell_key(cell_of(x, d.ox, d.inv_h), cell_of(y, d.oy, d.inv_h), cell_of(z, d.oz, d.inv_h));
    const int mask = d.T - 1;
    int s = (int)(mix64(key) & (unsigned long long)mask);
    while (true) {
      const unsigned long long prev = atomicCAS(&d.keys[s], EMPTY_KEY, key);
      if (prev == EMPTY_KEY || prev == key) break;
      s = (s + 1) & mask;
    }
    atomicAdd(&d.count[s], 1);
    d.slot[i] = s;
  }
}

// exclusive scan of one cloud's cell counts by one workgroup (tables up to GRID_SCAN_MAX cells; larger ones go through
// launch_scan): tiles of 4096 cells with a running carry
#define GRID_SCAN_MAX (1 << 18)
__global__ __launch_bounds__(1024) void k_grid_scan(const GridDesc* __restrict__ desc, const int* __restrict__ clouds) {
  __shared__ int part[1024];
  __shared__ int carry_s;
  const GridDesc& d = desc[clouds[blockIdx.x]];
  const int t = threadIdx.x, T = d.T;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < T; base += 4096) {
    int loc[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + 4 * t + k; loc[k] = i < T ? d.count[i] : 0; sum += loc[k]; }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int x = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += x;
      __syncthreads();
    }
    const int carry = carry_s;
    int run = carry + part[t] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + 4 * t + k; if (i < T) d.start[i] = run; run += loc[k]; }
    __syncthreads();
    if (t == 1023) carry_s = carry + part[1023];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_grid_scatter(const GridDesc* __restrict__ desc, const GridBlock* __restrict__ blocks) {
  const GridBlock b = blocks[blockIdx.x];
  const GridDesc& d = desc[b.cloud];
  const int i = b.first + threadIdx.x;
  if (i >= d.n) return;
  const int s = d.slot[i];
  const int pos = d.start[s] + atomicAdd(&d.cursor[s], 1);
  const float x = d.xyz[3 * i], y = d.xyz[3 * i + 1], z = d.xyz[3 * i + 2];
  d.sorted[pos] = make_float4(x, y, z, __int_as_float(i));
  if (d.pt4) d.pt4[i] = make_float4(x, y, z, d.tag[i]);
}

// ---- K26: re-posing the resident float clouds (Velodyne::Transform2LidarWorld / Transform2Local, sensors/Velodyne.cpp:1773-1848) ------------------
// pcl::transformPointCloud(cloud, cloud, Matrix4d) per point: every coordinate is float(((m0 x + m1 y) + m2 z) + m3) with the float point promoted to
// double — the statement the host mirror (host/pvlm_host.cpp: TransformCloud) and the oracle use, unfused (-ffp-contract=off), so that the device's
// clouds stay equal to the host's bit for bit through any number of World <-> Local round trips.  In place.  Clouds that carry a voxel grid also
// reduce their bounding box (what cloud_box() computes on the host for an upload) and the first non-finite point: floats as order-preserving
// 32-bit words, one atomic per wave and coordinate.
struct XformCloud { float* xyz; int n; int scan; int box; int pad; };   // box: index of the cloud's 7-word box record, -1 = no grid
struct XformBlock { int cloud, first; };                                // 256 points of one cloud
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float ord2f(unsigned o) { const unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; std::memcpy(&f, &u, 4); return f; }
__global__ __launch_bounds__(256) void k_scan_transform(const XformCloud* __restrict__ clouds, const XformBlock* __restrict__ blocks, const double* __restrict__ T12,
                                                        unsigned* __restrict__ boxes) {
  const XformBlock b = blocks[blockIdx.x];
  const XformCloud c = clouds[b.cloud];
  const double* m = T12 + 12 * (size_t)c.scan;
  const int i = b.first + (int)threadIdx.x;
  unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  int bad = 0x7FFFFFFF;
  if (i < c.n) {
    float* p = c.xyz + 3 * (size_t)i;
    const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = (float)(((m[4 * k] * x + m[4 * k + 1] * y) + m[4 * k + 2] * z) + m[4 * k + 3]);
    p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    if (c.box >= 0) {
      if (!(isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]))) bad = i;
      else {
#pragma unroll
        for (int k = 0; k < 3; ++k) lo[k] = hi[k] = f2ord(o[k]);
      }
    }
  }
  if (c.box < 0) return;                       // uniform per workgroup
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], (unsigned)__shfl_xor((int)lo[k], off, 64)); hi[k] = max(hi[k], (unsigned)__shfl_xor((int)hi[k], off, 64)); }
    bad = min(bad, __shfl_xor(bad, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned* bx = boxes + 7 * (size_t)c.box;
#pragma unroll
    for (int k = 0; k < 3; ++k) { if (lo[k] != 0xFFFFFFFFu) atomicMin(bx + k, lo[k]); if (hi[k] != 0u) atomicMax(bx + 3 + k, hi[k]); }
    if (bad != 0x7FFFFFFF) atomicMin((int*)(bx + 6), bad);
  }
}

// ---- K2 / K3: per-query bodies in pvlm_assoc_core.h (TopK, knn_search, Fit10, world2local) --------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_knn_queries(CloudView cv, const float* __restrict__ q, int nq, float max_dist,
                                                     int* __restrict__ idx, float* __restrict__ sqd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  TopK<K> tk;
  const float thr2 = max_dist * max_dist;
  knn_search<K>(cv, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_dist, thr2, tk);
#pragma unroll
  for (int k = 0; k < K; ++k) { idx[(size_t)i * K + k] = tk.index(k); sqd[(size_t)i * K + k] = tk.dist(k); }
}

// Occupancy targets (waves per SIMD) of the two kernels: A/B-measured with panovlm_amd.build --variant (profiles/r3_assoc_variants.txt)
#ifndef PVLM_K2_WAVES
#define PVLM_K2_WAVES 8     // round 4, after the row-logic diet (76 VGPRs unconstrained): 6 / 7 / 8 waves -> 1313 / 1229 / 1188 us per dispatch (voxel), 6377 / 5974 / 5684 (raw):
                            // the search waits on dependent loads (cell table -> candidates), more resident waves hide more of it than the few spilled registers cost
#endif
#ifndef PVLM_K3_WAVES
#define PVLM_K3_WAVES 2     // 195 VGPRs; 3 waves = 168 VGPRs + 15 spilled doubles
#endif

// K2 — grid: x = chunk of 256 queries, y = pair in batch.  Exact 10-NN of every query; slot 9 is -1
// when fewer than 10 targets lie within dist_threshold (LidarFeatureAssociate.cpp:577).  Kept apart
// from K3 so that the register-hungry fp64 fits do not set the occupancy of the search.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PVLM_K2_WAVES, 8))) void k_knn_pairs(const PairDesc* __restrict__ pairs, float dist_threshold, int* __restrict__ nn_tmp, long long tmp_rows) {
  const PairDesc& pd = pairs[blockIdx.y];
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= pd.nq) return;
  const float qx = pd.q_xyz[3 * q], qy = pd.q_xyz[3 * q + 1], qz = pd.q_xyz[3 * q + 2];
  TopK<10> tk;
  knn_search<10>(pd.ref, qx, qy, qz, dist_threshold, dist_threshold * dist_threshold, tk);
  // neighbour table is column-major (slot k of every query contiguous): coalesced 256-byte stores per wave
#pragma unroll
  for (int k = 0; k < 10; ++k) nn_tmp[(size_t)k * tmp_rows + pd.tmp_base + q] = tk.index(k);
}

// K3 — class test, collinearity test, 10x3 plane fit, and the accepted records written straight into the pair's segment of the residual set,
// in query order (LidarFeatureAssociate.cpp:578-629 emits in query order).
//
// Ordered compaction inside the kernel.  A workgroup fits one chunk of 512 queries (two passes of 256) of one pair at a time; the row of an accepted query is
//     segment start + accepted queries of the pair's earlier chunks + its rank inside the chunk.
// The middle term comes from a chain over the pair's chunks (decoupled look-back): every chunk publishes its own count as soon as its fits are
// done (one 8-byte word {state, count}, relaxed agent-scope atomics on both sides — the word is its own payload, no fence); the counts of the
// chunks before it are summed back to the nearest one that already holds an inclusive prefix (the first wave reads 64 words per round trip),
// and the chunk publishes its own inclusive prefix.  The last chunk of a pair leaves the pair's total for the host.
// A workgroup must not WAIT for that sum with its registers allocated: fits take anything from a class test to a full QR, and at two
// workgroups per CU a finished chunk idling behind a slow predecessor cost 40 % (1820 against 1296 us per dispatch without the chain).  So
// the workgroups are persistent, take chunks from a ticket counter (the chunks a chunk depends on have always been taken), park the records
// of the chunk just fitted in LDS (two buffers of 28 KB), fit the NEXT chunk, and only then place the parked one — by then its predecessors have long
// published.
// Round 4 wrote the records of EVERY query to scratch (56 B), copied the chunk counts to the host, sized the block there, uploaded the
// destinations and compacted in a second kernel (k_compact: 282 us of the 2.7 ms per 16.7 M queries, HBM-bound on the round trip of the records);
// now the segment of a pair has room for all of its queries (the set costs what the queries cost, not what was accepted: 7.5 instead of 5.6 GB
// for the bench's 134 M queries) and the host only learns the per-pair totals.
#define PVLM_CHAIN_AGG (1ull << 62)
#define PVLM_CHAIN_INC (1ull << 63)
#ifndef PVLM_K3_SUB
#define PVLM_K3_SUB 2                          // exact kernel: a chunk = PVLM_K3_SUB x 256 queries: one ticket, one chain word, one look-back and two barriers per chunk
#endif
#ifndef PVLM_K3F_SUB
#define PVLM_K3F_SUB 2                         // fast kernel: measured below
#endif
// EXACT = false (the default of pvlm_assoc_point2plane): the plane of a query comes from Fit10::form_plane_fast — normal equations + an a-posteriori bound, a
// quarter of the QR's instructions and half of its registers — whenever that routine can certify the reference's accept / reject decision; the few
// queries it refuses (the largest distance within ~1e-6 of the tolerance, ill-conditioned neighbourhoods) take the QR, counted in ticket[1].
// EXACT = true (PVLM_FLAG_ASSOC_EXACT_FIT): the QR for every query — records bit-identical to a non-FMA x86-64 build of the reference.
#ifndef PVLM_K3F_WAVES
#define PVLM_K3F_WAVES 3
#endif
template <bool EXACT, int SUB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EXACT ? PVLM_K3_WAVES : PVLM_K3F_WAVES, 8))) void k_fit_pairs(const PairDesc* __restrict__ pairs, double plane_tol, const int* __restrict__ nn_tmp, long long tmp_rows,
                                                   double* __restrict__ cols, long long n_dev, unsigned long long* __restrict__ chain, int* __restrict__ pair_count,
                                                   int* __restrict__ ticket, int* __restrict__ qidx_out, int* __restrict__ nn_out, int chunks_x, int total) {
  constexpr int PVLM_K3_CHUNK = 256 * SUB;
  static_assert(SUB == 1 || SUB == 2, "the sub-chunk loop below keeps its per-pass results in slots 0 and SUB - 1");
  __shared__ int s_vid, s_vid2;
  __shared__ int wc[SUB][4];
  __shared__ long long s_prefix;
  __shared__ double s_rec[2][SUB][4][256];      // two buffers: the plane coefficients of the chunk being fitted and of the parked one (the query's own
                                                // local coordinates — three more doubles — are recomputed when the row is placed: 18 flops against 12 KB of LDS per buffer)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // the parked chunk: wave-uniform (pair, chunk, count, buffer) and per thread and sub-chunk (accepted, rank inside the chunk)
  int prev_pair = -1, prev_chunk = 0, prev_count = 0, prev_buf = 0;
  int prev_rank[SUB];
  bool prev_accept[SUB];
#pragma unroll
  for (int u = 0; u < SUB; ++u) { prev_rank[u] = 0; prev_accept[u] = false; }
  // rows of the parked chunk: look-back, inclusive prefix, records out of LDS
  auto place = [&]() {
    const PairDesc& pp = pairs[prev_pair];
    if (wv == 0) {
      unsigned long long* my = chain + pp.chunk_base + prev_chunk;
      long long prefix = 0;
      if (prev_chunk > 0) {
        for (int base = prev_chunk - 1;; base -= 64) {
          const int c = base - lane;
          unsigned long long w = PVLM_CHAIN_INC;                 // before the pair's first chunk: an inclusive prefix of zero
          if (c >= 0)
            while (((w = __hip_atomic_load(chain + pp.chunk_base + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & (PVLM_CHAIN_AGG | PVLM_CHAIN_INC)) == 0ull) __builtin_amdgcn_s_sleep(1);
          const unsigned long long inc = __ballot((w & PVLM_CHAIN_INC) != 0ull);   // never empty in the last window (c < 0 lanes)
          const int first = inc ? __builtin_ctzll(inc) : 64;
          long long v = lane <= first ? (long long)(w & 0xFFFFFFFFull) : 0ll;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
          prefix += v;
          if (inc) break;
        }
        if (lane == 0) __hip_atomic_store(my, PVLM_CHAIN_INC | (unsigned long long)(prefix + prev_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        if ((prev_chunk + 1) * PVLM_K3_CHUNK >= pp.nq) pair_count[prev_pair] = (int)(prefix + prev_count);
        s_prefix = prefix;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SUB; ++u) {
      if (prev_accept[u]) {
        const long long d = pp.dst_row + s_prefix + prev_rank[u];
        const int q = prev_chunk * PVLM_K3_CHUNK + u * 256 + (int)threadIdx.x;
        double pl[3];
        world2local(pp.Rn, pp.tn, (double)pp.q_xyz[3 * q], (double)pp.q_xyz[3 * q + 1], (double)pp.q_xyz[3 * q + 2], pl);
#pragma unroll
        for (int c = 0; c < 3; ++c) cols[(size_t)c * n_dev + d] = pl[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) cols[(size_t)(3 + c) * n_dev + d] = s_rec[prev_buf][u][c][threadIdx.x];
        if (qidx_out) {
          qidx_out[d] = q;
#pragma unroll
          for (int k = 0; k < 10; ++k) nn_out[d * 10 + k] = nn_tmp[(size_t)k * tmp_rows + pp.tmp_base + q];
        }
      }
    }
    __syncthreads();   // s_prefix is free again
  };
  int buf = 0;
  int n_refused = 0;                         // queries of this thread the fast fit left to the QR: summed per wave when the workgroup retires
  // ---- one query: the row of the neighbour table, the ten neighbours, the fits -------------------------------------------------------------------------
  auto fit = [&](const PairDesc& pd, int q, int u) -> bool {
    if (q >= pd.nq) return false;
    const long long row = pd.tmp_base + q;
    int id[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) id[k] = nn_tmp[(size_t)k * tmp_rows + row];
    if (id[9] < 0) return false;                                           // fewer than ten neighbours in reach (:577)
    const float qtag = pd.q_tag[q];
    double Rt[3], plane[4];
    world2local_rt(pd.Rr, pd.tr, Rt);
    bool ok;
    bool decided = false;
    if (!EXACT) {
      // The fast kernel keeps no point: it gathers the ten neighbours TWICE (the second time out of the cache) — first into the raw second moments, from which the
      // collinearity screen and the normal-equation solve come, then into the residuals of that solve, from which the certified accept / reject decision comes
      // (Fit10::FastFit).  Sixty registers less than the arrays: twice the resident waves for a kernel that waited for its gathers two thirds of its cycles.  Whatever the
      // moments cannot decide — the screen inside its guard band, an ill-conditioned system, a largest distance within ~1e-6 of the tolerance — takes the exact path below.
      Fit10::FastFit F;
      int same = 0;
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const Point4 t = pd.ref.pt4[id[k]];                                // (x, y, z, tag) of the neighbour: one 16-byte gather
        same += (t.w == qtag);
        double l[3];
        world2local_pt(pd.Rr, Rt, (double)t.x, (double)t.y, (double)t.z, l);
        F.add(l[0], l[1], l[2]);
      }
      if (same != 10) return false;  // :583-591
      const int line = F.collinear(3.0);
      if (line == 1) return false;   // :592-596: collinear neighbourhoods are rejected
      if (line == 0 && F.solve(plane_tol)) {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const Point4 t = pd.ref.pt4[id[k]];
          double l[3];
          world2local_pt(pd.Rr, Rt, (double)t.x, (double)t.y, (double)t.z, l);
          F.residual(l[0], l[1], l[2]);
        }
        const int fast = F.decide(plane_tol, plane);
        if (fast >= 0) { ok = fast != 0; decided = true; }
      }
      if (!decided) ++n_refused;
    }
    if (!decided) {
      // the reference's own arithmetic: the ten points in the reference scan's frame, FormLine's eigen decision, the pivoted Householder QR
      double px[10], py[10], pz[10];
      int same = 0;
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const Point4 t = pd.ref.pt4[id[k]];                                // (x, y, z, tag) of the neighbour: one 16-byte gather (round 4: four 4-byte ones)
        same += (t.w == qtag);
        double l[3];
        world2local_pt(pd.Rr, Rt, (double)t.x, (double)t.y, (double)t.z, l);
        px[k] = l[0]; py[k] = l[1]; pz[k] = l[2];
      }
      if (same != 10) return false;  // :583-591
      // :592-596 accepts when the plane fits AND the ten points are not collinear.  Both tests are side-effect free, so the
      // cheap one runs first: the scatter matrix + closed-form screen is ~250 flops, the 10x3 pivoted QR ~2 000 instructions,
      // and with raw scans as targets 94 % of the queries die at the collinearity test (ten neighbours along one ring).
      // A wave whose lanes are all collinear never enters the QR.  (Re-packing the survivors of a workgroup so that whole waves skip the QR
      // was built and measured: slower, 1394 vs 1322 us voxel, 1599 vs 1316 us raw; profiles/r4_assoc_variants.txt.)
      if (Fit10::is_line(px, py, pz, 3.0)) return false;
      if (EXACT) ok = Fit10::form_plane(px, py, pz, plane_tol, plane);
      else {
        // the QR in place on the coordinate arrays, the ten points fetched again for the accept test: the fall-back does not set the
        // register budget of the kernel (form_plane keeps 70 doubles alive)
        double x[3];
        Fit10::form_plane_solve(px, py, pz, x);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const Point4 t = pd.ref.pt4[id[k]];
          double l[3];
          world2local_pt(pd.Rr, Rt, (double)t.x, (double)t.y, (double)t.z, l);
          px[k] = l[0]; py[k] = l[1]; pz[k] = l[2];
        }
        ok = Fit10::form_plane_accept(x, px, py, pz, plane_tol, plane);
      }
    }
    if (ok) {
      double* r = &s_rec[buf][u][0][threadIdx.x];           // straight into this chunk's buffer (the parked chunk sits in the other one)
      r[0 * 256] = plane[0]; r[1 * 256] = plane[1]; r[2 * 256] = plane[2]; r[3 * 256] = plane[3];
    }
    return ok;
  };
  // Tickets are drawn ahead by thread 0: the atomic's latency is off the path.  (A software pipeline over the queries — the ten gathers of query k + 1 and the
  // table row of query k + 2 in flight while query k is fitted — was built for the fast kernel, which waits for those two dependent loads two thirds of its
  // cycles (SQ_WAIT_ANY 0.66, SIMDs 45 % busy): 100 more live registers, spills inside the loop, 1261 against 931 us per dispatch.  Removed.)
  int ahead = 0, cur_vid, nxt_vid;
  if (threadIdx.x == 0) { s_vid = atomicAdd(ticket, 1); s_vid2 = atomicAdd(ticket, 1); ahead = atomicAdd(ticket, 1); }
  __syncthreads();
  cur_vid = __builtin_amdgcn_readfirstlane(s_vid); nxt_vid = __builtin_amdgcn_readfirstlane(s_vid2);
  __syncthreads();
  for (;;) {
    if (cur_vid >= total) break;
    const int pair = cur_vid / chunks_x, chunk = cur_vid - pair * chunks_x;
    const PairDesc& pd = pairs[pair];
    const bool live = chunk * PVLM_K3_CHUNK < pd.nq;        // the launch covers chunks_x chunks per pair: a short pair leaves empty ones behind
    bool accept[SUB];
    unsigned long long bal[SUB];
#pragma unroll 1
    for (int u = 0; u < SUB; ++u) {
      const bool ok = live && fit(pd, chunk * PVLM_K3_CHUNK + u * 256 + (int)threadIdx.x, u);
      if (u == 0) { accept[0] = ok; bal[0] = __ballot(ok); if (lane == 0) wc[0][wv] = __popcll(bal[0]); }
      else { accept[SUB - 1] = ok; bal[SUB - 1] = __ballot(ok); if (lane == 0) wc[SUB - 1][wv] = __popcll(bal[SUB - 1]); }
    }
    if (threadIdx.x == 0) { s_vid = ahead; ahead = atomicAdd(ticket, 1); }      // the chunk after the next one
    __syncthreads();
    cur_vid = nxt_vid; nxt_vid = __builtin_amdgcn_readfirstlane(s_vid);
    if (!live) { __syncthreads(); continue; }                                     // (s_vid is read before the next round overwrites it)
    int count = 0, rank[SUB];
#pragma unroll
    for (int u = 0; u < SUB; ++u) {
      int base = count;
      for (int w = 0; w < wv; ++w) base += wc[u][w];
      rank[u] = base + __popcll(bal[u] & ((1ull << lane) - 1ull));
      count += wc[u][0] + wc[u][1] + wc[u][2] + wc[u][3];
    }
    // the chunk's own count is public at once: a first chunk's is its inclusive prefix
    if (threadIdx.x == 0)
      __hip_atomic_store(chain + pd.chunk_base + chunk, (chunk == 0 ? PVLM_CHAIN_INC : PVLM_CHAIN_AGG) | (unsigned long long)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev_pair >= 0) place();           // ends with a barrier: wc and s_vid are free
    else __syncthreads();
    prev_pair = pair; prev_chunk = chunk; prev_count = count; prev_buf = buf;
#pragma unroll
    for (int u = 0; u < SUB; ++u) { prev_rank[u] = rank[u]; prev_accept[u] = accept[u]; }
    buf ^= 1;
  }
  if (prev_pair >= 0) place();
  if (!EXACT) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_refused += __shfl_xor(n_refused, off, 64);
    if (lane == 0 && n_refused) atomicAdd(ticket + 1, n_refused);
  }
}

// A sparse column block made dense: the accepted rows of every pair copied into a block that holds only them (segment p: rows [dst[p], dst[p] + count[p])).
// With raw scans as targets 94 % of the queries are rejected: the block of a batch reserves a row per QUERY (the ordered placement needs no sizing round trip)
// and would keep sixteen times the memory it uses for as long as the residual set lives.
struct CompactSeg { long long src, dst; int count, pad; };
__global__ __launch_bounds__(256) void k_compact_block(const CompactSeg* __restrict__ segs, const double* __restrict__ src, long long src_rows, double* __restrict__ dst, long long dst_rows) {
  const CompactSeg sg = segs[blockIdx.y];
  for (int r = blockIdx.x * 256 + threadIdx.x; r < sg.count; r += gridDim.x * 256)
#pragma unroll
    for (int c = 0; c < 7; ++c) dst[(size_t)c * dst_rows + sg.dst + r] = src[(size_t)c * src_rows + sg.src + r];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static void launch_scan(pvlm_ctx* ctx, int T, const int* count, int* start, int* tiles) {
  const int nt = (T + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(nt), dim3(256), 0, ctx->stream, T, count, tiles);
  hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, nt, tiles);
  hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(256), 0, ctx->stream, T, count, tiles, start);
}

// build-time scratch, released on every exit path
struct DevScratch {
  pvlm_ctx* ctx;
  std::vector<void*> ptrs;
  explicit DevScratch(pvlm_ctx* c) : ctx(c) {}
  ~DevScratch() { for (void* p : ptrs) pvlm_i_free(ctx, p); }
  template <typename T> pvlm_status alloc(T** p, size_t count) {
    const pvlm_status st = pvlm_i_alloc(ctx, p, count);
    if (!st) ptrs.push_back(*p);
    return st;
  }
};

// largest cloud the voxel grid addresses with 32-bit cell tables (hashed table: 2 n slots rounded up to a power of two)
#define PVLM_MAX_CLOUD_POINTS (256 << 20)

// ---- scan upload: ONE staging copy, ONE slab and ONE set of grid-build launches for a whole batch of scans -----------
// (round 1 queued 7 pageable copies, 6 memsets and 10 kernels per scan and synchronised after each: 0.48 ms per scan,
// 0.22 s of a Room-scale EstimatePose that re-uploads its 454 re-posed scans at every outer iteration)
struct CloudPlan {
  int n = 0;
  const float* xyz = nullptr; const float* tag = nullptr;
  int xyz_stride = 3, tag_stride = 1;            // floats between consecutive points / tags of the CALLER's arrays (pvlm_scan_desc::point_stride_floats)
  bool grid = false;
  float h = 0.f, origin[3] = {0, 0, 0};
  int dense = 0, nx = 0, ny = 0, nz = 0, xf = 1;
  long long T = 0;
  // byte offsets: the points slab (o_xyz, o_tag), the grid slab (o_count .. o_pt4) and the build scratch (s_*)
  size_t o_xyz = 0, o_tag = 0, o_count = 0, o_keys = 0, o_start = 0, o_sorted = 0, o_pt4 = 0, s_cursor = 0, s_slot = 0;
};

// bounding box of a cloud and its first non-finite point (a NaN never updates a min / max, so every coordinate is tested): the one pass
// over the points that planning needs — taken for all clouds of a batch side by side (pvlm_scan_upload_batch), reported serially
struct CloudBox { float mn[3], mx[3]; int bad_point; };
static void cloud_box(int n, const float* xyz, CloudBox& b, int stride = 3) {
  for (int k = 0; k < 3; ++k) { b.mn[k] = FLT_MAX; b.mx[k] = -FLT_MAX; }
  b.bad_point = -1;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      const float v = xyz[(size_t)stride * i + k];
      if (!std::isfinite(v)) { b.bad_point = i; return; }
      if (v < b.mn[k]) b.mn[k] = v;
      if (v > b.mx[k]) b.mx[k] = v;
    }
}
static pvlm_status cloud_plan(pvlm_ctx* ctx, CloudPlan& c, int n, const float* xyz, const float* tag, bool grid, const CloudBox* box = nullptr) {
  c.n = n; c.xyz = xyz; c.tag = tag; c.grid = grid && n > 0;
  if (n <= 0) return PVLM_OK;
  if (n > PVLM_MAX_CLOUD_POINTS) { PVLM_SET_ERR(ctx, "cloud of %d points exceeds the supported maximum of %d", n, PVLM_MAX_CLOUD_POINTS); return PVLM_ERR_ARG; }
  CloudBox own;
  if (!box) { cloud_box(n, xyz, own); box = &own; }
  if (box->bad_point >= 0) { PVLM_SET_ERR(ctx, "cloud contains a non-finite coordinate (point %d)", box->bad_point); return PVLM_ERR_ARG; }
  const float* mn = box->mn; const float* mx = box->mx;
  if (!c.grid) return PVLM_OK;
  // cell edge: surface-like clouds, aim at ~4 points per occupied cell
  float e[3];
  for (int k = 0; k < 3; ++k) { if (!(mx[k] - mn[k] < 1e7f)) { PVLM_SET_ERR(ctx, "cloud extent exceeds 1e7"); return PVLM_ERR_ARG; } e[k] = std::max(mx[k] - mn[k], 0.05f); }
  const float area = 2.f * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
  // 0.65 x the edge that gives 4 points per occupied cell: measured optimum of the pruned search (profiles/r3_assoc_variants.txt:
  // 0.5 / 0.65 / 0.8 / 1.0 -> 13.2 / 11.9 / 12.6 / 12.9 ms per 67 M queries; round 2's unpruned search sat on a flat optimum at 1.0)
  float h = 0.65f * std::sqrt(4.f * area / (float)n);
  const char* env = getenv("PVLM_CELL");
  if (env && atof(env) > 0) h = (float)atof(env);
  if (const char* sc = getenv("PVLM_CELL_SCALE")) if (atof(sc) > 0) h *= (float)atof(sc);     // measured variants of the heuristic
  h = std::min(std::max(h, 0.02f), 4.0f);
  c.h = h;
  for (int k = 0; k < 3; ++k) c.origin[k] = mn[k] - h;
  const float inv_h = 1.0f / h;
  long long dims[3];
  for (int k = 0; k < 3; ++k) dims[k] = (long long)std::ceil((mx[k] - c.origin[k]) * inv_h) + 2;
  const long long ncells = dims[0] * dims[1] * dims[2];
  const bool dense = ncells <= std::max<long long>(64ll * n, 4096) && ncells <= (4ll << 20) && !getenv("PVLM_FORCE_HASH");
  if (dense) {
    // cells xf times finer along x, the direction in which a row is one contiguous run (pvlm_assoc_core.h): as fine as the
    // table limits allow, 4 by default (PVLM_CELL_XF)
    int xf = 4;
    if (const char* e = getenv("PVLM_CELL_XF")) xf = std::min(std::max(atoi(e), 1), 16);
    while (xf > 1 && !(ncells * xf <= std::max<long long>(64ll * n, 4096) && ncells * xf <= (4ll << 20))) --xf;
    c.xf = xf;
    c.T = ncells * xf + 1; c.dense = 1; c.nx = (int)dims[0] * xf; c.ny = (int)dims[1]; c.nz = (int)dims[2];
  } else { c.T = 1024; while (c.T < 2ll * n) c.T <<= 1; }
  return PVLM_OK;
}

struct ScanPlan {
  CloudPlan flat, less, corner;
  size_t o_p2s_off = 0, o_p2s_ids = 0, o_seg_xyz = 0;
  size_t n_p2s_ids = 0, n_seg_pts = 0;
};

static CloudView view_of(const pvlm_cloud& c);

static CloudView view_of(const pvlm_cloud& c) {
  CloudView v;
  v.sorted = reinterpret_cast<const Point4*>(c.d_sorted); v.keys = c.d_keys; v.cell_start = c.d_cell_start; v.cell_count = c.d_cell_count;
  v.xyz = c.d_xyz; v.tag = c.d_tag; v.pt4 = reinterpret_cast<const Point4*>(c.d_pt4); v.n = c.n; v.mask = c.table_size - 1;
  v.dense = c.dense; v.nx = c.nx; v.ny = c.ny; v.nz = c.nz; v.xf = c.dense ? std::max(c.xf, 1) : 1;
  v.ox = c.origin[0]; v.oy = c.origin[1]; v.oz = c.origin[2]; v.h = c.cell; v.inv_h = c.cell > 0 ? 1.0f / c.cell : 0.f;
  return v;
}

// (re)sizes the two pipeline slots of the association scratch; grow-only, so a steady state allocates nothing
static pvlm_status assoc_ws_ensure(pvlm_ctx* ctx, long long rows, int chunks, int pairs) {
  pvlm_assoc_ws& w = ctx->assoc_ws;
  rows = std::max<long long>(rows, 1); chunks = std::max(chunks, 1); pairs = std::max(pairs, 1);
  if (w.rows >= rows && w.chunks >= chunks && w.pairs >= pairs) return PVLM_OK;
  rows = std::max(rows, w.rows); chunks = std::max(chunks, w.chunks); pairs = std::max(pairs, w.pairs);
  PVLM_TRY_SYNC(ctx);
  pvlm_i_assoc_ws_free(ctx);
  pvlm_status st = PVLM_OK;
  for (int s = 0; s < 2 && !st; ++s) {
    if (!st) st = pvlm_i_alloc(ctx, &w.d_nn[s], (size_t)rows * 10);
    if (!st) st = pvlm_i_alloc(ctx, &w.d_chain[s], (size_t)chunks + (size_t)pairs / 2 + 3);
    if (!st) st = pvlm_i_alloc_bytes(ctx, &w.d_desc[s], (size_t)pairs * sizeof(PairDesc));
    if (!st && (hipHostMalloc((void**)&w.h_count[s], ((size_t)pairs + 2) * sizeof(int), hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc(&w.h_desc[s], (size_t)pairs * sizeof(PairDesc), hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&w.ev[s], hipEventDisableTiming) != hipSuccess)) {
      PVLM_SET_ERR(ctx, "association staging: pinned host allocation failed");
      st = PVLM_ERR_NOMEM;
    }
  }
  if (st) { pvlm_i_assoc_ws_free(ctx); return st; }
  w.rows = rows; w.chunks = chunks; w.pairs = pairs; w.desc_bytes = (size_t)pairs * sizeof(PairDesc);
  return PVLM_OK;
}

// ---- K1, host side: the voxel grids of a list of device-resident clouds — ONE slab, one table copy, three memsets, three launches (+ the long
// tables), one synchronisation.  Shared by the upload (clouds just copied up) and by the re-pose (pvlm_scan_transform_batch: clouds transformed
// in place, bounding boxes reduced on the device).  Grid slab: [descriptors | point blocks | short-table list | cell counts (zeroed) | hash keys
// (0xFF) | cell starts, sorted points, pt4];  scratch: [cursors (zeroed) | cell / slot of every point].  The plans' o_count .. o_pt4 are offsets into
// the slab returned in *out_slab (null when no cloud has a grid).
struct GridJob { CloudPlan* plan; const float* d_xyz; const float* d_tag; };
static pvlm_status grids_build(pvlm_ctx* ctx, const std::vector<GridJob>& jobs, char** out_slab) {
  *out_slab = nullptr;
  const int n_grids = (int)jobs.size();
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  auto place = [&](size_t& off, size_t& cursor, size_t bytes) { off = cursor; cursor = align(cursor + bytes); };
  if (n_grids == 0) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { PVLM_SET_ERR(ctx, "scan upload: device error"); return PVLM_ERR_HIP; }
    return PVLM_OK;
  }
  size_t n_blocks = 0, n_small = 0;
  for (const GridJob& j : jobs) { n_blocks += ((size_t)j.plan->n + 255) / 256; n_small += j.plan->T <= GRID_SCAN_MAX; }
  size_t slab = 0, scratch = 0, o_desc = 0, o_blocks = 0, o_small = 0;
  place(o_desc, slab, (size_t)n_grids * sizeof(GridDesc));
  place(o_blocks, slab, std::max<size_t>(n_blocks, 1) * sizeof(GridBlock));
  place(o_small, slab, std::max<size_t>(n_small, 1) * sizeof(int));
  const size_t tables_bytes = slab, o_count0 = slab;
  for (const GridJob& j : jobs) place(j.plan->o_count, slab, (size_t)j.plan->T * 4);
  const size_t count_bytes = slab - o_count0, o_keys0 = slab;
  for (const GridJob& j : jobs) if (!j.plan->dense) place(j.plan->o_keys, slab, (size_t)j.plan->T * 8);
  const size_t keys_bytes = slab - o_keys0;
  for (const GridJob& j : jobs) {
    CloudPlan& c = *j.plan;
    place(c.o_start, slab, (size_t)c.T * 4);
    place(c.o_sorted, slab, ((size_t)c.n + 1) * 16);          // + 1: the search may read (never use) one record past a run
    if (j.d_tag) place(c.o_pt4, slab, (size_t)c.n * 16);
  }
  for (const GridJob& j : jobs) place(j.plan->s_cursor, scratch, (size_t)j.plan->T * 4);
  const size_t cursor_bytes = scratch;
  for (const GridJob& j : jobs) place(j.plan->s_slot, scratch, (size_t)j.plan->n * 4);
  // pinned mirror of the tables (grow-only; idle here: every build ends with a synchronisation)
  if (ctx->grid_bytes < tables_bytes) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { PVLM_SET_ERR(ctx, "voxel-grid build: device error"); return PVLM_ERR_HIP; }
    if (ctx->h_grid) (void)hipHostFree(ctx->h_grid);
    ctx->h_grid = nullptr; ctx->grid_bytes = 0;
    const size_t want = tables_bytes + tables_bytes / 2;
    if (hipHostMalloc(&ctx->h_grid, want, hipHostMallocDefault) != hipSuccess) { PVLM_SET_ERR(ctx, "voxel-grid build: %zu bytes of pinned staging unavailable", want); return PVLM_ERR_NOMEM; }
    ctx->grid_bytes = want;
  }
  char* d_slab = nullptr; char* d_scr = nullptr;
  pvlm_status st = pvlm_i_alloc_bytes(ctx, (void**)&d_slab, std::max<size_t>(slab, 256));
  if (st) { (void)hipStreamSynchronize(ctx->stream); return st; }
  if ((st = pvlm_i_alloc_bytes(ctx, (void**)&d_scr, std::max<size_t>(scratch, 256)))) { (void)hipStreamSynchronize(ctx->stream); pvlm_i_free(ctx, d_slab); return st; }
  auto bail = [&](pvlm_status e) { (void)hipStreamSynchronize(ctx->stream); pvlm_i_free(ctx, d_slab); pvlm_i_free(ctx, d_scr); return e; };
  char* h = (char*)ctx->h_grid;
  GridDesc* hd = (GridDesc*)(h + o_desc); GridBlock* hb = (GridBlock*)(h + o_blocks); int* hs = (int*)(h + o_small);
  std::vector<int> big;                                         // tables too long for one workgroup: scanned by launch_scan
  size_t nb = 0; int ns = 0;
  for (int g = 0; g < n_grids; ++g) {
    const CloudPlan& c = *jobs[(size_t)g].plan;
    GridDesc& D = hd[g];
    D.xyz = jobs[(size_t)g].d_xyz; D.n = c.n; D.dense = c.dense; D.nx = c.nx; D.ny = c.ny; D.nz = c.nz; D.T = (int)c.T;
    D.ox = c.origin[0]; D.oy = c.origin[1]; D.oz = c.origin[2]; D.inv_h = 1.0f / c.h; D.inv_hx = D.inv_h * (float)c.xf;
    D.keys = c.dense ? nullptr : (unsigned long long*)(d_slab + c.o_keys);
    D.count = (int*)(d_slab + c.o_count); D.start = (int*)(d_slab + c.o_start); D.sorted = (float4*)(d_slab + c.o_sorted);
    D.cursor = (int*)(d_scr + c.s_cursor); D.slot = (int*)(d_scr + c.s_slot);
    D.tag = jobs[(size_t)g].d_tag; D.pt4 = D.tag ? (float4*)(d_slab + c.o_pt4) : nullptr;
    for (int f = 0; f < c.n; f += 256) hb[nb++] = GridBlock{g, f};
    if (c.T <= GRID_SCAN_MAX) hs[ns++] = g; else big.push_back(g);
  }
  hipError_t e = hipMemcpyAsync(d_slab, h, tables_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && count_bytes) e = hipMemsetAsync(d_slab + o_count0, 0, count_bytes, ctx->stream);
  if (e == hipSuccess && keys_bytes) e = hipMemsetAsync(d_slab + o_keys0, 0xFF, keys_bytes, ctx->stream);
  if (e == hipSuccess && cursor_bytes) e = hipMemsetAsync(d_scr, 0, cursor_bytes, ctx->stream);
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "voxel-grid build: copy / memset failed: %s", hipGetErrorString(e)); return bail(PVLM_ERR_HIP); }
  DevScratch tiles_scratch(ctx);
  if (nb) {
    const GridDesc* dd = (const GridDesc*)(d_slab + o_desc); const GridBlock* db = (const GridBlock*)(d_slab + o_blocks);
    hipLaunchKernelGGL(k_grid_count, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dd, db);
    if (ns) hipLaunchKernelGGL(k_grid_scan, dim3((unsigned)ns), dim3(1024), 0, ctx->stream, dd, (const int*)(d_slab + o_small));
    for (int g : big) {
      const CloudPlan& c = *jobs[(size_t)g].plan;
      int* d_tiles = nullptr;
      if ((st = tiles_scratch.alloc(&d_tiles, (size_t)((c.T + SCAN_TILE - 1) / SCAN_TILE) + 1))) return bail(st);
      launch_scan(ctx, (int)c.T, (const int*)(d_slab + c.o_count), (int*)(d_slab + c.o_start), d_tiles);
    }
    hipLaunchKernelGGL(k_grid_scatter, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dd, db);
    e = hipGetLastError();
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "voxel-grid build failed: %s", hipGetErrorString(e)); return bail(PVLM_ERR_HIP); }
  }
  pvlm_i_trace("grids_build: tables + kernels queued");
  // the staging buffers are reused by the next call and the scratch goes back to the pool: wait once for the whole batch
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { PVLM_SET_ERR(ctx, "voxel-grid build: device error"); return bail(PVLM_ERR_HIP); }
  pvlm_i_free(ctx, d_scr);
  pvlm_i_trace("grids_build: synchronised");
  *out_slab = d_slab;
  return PVLM_OK;
}
// points a cloud's grid members into the grid slab its plan was laid out in
static void grid_bind(pvlm_cloud& c, const CloudPlan& p, char* d_grid, bool has_tag) {
  c.cell = p.h; for (int q = 0; q < 3; ++q) c.origin[q] = p.origin[q];
  c.table_size = (int)p.T; c.dense = p.dense; c.nx = p.nx; c.ny = p.ny; c.nz = p.nz; c.xf = p.xf;
  c.d_keys = p.dense ? nullptr : (unsigned long long*)(d_grid + p.o_keys);
  c.d_cell_start = (int*)(d_grid + p.o_start); c.d_cell_count = (int*)(d_grid + p.o_count); c.d_sorted = (float4*)(d_grid + p.o_sorted);
  c.d_pt4 = has_tag ? (float4*)(d_grid + p.o_pt4) : nullptr;
  c.grid_stale = false;
}
static void slab_unref(pvlm_ctx* ctx, pvlm_scan_slab*& sl) {
  if (sl && --sl->refs == 0) { pvlm_i_free(ctx, sl->base); delete sl; }
  sl = nullptr;
}

// K28 — the searches of FindNeighbors for all scans: row i = the centres ordered by (float squared distance from centre i, position).  A workgroup per row: the keys of
// the row in LDS (P = the next power of two >= n, the tail padded with ~0), a bitonic sort (P / 2 compare-exchanges per step, log2 P (log2 P + 1) / 2 steps), the
// positions of the first n keys out (2 bytes each: 5 MB for the 1593 scans of Floor where the keys themselves are 20 — the caller recomputes a distance it needs).  The distance is the reference's float chain without contraction: ((dx*dx) + dy*dy) + dz*dz.
template <int P>
__global__ __launch_bounds__(1024) void k_centre_orders(int n, const float* __restrict__ xyz, unsigned short* __restrict__ order_out) {
  __shared__ unsigned long long key[P];
  const int i = blockIdx.x, t = threadIdx.x;
  const float qx = xyz[3 * i], qy = xyz[3 * i + 1], qz = xyz[3 * i + 2];
  for (int j = t; j < P; j += 1024) {
    unsigned long long k = ~0ull;
    if (j < n) {
      const float dx = __fsub_rn(qx, xyz[3 * j]), dy = __fsub_rn(qy, xyz[3 * j + 1]), dz = __fsub_rn(qz, xyz[3 * j + 2]);
      float s2 = __fmul_rn(dx, dx);
      s2 = __fadd_rn(s2, __fmul_rn(dy, dy));
      s2 = __fadd_rn(s2, __fmul_rn(dz, dz));
      k = ((unsigned long long)__float_as_uint(s2) << 32) | (unsigned)j;
    }
    key[j] = k;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int p = t; p < P / 2; p += 1024) {
        const int lo = 2 * p - (p & (stride - 1));          // the lower partner of the p-th pair of this step
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = key[lo], b = key[hi];
        if ((a > b) == up) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  for (int j = t; j < n; j += 1024) order_out[(size_t)i * n + j] = (unsigned short)(key[j] & 0xffffu);     // the positions in key order: 2 bytes each (n <= 4096)
}

extern "C" {

pvlm_status pvlm_centre_orders(pvlm_ctx* ctx, int n, const float* xyz, uint16_t* order) {
  if (!ctx || n < 0 || (n > 0 && (!xyz || !order))) return PVLM_ERR_ARG;
  if (n == 0) return PVLM_OK;
  if (n > 4096) { PVLM_SET_ERR(ctx, "pvlm_centre_orders: %d centres (at most 4096)", n); return PVLM_ERR_CAPACITY; }
  for (int j = 0; j < 3 * n; ++j) if (!(xyz[j] == xyz[j]) || std::isinf(xyz[j])) { PVLM_SET_ERR(ctx, "pvlm_centre_orders: a centre is not finite"); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t bytes = (size_t)n * n * sizeof(unsigned short);
  float* d_xyz = nullptr; unsigned short* d_keys = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_xyz, (size_t)3 * n);
  if (!st) st = pvlm_i_alloc(ctx, &d_keys, (size_t)n * n);
  if (!st) st = pvlm_i_h2d_q(ctx, d_xyz, xyz, (size_t)3 * n * sizeof(float));
  if (!st) {
    int P = 64; while (P < n) P <<= 1;
    if (P <= 1024) hipLaunchKernelGGL(k_centre_orders<1024>, dim3((unsigned)n), dim3(1024), 0, ctx->stream, n, (const float*)d_xyz, d_keys);
    else if (P == 2048) hipLaunchKernelGGL(k_centre_orders<2048>, dim3((unsigned)n), dim3(1024), 0, ctx->stream, n, (const float*)d_xyz, d_keys);
    else hipLaunchKernelGGL(k_centre_orders<4096>, dim3((unsigned)n), dim3(1024), 0, ctx->stream, n, (const float*)d_xyz, d_keys);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_centre_orders: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
    if (!st) st = pvlm_i_d2h_q(ctx, order, d_keys, bytes);        // through the context's pinned arena: no buffer of its own to allocate (hipHostMalloc of 5 MB: 10 ms)
  }
  { const pvlm_status s2 = pvlm_i_sync(ctx); if (!st) st = s2; }
  pvlm_i_free(ctx, d_xyz); pvlm_i_free(ctx, d_keys);
  return st;
}

pvlm_status pvlm_scan_upload(pvlm_ctx* ctx, const pvlm_scan_desc* d, pvlm_scan** out) { return pvlm_scan_upload_batch(ctx, 1, d, out); }

static pvlm_status scan_upload_batch_impl(pvlm_ctx* ctx, int n_scans, const pvlm_scan_desc* descs, pvlm_scan** out);
// no exception crosses the C ABI: a worker's or a container's std::bad_alloc / std::system_error becomes a status
pvlm_status pvlm_scan_upload_batch(pvlm_ctx* ctx, int n_scans, const pvlm_scan_desc* descs, pvlm_scan** out) {
  try {
    return scan_upload_batch_impl(ctx, n_scans, descs, out);
  } catch (const std::bad_alloc&) {
    if (ctx) { (void)hipStreamSynchronize(ctx->stream); PVLM_SET_ERR(ctx, "pvlm_scan_upload_batch: out of host memory"); }
    return PVLM_ERR_NOMEM;
  } catch (const std::exception& e) {
    if (ctx) { (void)hipStreamSynchronize(ctx->stream); PVLM_SET_ERR(ctx, "pvlm_scan_upload_batch: %s", e.what()); }
    return PVLM_ERR_HIP;
  } catch (...) {
    if (ctx) { (void)hipStreamSynchronize(ctx->stream); PVLM_SET_ERR(ctx, "pvlm_scan_upload_batch: unexpected host exception"); }
    return PVLM_ERR_HIP;
  }
}
static pvlm_status scan_upload_batch_impl(pvlm_ctx* ctx, int n_scans, const pvlm_scan_desc* descs, pvlm_scan** out) {
  if (!ctx || n_scans < 0 || (n_scans > 0 && (!descs || !out))) return PVLM_ERR_ARG;
  for (int k = 0; k < n_scans; ++k) out[k] = nullptr;
  if (n_scans == 0) return PVLM_OK;
  for (int k = 0; k < n_scans; ++k) {
    const pvlm_scan_desc* d = &descs[k];
    if (!d->R_wl || !d->t_wl || d->n_surf_flat < 0 || d->n_surf_less_flat < 0 || d->n_corner < 0 || d->n_segments < 0 ||
        (d->n_surf_flat > 0 && (!d->surf_flat_xyz || !d->surf_flat_tag)) ||
        (d->n_surf_less_flat > 0 && (!d->surf_less_flat_xyz || !d->surf_less_flat_tag)) || (d->n_corner > 0 && !d->corner_xyz) ||
        (d->n_segments > 0 && (!d->segment_size || !d->segment_coeffs)) || d->point_stride_floats < 0 || (d->point_stride_floats > 0 && d->point_stride_floats < 3)) {
      PVLM_SET_ERR(ctx, "pvlm_scan_upload: inconsistent descriptor (scan %d of the batch)", k);
      return PVLM_ERR_ARG;
    }
  }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_i_trace("scan_upload_batch: enter");
  // ---- 1. plan: per-cloud grid parameters (host pass over the points: bounding box + finiteness), host-side tables
  std::vector<ScanPlan> plan((size_t)n_scans);
  std::vector<pvlm_scan*> scans((size_t)n_scans, nullptr);
  auto fail = [&](pvlm_status st) { for (pvlm_scan* s : scans) delete s; return st; };
  pvlm_status st = PVLM_OK;
  // the pass over every point (bounding boxes, finiteness): scan-parallel for a batch (with the threaded staging copy below: 39 -> 25 ms per
  // call for the 1593 scans of Floor)
  std::vector<CloudBox> boxes((size_t)n_scans * 3);
  {
    auto box_of = [&](int k) {
      const pvlm_scan_desc* d = &descs[k];
      const int ps = d->point_stride_floats > 0 ? d->point_stride_floats : 3;
      if (d->n_surf_flat > 0 && d->n_surf_flat <= PVLM_MAX_CLOUD_POINTS) cloud_box(d->n_surf_flat, d->surf_flat_xyz, boxes[(size_t)k * 3], ps);
      if (d->n_surf_less_flat > 0 && d->n_surf_less_flat <= PVLM_MAX_CLOUD_POINTS) cloud_box(d->n_surf_less_flat, d->surf_less_flat_xyz, boxes[(size_t)k * 3 + 1], ps);
      if (d->n_corner > 0 && d->n_corner <= PVLM_MAX_CLOUD_POINTS) cloud_box(d->n_corner, d->corner_xyz, boxes[(size_t)k * 3 + 2], ps);
    };
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({pvlm_thread_cap(), (size_t)n_scans / 64 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
    std::atomic<int> next{0};
    auto work = [&]() { for (int k = next++; k < n_scans; k = next++) box_of(k); };
    pvlm_run_workers(n_threads, work);
  }
  for (int k = 0; k < n_scans && !st; ++k) {
    const pvlm_scan_desc* d = &descs[k];
    pvlm_scan* s = new (std::nothrow) pvlm_scan();
    if (!s) return fail(PVLM_ERR_NOMEM);
    scans[(size_t)k] = s;
    s->id = d->id;
    std::memcpy(s->R_wl, d->R_wl, sizeof(s->R_wl));
    std::memcpy(s->t_wl, d->t_wl, sizeof(s->t_wl));
    ScanPlan& P = plan[(size_t)k];
    st = cloud_plan(ctx, P.flat, d->n_surf_flat, d->surf_flat_xyz, d->surf_flat_tag, false, &boxes[(size_t)k * 3]);
    if (!st) st = cloud_plan(ctx, P.less, d->n_surf_less_flat, d->surf_less_flat_xyz, d->surf_less_flat_tag, true, &boxes[(size_t)k * 3 + 1]);
    if (!st) st = cloud_plan(ctx, P.corner, d->n_corner, d->corner_xyz, nullptr, true, &boxes[(size_t)k * 3 + 2]);
    if (st) break;
    if (d->point_stride_floats > 3) {
      for (CloudPlan* c : {&P.flat, &P.less, &P.corner}) { c->xyz_stride = d->point_stride_floats; c->tag_stride = d->point_stride_floats; }
    }
    if (d->n_corner > 0) {
      if (d->p2s_offsets) {
        s->h_p2s_off.assign(d->p2s_offsets, d->p2s_offsets + d->n_corner + 1);
        bool monotone = s->h_p2s_off.front() == 0;
        for (int q = 0; q < d->n_corner && monotone; ++q) monotone = s->h_p2s_off[(size_t)q] <= s->h_p2s_off[(size_t)q + 1];
        const int tot = s->h_p2s_off.back();
        if (!monotone || (tot > 0 && !d->p2s_ids)) { PVLM_SET_ERR(ctx, "point_to_segment offsets must start at 0 and not decrease; p2s_ids must be given (scan %d of the batch)", k); st = PVLM_ERR_ARG; break; }
        if (tot > 0) s->h_p2s_ids.assign(d->p2s_ids, d->p2s_ids + tot);
        for (int v : s->h_p2s_ids) if (v < 0 || v >= d->n_segments) { PVLM_SET_ERR(ctx, "point_to_segment id %d out of range", v); st = PVLM_ERR_ARG; break; }
      } else {
        s->h_p2s_off.assign((size_t)d->n_corner + 1, 0);
      }
      P.n_p2s_ids = s->h_p2s_ids.size();
    }
    if (!st && d->n_segments > 0) {
      s->n_segments = d->n_segments;
      s->h_seg_size.assign(d->segment_size, d->segment_size + d->n_segments);
      s->h_seg_coeffs.assign(d->segment_coeffs, d->segment_coeffs + 6 * (size_t)d->n_segments);
      if (d->end_points) s->h_end_points.assign(d->end_points, d->end_points + 6 * (size_t)d->n_segments);
      if (d->seg_points_xyz) {
        s->h_seg_pt_off.assign((size_t)d->n_segments + 1, 0);
        for (int q = 0; q < d->n_segments; ++q) {
          if (d->segment_size[q] < 0) { PVLM_SET_ERR(ctx, "negative segment size"); st = PVLM_ERR_ARG; break; }
          s->h_seg_pt_off[(size_t)q + 1] = s->h_seg_pt_off[(size_t)q] + d->segment_size[q];
        }
        P.n_seg_pts = (size_t)s->h_seg_pt_off.back();
      }
    }
  }
  if (st) return fail(st);
  pvlm_i_trace("scan_upload_batch: plan (bbox, host tables)");
  // ---- 2. layout.  Points slab (lives as long as the scans; pvlm_scan_transform_batch re-poses it in place): the uploaded arrays.  The voxel
  //         grids go into a slab of their own (grids_build), replaced whenever the clouds are re-posed.
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t up = 0;
  auto place = [&](size_t& off, size_t& cursor, size_t bytes) { off = cursor; cursor = align(cursor + bytes); };
  auto each_cloud = [&](auto&& f) { for (ScanPlan& P : plan) { f(P.flat); f(P.less); f(P.corner); } };
  each_cloud([&](CloudPlan& c) {
    if (c.n <= 0) return;
    place(c.o_xyz, up, (size_t)c.n * 12);
    if (c.tag) place(c.o_tag, up, (size_t)c.n * 4);
  });
  for (int k = 0; k < n_scans; ++k) {
    ScanPlan& P = plan[(size_t)k];
    if (!scans[(size_t)k]->h_p2s_off.empty()) { place(P.o_p2s_off, up, scans[(size_t)k]->h_p2s_off.size() * 4); place(P.o_p2s_ids, up, std::max<size_t>(P.n_p2s_ids, 1) * 4); }
    if (!scans[(size_t)k]->h_seg_pt_off.empty()) place(P.o_seg_xyz, up, std::max<size_t>(P.n_seg_pts, 1) * 12);
  }
  const size_t up_bytes = up;
  // ---- 3. allocate: the slab from the pool, the staging mirror of the uploaded arrays (pinned, grow-only)
  char* d_slab = nullptr;
  if ((st = pvlm_i_alloc_bytes(ctx, (void**)&d_slab, std::max<size_t>(up_bytes, 256)))) return fail(st);
  auto bail = [&](pvlm_status e) { hipStreamSynchronize(ctx->stream); pvlm_i_free(ctx, d_slab); return fail(e); };
  // staging window: the whole front when it fits, else PVLM_UPLOAD_STAGE_MB (default 64) at a time
  size_t window = (size_t)64 << 20;
  if (const char* env = getenv("PVLM_UPLOAD_STAGE_MB")) if (atof(env) > 0) window = (size_t)(atof(env) * 1048576.0);
  window = align(std::max<size_t>(window, 4096));
  const size_t need = std::min(up_bytes, window);
  if (ctx->up_bytes < need) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(PVLM_ERR_HIP);
    if (ctx->h_up) (void)hipHostFree(ctx->h_up);
    ctx->h_up = nullptr; ctx->up_bytes = 0;
    const size_t want = std::min(need + need / 2, window);
    if (hipHostMalloc(&ctx->h_up, want, hipHostMallocDefault) != hipSuccess) { PVLM_SET_ERR(ctx, "scan upload: %zu bytes of pinned staging unavailable", want); return bail(PVLM_ERR_NOMEM); }
    ctx->up_bytes = want;
  }
  char* h = (char*)ctx->h_up;
  pvlm_i_trace("scan_upload_batch: layout + allocations");
  // ---- 4. the uploaded arrays as a list of (offset, source) segments, in ascending offset order
  struct Seg { size_t off; const void* src; size_t bytes; size_t elem = 0, stride = 0; };   // elem != 0: `bytes / elem` records of `elem` bytes, `stride` bytes apart at the source
  std::vector<Seg> segs;
  each_cloud([&](CloudPlan& c) {
    if (c.n <= 0) return;
    if (c.xyz_stride == 3) segs.push_back({c.o_xyz, c.xyz, (size_t)c.n * 12});
    else segs.push_back({c.o_xyz, c.xyz, (size_t)c.n * 12, 12, (size_t)c.xyz_stride * 4});
    if (c.tag) { if (c.tag_stride == 1) segs.push_back({c.o_tag, c.tag, (size_t)c.n * 4}); else segs.push_back({c.o_tag, c.tag, (size_t)c.n * 4, 4, (size_t)c.tag_stride * 4}); }
  });
  for (int k = 0; k < n_scans; ++k) {
    ScanPlan& P = plan[(size_t)k]; pvlm_scan* s = scans[(size_t)k];
    if (!s->h_p2s_off.empty()) {
      segs.push_back({P.o_p2s_off, s->h_p2s_off.data(), s->h_p2s_off.size() * 4});
      if (P.n_p2s_ids) segs.push_back({P.o_p2s_ids, s->h_p2s_ids.data(), P.n_p2s_ids * 4});
    }
    if (!s->h_seg_pt_off.empty() && P.n_seg_pts) segs.push_back({P.o_seg_xyz, descs[k].seg_points_xyz, P.n_seg_pts * 12});
  }
  // ---- 5. one copy (window by window), then the grids of the batch
  hipError_t e = hipSuccess;
  size_t first_seg = 0;
  pvlm_i_trace("scan_upload_batch: segment lists");
  // a window goes down in PIECES (PVLM_UPLOAD_PIECE_MB, default 8): the copy of a piece is queued as soon as the host threads have staged it and runs while they stage
  // the next one (staging at ~14 GB/s of host copy bandwidth and the link at ~55 GB/s used to take turns: 15 + 8 ms of a Floor-sized upload)
  size_t piece = (size_t)8 << 20;
  if (const char* env = getenv("PVLM_UPLOAD_PIECE_MB")) if (atof(env) > 0) piece = (size_t)(atof(env) * 1048576.0);
  piece = align(std::max<size_t>(piece, 4096));
  for (size_t lo = 0; lo < up_bytes && e == hipSuccess; lo += ctx->up_bytes) {
    const size_t hi = std::min(up_bytes, lo + ctx->up_bytes);
    if (lo > 0) e = hipStreamSynchronize(ctx->stream);            // the window is refilled: the previous copies must have left it
    // the pieces of this window and, per piece, the segments that reach into it
    struct Piece { size_t pa, pb, first, last, item0; };
    std::vector<Piece> pieces;
    size_t n_items = 0;
    for (size_t pa = lo; pa < hi; pa += piece) {
      const size_t pb = std::min(hi, pa + piece);
      while (first_seg < segs.size() && segs[first_seg].off + segs[first_seg].bytes <= pa) ++first_seg;
      size_t last_seg = first_seg;
      while (last_seg < segs.size() && segs[last_seg].off < pb) ++last_seg;
      pieces.push_back(Piece{pa, pb, first_seg, last_seg, n_items});
      n_items += last_seg - first_seg;
    }
    auto stage = [&](size_t q, size_t pa, size_t pb) {
      const size_t a = std::max(segs[q].off, pa), b = std::min(segs[q].off + segs[q].bytes, pb);
      if (b <= a) return;
      if (!segs[q].elem) { std::memcpy(h + (a - lo), (const char*)segs[q].src + (a - segs[q].off), b - a); return; }
      // strided records (pcl::PointXYZI-style arrays): a piece boundary may fall inside a record — the (at most two) cut records byte by byte, the whole
      // records between them in a loop of fixed-size copies without a division per record (round 6: 24 -> 6 ms of a Floor-sized upload were this gather)
      const size_t el = segs[q].elem, sd = segs[q].stride;
      const char* src = (const char*)segs[q].src;
      size_t p = a;
      auto partial = [&](size_t upto) {
        while (p < upto) {
          const size_t rel = p - segs[q].off, rec = rel / el, in = rel - rec * el, take = std::min(el - in, upto - p);
          std::memcpy(h + (p - lo), src + rec * sd + in, take);
          p += take;
        }
      };
      const size_t first_whole = segs[q].off + ((a - segs[q].off + el - 1) / el) * el;      // first record boundary at or after a
      if (first_whole >= b) { partial(b); return; }
      partial(first_whole);
      const size_t n_whole = (b - first_whole) / el;
      const char* sp = src + ((first_whole - segs[q].off) / el) * sd;
      char* dp = h + (first_whole - lo);
      if (el == 12) for (size_t r = 0; r < n_whole; ++r, sp += sd, dp += 12) std::memcpy(dp, sp, 12);
      else if (el == 4) for (size_t r = 0; r < n_whole; ++r, sp += sd, dp += 4) std::memcpy(dp, sp, 4);
      else for (size_t r = 0; r < n_whole; ++r, sp += sd, dp += el) std::memcpy(dp, sp, el);
      p = first_whole + n_whole * el;
      partial(b);
    };
    const size_t n_threads = (hi - lo) < ((size_t)8 << 20) ? 1 : std::max<size_t>(1, std::min<size_t>({(size_t)8, n_items / 64 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())}));
    if (n_threads == 1) {
      for (const Piece& pc : pieces) {
        for (size_t q = pc.first; q < pc.last; ++q) stage(q, pc.pa, pc.pb);
        if (e == hipSuccess) e = hipMemcpyAsync(d_slab + pc.pa, h + (pc.pa - lo), pc.pb - pc.pa, hipMemcpyHostToDevice, ctx->stream);
      }
    } else {
      // ONE set of threads for the window: they take (piece, segment) items in order from a counter; the calling thread queues the copy of a piece when its last item
      // is done (the segments land in disjoint ranges of the pinned window; one core moves ~10 GB/s)
      std::vector<std::atomic<size_t>> done(pieces.size());
      for (auto& d : done) d.store(0, std::memory_order_relaxed);
      std::atomic<size_t> next{0};
      std::atomic<bool> failed{false};
      auto work = [&]() {
        size_t pi = 0;
        for (size_t it = next++; it < n_items; it = next++) {
          while (pi + 1 < pieces.size() && pieces[pi + 1].item0 <= it) ++pi;
          const Piece& pc = pieces[pi];
          try { stage(pc.first + (it - pc.item0), pc.pa, pc.pb); } catch (...) { failed.store(true); }
          done[pi].fetch_add(1, std::memory_order_release);
        }
      };
      std::vector<std::thread> pool;
      try { pool.reserve(n_threads); for (size_t t = 0; t < n_threads; ++t) pool.emplace_back(work); } catch (...) {}
      if (pool.empty()) work();                                   // no thread to be had: the calling thread stages everything
      for (size_t pi = 0; pi < pieces.size(); ++pi) {
        const Piece& pc = pieces[pi];
        while (done[pi].load(std::memory_order_acquire) < pc.last - pc.first) std::this_thread::yield();
        if (e == hipSuccess) e = hipMemcpyAsync(d_slab + pc.pa, h + (pc.pa - lo), pc.pb - pc.pa, hipMemcpyHostToDevice, ctx->stream);
      }
      for (std::thread& t : pool) t.join();
      if (failed.load() && e == hipSuccess) { PVLM_SET_ERR(ctx, "scan upload: staging failed"); return bail(PVLM_ERR_HIP); }
    }
  }
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "scan upload: copy failed: %s", hipGetErrorString(e)); return bail(PVLM_ERR_HIP); }
  std::vector<GridJob> jobs;
  each_cloud([&](CloudPlan& c) { if (c.grid) jobs.push_back(GridJob{&c, (const float*)(d_slab + c.o_xyz), c.tag ? (const float*)(d_slab + c.o_tag) : nullptr}); });
  pvlm_i_trace("scan_upload_batch: staged, copies queued");
  char* d_grid = nullptr;
  if ((st = grids_build(ctx, jobs, &d_grid))) { pvlm_i_free(ctx, d_slab); return fail(st); }      // synchronises: the staging window is free again
  // ---- 6. hand out the scans: views into the two shared slabs, each released with the last of its scans
  pvlm_scan_slab* sl = new (std::nothrow) pvlm_scan_slab();
  pvlm_scan_slab* gl = d_grid ? new (std::nothrow) pvlm_scan_slab() : nullptr;
  if (!sl || (d_grid && !gl)) { delete sl; delete gl; pvlm_i_free(ctx, d_slab); pvlm_i_free(ctx, d_grid); return fail(PVLM_ERR_NOMEM); }
  sl->base = d_slab; sl->refs = n_scans;
  if (gl) { gl->base = d_grid; gl->refs = n_scans; }
  auto bind = [&](pvlm_cloud& c, const CloudPlan& p) {
    c.n = std::max(p.n, 0);
    if (p.n <= 0) return;
    c.d_xyz = (float*)(d_slab + p.o_xyz);
    if (p.tag) c.d_tag = (float*)(d_slab + p.o_tag);
    if (p.grid) grid_bind(c, p, d_grid, p.tag != nullptr);
  };
  for (int k = 0; k < n_scans; ++k) {
    pvlm_scan* s = scans[(size_t)k]; const ScanPlan& P = plan[(size_t)k];
    s->slab = sl; s->grid_slab = gl;
    bind(s->flat, P.flat); bind(s->less, P.less); bind(s->corner, P.corner);
    if (!s->h_p2s_off.empty()) { s->d_p2s_off = (int*)(d_slab + P.o_p2s_off); s->d_p2s_ids = (int*)(d_slab + P.o_p2s_ids); }
    if (!s->h_seg_pt_off.empty()) s->d_seg_xyz = (float*)(d_slab + P.o_seg_xyz);
    out[k] = s;
  }
  return PVLM_OK;
}

pvlm_status pvlm_scan_set_pose(pvlm_ctx* ctx, pvlm_scan* s, const double* R_wl, const double* t_wl) {
  if (!ctx || !s || !R_wl || !t_wl) return PVLM_ERR_ARG;
  std::memcpy(s->R_wl, R_wl, sizeof(s->R_wl));
  std::memcpy(s->t_wl, t_wl, sizeof(s->t_wl));
  return PVLM_OK;
}

static pvlm_status scan_transform_batch_impl(pvlm_ctx* ctx, int n_scans, pvlm_scan* const* scans, const double* T12, int rebuild_grids);
pvlm_status pvlm_scan_transform_batch(pvlm_ctx* ctx, int n_scans, pvlm_scan* const* scans, const double* T_rowmajor12, int rebuild_grids) {
  try {
    return scan_transform_batch_impl(ctx, n_scans, scans, T_rowmajor12, rebuild_grids);
  } catch (const std::bad_alloc&) {
    if (ctx) { (void)hipStreamSynchronize(ctx->stream); PVLM_SET_ERR(ctx, "pvlm_scan_transform_batch: out of host memory"); }
    return PVLM_ERR_NOMEM;
  } catch (...) {
    if (ctx) { (void)hipStreamSynchronize(ctx->stream); PVLM_SET_ERR(ctx, "pvlm_scan_transform_batch: unexpected host exception"); }
    return PVLM_ERR_HIP;
  }
}
static pvlm_status scan_transform_batch_impl(pvlm_ctx* ctx, int n_scans, pvlm_scan* const* scans, const double* T12, int rebuild_grids) {
  if (!ctx || n_scans < 0 || (n_scans > 0 && (!scans || !T12))) return PVLM_ERR_ARG;
  if (n_scans == 0) return PVLM_OK;
  for (int k = 0; k < n_scans; ++k) if (!scans[k]) { PVLM_SET_ERR(ctx, "pvlm_scan_transform_batch: null scan (%d of the batch)", k); return PVLM_ERR_ARG; }
  for (int k = 0; k < 12 * n_scans; ++k) if (!std::isfinite(T12[k])) { PVLM_SET_ERR(ctx, "pvlm_scan_transform_batch: non-finite transform (scan %d of the batch)", k / 12); return PVLM_ERR_ARG; }
  {   // a scan listed twice would be transformed twice
    std::vector<const pvlm_scan*> seen(scans, scans + n_scans);
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) { PVLM_SET_ERR(ctx, "pvlm_scan_transform_batch: a scan is listed twice"); return PVLM_ERR_ARG; }
  }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_i_trace("scan_transform_batch: enter");
  // ---- 1. the clouds to move: flat, less, corner and the segments' points of every scan; less / corner carry a grid
  std::vector<XformCloud> hc; std::vector<XformBlock> hb;
  struct GridRef { pvlm_scan* scan; pvlm_cloud* cloud; };
  std::vector<GridRef> grids;
  for (int k = 0; k < n_scans; ++k) {
    pvlm_scan* s = scans[k];
    auto add = [&](float* xyz, int n, pvlm_cloud* grid_cloud) {
      if (n <= 0 || !xyz) return;
      int box = -1;
      if (rebuild_grids && grid_cloud) { box = (int)grids.size(); grids.push_back(GridRef{s, grid_cloud}); }
      const int ci = (int)hc.size();
      hc.push_back(XformCloud{xyz, n, k, box, 0});
      for (int f = 0; f < n; f += 256) hb.push_back(XformBlock{ci, f});
    };
    add(s->flat.d_xyz, s->flat.n, nullptr);
    add(s->less.d_xyz, s->less.n, &s->less);
    add(s->corner.d_xyz, s->corner.n, &s->corner);
    if (s->d_seg_xyz && !s->h_seg_pt_off.empty()) add(s->d_seg_xyz, s->h_seg_pt_off.back(), nullptr);
  }
  if (hb.empty()) return PVLM_OK;
  // ---- 2. tables up, one launch
  DevScratch scratch(ctx);
  XformCloud* d_clouds = nullptr; XformBlock* d_blocks = nullptr; double* d_T = nullptr; unsigned* d_boxes = nullptr;
  pvlm_status st = scratch.alloc(&d_clouds, hc.size());
  if (!st) st = scratch.alloc(&d_blocks, hb.size());
  if (!st) st = scratch.alloc(&d_T, (size_t)n_scans * 12);
  if (!st) st = scratch.alloc(&d_boxes, std::max<size_t>(grids.size(), 1) * 7);
  if (!st) st = pvlm_i_h2d_q(ctx, d_clouds, hc.data(), hc.size() * sizeof(XformCloud));
  if (!st) st = pvlm_i_h2d_q(ctx, d_blocks, hb.data(), hb.size() * sizeof(XformBlock));
  if (!st) st = pvlm_i_h2d_q(ctx, d_T, T12, (size_t)n_scans * 12 * sizeof(double));
  std::vector<unsigned> hbox(std::max<size_t>(grids.size(), 1) * 7);
  for (size_t g = 0; g < grids.size(); ++g) { for (int q = 0; q < 3; ++q) { hbox[7 * g + q] = 0xFFFFFFFFu; hbox[7 * g + 3 + q] = 0u; } hbox[7 * g + 6] = 0x7FFFFFFFu; }
  if (!st && !grids.empty()) st = pvlm_i_h2d_q(ctx, d_boxes, hbox.data(), grids.size() * 7 * sizeof(unsigned));
  if (st) { (void)pvlm_i_sync(ctx); return st; }
  hipLaunchKernelGGL(k_scan_transform, dim3((unsigned)hb.size()), dim3(256), 0, ctx->stream, d_clouds, d_blocks, d_T, d_boxes);
  { const hipError_t e = hipGetLastError(); if (e != hipSuccess) { (void)pvlm_i_sync(ctx); PVLM_SET_ERR(ctx, "pvlm_scan_transform_batch: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; } }
  if (!rebuild_grids)
    for (int k = 0; k < n_scans; ++k) { scans[k]->less.grid_stale = scans[k]->less.n > 0; scans[k]->corner.grid_stale = scans[k]->corner.n > 0; }
  if (grids.empty()) { return pvlm_i_sync(ctx); }            // the scratch tables go back to the pool: stream-ordered, but the staged copies must have left the arena
  // ---- 3. boxes back, the grid plans of an upload of the same floats, the grids
  if ((st = pvlm_i_d2h_q(ctx, hbox.data(), d_boxes, grids.size() * 7 * sizeof(unsigned)))) { (void)pvlm_i_sync(ctx); return st; }
  if ((st = pvlm_i_sync(ctx))) return st;
  pvlm_i_trace("scan_transform_batch: transformed, boxes on the host");
  std::vector<CloudPlan> plans(grids.size());
  std::vector<GridJob> jobs; jobs.reserve(grids.size());
  for (size_t g = 0; g < grids.size(); ++g) {
    CloudBox box;
    for (int q = 0; q < 3; ++q) { box.mn[q] = ord2f(hbox[7 * g + q]); box.mx[q] = ord2f(hbox[7 * g + 3 + q]); }
    box.bad_point = hbox[7 * g + 6] == 0x7FFFFFFFu ? -1 : (int)hbox[7 * g + 6];
    pvlm_cloud& c = *grids[g].cloud;
    if ((st = cloud_plan(ctx, plans[g], c.n, nullptr, c.d_tag, true, &box))) return st;     // the clouds stay transformed: the caller destroys the scans
    jobs.push_back(GridJob{&plans[g], c.d_xyz, c.d_tag});
  }
  char* d_grid = nullptr;
  if ((st = grids_build(ctx, jobs, &d_grid))) return st;
  pvlm_scan_slab* gl = new (std::nothrow) pvlm_scan_slab();
  if (!gl) { pvlm_i_free(ctx, d_grid); return PVLM_ERR_NOMEM; }
  gl->base = d_grid; gl->refs = 0;
  for (size_t g = 0; g < grids.size(); ++g) grid_bind(*grids[g].cloud, plans[g], d_grid, grids[g].cloud->d_tag != nullptr);
  for (int k = 0; k < n_scans; ++k) {
    pvlm_scan* s = scans[k];
    if (s->less.n <= 0 && s->corner.n <= 0) continue;         // no grid of this scan in the new slab
    slab_unref(ctx, s->grid_slab);                            // the old tables: stream-ordered reuse, every kernel reading them has been queued
    s->grid_slab = gl; ++gl->refs;
  }
  if (gl->refs == 0) { pvlm_i_free(ctx, d_grid); delete gl; }
  return PVLM_OK;
}

static const pvlm_cloud* scan_cloud(const pvlm_scan* s, int which) { return which == 0 ? &s->flat : which == 1 ? &s->less : which == 2 ? &s->corner : nullptr; }
pvlm_status pvlm_scan_cloud_info(const pvlm_scan* s, int which, pvlm_grid_info* info) {
  if (!s || !info || which < 0 || which > 3) return PVLM_ERR_ARG;
  std::memset(info, 0, sizeof(*info));
  if (which == 3) { info->n = s->h_seg_pt_off.empty() || !s->d_seg_xyz ? 0 : s->h_seg_pt_off.back(); return PVLM_OK; }
  const pvlm_cloud& c = *scan_cloud(s, which);
  info->n = c.n;
  info->has_grid = c.d_sorted != nullptr; info->stale = c.grid_stale;
  if (info->has_grid) {
    info->dense = c.dense; info->nx = c.nx; info->ny = c.ny; info->nz = c.nz; info->xf = c.xf; info->table_size = c.table_size; info->cell = c.cell;
    for (int q = 0; q < 3; ++q) info->origin[q] = c.origin[q];
  }
  return PVLM_OK;
}
pvlm_status pvlm_scan_cloud_fetch(pvlm_ctx* ctx, const pvlm_scan* s, int which, float* xyz, int* cell_count, int* cell_start, unsigned long long* keys,
                                  float* sorted_xyzi) {
  if (!ctx || !s || which < 0 || which > 3) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_status st = PVLM_OK;
  if (which == 3) {
    const int n = s->h_seg_pt_off.empty() || !s->d_seg_xyz ? 0 : s->h_seg_pt_off.back();
    if (xyz && n > 0) st = pvlm_i_d2h_q(ctx, xyz, s->d_seg_xyz, (size_t)n * 12);
  } else {
    const pvlm_cloud& c = *scan_cloud(s, which);
    if (xyz && c.n > 0) st = pvlm_i_d2h_q(ctx, xyz, c.d_xyz, (size_t)c.n * 12);
    if (c.d_sorted) {
      if (!st && cell_count) st = pvlm_i_d2h_q(ctx, cell_count, c.d_cell_count, (size_t)c.table_size * 4);
      if (!st && cell_start) st = pvlm_i_d2h_q(ctx, cell_start, c.d_cell_start, (size_t)c.table_size * 4);
      if (!st && keys && c.d_keys) st = pvlm_i_d2h_q(ctx, keys, c.d_keys, (size_t)c.table_size * 8);
      if (!st && sorted_xyzi) st = pvlm_i_d2h_q(ctx, sorted_xyzi, c.d_sorted, (size_t)c.n * 16);
    }
  }
  { const pvlm_status s2 = pvlm_i_sync(ctx); if (!st) st = s2; }
  return st;
}

pvlm_status pvlm_scan_destroy(pvlm_ctx* ctx, pvlm_scan* s) {
  if (!ctx) return PVLM_ERR_ARG;
  if (!s) return PVLM_OK;
  hipSetDevice(ctx->device);
  slab_unref(ctx, s->slab); slab_unref(ctx, s->grid_slab);   // every array of the scan lives in the two slabs its batch shares
  delete s;
  return PVLM_OK;
}

pvlm_status pvlm_knn(pvlm_ctx* ctx, const pvlm_scan* scan, int which, const float* queries, int nq, int k, float max_dist, int32_t* idx,
                     float* sqd) {
  if (!ctx || !scan || nq < 0 || (nq > 0 && (!queries || !idx || !sqd)) || (which != 0 && which != 1) || !(max_dist > 0)) return PVLM_ERR_ARG;
  if (k != 5 && k != 10) { PVLM_SET_ERR(ctx, "pvlm_knn: k must be 5 or 10 (the values the reference uses)"); return PVLM_ERR_ARG; }
  if (nq == 0) return PVLM_OK;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const pvlm_cloud& c = which == 0 ? scan->less : scan->corner;
  if (c.grid_stale) { PVLM_SET_ERR(ctx, "pvlm_knn: the scan's clouds were transformed without rebuilding their grids (pvlm_scan_transform_batch with rebuild_grids = 0)"); return PVLM_ERR_STATE; }
  float* d_q = nullptr; int* d_idx = nullptr; float* d_sqd = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_q, (size_t)nq * 3);
  if (!st) st = pvlm_i_alloc(ctx, &d_idx, (size_t)nq * k);
  if (!st) st = pvlm_i_alloc(ctx, &d_sqd, (size_t)nq * k);
  hipError_t e = hipSuccess;
  if (!st) st = pvlm_i_h2d_q(ctx, d_q, queries, (size_t)nq * 3 * sizeof(float));
  if (!st) {
    {
      const CloudView cv = view_of(c);
      if (k == 10) hipLaunchKernelGGL(k_knn_queries<10>, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, cv, d_q, nq, max_dist, d_idx, d_sqd);
      else hipLaunchKernelGGL(k_knn_queries<5>, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, cv, d_q, nq, max_dist, d_idx, d_sqd);
      e = hipGetLastError();
    }
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_knn: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
    if (!st) st = pvlm_i_d2h_q(ctx, idx, d_idx, (size_t)nq * k * sizeof(int));
    if (!st) st = pvlm_i_d2h_q(ctx, sqd, d_sqd, (size_t)nq * k * sizeof(float));
  }
  { const pvlm_status s2 = pvlm_i_sync(ctx); if (!st) st = s2; }
  pvlm_i_free(ctx, d_q); pvlm_i_free(ctx, d_idx); pvlm_i_free(ctx, d_sqd);
  return st;
}

// The call is a two-slot software pipeline over batches of pairs (<= PVLM_ASSOC_BATCH_ROWS query rows, 16 M by
// default = 1.6 GB of scratch per slot, whatever the size of the pair list):
//   issue(b):  K2 + K3 of batch b into slot b & 1, accept counts -> pinned host memory, event
//   finish(b): wait for the event (the GPU is already busy with batch b + 1), size the batch's column block
//              exactly, take it from the context's pool, upload the destination rows, launch the ordered compaction
// so the scratch is bounded, the output is exactly sized, nothing is allocated from the driver in a steady state
// (re-association of every outer iteration reuses the pool) and the host never waits on an idle GPU.
pvlm_status pvlm_assoc_point2plane(pvlm_ctx* ctx, int n_pairs, pvlm_scan* const* ref, pvlm_scan* const* nei, double plane_tolerance,
                                   float dist_threshold, pvlm_functor kind, unsigned flags, double weight, pvlm_resset** out) {
  if (!ctx || !out || n_pairs < 0 || (n_pairs > 0 && (!ref || !nei))) return PVLM_ERR_ARG;
  *out = nullptr;
  if (kind != PVLM_POINT2PLANE_ANGLE && kind != PVLM_POINT2PLANE_METER) { PVLM_SET_ERR(ctx, "kind must be a point-to-plane functor"); return PVLM_ERR_ARG; }
  if (!(dist_threshold > 0)) { PVLM_SET_ERR(ctx, "dist_threshold must be positive"); return PVLM_ERR_ARG; }
  for (int p = 0; p < n_pairs; ++p) if (!ref[p] || !nei[p]) { PVLM_SET_ERR(ctx, "null scan in pair %d", p); return PVLM_ERR_ARG; }
  for (int p = 0; p < n_pairs; ++p)
    if (ref[p]->less.grid_stale) { PVLM_SET_ERR(ctx, "pair %d: the reference scan's clouds were transformed without rebuilding their grids (pvlm_scan_transform_batch with rebuild_grids = 0)", p); return PVLM_ERR_STATE; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const bool keep_idx = (flags & 0x100u) != 0;
  static const bool force_exact = getenv("PVLM_ASSOC_EXACT_FIT") != nullptr;     // A/B runs of hosts that do not pass the flag
  const bool exact_fit = (flags & PVLM_FLAG_ASSOC_EXACT_FIT) != 0 || force_exact;

  pvlm_resset* rs = new (std::nothrow) pvlm_resset();
  if (!rs) return PVLM_ERR_NOMEM;
  rs->kind = kind; rs->flags = flags & 0xFFu; rs->weight = weight; rs->n_pairs = n_pairs; rs->ncols = 7;
  rs->h_ref.resize(n_pairs); rs->h_nei.resize(n_pairs);
  for (int p = 0; p < n_pairs; ++p) { rs->h_ref[p] = ref[p]->id; rs->h_nei[p] = nei[p]->id; }
  rs->h_out_start.assign(n_pairs + 1, 0);
  rs->h_seg_start.assign(n_pairs + 1, 0);
  rs->h_pair_block.assign(n_pairs, 0);

  std::vector<PairDesc> descs(n_pairs);
  for (int p = 0; p < n_pairs; ++p) {
    PairDesc& d = descs[p];
    d.ref = view_of(ref[p]->less);
    d.q_xyz = nei[p]->flat.d_xyz; d.q_tag = nei[p]->flat.d_tag; d.nq = nei[p]->flat.n;
    std::memcpy(d.Rr, ref[p]->R_wl, 72); std::memcpy(d.tr, ref[p]->t_wl, 24);
    std::memcpy(d.Rn, nei[p]->R_wl, 72); std::memcpy(d.tn, nei[p]->t_wl, 24);
    // a target cloud with fewer than 10 points can never satisfy the k = 10 search
    if (d.ref.n < 10) d.nq = 0;
  }
  static const long long compact_min_bytes = getenv("PVLM_ASSOC_COMPACT_MIN_MB") ? (long long)(atof(getenv("PVLM_ASSOC_COMPACT_MIN_MB")) * 1048576.0) : (64ll << 20);
  const int k3_chunk = 256 * (exact_fit ? PVLM_K3_SUB : PVLM_K3F_SUB);       // queries per chunk of the plane-fit kernel that will run
  long long budget_rows = 16ll << 20;
  if (const char* env = getenv("PVLM_ASSOC_BATCH_ROWS")) { const long long v = atoll(env); if (v > 0) budget_rows = v; }
  struct Batch { int p0, p1; long long rows; int chunks; int bmax; };
  std::vector<Batch> batches;
  long long cap_rows = 0; int cap_chunks = 0, cap_pairs = 0;
  // Fast mode: the FIRST batch is a probe of about a million queries.  Its share of queries the fast fit had to leave to the QR decides the kernel of the other
  // batches: a cloud whose neighbourhoods the normal equations cannot handle (raw scans as targets: ten neighbours along one ring, planes through the sensor) pays
  // for the fast attempt AND the QR — there the exact kernel is the faster one (1.2 against 1.5-2.3 ms per 16.7 M queries).  Decided from the data alone: the same
  // call gives the same set.
  static const long long probe_rows = getenv("PVLM_ASSOC_PROBE_ROWS") ? atoll(getenv("PVLM_ASSOC_PROBE_ROWS")) : (1ll << 20);
  const bool probing = !exact_fit && probe_rows > 0 && PVLM_K3_SUB == PVLM_K3F_SUB;      // (both kernels lay their chunks out alike: the choice is per batch)
  for (int p = 0; p < n_pairs;) {
    Batch b{p, p, 0, 0, 0};
    const long long budget = (probing && batches.empty()) ? std::min(budget_rows, probe_rows) : budget_rows;
    while (b.p1 < n_pairs && (b.p1 == b.p0 || (b.rows + descs[b.p1].nq <= budget && b.p1 - b.p0 < 32768))) {
      descs[b.p1].tmp_base = b.rows;
      descs[b.p1].chunk_base = b.chunks;
      b.rows += descs[b.p1].nq;
      b.chunks += (descs[b.p1].nq + k3_chunk - 1) / k3_chunk;
      b.bmax = std::max(b.bmax, descs[b.p1].nq);
      ++b.p1;
    }
    cap_rows = std::max(cap_rows, b.rows); cap_chunks = std::max(cap_chunks, b.chunks); cap_pairs = std::max(cap_pairs, b.p1 - b.p0);
    batches.push_back(b);
    p = b.p1;
  }
  pvlm_status st = assoc_ws_ensure(ctx, cap_rows, cap_chunks, cap_pairs);
  if (st) { pvlm_i_resset_free(ctx, rs); return st; }
  pvlm_assoc_ws& ws = ctx->assoc_ws;

  bool batch_exact = exact_fit;                      // which plane-fit kernel the batches issued from now on run
  // per batch: the column block is taken at ISSUE time with room for every query of the batch (segment of pair p: seg_rows(nq) rows at dst_row)
  std::vector<long long> batch_R(batches.size(), 0);
  for (size_t bi = 0; bi < batches.size(); ++bi) {
    long long row = 0;
    for (int p = batches[bi].p0; p < batches[bi].p1; ++p) { descs[p].dst_row = row; row += pvlm_i_seg_rows(descs[p].nq); }
    batch_R[bi] = std::max<long long>(row, 16);
  }
  auto issue = [&](int bi) -> pvlm_status {
    const Batch& b = batches[bi];
    const int s = bi & 1, nb = b.p1 - b.p0;
    const long long R = batch_R[bi];
    double* d_block = nullptr;
    pvlm_status sa = pvlm_i_alloc(ctx, &d_block, (size_t)R * 7);
    if (sa) return sa;
    rs->col_blocks.push_back(d_block); rs->block_rows.push_back(R); rs->block_n.push_back(0);
    rs->n_dev += R;
    int32_t *d_q = nullptr, *d_n = nullptr;
    if (keep_idx) {
      if ((sa = pvlm_i_alloc(ctx, &d_q, (size_t)R))) return sa;
      rs->d_qidx.push_back(d_q);
      if ((sa = pvlm_i_alloc(ctx, &d_n, (size_t)R * 10))) return sa;
      rs->d_nn.push_back(d_n);
    }
    std::memcpy(ws.h_desc[s], &descs[b.p0], (size_t)nb * sizeof(PairDesc));
    PVLM_HIP(ctx, hipMemcpyAsync(ws.d_desc[s], ws.h_desc[s], (size_t)nb * sizeof(PairDesc), hipMemcpyHostToDevice, ctx->stream));
    const PairDesc* d_desc = static_cast<const PairDesc*>(ws.d_desc[s]);
    // chain words of the batch's chunks, then the per-pair totals (ints), then the ticket: zeroed together
    unsigned long long* d_chain = ws.d_chain[s];
    int* d_count = reinterpret_cast<int*>(d_chain + b.chunks);
    int* d_ticket = d_count + nb;
    PVLM_HIP(ctx, hipMemsetAsync(d_chain, 0, (size_t)b.chunks * sizeof(unsigned long long) + ((size_t)nb + 2) * sizeof(int), ctx->stream));   // + ticket, + the count of exact fits
    if (b.bmax > 0) {
      pvlm_prof_scope prof(ctx, 2);
      // (an LDS-staged variant of the search was built and measured in round 2: 35.0 vs 33.6 ms for 134 M queries — the
      // search is bound by instruction issue, not by memory latency; numbers in DESIGN.md, code removed in round 3)
      hipLaunchKernelGGL(k_knn_pairs, dim3((b.bmax + 255) / 256, nb), dim3(256), 0, ctx->stream, d_desc, dist_threshold, ws.d_nn[s], ws.rows);
      const int chunks_x = (b.bmax + k3_chunk - 1) / k3_chunk;
      const long long total = (long long)chunks_x * nb;
      if (total > 0x7fffffffll) { PVLM_SET_ERR(ctx, "association batch too large"); return PVLM_ERR_ARG; }
      static const int k3_blocks = [] { const char* e = getenv("PVLM_K3_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 1024; }();   // persistent workgroups (two fit per CU)
      static const int k3f_blocks = [] { const char* e = getenv("PVLM_K3F_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 256 * PVLM_K3F_WAVES; }();   // the fast kernel: PVLM_K3F_WAVES workgroups per CU
      if (batch_exact)
        hipLaunchKernelGGL((k_fit_pairs<true, PVLM_K3_SUB>), dim3((unsigned)std::min<long long>(total, k3_blocks)), dim3(256), 0, ctx->stream, d_desc, plane_tolerance, ws.d_nn[s], ws.rows, d_block, R, d_chain,
                           d_count, d_ticket, d_q, d_n, chunks_x, (int)total);
      else
        hipLaunchKernelGGL((k_fit_pairs<false, PVLM_K3F_SUB>), dim3((unsigned)std::min<long long>(total, k3f_blocks)), dim3(256), 0, ctx->stream, d_desc, plane_tolerance, ws.d_nn[s], ws.rows, d_block, R, d_chain,
                           d_count, d_ticket, d_q, d_n, chunks_x, (int)total);
      PVLM_HIP(ctx, hipGetLastError());
    }
    PVLM_HIP(ctx, hipMemcpyAsync(ws.h_count[s], d_count, ((size_t)nb + 2) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));   // per-pair totals, the ticket, the exact fits
    PVLM_HIP(ctx, hipEventRecord(ws.ev[s], ctx->stream));
    return PVLM_OK;
  };
  auto finish = [&](int bi) -> pvlm_status {
    const Batch& b = batches[bi];
    const int s = bi & 1;
    PVLM_HIP(ctx, hipEventSynchronize(ws.ev[s]));
    long long block_n = 0;
    for (int p = b.p0; p < b.p1; ++p) {
      const long long m = descs[p].nq > 0 ? ws.h_count[s][p - b.p0] : 0;
      rs->h_pair_block[p] = bi;
      rs->h_seg_start[p] = descs[p].dst_row;
      rs->h_out_start[p + 1] = rs->h_out_start[p] + m;
      block_n += m;
    }
    rs->block_n[(size_t)bi] = block_n;
    rs->assoc_exact_fits += ws.h_count[s][b.p1 - b.p0 + 1];
    // a block that is mostly empty is replaced by a dense one (queued behind the kernels that filled it; the next batch is already running)
    const long long R = batch_R[(size_t)bi];
    if (!keep_idx && R * 56 >= compact_min_bytes && block_n * 2 < R) {
      std::vector<CompactSeg> segs((size_t)(b.p1 - b.p0));
      long long row = 0, longest = 0;
      for (int p = b.p0; p < b.p1; ++p) {
        const long long m = rs->h_out_start[(size_t)p + 1] - rs->h_out_start[(size_t)p];
        segs[(size_t)(p - b.p0)] = CompactSeg{descs[(size_t)p].dst_row, row, (int)m, 0};
        rs->h_seg_start[(size_t)p] = row;
        row += pvlm_i_seg_rows(m); longest = std::max(longest, m);
      }
      const long long R2 = std::max<long long>(row, 16);
      double* d_new = nullptr; CompactSeg* d_segs = nullptr;
      pvlm_status sc = pvlm_i_alloc(ctx, &d_new, (size_t)R2 * 7);
      if (!sc) sc = pvlm_i_alloc(ctx, &d_segs, segs.size());
      if (!sc) sc = pvlm_i_h2d_q(ctx, d_segs, segs.data(), segs.size() * sizeof(CompactSeg));
      if (sc) { pvlm_i_free(ctx, d_new); pvlm_i_free(ctx, d_segs); return sc; }
      hipLaunchKernelGGL(k_compact_block, dim3((unsigned)std::max<long long>(1, std::min<long long>((longest + 255) / 256, 64)), (unsigned)segs.size()), dim3(256), 0, ctx->stream,
                         (const CompactSeg*)d_segs, (const double*)rs->col_blocks[(size_t)bi], R, d_new, R2);
      PVLM_HIP(ctx, hipGetLastError());
      pvlm_i_free(ctx, rs->col_blocks[(size_t)bi]); pvlm_i_free(ctx, d_segs);     // stream-ordered: reused only by work queued after the copy
      rs->col_blocks[(size_t)bi] = d_new; rs->block_rows[(size_t)bi] = R2; rs->n_dev += R2 - R;
    }
    return PVLM_OK;
  };
  const int B = (int)batches.size();
  int finished = 0;                                  // batches [0, finished) are accounted for
  for (int bi = 0; bi < B && !st; ++bi) {
    st = issue(bi);
    if (!st && bi == 0 && probing && B > 1) {        // the probe: waited for at once (a few hundred microseconds with the GPU idle, once per call)
      st = finish(0); finished = 1;
      if (!st && batches[0].rows > 0 && (double)rs->assoc_exact_fits > 0.02 * (double)batches[0].rows) batch_exact = true;
    }
    if (!st && bi >= 1 && finished < bi) { st = finish(bi - 1); finished = bi; }
  }
  if (!st && finished < B) st = finish(B - 1);
  rs->assoc_exact_kernel_batches = batch_exact && !exact_fit ? B - 1 : (exact_fit ? B : 0);
  if (!st) {
    rs->n = rs->h_out_start[n_pairs];
    st = pvlm_i_resset_finalize(ctx, rs);   // uploads the segment table + work list; the call's one full synchronisation
  }
  if (st) { hipStreamSynchronize(ctx->stream); pvlm_i_resset_free(ctx, rs); return st; }
  *out = rs;
  return PVLM_OK;
}

pvlm_status pvlm_assoc_point2plane_stats(const pvlm_resset* rs, int64_t* exact_fits) {
  if (!rs) return PVLM_ERR_ARG;
  if (exact_fits) *exact_fits = rs->assoc_exact_fits;
  return PVLM_OK;
}
pvlm_status pvlm_assoc_point2plane_stats2(const pvlm_resset* rs, int64_t* exact_fits, int* batches, int* exact_kernel_batches) {
  if (!rs) return PVLM_ERR_ARG;
  if (exact_fits) *exact_fits = rs->assoc_exact_fits;
  if (batches) *batches = (int)rs->col_blocks.size();
  if (exact_kernel_batches) *exact_kernel_batches = rs->assoc_exact_kernel_batches;
  return PVLM_OK;
}

pvlm_status pvlm_assoc_point2plane_debug(pvlm_ctx* ctx, const pvlm_resset* rs, int32_t* qidx, int32_t* nn) {
  if (!ctx || !rs) return PVLM_ERR_ARG;
  if (rs->n > 0 && rs->d_qidx.empty()) { PVLM_SET_ERR(ctx, "indices were not kept: pass flag 0x100 to pvlm_assoc_point2plane"); return PVLM_ERR_STATE; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (rs->n == 0) return PVLM_OK;
  // the debug arrays share the layout of the column blocks: the rows of pair p sit at its segment start
  pvlm_status st = PVLM_OK;
  for (int p = 0; p < rs->n_pairs && !st; ++p) {
    const int64_t o = rs->h_out_start[(size_t)p], m = rs->h_out_start[(size_t)p + 1] - o, seg = rs->h_seg_start[(size_t)p];
    const size_t b = (size_t)rs->h_pair_block[(size_t)p];
    if (m <= 0) continue;
    if (qidx) st = pvlm_i_d2h_q(ctx, qidx + o, rs->d_qidx[b] + seg, (size_t)m * sizeof(int));
    if (!st && nn) st = pvlm_i_d2h_q(ctx, nn + o * 10, rs->d_nn[b] + seg * 10, (size_t)m * 10 * sizeof(int));
  }
  { const pvlm_status s2 = pvlm_i_sync(ctx); if (!st) st = s2; }
  return st;
}

}  // extern "C"

// pvlm_preload: HIP loads the code object of a translation unit at the first launch of one of its kernels (15 ms for the larger ones) — an empty launch from here
// moves that out of the first call that needs this file's kernels
__global__ void k_preload_assoc() {}
void pvlm_i_preload_assoc(hipStream_t s) { hipLaunchKernelGGL(k_preload_assoc, dim3(1), dim3(1), 0, s); }
