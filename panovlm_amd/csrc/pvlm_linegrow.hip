// K27 — the growth phase of the LiDAR line extraction for a batch of scans:
//   ExtractLineFeatures / ExpandLine   sensors/LidarLineExtraction.cpp:296-389, :10-70     (called by Velodyne::EdgeToLine, sensors/Velodyne.cpp:1269-1324)
// Upstream grows one segment after the other, scan by scan, on the host (2 000 principal-axis fits per scan, each needed by the greedy growth: 1.3 ms per scan, the
// largest host item of the feature batch).  The growth of the segment that starts from edge point i with its neighbours (a, b) never looks at what other segments
// took — only upstream's WALK over the start points does (a point an earlier segment took is skipped, `visited`) — so every (i, a, b) is an independent task
// (csrc/pvlm_linegrow_core.h), and the walk is replayed over finished tasks:
//   k_edge_knn         one thread per edge point: its 5 nearest edge points of the same scan, brute force in index order (a scan has a few hundred edge points), the
//                      (d2, index) order and float sums of the host's table
//   k_line_grow_walk   one workgroup per scan, in ROUNDS: the next POINTS start points the walk has not skipped yet, six tasks each, one task per lane (the two id
//                      lists of a task in LDS, uint16 ids); then the walk itself over the round's points in index order — a point taken by a segment kept earlier
//                      in the round is dropped with its tasks (speculation lost), the segments of the others are kept (>= 5 members), their members marked, their
//                      records written.  Exactly upstream's segments in upstream's order, whatever the round size.
// Growing EVERY task first and walking afterwards (one launch, 680 000 lanes) costs 6 x the fits of the serial walk and returns 370 000 segments per 454 scans of
// which the walk keeps 6 %: 15-19 ms of kernel + 25 ms of copies and sorting, measured (profiles/r6_k27_variants.txt); the rounds spend ~2 x the serial walk's fits.
// Double arithmetic without contraction (NO_CONTRACT in build.py), no libm on the device (the turn test compares cosines against thresholds from the host's acos).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "pvlm_internal.h"
#include "pvlm_linegrow_core.h"

using namespace pvlm_linegrow;

namespace {

struct GrowScan { int pt_off, n, k, pad; };

__global__ __launch_bounds__(256) void k_edge_knn(const GrowScan* __restrict__ scans, const int* __restrict__ pt_scan, int n_points, const float4* __restrict__ xyz,
                                                  int* __restrict__ nn_idx, float* __restrict__ nn_sqd) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n_points) return;
  const GrowScan s = scans[pt_scan[g]];
  int idx[kK]; float sqd[kK];
  neighbours_of(reinterpret_cast<const float*>(xyz + s.pt_off), 4, s.n, g - s.pt_off, s.k, idx, sqd);
#pragma unroll
  for (int j = 0; j < kK; ++j) { nn_idx[(size_t)g * kK + j] = idx[j]; nn_sqd[(size_t)g * kK + j] = sqd[j]; }
}

// the two id lists of a task in LDS, [slot][thread] (uint16: two threads per bank word)
template <int THREADS>
struct LdsLists {
  unsigned short* m_; unsigned short* o_;
  __device__ unsigned short& m(int k) { return m_[k * THREADS]; }
  __device__ unsigned short& o(int k) { return o_[k * THREADS]; }
};

struct SegRecord { int scan, order, task, count, mem_off, pad; double coeff[6]; };      // order: position of the segment in the scan's walk

constexpr int kMaxScanPoints = 8192;   // `visited` flags of a scan in LDS (a byte each); a larger edge cloud is left to the host
constexpr int kLdsPoints = 2048;       // edge points of a scan kept in LDS (32 KB as float4); a larger scan reads them from global memory

// WAVES waves per workgroup, 10 start points (60 tasks) per wave and round.
template <int WAVES, bool IN_LDS>
__global__ __launch_bounds__(WAVES * 64) void k_line_grow_walk(const GrowScan* __restrict__ scans, const float4* __restrict__ xyz, const int* __restrict__ nn_idx,
                                                               const float* __restrict__ nn_sqd, Turn turn, SegRecord* __restrict__ segs, int seg_cap,
                                                               int* __restrict__ members, int mem_cap,
                                                               int* __restrict__ counters /* [0] segments [1] members [2] pool overflow [3] tasks run */,
                                                               int* __restrict__ scan_status) {
  constexpr int THREADS = WAVES * 64, POINTS = WAVES * 10;
  __shared__ unsigned short s_m[kMaxMembers * THREADS], s_o[kMaxMembers * THREADS];
  __shared__ unsigned char s_visited[kMaxScanPoints];
  __shared__ float4 s_xyz[IN_LDS ? kLdsPoints : 1];                                  // the scan's edge points (every loop of the growth fetches the points its lists name)
  __shared__ int s_start[POINTS];
  __shared__ int s_count, s_cursor, s_order, s_fail;
  const int sc = blockIdx.x;
  const GrowScan s = scans[sc];
  if (s.n <= 0) return;
  if (s.n > kMaxScanPoints) { if (threadIdx.x == 0) scan_status[sc] = kOverflow; return; }
  for (int k = threadIdx.x; k < s.n; k += THREADS) s_visited[k] = 0;
  if (IN_LDS) for (int k = threadIdx.x; k < s.n; k += THREADS) s_xyz[k] = xyz[s.pt_off + k];
  if (threadIdx.x == 0) { s_cursor = 0; s_order = 0; s_fail = 0; }
  __syncthreads();
  const Cloud C{IN_LDS ? reinterpret_cast<const float*>(s_xyz) : reinterpret_cast<const float*>(xyz + s.pt_off), 4, s.n, s.k, nn_idx + (size_t)s.pt_off * kK, nn_sqd + (size_t)s.pt_off * kK};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = lane < 60 ? wave * 10 + lane / kCombos : -1;        // which of the round's start points this thread works for (lanes 60 .. 63 idle)
  const int c = lane % kCombos;
  int a, b; combo(c, &a, &b);
  LdsLists<THREADS> w{s_m + threadIdx.x, s_o + threadIdx.x};
  int tasks_run = 0;
  for (;;) {
    // ---- the round's start points: the next POINTS points nothing has taken yet, in index order
    if (threadIdx.x == 0) {
      int have = 0, at = s_cursor;
      for (; at < s.n && have < POINTS; ++at) if (!s_visited[at]) s_start[have++] = at;
      s_count = have; s_cursor = at;
    }
    __syncthreads();
    const int count = s_count;
    if (count == 0) break;
    // ---- one task per lane
    int st = kNone, n_mem = 0; double coeff[6];
    if (slot >= 0 && slot < count) { st = grow_task(C, turn, s_start[slot], a, b, w, &n_mem, coeff); ++tasks_run; }
    if (st == kOverflow || st == kUndecided) { atomicMax(&scan_status[sc], st); s_fail = 1; }
    __syncthreads();
    if (s_fail) break;
    // ---- the walk over the round's points, in index order: a point taken by a segment kept earlier in this round is dropped with its tasks
    for (int p = 0; p < count; ++p) {
      const int i = s_start[p];
      const bool taken = s_visited[i] != 0;                                   // uniform: read by everyone before anyone of this step writes
      __syncthreads();
      if (!taken) {
        if (threadIdx.x == 0) s_visited[i] = 1;
        if (slot == p) {
          // the six lanes of the point are neighbours in one wave: the kept ones number themselves in combination order
          const unsigned long long kept = __ballot(st == kSegment);
          const int first = (lane / kCombos) * kCombos;
          const unsigned long long mine = (kept >> first) & 0x3Full;
          if (st == kSegment) {
            const int rank = __popcll(mine & ((1ull << c) - 1ull));
            const int leader = __ffsll((long long)mine) - 1;
            int base = 0;
            if (c == leader) base = atomicAdd(&s_order, __popcll(mine));      // the first kept lane of the point takes the numbers for all
            base = __shfl(base, first + leader, 64);
            const int g_slot = atomicAdd(&counters[0], 1);
            const int off = atomicAdd(&counters[1], n_mem);
            if (g_slot >= seg_cap || off + n_mem > mem_cap) atomicMax(&counters[2], 1);
            else {
              SegRecord r; r.scan = sc; r.order = base + rank; r.task = i * kCombos + c; r.count = n_mem; r.mem_off = off; r.pad = 0;
              for (int k = 0; k < 6; ++k) r.coeff[k] = coeff[k];
              segs[g_slot] = r;
              for (int k = 0; k < n_mem; ++k) { const int id = w.m(k); members[off + k] = id; s_visited[id] = 1; }
            }
          }
        }
      }
      __syncthreads();
    }
  }
  if (tasks_run) atomicAdd(&counters[3], tasks_run);
}

}  // namespace

struct pvlm_line_grow {
  int n_scans = 0;
  std::vector<int> status, n_points;
  std::vector<int> seg_first;                 // n_scans + 1: segments of scan s = [seg_first[s], seg_first[s + 1]), in walk order
  std::vector<int> seg_task, seg_off;         // per segment: task = i * 6 + combo; members [seg_off[q], seg_off[q + 1])
  std::vector<int> members;
  std::vector<double> coeffs;                 // 6 per segment
  double kernel_ms = 0;                       // k_edge_knn + k_line_grow_walk (HIP events)
  long long tasks_run = 0;
  // between pvlm_line_grow_begin and pvlm_line_grow_finish: the device blocks of the batch in flight
  bool pending = false;
  float4* d_xyz = nullptr; int* d_pt_scan = nullptr; GrowScan* d_scans = nullptr; int* d_nn_idx = nullptr; float* d_nn_sqd = nullptr;
  SegRecord* d_segs = nullptr; int* d_members = nullptr; int* d_counters = nullptr; int* d_status = nullptr;
  int seg_cap = 0, mem_cap = 0;
};

static void grow_release(pvlm_ctx* ctx, pvlm_line_grow* G) {
  pvlm_i_free(ctx, G->d_xyz); pvlm_i_free(ctx, G->d_pt_scan); pvlm_i_free(ctx, G->d_scans); pvlm_i_free(ctx, G->d_nn_idx); pvlm_i_free(ctx, G->d_nn_sqd);
  pvlm_i_free(ctx, G->d_segs); pvlm_i_free(ctx, G->d_members); pvlm_i_free(ctx, G->d_counters); pvlm_i_free(ctx, G->d_status);
  G->d_xyz = nullptr; G->d_pt_scan = nullptr; G->d_scans = nullptr; G->d_nn_idx = nullptr; G->d_nn_sqd = nullptr; G->d_segs = nullptr; G->d_members = nullptr;
  G->d_counters = nullptr; G->d_status = nullptr;
}
// a grow-only pinned buffer of the context
static bool grow_pinned(void** buf, size_t* have, size_t need) {
  if (*have >= need) return true;
  if (*buf) (void)hipHostFree(*buf);
  *buf = nullptr; *have = 0;
  const size_t want = need + need / 2 + 4096;
  if (hipHostMalloc(buf, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *buf = nullptr; return false; }
  *have = want;
  return true;
}

extern "C" {

// Queues the batch on the context's growth stream (uploads from a pinned buffer, both kernels, counters and statuses back) and returns; nothing here waits for the
// device.  The growth stream is ordered behind whatever the main stream has queued so far (the pool hands out blocks the main stream may still be using).
pvlm_status pvlm_line_grow_begin(pvlm_ctx* ctx, int n_scans, const pvlm_edge_cloud* clouds, pvlm_line_grow** out) {
  if (!ctx || !out || n_scans < 0 || (n_scans > 0 && !clouds)) return PVLM_ERR_ARG;
  *out = nullptr;
  for (int s = 0; s < n_scans; ++s)
    if (clouds[s].n < 0 || clouds[s].n > 65535 || (clouds[s].n > 0 && (!clouds[s].xyz || clouds[s].stride_floats < 3))) {
      PVLM_SET_ERR(ctx, "pvlm_line_grow_begin: edge cloud %d: 0 .. 65535 points with a stride of at least 3 floats", s);
      return PVLM_ERR_ARG;
    }
  if (ctx->grow_in_flight) { PVLM_SET_ERR(ctx, "pvlm_line_grow_begin: the previous batch of this context has not been finished"); return PVLM_ERR_STATE; }
  if (ctx->capturing) { PVLM_SET_ERR(ctx, "pvlm_line_grow_begin inside a graph capture"); return PVLM_ERR_STATE; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_line_grow* G = nullptr;
  try {
    G = new pvlm_line_grow();
    G->n_scans = n_scans;
    G->status.assign((size_t)n_scans, 0); G->n_points.assign((size_t)n_scans, 0); G->seg_first.assign((size_t)n_scans + 1, 0);
    G->seg_off.assign(1, 0);
    size_t n_points = 0;
    for (int s = 0; s < n_scans; ++s) { G->n_points[(size_t)s] = clouds[s].n; n_points += (size_t)clouds[s].n; }
    if (n_points == 0) { *out = G; return PVLM_OK; }
    if (n_points > (size_t)0x3fffffff) { delete G; PVLM_SET_ERR(ctx, "pvlm_line_grow_begin: too many edge points"); return PVLM_ERR_ARG; }
    if (!ctx->grow_stream) {
      bool ok = hipStreamCreateWithFlags(&ctx->grow_stream, hipStreamNonBlocking) == hipSuccess;
      for (int k = 0; k < 3 && ok; ++k) ok = hipEventCreate(&ctx->grow_ev[k]) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); delete G; PVLM_SET_ERR(ctx, "pvlm_line_grow_begin: no stream / events"); return PVLM_ERR_HIP; }
    }
    // pinned input: [xyz (float4) | point -> scan | scan table]; pinned output: [counters (4) | statuses | segments | members]
    const size_t o_pt = n_points * 16, o_sc = o_pt + n_points * 4, in_bytes = o_sc + (size_t)n_scans * sizeof(GrowScan);
    // a kept segment belongs to one of the six tasks of a start point: at most 6 n segments, sized for one per point (more: PVLM_ERR_REFUSED, the caller grows on
    // the host); the member pool for 16 members per edge point (a walk keeps ~0.3 segments of ~16 members per edge point)
    G->seg_cap = (int)std::min<size_t>(n_points + 1024, (size_t)1 << 28); G->mem_cap = (int)std::min<size_t>(n_points * 16 + 4096, (size_t)1 << 30);
    const size_t out_bytes = 16 + (size_t)n_scans * 4 + (size_t)G->seg_cap * sizeof(SegRecord) + (size_t)G->mem_cap * 4;
    if (!grow_pinned(&ctx->h_grow_in, &ctx->grow_in_bytes, in_bytes) || !grow_pinned(&ctx->h_grow_out, &ctx->grow_out_bytes, out_bytes)) {
      delete G; PVLM_SET_ERR(ctx, "pvlm_line_grow_begin: pinned staging unavailable"); return PVLM_ERR_NOMEM;
    }
    char* hin = static_cast<char*>(ctx->h_grow_in);
    float* xyz = reinterpret_cast<float*>(hin); int* pt_scan = reinterpret_cast<int*>(hin + o_pt); GrowScan* scans = reinterpret_cast<GrowScan*>(hin + o_sc);
    size_t at = 0; int n_max = 0;
    for (int s = 0; s < n_scans; ++s) {
      const int n = clouds[s].n;
      scans[s] = GrowScan{(int)at, n, std::min(kK, n), 0};
      n_max = std::max(n_max, n);
      for (int i = 0; i < n; ++i) {
        const float* p = clouds[s].xyz + (size_t)i * (size_t)clouds[s].stride_floats;
        float* q = xyz + (at + (size_t)i) * 4;
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = 0.f;
        pt_scan[at + (size_t)i] = s;
      }
      at += (size_t)n;
    }
    static const Turn turn = turn_thresholds();
    pvlm_status st = pvlm_i_alloc(ctx, &G->d_xyz, n_points);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_pt_scan, n_points);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_scans, (size_t)n_scans);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_nn_idx, n_points * kK);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_nn_sqd, n_points * kK);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_segs, (size_t)G->seg_cap);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_members, (size_t)G->mem_cap);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_counters, 4);
    if (!st) st = pvlm_i_alloc(ctx, &G->d_status, (size_t)n_scans);
    if (st) { grow_release(ctx, G); delete G; return st; }
    hipStream_t gs = ctx->grow_stream;
    hipError_t e = hipEventRecord(ctx->grow_ev[0], ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(gs, ctx->grow_ev[0], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(G->d_xyz, xyz, n_points * 16, hipMemcpyHostToDevice, gs);
    if (e == hipSuccess) e = hipMemcpyAsync(G->d_pt_scan, pt_scan, n_points * 4, hipMemcpyHostToDevice, gs);
    if (e == hipSuccess) e = hipMemcpyAsync(G->d_scans, scans, (size_t)n_scans * sizeof(GrowScan), hipMemcpyHostToDevice, gs);
    if (e == hipSuccess) e = hipMemsetAsync(G->d_counters, 0, 4 * sizeof(int), gs);
    if (e == hipSuccess) e = hipMemsetAsync(G->d_status, 0, (size_t)n_scans * sizeof(int), gs);
    if (e == hipSuccess) {
      hipEventRecord(ctx->grow_ev[1], gs);
      hipLaunchKernelGGL(k_edge_knn, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, gs, (const GrowScan*)G->d_scans, (const int*)G->d_pt_scan, (int)n_points,
                         (const float4*)G->d_xyz, G->d_nn_idx, G->d_nn_sqd);
      static const int waves = [] { const char* v = getenv("PVLM_K27_WAVES"); const int k = v ? atoi(v) : 0; return k == 1 || k == 2 || k == 4 ? k : 2; }();
      const bool in_lds = n_max <= kLdsPoints;
#define PVLM_K27_LAUNCH(W, L) hipLaunchKernelGGL((k_line_grow_walk<W, L>), dim3((unsigned)n_scans), dim3(W * 64), 0, gs, (const GrowScan*)G->d_scans, (const float4*)G->d_xyz, \
                                                 (const int*)G->d_nn_idx, (const float*)G->d_nn_sqd, turn, G->d_segs, G->seg_cap, G->d_members, G->mem_cap, G->d_counters, G->d_status)
      if (waves == 1) { if (in_lds) PVLM_K27_LAUNCH(1, true); else PVLM_K27_LAUNCH(1, false); }
      else if (waves == 2) { if (in_lds) PVLM_K27_LAUNCH(2, true); else PVLM_K27_LAUNCH(2, false); }
      else { if (in_lds) PVLM_K27_LAUNCH(4, true); else PVLM_K27_LAUNCH(4, false); }
#undef PVLM_K27_LAUNCH
      hipEventRecord(ctx->grow_ev[2], gs);
      e = hipGetLastError();
    }
    char* hout = static_cast<char*>(ctx->h_grow_out);
    if (e == hipSuccess) e = hipMemcpyAsync(hout, G->d_counters, 16, hipMemcpyDeviceToHost, gs);
    if (e == hipSuccess) e = hipMemcpyAsync(hout + 16, G->d_status, (size_t)n_scans * 4, hipMemcpyDeviceToHost, gs);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(gs);
      PVLM_SET_ERR(ctx, "pvlm_line_grow_begin: %s", hipGetErrorString(e));
      grow_release(ctx, G); delete G;
      return PVLM_ERR_HIP;
    }
    G->pending = true;
    ctx->grow_in_flight = true;
  } catch (const std::bad_alloc&) {
    if (G) { if (ctx->grow_stream) (void)hipStreamSynchronize(ctx->grow_stream); grow_release(ctx, G); delete G; }
    PVLM_SET_ERR(ctx, "pvlm_line_grow_begin: out of host memory");
    return PVLM_ERR_NOMEM;
  }
  *out = G;
  return PVLM_OK;
}

// Waits for the batch, brings the kept segments down and puts them in walk order.  On failure the object stays valid for pvlm_line_grow_destroy only.
pvlm_status pvlm_line_grow_finish(pvlm_ctx* ctx, pvlm_line_grow* G) {
  if (!ctx || !G) return PVLM_ERR_ARG;
  if (!G->pending) return PVLM_OK;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  hipStream_t gs = ctx->grow_stream;
  pvlm_status st = PVLM_OK;
  auto done = [&](pvlm_status r) { grow_release(ctx, G); G->pending = false; ctx->grow_in_flight = false; return r; };
  if (hipStreamSynchronize(gs) != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_line_grow_finish: device error"); (void)hipGetLastError(); return done(PVLM_ERR_HIP); }
  const char* hout = static_cast<const char*>(ctx->h_grow_out);
  int counters[4];
  std::memcpy(counters, hout, sizeof counters);
  std::memcpy(G->status.data(), hout + 16, (size_t)G->n_scans * 4);
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ctx->grow_ev[1], ctx->grow_ev[2]) == hipSuccess) G->kernel_ms = ms; else (void)hipGetLastError();
  G->tasks_run = counters[3];
  if (counters[2]) { PVLM_SET_ERR(ctx, "pvlm_line_grow_finish: segment pool exhausted (%d segments, %d members)", counters[0], counters[1]); return done(PVLM_ERR_REFUSED); }
  try {
    const size_t n_seg = (size_t)counters[0], n_mem = (size_t)counters[1];
    const SegRecord* segs = reinterpret_cast<const SegRecord*>(hout + 16 + (((size_t)G->n_scans * 4 + 7) & ~(size_t)7));
    const int* pool = reinterpret_cast<const int*>(reinterpret_cast<const char*>(segs) + n_seg * sizeof(SegRecord));
    if (n_seg > 0) {
      hipError_t e = hipMemcpyAsync(const_cast<SegRecord*>(segs), G->d_segs, n_seg * sizeof(SegRecord), hipMemcpyDeviceToHost, gs);
      if (e == hipSuccess && n_mem > 0) e = hipMemcpyAsync(const_cast<int*>(pool), G->d_members, n_mem * sizeof(int), hipMemcpyDeviceToHost, gs);
      if (e == hipSuccess) e = hipStreamSynchronize(gs);
      if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_line_grow_finish: %s", hipGetErrorString(e)); return done(PVLM_ERR_HIP); }
    }
    // the segments of a scan in walk order (their slots came from an atomic counter, their order numbers from the scan's workgroup)
    for (size_t q = 0; q < n_seg; ++q) G->seg_first[(size_t)segs[q].scan + 1]++;
    for (int s = 0; s < G->n_scans; ++s) G->seg_first[(size_t)s + 1] += G->seg_first[(size_t)s];
    std::vector<int> order(n_seg);
    for (size_t q = 0; q < n_seg; ++q) order[(size_t)G->seg_first[(size_t)segs[q].scan] + (size_t)segs[q].order] = (int)q;
    G->seg_task.resize(n_seg); G->coeffs.resize(n_seg * 6); G->members.resize(n_mem); G->seg_off.resize(n_seg + 1);
    size_t at = 0;
    for (size_t k = 0; k < n_seg; ++k) {
      const SegRecord& r = segs[(size_t)order[k]];
      G->seg_task[k] = r.task;
      std::memcpy(&G->members[at], &pool[(size_t)r.mem_off], (size_t)r.count * sizeof(int));
      at += (size_t)r.count;
      G->seg_off[k + 1] = (int)at;
      std::memcpy(&G->coeffs[6 * k], r.coeff, 6 * sizeof(double));
    }
  } catch (const std::bad_alloc&) {
    PVLM_SET_ERR(ctx, "pvlm_line_grow_finish: out of host memory");
    st = PVLM_ERR_NOMEM;
  }
  return done(st);
}

pvlm_status pvlm_line_grow_batch(pvlm_ctx* ctx, int n_scans, const pvlm_edge_cloud* clouds, pvlm_line_grow** out) {
  if (!out) return PVLM_ERR_ARG;
  pvlm_status st = pvlm_line_grow_begin(ctx, n_scans, clouds, out);
  if (st) return st;
  st = pvlm_line_grow_finish(ctx, *out);
  if (st) { pvlm_line_grow_destroy(ctx, *out); *out = nullptr; }
  return st;
}

pvlm_status pvlm_line_grow_scan(const pvlm_line_grow* g, int scan, pvlm_line_grow_result* r) {
  if (!g || !r || scan < 0 || scan >= g->n_scans || g->pending) return PVLM_ERR_ARG;
  const int first = g->seg_first[(size_t)scan], last = g->seg_first[(size_t)scan + 1];
  r->status = g->status[(size_t)scan];
  r->n_points = g->n_points[(size_t)scan];
  r->n_segments = last - first;
  r->seg_task = g->seg_task.data() + first;
  r->seg_offset = g->seg_off.data() + first;
  r->members = g->members.data();
  r->coeffs = g->coeffs.data() + 6 * (size_t)first;
  r->kernel_ms = g->kernel_ms;
  r->tasks_run = g->tasks_run;
  return PVLM_OK;
}

pvlm_status pvlm_line_grow_destroy(pvlm_ctx* ctx, pvlm_line_grow* g) {
  if (!ctx) return PVLM_ERR_ARG;
  if (!g) return PVLM_OK;
  if (g->pending) {                       // begun and never finished: its kernels end before its blocks go back to the pool
    if (pvlm_i_bind(ctx) == PVLM_OK && ctx->grow_stream) (void)hipStreamSynchronize(ctx->grow_stream);
    grow_release(ctx, g);
    ctx->grow_in_flight = false;
  }
  delete g;
  return PVLM_OK;
}

}  // extern "C"

// pvlm_preload: HIP loads the code object of a translation unit at the first launch of one of its kernels (15 ms for the larger ones) — an empty launch from here
// moves that out of the first call that needs this file's kernels
__global__ void k_preload_linegrow() {}
void pvlm_i_preload_linegrow(hipStream_t s) { hipLaunchKernelGGL(k_preload_linegrow, dim3(1), dim3(1), 0, s); }
