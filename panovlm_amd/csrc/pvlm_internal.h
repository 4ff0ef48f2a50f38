// Internal definitions shared by the translation units of libpvlm.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pvlm.h"

#define PVLM_VERSION_STR "panovlm_amd 0.1 (gfx950)"

// Device pose table row: [R_lw row-major (9) | J_l(aa_lw) row-major (9) | t_lw (3)]
#define PVLM_POSE_TAB 21
// Per-pair table row: [R_rn(9) | t_rn(3) | t_rw(3) | Jl_r(9) | M_n = -R_rn*Jl_n (9)]
#define PVLM_PAIR_TAB 33
// Fused partial: [S upper triangle (21) | gv (6) | cost (1)]
#define PVLM_PARTIAL 28

// Device memory of a context comes from a caching sub-allocator: hipMalloc maps pages at 40-70 ms per GB on MI355X
// (tools/micro/alloc_cost.hip: 16 GB 0.64 s, 48 GB 1.9-3.5 s), which made the association call 27x slower than
// its kernels.  Slabs are obtained with hipMalloc (or reserved up front by pvlm_reserve), carved best-fit,
// coalesced on free and kept until pvlm_trim / pvlm_destroy.  Reuse is stream-ordered: every access of the
// library goes through the context's one stream, so a freed range may be handed out again immediately.
struct pvlm_pool {
  struct Range { size_t size; int slab; };
  struct Slab { char* base; size_t size; };
  std::vector<Slab> slabs;
  std::map<char*, Range> free_ranges;             // address-ordered, coalesced inside a slab
  std::unordered_map<const void*, Range> live;
  size_t reserved = 0, in_use = 0, peak = 0;
  uint64_t device_allocs = 0;                     // hipMalloc calls made (a steady state makes none)
  bool disabled = false;                          // PVLM_NO_POOL=1: plain hipMalloc / hipFree
};

// scratch of pvlm_assoc_point2plane that survives the call (two pipeline slots + pinned host staging)
struct pvlm_assoc_ws {
  long long rows = 0; int chunks = 0, pairs = 0;  // capacity of one slot
  int* d_nn[2] = {nullptr, nullptr};              // 10 x rows: the neighbour table K2 hands to K3
  unsigned long long* d_chain[2] = {nullptr, nullptr};   // chunks + pairs + 1 words: K3's look-back chain, the per-pair totals (as ints) behind it, the ticket
  void* d_desc[2] = {nullptr, nullptr};
  int* h_count[2] = {nullptr, nullptr};           // pinned: accepted rows per pair
  void* h_desc[2] = {nullptr, nullptr};           // pinned
  size_t desc_bytes = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
};

// Pinned staging arena.  A copy between a caller's pageable buffer and the device makes the runtime lock and unlock the
// caller's pages (measured: 10-20 ms for 3 MB); through the arena it is a host memcpy + a truly asynchronous DMA.
// H2D: the bytes are copied into the arena when the copy is queued (the source may be reused at once).  D2H: the copy lands
// in the arena and reaches the caller's buffer at the next pvlm_i_sync.  The arena rewinds at pvlm_i_sync; when it is full
// it synchronises itself.
struct pvlm_stage {
  char* base = nullptr;
  size_t size = 0, cursor = 0;
  struct Deferred { void* dst; const void* src; size_t bytes; };
  std::vector<Deferred> deferred;
};

struct pvlm_ctx {
  int device = 0;
  pvlm_pool pool;
  pvlm_assoc_ws assoc_ws;
  uint64_t resset_serial = 0;  // every residual set gets a fresh serial (pvlm_neq caches its binding by it)
  bool capturing = false;      // between pvlm_graph_begin and pvlm_graph_end: no allocation, no synchronisation
  // persistent packed buffer of pvlm_neq_accumulate (host-pointer form)
  double* d_neq_tmp = nullptr; size_t neq_tmp_count = 0;
  // pinned staging of pvlm_scan_upload[_batch] (grow-only)
  void* h_up = nullptr; size_t up_bytes = 0;
  // pinned staging of the voxel-grid build tables (descriptors, point blocks; grow-only): its own buffer, so that the tables of a build queued
  // behind an upload's front copy do not have to wait for that copy to leave h_up
  void* h_grid = nullptr; size_t grid_bytes = 0;
  // pinned buffers destroyed pvlm_ring_batches leave behind for the next ones (hipHostMalloc of a Room batch's 260 MB: 51 ms).  Several: the host
  // mirror runs a call's scans as a sequence of batches whose results stay alive until the call ends (the picks of one overlap the device stages of the next)
  static constexpr int kRingPool = 16;
  void* h_ring[kRingPool] = {}; size_t ring_bytes[kRingPool] = {}; int ring_pool = 0;
  // pinned staging arena of every other host <-> device copy (pvlm_i_h2d_q / pvlm_i_d2h_q / pvlm_i_sync)
  pvlm_stage stage;
  hipStream_t own_stream = nullptr;
  hipStream_t aux_stream = nullptr;   // second stream of the look-ahead Cholesky (K10), created on first use
  hipEvent_t aux_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // K27 (pvlm_line_grow_begin / _finish): its own stream — the line growth of one device batch of scans runs beside the range-image stages of the next —, three
  // events (ordering behind the main stream, kernel timing) and two grow-only pinned buffers (inputs; counters, statuses and the kept segments coming back)
  hipStream_t grow_stream = nullptr;
  hipEvent_t grow_ev[3] = {nullptr, nullptr, nullptr};
  void* h_grow_in = nullptr; size_t grow_in_bytes = 0;
  void* h_grow_out = nullptr; size_t grow_out_bytes = 0;
  bool grow_in_flight = false;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  int cu_count = 0;
  // pose table
  int n_poses = 0, cap_poses = 0;
  bool poses_set = false;
  uint64_t pose_epoch = 0;  // bumps on every pvlm_set_poses
  double* d_aa = nullptr;
  double* d_t = nullptr;
  double* d_pose_tab = nullptr;
  // grow-only device workspace of the dense solver (K10): a Room-sized system is 237 MB, allocating it per LM step costs ms
  void* d_ws = nullptr;
  size_t ws_bytes = 0;
  void* spd_plan = nullptr;           // tile-sparse plan of the last pvlm_spd_solve_blocks structure (csrc/pvlm_linalg.hip), freed by pvlm_i_spd_plan_release
  void* spd_prefetch = nullptr;       // plan being made ahead on a host thread (pvlm_spd_plan_prefetch), joined and freed by pvlm_i_spd_plan_release
  long long spd_prefetch_hits = 0;    // solves that took their plan from a prefetch
  // pvlm_host_alloc / pvlm_host_free: the sizes of the live blocks and ONE freed block kept for the next allocation (the LM driver takes a pinned landing buffer of the
  // same size per Solve: hipHostMalloc + hipHostFree of it cost 2-3 ms each time)
  std::vector<std::pair<void*, size_t>> host_live; void* host_spare = nullptr; size_t host_spare_bytes = 0;
  int spd_one_launch = 1;             // pvlm_spd_one_launch: 1 = the tile-sparse factorisation as ONE launch whose workgroups wait for each other (k_nd_flow), 0 = level launches
  int spd_withhold_task = -1;         // test hook (pvlm_spd_one_launch(ctx, 2 + task, ..)): this task of the one launch never publishes its tile — the recovery path's test
  long long spd_fallbacks = 0;        // solves redone with the level launches because a wait inside the one launch ran into its limit (a GPU shared with other processes)
  // per-kernel profiling (pvlm_profile_*): pending (start, stop) event pairs per kernel class
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending[4];
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[4] = {0, 0, 0, 0};
  int64_t prof_n[4] = {0, 0, 0, 0};
};

// RAII bracket: records events around a kernel launch when profiling is on.
struct pvlm_prof_scope {
  pvlm_ctx* ctx; int which; hipEvent_t e0 = nullptr, e1 = nullptr;
  pvlm_prof_scope(pvlm_ctx* c, int w);
  ~pvlm_prof_scope();
};

struct pvlm_resset {
  int kind = 0;
  unsigned flags = 0;
  double weight = 1.0;
  int64_t n = 0;       // residual blocks (compact / host order)
  int64_t n_dev = 0;   // padded device rows over all column blocks (every segment starts at an even row)
  uint64_t serial = 0; // unique per set and context (see pvlm_ctx::resset_serial)
  int n_pairs = 0;
  int ncols = 0;       // SoA columns
  // Column storage: one or more blocks; block b is ncols x block_rows[b] doubles, column-major (column c at
  // block + c*block_rows[b]).  A pair segment lives in exactly one block.  An uploaded set has one block; the
  // association writes one block per batch of pairs, so that its scratch stays bounded and no pass over all
  // pairs has to finish before the first correspondence is written.
  std::vector<double*> col_blocks;
  std::vector<int64_t> block_rows;          // even
  std::vector<int> h_pair_block;            // n_pairs: block of the pair
  const double** d_pair_cols = nullptr;     // n_pairs: address of the pair's first row in column 0
  int64_t* d_pair_stride = nullptr;         // n_pairs: column stride (rows) of the pair's block
  int64_t* d_out_start = nullptr; // n_pairs+1, compact offsets
  int* d_ref = nullptr;
  int* d_nei = nullptr;
  // block work list: block b handles rows [chunk*chunk_rows, ...) of pair blk_pair[b]
  int n_blocks = 0;
  int chunk_rows = 0;
  bool wave_units = false;   // the fused kernel takes one work-list entry per WAVE (short segments), see pvlm_i_resset_finalize
  int* d_blk_pair = nullptr;
  int* d_blk_chunk = nullptr;
  int* d_pair_blk_start = nullptr;  // n_pairs+1
  double* d_pair_tab = nullptr;     // n_pairs x PVLM_PAIR_TAB
  uint64_t pair_tab_epoch = ~0ull;
  std::vector<int> h_pair_blk_start;  // host mirror of d_pair_blk_start
  double* d_stage = nullptr;        // bounded staging buffer of the host-delivering evaluations (grow-only)
  size_t stage_doubles = 0;
  double* d_partials = nullptr;     // n_blocks x PVLM_PARTIAL
  double* d_pair_blocks = nullptr;  // n_pairs x PVLM_PAIR_BLOCK (scratch for neq accumulate)
  // host mirrors of the segment table (h_seg_start: first row of the pair INSIDE its block, n_pairs entries)
  std::vector<int64_t> h_seg_start, h_out_start;
  std::vector<int> h_ref, h_nei;
  // optional association debug, per column block (compact order inside the block; blocks are in pair order)
  std::vector<int32_t*> d_qidx;  // rows of the block
  std::vector<int32_t*> d_nn;    // rows x 10
  std::vector<int64_t> block_n;  // accepted rows of the block
  int assoc_exact_kernel_batches = 0;   // batches of the association that ran the exact plane-fit kernel (all of them with PVLM_FLAG_ASSOC_EXACT_FIT; all but the probe when the probe's refusal rate said so)
  int64_t assoc_exact_fits = 0;  // queries of the association whose plane came from the exact QR because the certified fast fit refused (0 with PVLM_FLAG_ASSOC_EXACT_FIT)
};

struct pvlm_neq {
  int n_poses = 0, n_upairs = 0;
  std::vector<int> ui, uj;
  // binding to a residual set (rebuilt when the set changes).  Keyed by the set's serial, not only by its
  // address: a set destroyed and re-created by a re-association routinely gets the old address back.
  const pvlm_resset* bound = nullptr;
  uint64_t bound_serial = 0;
  int* d_diag_off = nullptr;   // n_poses+1 : CSR of (pair, role) incident to each pose
  int* d_diag_items = nullptr; // item = pair*2 + role (0 = pose is ref, 1 = pose is nei)
  int* d_off_off = nullptr;    // n_upairs+1
  int* d_off_items = nullptr;  // item = pair*2 + transposed
  int64_t n_diag_items = 0, n_off_items = 0;
  double* d_packed = nullptr;  // the structure's own packed buffer (pvlm_neq_accumulate_async), allocated on first use
};

struct pvlm_cloud {
  int n = 0;
  float* d_xyz = nullptr;   // interleaved x y z, original order
  float* d_tag = nullptr;
  // voxel hash (built at upload): points sorted by cell
  float cell = 0.f;
  float origin[3] = {0, 0, 0};
  int table_size = 0;            // hash: power-of-two slots; dense: ncells + 1
  int dense = 0, nx = 0, ny = 0, nz = 0;  // dense grid when the bounding box is small enough (nx: fine cells, see xf)
  int xf = 1;                             // dense grid: cells are xf times finer along x
  unsigned long long* d_keys = nullptr;  // table_size, ~0 = empty
  int* d_cell_start = nullptr;   // table_size
  int* d_cell_count = nullptr;   // table_size
  float4* d_sorted = nullptr;    // n: (x,y,z, as_float(original index))
  float4* d_pt4 = nullptr;       // n: (x,y,z,tag) in original order (clouds with tags)
  int* d_sorted_cell = nullptr;  // unused placeholder
  bool grid_stale = false;       // the points were transformed in place without a rebuild of the grid (pvlm_scan_transform_batch, rebuild_grids = 0)
};

struct pvlm_scan_slab { void* base = nullptr; int refs = 0; };   // one device allocation shared by the scans of an upload (or re-pose) batch

struct pvlm_scan {
  int id = 0;
  pvlm_scan_slab* slab = nullptr;        // the uploaded arrays: float clouds (transformed in place by pvlm_scan_transform_batch), tags, point_to_segment, segment points
  pvlm_scan_slab* grid_slab = nullptr;   // the voxel grids built from them (cell tables, sorted points, pt4): replaced whenever the clouds are re-posed
  double R_wl[9];
  double t_wl[3];
  pvlm_cloud flat, less, corner;
  int n_segments = 0;
  std::vector<int> h_p2s_off, h_p2s_ids, h_seg_size;
  std::vector<double> h_seg_coeffs, h_end_points;
  int* d_p2s_off = nullptr;
  int* d_p2s_ids = nullptr;
  float* d_seg_xyz = nullptr;          // optional: points of every segment, one segment after the other (world frame)
  std::vector<int> h_seg_pt_off;       // n_segments + 1 offsets into d_seg_xyz (points)
};

#define PVLM_SET_ERR(ctx, ...)                                   \
  do {                                                           \
    char _b[512];                                                \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                       \
    (ctx)->err = _b;                                             \
  } while (0)

#define PVLM_HIP(ctx, call)                                                                        \
  do {                                                                                             \
    hipError_t _e = (call);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      PVLM_SET_ERR(ctx, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e));        \
      return (_e == hipErrorOutOfMemory) ? PVLM_ERR_NOMEM : PVLM_ERR_HIP;                          \
    }                                                                                              \
  } while (0)

// PVLM_TRACE=<file>: appends "label +ms since the previous mark" lines — host-side wall clock between marks inside the
// multi-step entry points (where does a call's time go when its kernels take microseconds?)
void pvlm_i_trace(const char* label);
// helpers implemented in pvlm_ctx.hip
pvlm_status pvlm_i_bind(pvlm_ctx* ctx);  // hipSetDevice(ctx->device)
// Rows a pair's segment occupies inside a column block: a multiple of 16 rows = 128 B, so that every column of every pair starts on
// a cache line (the pool hands out 256-B aligned blocks and the column stride is a sum of padded segments).  With segments padded
// to 2 rows only (round 1) seven of eight pairs started inside a line: every 1 KiB wave access of k_eval_fused touched 9 lines
// instead of 8 (FETCH_SIZE 1.07 x the algorithmic bytes once the loads are non-temporal, profiles/r2_pmc_traffic_default.json).
inline int64_t pvlm_i_seg_rows(int64_t rows) { return (rows + 15) & ~int64_t(15); }
pvlm_status pvlm_i_alloc_bytes(pvlm_ctx* ctx, void** p, size_t bytes);   // from the context's pool
void pvlm_i_free(pvlm_ctx* ctx, const void* p);                           // back to the pool (NULL ok; foreign pointers go to hipFree)
void pvlm_i_pool_release(pvlm_ctx* ctx, bool all);                        // hipFree the fully free slabs (all: every slab, at destroy)
template <typename T>
inline pvlm_status pvlm_i_alloc(pvlm_ctx* ctx, T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  return pvlm_i_alloc_bytes(ctx, (void**)p, count * sizeof(T));
}
// queued copies through the staging arena + the synchronisation that completes them (see pvlm_stage)
pvlm_status pvlm_i_h2d_q(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes);
pvlm_status pvlm_i_d2h_q(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes);
pvlm_status pvlm_i_sync(pvlm_ctx* ctx);
// plain stream synchronisation of an entry point that is not capturable: PVLM_ERR_STATE inside a graph capture (a synchronisation
// would invalidate the capture), otherwise hipStreamSynchronize
pvlm_status pvlm_i_stream_sync(pvlm_ctx* ctx);
#define PVLM_TRY_SYNC(ctx) do { const pvlm_status _s = pvlm_i_stream_sync(ctx); if (_s) return _s; } while (0)
// host -> device through the context stream (pageable source: returns when the source may be reused)
pvlm_status pvlm_i_h2d(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes);
pvlm_status pvlm_i_d2h(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes);
void pvlm_i_assoc_ws_free(pvlm_ctx* ctx);
void pvlm_i_spd_plan_release(pvlm_ctx* ctx);
void pvlm_i_preload_assoc(hipStream_t s);
void pvlm_i_preload_ba(hipStream_t s);
void pvlm_i_preload_eval(hipStream_t s);
void pvlm_i_preload_linalg(hipStream_t s);
void pvlm_i_preload_linegrow(hipStream_t s);
void pvlm_i_preload_lines(hipStream_t s);
void pvlm_i_preload_mvs(hipStream_t s);
void pvlm_i_preload_ring(hipStream_t s);
void pvlm_i_preload_undistort(hipStream_t s);
// builds work list + scratch for a resset whose segment table is final (h_* mirrors filled)
pvlm_status pvlm_i_resset_finalize(pvlm_ctx* ctx, pvlm_resset* rs);
pvlm_status pvlm_i_resset_free(pvlm_ctx* ctx, pvlm_resset* rs);
int pvlm_i_ncols(int kind);
int pvlm_i_stride(int kind);
