// Internal definitions shared by the translation units of libpvlm.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/pvlm.h"

#define PVLM_VERSION_STR "panovlm_amd 0.1 (gfx950)"

// Device pose table row: [R_lw row-major (9) | J_l(aa_lw) row-major (9) | t_lw (3)]
#define PVLM_POSE_TAB 21
// Per-pair table row: [R_rn(9) | t_rn(3) | t_rw(3) | Jl_r(9) | M_n = -R_rn*Jl_n (9)]
#define PVLM_PAIR_TAB 33
// Fused partial: [S upper triangle (21) | gv (6) | cost (1)]
#define PVLM_PARTIAL 28

struct pvlm_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
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
  // per-kernel profiling (pvlm_profile_*): pending (start, stop) event pairs per kernel class
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending[3];
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[3] = {0, 0, 0};
  int64_t prof_n[3] = {0, 0, 0};
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
  int64_t n_dev = 0;   // padded device rows (every segment starts at an even row)
  int n_pairs = 0;
  int ncols = 0;       // SoA columns
  double* d_cols = nullptr;       // ncols x n_dev, column-major (column c at d_cols + c*n_dev)
  int64_t* d_seg_start = nullptr; // n_pairs+1, padded device row offsets (host-uploaded or device-built)
  int64_t* d_out_start = nullptr; // n_pairs+1, compact offsets
  int* d_ref = nullptr;
  int* d_nei = nullptr;
  // block work list: block b handles rows [chunk*chunk_rows, ...) of pair blk_pair[b]
  int n_blocks = 0;
  int chunk_rows = 0;
  int* d_blk_pair = nullptr;
  int* d_blk_chunk = nullptr;
  int* d_pair_blk_start = nullptr;  // n_pairs+1
  double* d_pair_tab = nullptr;     // n_pairs x PVLM_PAIR_TAB
  uint64_t pair_tab_epoch = ~0ull;
  double* d_partials = nullptr;     // n_blocks x PVLM_PARTIAL
  double* d_pair_blocks = nullptr;  // n_pairs x PVLM_PAIR_BLOCK (scratch for neq accumulate)
  // host mirrors of the segment table
  std::vector<int64_t> h_seg_start, h_out_start;
  std::vector<int> h_ref, h_nei;
  // optional association debug
  int32_t* d_qidx = nullptr;  // n (compact)
  int32_t* d_nn = nullptr;    // n x 10 (compact)
};

struct pvlm_neq {
  int n_poses = 0, n_upairs = 0;
  std::vector<int> ui, uj;
  // binding to a residual set (rebuilt when the set changes)
  const pvlm_resset* bound = nullptr;
  int* d_diag_off = nullptr;   // n_poses+1 : CSR of (pair, role) incident to each pose
  int* d_diag_items = nullptr; // item = pair*2 + role (0 = pose is ref, 1 = pose is nei)
  int* d_off_off = nullptr;    // n_upairs+1
  int* d_off_items = nullptr;  // item = pair*2 + transposed
  int64_t n_diag_items = 0, n_off_items = 0;
};

struct pvlm_cloud {
  int n = 0;
  float* d_xyz = nullptr;   // SoA: x[n] y[n] z[n]  (original order)
  float* d_tag = nullptr;
  // voxel hash (built at upload): points sorted by cell
  float cell = 0.f;
  float origin[3] = {0, 0, 0};
  int table_size = 0;            // hash: power-of-two slots; dense: ncells + 1
  int dense = 0, nx = 0, ny = 0, nz = 0;  // dense grid when the bounding box is small enough
  unsigned long long* d_keys = nullptr;  // table_size, ~0 = empty
  int* d_cell_start = nullptr;   // table_size
  int* d_cell_count = nullptr;   // table_size
  float4* d_sorted = nullptr;    // n: (x,y,z, as_float(original index))
  int* d_sorted_cell = nullptr;  // unused placeholder
};

struct pvlm_scan {
  int id = 0;
  double R_wl[9];
  double t_wl[3];
  pvlm_cloud flat, less, corner;
  int n_segments = 0;
  std::vector<int> h_p2s_off, h_p2s_ids, h_seg_size;
  std::vector<double> h_seg_coeffs, h_end_points;
  int* d_p2s_off = nullptr;
  int* d_p2s_ids = nullptr;
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

// helpers implemented in pvlm_ctx.hip
pvlm_status pvlm_i_bind(pvlm_ctx* ctx);  // hipSetDevice(ctx->device)
template <typename T>
inline pvlm_status pvlm_i_alloc(pvlm_ctx* ctx, T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  PVLM_HIP(ctx, hipMalloc((void**)p, count * sizeof(T)));
  return PVLM_OK;
}
// builds work list + scratch for a resset whose segment table is final (h_* mirrors filled)
pvlm_status pvlm_i_resset_finalize(pvlm_ctx* ctx, pvlm_resset* rs);
pvlm_status pvlm_i_resset_free(pvlm_ctx* ctx, pvlm_resset* rs);
int pvlm_i_ncols(int kind);
int pvlm_i_stride(int kind);
