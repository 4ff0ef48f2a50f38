// Context, pose table, residual-set and normal-equation bookkeeping of libpvlm.so.
// Host-side plumbing only; the kernels live in pvlm_eval.hip / pvlm_assoc.hip / pvlm_lines.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "pvlm_internal.h"

pvlm_status pvlm_i_bind(pvlm_ctx* ctx) {
  PVLM_HIP(ctx, hipSetDevice(ctx->device));
  return PVLM_OK;
}

pvlm_prof_scope::pvlm_prof_scope(pvlm_ctx* c, int w) : ctx(c), which(w) {
  if (!ctx->profiling) return;
  auto get = [&]() -> hipEvent_t {
    if (!ctx->prof_pool.empty()) { hipEvent_t e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  };
  e0 = get(); e1 = get();
  if (e0 && e1) (void)hipEventRecord(e0, ctx->stream);
}
pvlm_prof_scope::~pvlm_prof_scope() {
  if (!e0 || !e1) return;
  (void)hipEventRecord(e1, ctx->stream);
  ctx->prof_pending[which].push_back({e0, e1});
}

int pvlm_i_ncols(int kind) {
  switch (kind) {
    case PVLM_POINT2PLANE_METER: case PVLM_POINT2PLANE_ANGLE: return 7;
    case PVLM_POINT2LINE_METER: case PVLM_POINT2LINE_ANGLE: return 9;
    case PVLM_PLANE2PLANE_GLOBAL: return 10;
    case PVLM_PLANE_IOU: return 12;
    default: return -1;
  }
}
int pvlm_i_stride(int kind) { return pvlm_i_ncols(kind); }

extern "C" {

const char* pvlm_version(void) { return PVLM_VERSION_STR; }

pvlm_status pvlm_create(int device, pvlm_ctx** out) {
  if (!out) return PVLM_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return PVLM_ERR_HIP;
  pvlm_ctx* ctx = new (std::nothrow) pvlm_ctx();
  if (!ctx) return PVLM_ERR_NOMEM;
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    delete ctx;
    return PVLM_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  *out = ctx;
  return PVLM_OK;
}

pvlm_status pvlm_destroy(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  hipFree(ctx->d_aa); hipFree(ctx->d_t); hipFree(ctx->d_pose_tab); hipFree(ctx->d_ws);
  hipEventDestroy(ctx->ev0); hipEventDestroy(ctx->ev1);
  for (int w = 0; w < 3; ++w) for (auto& pr : ctx->prof_pending[w]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  for (hipEvent_t e : ctx->prof_pool) hipEventDestroy(e);
  hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return PVLM_OK;
}

const char* pvlm_last_error(const pvlm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

pvlm_status pvlm_set_stream(pvlm_ctx* ctx, void* s) {
  if (!ctx) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // nothing of ours may still be in flight on the old stream
  ctx->stream = (hipStream_t)s;
  return PVLM_OK;
}

pvlm_status pvlm_use_own_stream(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = ctx->own_stream;
  return PVLM_OK;
}

pvlm_status pvlm_synchronize(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PVLM_OK;
}

pvlm_status pvlm_timer_start(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return PVLM_OK;
}

pvlm_status pvlm_timer_stop(pvlm_ctx* ctx, float* ms) {
  if (!ctx || !ms) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  PVLM_HIP(ctx, hipEventSynchronize(ctx->ev1));
  PVLM_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return PVLM_OK;
}

pvlm_status pvlm_device_info(pvlm_ctx* ctx, int* cu, int64_t* hbm, char* name, int cap) {
  if (!ctx) return PVLM_ERR_ARG;
  hipDeviceProp_t prop;
  PVLM_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (cu) *cu = prop.multiProcessorCount;
  if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
  if (name && cap > 0) { std::strncpy(name, prop.gcnArchName, cap - 1); name[cap - 1] = 0; }
  return PVLM_OK;
}

pvlm_status pvlm_profile_enable(pvlm_ctx* ctx, int on) {
  if (!ctx) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int w = 0; w < 3; ++w) {
    for (auto& pr : ctx->prof_pending[w]) { ctx->prof_pool.push_back(pr.first); ctx->prof_pool.push_back(pr.second); }
    ctx->prof_pending[w].clear();
    ctx->prof_ms[w] = 0; ctx->prof_n[w] = 0;
  }
  ctx->profiling = on != 0;
  return PVLM_OK;
}

pvlm_status pvlm_profile_read(pvlm_ctx* ctx, int which, double* total_ms, int64_t* launches) {
  if (!ctx || which < 0 || which > 2) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& pr : ctx->prof_pending[which]) {
    float ms = 0.f;
    PVLM_HIP(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
    ctx->prof_ms[which] += ms; ctx->prof_n[which] += 1;
    ctx->prof_pool.push_back(pr.first); ctx->prof_pool.push_back(pr.second);
  }
  ctx->prof_pending[which].clear();
  if (total_ms) *total_ms = ctx->prof_ms[which];
  if (launches) *launches = ctx->prof_n[which];
  return PVLM_OK;
}

// ------------------------------------------------------------------------------------------------
// residual sets
// ------------------------------------------------------------------------------------------------
pvlm_status pvlm_resset_info(const pvlm_resset* rs, int64_t* n, int* n_pairs, int* kind, unsigned* flags) {
  if (!rs) return PVLM_ERR_ARG;
  if (n) *n = rs->n;
  if (n_pairs) *n_pairs = rs->n_pairs;
  if (kind) *kind = rs->kind;
  if (flags) *flags = rs->flags;
  return PVLM_OK;
}

}  // extern "C"

pvlm_status pvlm_i_resset_free(pvlm_ctx* ctx, pvlm_resset* rs) {
  if (!rs) return PVLM_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  hipFree(rs->d_cols); hipFree(rs->d_seg_start); hipFree(rs->d_out_start); hipFree(rs->d_ref); hipFree(rs->d_nei);
  hipFree(rs->d_blk_pair); hipFree(rs->d_blk_chunk); hipFree(rs->d_pair_blk_start); hipFree(rs->d_pair_tab);
  hipFree(rs->d_partials); hipFree(rs->d_pair_blocks); hipFree(rs->d_qidx); hipFree(rs->d_nn);
  delete rs;
  return PVLM_OK;
}

// Needs: kind, n_pairs, h_seg_start, h_out_start, h_ref, h_nei filled; d_cols allocated & filled;
// d_seg_start / d_out_start / d_ref / d_nei uploaded.  Builds the block work list and scratch.
pvlm_status pvlm_i_resset_finalize(pvlm_ctx* ctx, pvlm_resset* rs) {
  const int P = rs->n_pairs;
  int64_t total = rs->n_dev;
  int64_t chunk = ((total / 4096 + 511) / 512) * 512;
  chunk = std::max<int64_t>(512, std::min<int64_t>(16384, chunk));
  rs->chunk_rows = (int)chunk;
  std::vector<int> blk_pair, blk_chunk, pair_blk_start(P + 1, 0);
  for (int p = 0; p < P; ++p) {
    pair_blk_start[p] = (int)blk_pair.size();
    const int64_t len = rs->h_out_start[p + 1] - rs->h_out_start[p];
    const int nch = (int)((len + chunk - 1) / chunk);
    for (int c = 0; c < nch; ++c) { blk_pair.push_back(p); blk_chunk.push_back(c); }
  }
  pair_blk_start[P] = (int)blk_pair.size();
  rs->n_blocks = (int)blk_pair.size();
  pvlm_status st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_blk_pair, blk_pair.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_blk_chunk, blk_chunk.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_blk_start, pair_blk_start.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_tab, (size_t)std::max(P, 1) * PVLM_PAIR_TAB))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_partials, (size_t)std::max(rs->n_blocks, 1) * PVLM_PARTIAL))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_blocks, (size_t)std::max(P, 1) * PVLM_PAIR_BLOCK))) return st;
  if (rs->n_blocks) {
    PVLM_HIP(ctx, hipMemcpyAsync(rs->d_blk_pair, blk_pair.data(), blk_pair.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PVLM_HIP(ctx, hipMemcpyAsync(rs->d_blk_chunk, blk_chunk.data(), blk_chunk.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  }
  PVLM_HIP(ctx, hipMemcpyAsync(rs->d_pair_blk_start, pair_blk_start.data(), pair_blk_start.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
  rs->pair_tab_epoch = ~0ull;
  return PVLM_OK;
}

extern "C" {

pvlm_status pvlm_resset_upload(pvlm_ctx* ctx, pvlm_functor kind, unsigned flags, double weight, int64_t n, int n_pairs,
                               const int64_t* pair_offsets, const int* pair_ref, const int* pair_nei, const double* rows,
                               int stride, pvlm_resset** out) {
  if (!ctx || !out) return PVLM_ERR_ARG;
  *out = nullptr;
  const int ncols = pvlm_i_ncols(kind);
  if (ncols < 0) { PVLM_SET_ERR(ctx, "unknown functor kind %d", (int)kind); return PVLM_ERR_ARG; }
  if (n < 0 || n_pairs < 0 || (n > 0 && !rows) || (n_pairs > 0 && (!pair_offsets || !pair_ref || !pair_nei)) || stride != pvlm_i_stride(kind)) {
    PVLM_SET_ERR(ctx, "pvlm_resset_upload: bad arguments (n=%lld pairs=%d stride=%d, expected stride %d)", (long long)n, n_pairs, stride, pvlm_i_stride(kind));
    return PVLM_ERR_ARG;
  }
  if (n_pairs > 0 && (pair_offsets[0] != 0 || pair_offsets[n_pairs] != n)) { PVLM_SET_ERR(ctx, "pair_offsets must span [0,n]"); return PVLM_ERR_ARG; }
  for (int p = 0; p < n_pairs; ++p)
    if (pair_offsets[p + 1] < pair_offsets[p] || pair_ref[p] < 0 || pair_nei[p] < 0) { PVLM_SET_ERR(ctx, "segment %d malformed", p); return PVLM_ERR_ARG; }
  if (n_pairs == 0 && n != 0) { PVLM_SET_ERR(ctx, "rows without segments"); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;

  pvlm_resset* rs = new (std::nothrow) pvlm_resset();
  if (!rs) return PVLM_ERR_NOMEM;
  rs->kind = kind; rs->flags = flags; rs->weight = weight; rs->n = n; rs->n_pairs = n_pairs; rs->ncols = ncols;
  rs->h_out_start.assign(pair_offsets, pair_offsets + n_pairs + (n_pairs ? 1 : 0));
  if (n_pairs == 0) rs->h_out_start.assign(1, 0);
  rs->h_ref.assign(pair_ref, pair_ref + n_pairs);
  rs->h_nei.assign(pair_nei, pair_nei + n_pairs);
  rs->h_seg_start.resize(n_pairs + 1);
  int64_t o = 0;
  for (int p = 0; p < n_pairs; ++p) { rs->h_seg_start[p] = o; o += ((rs->h_out_start[p + 1] - rs->h_out_start[p]) + 1) & ~int64_t(1); }
  rs->h_seg_start[n_pairs] = o;
  rs->n_dev = o;

  // host staging: SoA, padded; per-functor normalisations the reference does in its constructors
  std::vector<double> cols((size_t)ncols * std::max<int64_t>(rs->n_dev, 1), 0.0);
  for (int p = 0; p < n_pairs; ++p) {
    for (int64_t i = rs->h_out_start[p]; i < rs->h_out_start[p + 1]; ++i) {
      const double* r = rows + (size_t)i * stride;
      const int64_t d = rs->h_seg_start[p] + (i - rs->h_out_start[p]);
      double c[12];
      switch (kind) {
        case PVLM_POINT2PLANE_METER: case PVLM_POINT2PLANE_ANGLE:
          for (int k = 0; k < 7; ++k) c[k] = r[k];
          break;
        case PVLM_POINT2LINE_METER: case PVLM_POINT2LINE_ANGLE: {
          // ctor: line_direction = (A - B).normalized()   (base/CostFunction.h:778-783, :845-850)
          double dx = r[3] - r[6], dy = r[4] - r[7], dz = r[5] - r[8];
          const double n2 = dx * dx + dy * dy + dz * dz;
          if (n2 > 0.0) { const double nn = std::sqrt(n2); dx /= nn; dy /= nn; dz /= nn; }
          for (int k = 0; k < 6; ++k) c[k] = r[k];
          c[6] = dx; c[7] = dy; c[8] = dz;
          break;
        }
        case PVLM_PLANE2PLANE_GLOBAL: {
          // ctor: plane_ref.normalize()   (base/CostFunction.h:357-362)
          const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
          const double nn = std::sqrt(n2);
          for (int k = 0; k < 3; ++k) c[k] = n2 > 0.0 ? r[k] / nn : r[k];
          for (int k = 3; k < 10; ++k) c[k] = r[k];
          break;
        }
        case PVLM_PLANE_IOU: {
          // ctor: ref_plane = plane / plane.head<3>().norm()   (base/CostFunction.h:453-460)
          const double nn = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
          for (int k = 0; k < 4; ++k) c[k] = r[k] / nn;
          for (int k = 4; k < 12; ++k) c[k] = r[k];
          break;
        }
      }
      for (int k = 0; k < ncols; ++k) cols[(size_t)k * rs->n_dev + d] = c[k];
    }
  }
  pvlm_status st = PVLM_OK;
#define TRY(x) do { if ((st = (x)) != PVLM_OK) { pvlm_i_resset_free(ctx, rs); return st; } } while (0)
  TRY(pvlm_i_alloc(ctx, &rs->d_cols, cols.size()));
  TRY(pvlm_i_alloc(ctx, &rs->d_seg_start, (size_t)n_pairs + 1));
  TRY(pvlm_i_alloc(ctx, &rs->d_out_start, (size_t)n_pairs + 1));
  TRY(pvlm_i_alloc(ctx, &rs->d_ref, (size_t)n_pairs));
  TRY(pvlm_i_alloc(ctx, &rs->d_nei, (size_t)n_pairs));
  auto cp = [&](void* d, const void* h, size_t bytes) -> pvlm_status {
    if (bytes == 0) return PVLM_OK;
    PVLM_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PVLM_OK;
  };
  TRY(cp(rs->d_cols, cols.data(), cols.size() * sizeof(double)));
  TRY(cp(rs->d_seg_start, rs->h_seg_start.data(), rs->h_seg_start.size() * sizeof(int64_t)));
  TRY(cp(rs->d_out_start, rs->h_out_start.data(), rs->h_out_start.size() * sizeof(int64_t)));
  TRY(cp(rs->d_ref, rs->h_ref.data(), rs->h_ref.size() * sizeof(int)));
  TRY(cp(rs->d_nei, rs->h_nei.data(), rs->h_nei.size() * sizeof(int)));
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { pvlm_i_resset_free(ctx, rs); return PVLM_ERR_HIP; }
  TRY(pvlm_i_resset_finalize(ctx, rs));
#undef TRY
  *out = rs;
  return PVLM_OK;
}

pvlm_status pvlm_resset_destroy(pvlm_ctx* ctx, pvlm_resset* rs) {
  if (!ctx) return PVLM_ERR_ARG;
  return pvlm_i_resset_free(ctx, rs);
}

pvlm_status pvlm_resset_download(pvlm_ctx* ctx, const pvlm_resset* rs, int64_t* pair_offsets, int* pair_ref, int* pair_nei, double* rows) {
  if (!ctx || !rs) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (pair_offsets) std::memcpy(pair_offsets, rs->h_out_start.data(), rs->h_out_start.size() * sizeof(int64_t));
  if (pair_ref && rs->n_pairs) std::memcpy(pair_ref, rs->h_ref.data(), rs->h_ref.size() * sizeof(int));
  if (pair_nei && rs->n_pairs) std::memcpy(pair_nei, rs->h_nei.data(), rs->h_nei.size() * sizeof(int));
  if (rows && rs->n > 0) {
    std::vector<double> cols((size_t)rs->ncols * rs->n_dev);
    PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PVLM_HIP(ctx, hipMemcpy(cols.data(), rs->d_cols, cols.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int stride = pvlm_i_stride(rs->kind);
    for (int p = 0; p < rs->n_pairs; ++p)
      for (int64_t i = rs->h_out_start[p]; i < rs->h_out_start[p + 1]; ++i) {
        const int64_t d = rs->h_seg_start[p] + (i - rs->h_out_start[p]);
        double* r = rows + (size_t)i * stride;
        for (int k = 0; k < rs->ncols; ++k) r[k] = cols[(size_t)k * rs->n_dev + d];
        if (rs->kind == PVLM_POINT2LINE_METER || rs->kind == PVLM_POINT2LINE_ANGLE) {
          // device keeps (A, unit direction); hand back B' = A - direction (same line)
          for (int k = 0; k < 3; ++k) r[6 + k] = r[3 + k] - r[6 + k];
        }
      }
  }
  return PVLM_OK;
}

// ------------------------------------------------------------------------------------------------
// normal-equation structure
// ------------------------------------------------------------------------------------------------
pvlm_status pvlm_neq_create(pvlm_ctx* ctx, int n_poses, int n_upairs, const int* ui, const int* uj, pvlm_neq** out) {
  if (!ctx || !out || n_poses < 0 || n_upairs < 0 || (n_upairs > 0 && (!ui || !uj))) return PVLM_ERR_ARG;
  *out = nullptr;
  for (int u = 0; u < n_upairs; ++u)
    if (!(0 <= ui[u] && ui[u] < uj[u] && uj[u] < n_poses)) { PVLM_SET_ERR(ctx, "upair %d must satisfy 0 <= i < j < n_poses", u); return PVLM_ERR_ARG; }
  pvlm_neq* q = new (std::nothrow) pvlm_neq();
  if (!q) return PVLM_ERR_NOMEM;
  q->n_poses = n_poses; q->n_upairs = n_upairs;
  q->ui.assign(ui, ui + n_upairs); q->uj.assign(uj, uj + n_upairs);
  *out = q;
  return PVLM_OK;
}

pvlm_status pvlm_neq_destroy(pvlm_ctx* ctx, pvlm_neq* q) {
  if (!ctx) return PVLM_ERR_ARG;
  if (!q) return PVLM_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  hipFree(q->d_diag_off); hipFree(q->d_diag_items); hipFree(q->d_off_off); hipFree(q->d_off_items);
  delete q;
  return PVLM_OK;
}

int64_t pvlm_neq_size(const pvlm_neq* q) { return q ? (int64_t)q->n_poses * 36 + (int64_t)q->n_upairs * 36 + (int64_t)q->n_poses * 6 + 1 : -1; }

}  // extern "C"
