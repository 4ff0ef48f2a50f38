// Context, pose table, residual-set and normal-equation bookkeeping of libpvlm.so.
// Host-side plumbing only; the kernels live in pvlm_eval.hip / pvlm_assoc.hip / pvlm_lines.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "pvlm_internal.h"
#include <chrono>

pvlm_status pvlm_i_bind(pvlm_ctx* ctx) {
  PVLM_HIP(ctx, hipSetDevice(ctx->device));
  return PVLM_OK;
}

// ------------------------------------------------------------------------------------------------
// device memory pool
// ------------------------------------------------------------------------------------------------
static const size_t kPoolAlign = 256;
static const size_t kSlabGranule = 2u << 20;
static const size_t kSmallSlab = 256u << 20;

static size_t round_up(size_t v, size_t g) { return (v + g - 1) / g * g; }

static void pool_insert_free(pvlm_pool& P, char* base, size_t size, int slab) {
  auto nx = P.free_ranges.lower_bound(base);
  if (nx != P.free_ranges.end() && nx->second.slab == slab && base + size == nx->first) {   // merge with the next range
    size += nx->second.size;
    nx = P.free_ranges.erase(nx);
  }
  if (nx != P.free_ranges.begin()) {
    auto pv = std::prev(nx);
    if (pv->second.slab == slab && pv->first + pv->second.size == base) { pv->second.size += size; return; }
  }
  P.free_ranges.emplace(base, pvlm_pool::Range{size, slab});
}

static bool pool_add_slab(pvlm_ctx* ctx, size_t bytes) {
  pvlm_pool& P = ctx->pool;
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
  P.device_allocs++;
  int id = -1;
  for (size_t i = 0; i < P.slabs.size(); ++i) if (!P.slabs[i].base) { id = (int)i; break; }
  if (id < 0) { id = (int)P.slabs.size(); P.slabs.push_back({nullptr, 0}); }
  P.slabs[id] = {(char*)p, bytes};
  P.reserved += bytes;
  pool_insert_free(P, (char*)p, bytes, id);
  return true;
}

void pvlm_i_pool_release(pvlm_ctx* ctx, bool all) {
  pvlm_pool& P = ctx->pool;
  (void)hipStreamSynchronize(ctx->stream);   // ranges freed stream-ordered may still be read by kernels in flight
  for (size_t i = 0; i < P.slabs.size(); ++i) {
    pvlm_pool::Slab& s = P.slabs[i];
    if (!s.base) continue;
    auto it = P.free_ranges.find(s.base);
    const bool whole = it != P.free_ranges.end() && it->second.slab == (int)i && it->second.size == s.size;
    if (!whole && !all) continue;
    if (whole) P.free_ranges.erase(it);
    else for (auto f = P.free_ranges.begin(); f != P.free_ranges.end();) f = (f->second.slab == (int)i) ? P.free_ranges.erase(f) : std::next(f);
    (void)hipFree(s.base);
    P.reserved -= s.size;
    s = {nullptr, 0};
  }
  if (all) { P.live.clear(); P.in_use = 0; }
}

pvlm_status pvlm_i_alloc_bytes(pvlm_ctx* ctx, void** out, size_t bytes) {
  *out = nullptr;
  pvlm_pool& P = ctx->pool;
  if (ctx->capturing) { PVLM_SET_ERR(ctx, "device allocation inside a graph capture (run the step once before pvlm_graph_begin)"); return PVLM_ERR_STATE; }
  if (P.disabled) { PVLM_HIP(ctx, hipMalloc(out, bytes)); P.device_allocs++; return PVLM_OK; }
  const size_t need = round_up(std::max<size_t>(bytes, 1), kPoolAlign);
  for (int attempt = 0; attempt < 3; ++attempt) {
    auto best = P.free_ranges.end();
    for (auto it = P.free_ranges.begin(); it != P.free_ranges.end(); ++it)
      if (it->second.size >= need && (best == P.free_ranges.end() || it->second.size < best->second.size)) best = it;
    if (best != P.free_ranges.end()) {
      char* base = best->first;
      const pvlm_pool::Range r = best->second;
      P.free_ranges.erase(best);
      if (r.size > need) P.free_ranges.emplace(base + need, pvlm_pool::Range{r.size - need, r.slab});
      P.live.emplace(base, pvlm_pool::Range{need, r.slab});
      P.in_use += need;
      P.peak = std::max(P.peak, P.in_use);
      *out = base;
      return PVLM_OK;
    }
    // miss: small requests share 256 MB slabs; a large one gets its own slab with 1/8 headroom, so that the next
    // outer iteration's slightly different size still fits the cached slab
    size_t slab = need <= kSmallSlab / 2 ? kSmallSlab : round_up(need + need / 8, kSlabGranule);
    if (attempt == 0 && need > kSmallSlab / 2) pvlm_i_pool_release(ctx, false);   // stale cached slabs could not serve it: give them back first
    if (pool_add_slab(ctx, slab)) continue;
    pvlm_i_pool_release(ctx, false);
    if (pool_add_slab(ctx, round_up(need, kSlabGranule))) continue;
    break;
  }
  PVLM_SET_ERR(ctx, "device allocation of %.3f GB failed (pool: %.3f GB reserved, %.3f GB in use)", bytes / 1e9, P.reserved / 1e9, P.in_use / 1e9);
  return PVLM_ERR_NOMEM;
}

void pvlm_i_free(pvlm_ctx* ctx, const void* p) {
  if (!p) return;
  pvlm_pool& P = ctx->pool;
  auto it = P.live.find(p);
  if (it == P.live.end()) {
    // PVLM_NO_POOL=1: every block is a plain hipMalloc.  With the pool on there are no foreign pointers: an unknown one is a double
    // free or an interior pointer — diagnosed, never handed to hipFree (its range may belong to another object by now)
    // no pool: the stream-ordered reuse argument does not apply to hipFree — wait for the work that may still read the buffer
    if (P.disabled) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(const_cast<void*>(p)); return; }
    PVLM_SET_ERR(ctx, "pvlm_i_free: %p is not a live block of this context's pool (double free or interior pointer)", p);
    fprintf(stderr, "[pvlm] %s\n", ctx->err.c_str());
    return;
  }
  const pvlm_pool::Range r = it->second;
  P.live.erase(it);
  P.in_use -= r.size;
  pool_insert_free(P, (char*)const_cast<void*>(p), r.size, r.slab);
}

static const size_t kStageBytes = (size_t)32 << 20;

// Inside a graph capture (pvlm_graph_begin .. pvlm_graph_end) only the _dev entry points may run: a staged copy would be captured as
// a memcpy node reading the recycled pinned arena (every replay would upload whatever bytes the arena holds by then), and a
// synchronisation invalidates the capture.  The three staging helpers refuse instead.
static pvlm_status refuse_in_capture(pvlm_ctx* ctx, const char* what) {
  PVLM_SET_ERR(ctx, "%s inside a graph capture: only the device-pointer (_dev) entry points may be captured", what);
  return PVLM_ERR_STATE;
}

pvlm_status pvlm_i_stream_sync(pvlm_ctx* ctx) {
  if (ctx->capturing) return refuse_in_capture(ctx, "host synchronisation");
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PVLM_OK;
}

pvlm_status pvlm_i_sync(pvlm_ctx* ctx) {
  if (ctx->capturing) return refuse_in_capture(ctx, "host synchronisation");
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  pvlm_stage& a = ctx->stage;
  if (e == hipSuccess) for (const pvlm_stage::Deferred& d : a.deferred) std::memcpy(d.dst, d.src, d.bytes);
  a.deferred.clear();
  a.cursor = 0;
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "stream synchronisation failed: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  return PVLM_OK;
}

static char* stage_take(pvlm_ctx* ctx, size_t bytes) {
  pvlm_stage& a = ctx->stage;
  if (!a.base) {
    if (hipHostMalloc((void**)&a.base, kStageBytes, hipHostMallocDefault) != hipSuccess) { a.base = nullptr; return nullptr; }
    a.size = kStageBytes;
  }
  bytes = (bytes + 63) & ~(size_t)63;
  if (bytes > a.size) return nullptr;
  if (a.cursor + bytes > a.size && pvlm_i_sync(ctx) != PVLM_OK) return nullptr;
  char* p = a.base + a.cursor;
  a.cursor += bytes;
  return p;
}

// Above kStageDirect a single copy goes the runtime's own way and is waited for: its pinning cost (milliseconds) is then
// small against the transfer, and its internal pipeline (10-12 GB/s) beats memcpy-then-DMA through the arena (8 GB/s:
// measured on pvlm_eval's 154 MB read-back, 94 vs 76 M eval/s).  Below it the arena wins by an order of magnitude
// (4 MB maps: 0.5 ms against 5 ms).
static const size_t kStageDirect = (size_t)64 << 20;

pvlm_status pvlm_i_h2d_q(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (ctx->capturing) return refuse_in_capture(ctx, "host-to-device copy");
  if (bytes > kStageDirect) {
    PVLM_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));     // pageable source: the caller may reuse it on return
    return PVLM_OK;
  }
  for (size_t done = 0; done < bytes;) {
    const size_t n = std::min(bytes - done, kStageBytes / 2);
    char* p = stage_take(ctx, n);
    if (!p) {   // no pinned memory to be had: the runtime's own pageable path, synchronous
      PVLM_HIP(ctx, hipMemcpyAsync((char*)dst + done, (const char*)src + done, bytes - done, hipMemcpyHostToDevice, ctx->stream));
      PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return PVLM_OK;
    }
    std::memcpy(p, (const char*)src + done, n);
    PVLM_HIP(ctx, hipMemcpyAsync((char*)dst + done, p, n, hipMemcpyHostToDevice, ctx->stream));
    done += n;
  }
  return PVLM_OK;
}

static pvlm_status d2h_queue(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (ctx->capturing) return refuse_in_capture(ctx, "device-to-host copy");
  if (bytes > kStageDirect) {
    PVLM_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PVLM_OK;
  }
  for (size_t done = 0; done < bytes;) {
    const size_t n = std::min(bytes - done, kStageBytes / 2);
    char* p = stage_take(ctx, n);
    if (!p) {
      PVLM_HIP(ctx, hipMemcpyAsync((char*)dst + done, (const char*)src + done, bytes - done, hipMemcpyDeviceToHost, ctx->stream));
      PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return PVLM_OK;
    }
    PVLM_HIP(ctx, hipMemcpyAsync(p, (const char*)src + done, n, hipMemcpyDeviceToHost, ctx->stream));
    ctx->stage.deferred.push_back({(char*)dst + done, p, n});
    done += n;
  }
  return PVLM_OK;
}

// A failed queueing leaves no deferred copy behind: the caller returns its error without reaching pvlm_i_sync, and a later
// synchronisation must not write into buffers that caller has released by then.
pvlm_status pvlm_i_d2h_q(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (ctx->capturing) return refuse_in_capture(ctx, "device-to-host copy");
  const pvlm_status st = d2h_queue(ctx, dst, src, bytes);
  if (st) {
    (void)hipStreamSynchronize(ctx->stream);
    ctx->stage.deferred.clear();
    ctx->stage.cursor = 0;
  }
  return st;
}

pvlm_status pvlm_i_h2d(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return PVLM_OK;
  const pvlm_status st = pvlm_i_h2d_q(ctx, dst, src, bytes);
  return st ? st : pvlm_i_sync(ctx);
}
pvlm_status pvlm_i_d2h(pvlm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return PVLM_OK;
  const pvlm_status st = pvlm_i_d2h_q(ctx, dst, src, bytes);
  return st ? st : pvlm_i_sync(ctx);
}

void pvlm_i_assoc_ws_free(pvlm_ctx* ctx) {
  pvlm_assoc_ws& w = ctx->assoc_ws;
  for (int s = 0; s < 2; ++s) {
    pvlm_i_free(ctx, w.d_nn[s]); pvlm_i_free(ctx, w.d_chain[s]); pvlm_i_free(ctx, w.d_desc[s]);
    if (w.h_count[s]) (void)hipHostFree(w.h_count[s]);
    if (w.h_desc[s]) (void)hipHostFree(w.h_desc[s]);
    if (w.ev[s]) (void)hipEventDestroy(w.ev[s]);
  }
  w = pvlm_assoc_ws();
}

pvlm_prof_scope::pvlm_prof_scope(pvlm_ctx* c, int w) : ctx(c), which(w) {
  if (!ctx->profiling || ctx->capturing) return;
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

void pvlm_i_trace(const char* label) {
  static const char* path = getenv("PVLM_TRACE");
  if (!path) return;
  static FILE* f = fopen(path, "a");
  static auto last = std::chrono::steady_clock::now();
  if (!f) return;
  const auto now = std::chrono::steady_clock::now();
  fprintf(f, "%-48s +%.3f ms\n", label, std::chrono::duration<double, std::milli>(now - last).count());
  fflush(f);
  last = now;
}

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
  ctx->pool.disabled = getenv("PVLM_NO_POOL") != nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  *out = ctx;
  return PVLM_OK;
}

pvlm_status pvlm_destroy(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->grow_stream) hipStreamSynchronize(ctx->grow_stream);      // a line growth begun and never finished: its kernels end before its blocks go
  pvlm_i_assoc_ws_free(ctx);
  if (ctx->h_up) (void)hipHostFree(ctx->h_up);
  if (ctx->host_spare) (void)hipHostFree(ctx->host_spare);
  if (ctx->h_grid) (void)hipHostFree(ctx->h_grid);
  for (int k = 0; k < ctx->ring_pool; ++k) (void)hipHostFree(ctx->h_ring[k]);
  pvlm_i_spd_plan_release(ctx);
  if (ctx->stage.base) (void)hipHostFree(ctx->stage.base);
  pvlm_i_free(ctx, ctx->d_aa); pvlm_i_free(ctx, ctx->d_t); pvlm_i_free(ctx, ctx->d_pose_tab); pvlm_i_free(ctx, ctx->d_ws); pvlm_i_free(ctx, ctx->d_neq_tmp);
  pvlm_i_pool_release(ctx, true);   // objects the caller leaked (scans, residual sets) die with their slabs
  hipEventDestroy(ctx->ev0); hipEventDestroy(ctx->ev1);
  for (int w = 0; w < 4; ++w) for (auto& pr : ctx->prof_pending[w]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  for (hipEvent_t e : ctx->prof_pool) hipEventDestroy(e);
  if (ctx->aux_stream) { hipStreamSynchronize(ctx->aux_stream); hipStreamDestroy(ctx->aux_stream); }
  if (ctx->grow_stream) { hipStreamSynchronize(ctx->grow_stream); hipStreamDestroy(ctx->grow_stream); }
  for (hipEvent_t e : ctx->grow_ev) if (e) hipEventDestroy(e);
  if (ctx->h_grow_in) (void)hipHostFree(ctx->h_grow_in);
  if (ctx->h_grow_out) (void)hipHostFree(ctx->h_grow_out);
  for (hipEvent_t e : ctx->aux_ev) if (e) hipEventDestroy(e);
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
  return pvlm_i_sync(ctx);   // PVLM_ERR_STATE inside a capture; completes the queued device-to-host copies (pvlm_neq_accumulate_async)
}

pvlm_status pvlm_reserve(pvlm_ctx* ctx, int64_t bytes) {
  if (!ctx || bytes < 0) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (ctx->pool.disabled || bytes == 0) return PVLM_OK;
  for (auto& f : ctx->pool.free_ranges) if ((int64_t)f.second.size >= bytes) return PVLM_OK;
  if (!pool_add_slab(ctx, round_up((size_t)bytes, kSlabGranule))) {
    PVLM_SET_ERR(ctx, "pvlm_reserve: hipMalloc of %.3f GB failed", bytes / 1e9);
    return PVLM_ERR_NOMEM;
  }
  return PVLM_OK;
}

__global__ void k_preload_ctx() {}
pvlm_status pvlm_preload(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  if (ctx->capturing) return PVLM_ERR_STATE;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  hipLaunchKernelGGL(k_preload_ctx, dim3(1), dim3(1), 0, ctx->stream);
  pvlm_i_preload_assoc(ctx->stream);
  pvlm_i_preload_ba(ctx->stream);
  pvlm_i_preload_eval(ctx->stream);
  pvlm_i_preload_linalg(ctx->stream);
  pvlm_i_preload_linegrow(ctx->stream);
  pvlm_i_preload_lines(ctx->stream);
  pvlm_i_preload_mvs(ctx->stream);
  pvlm_i_preload_ring(ctx->stream);
  pvlm_i_preload_undistort(ctx->stream);
  PVLM_HIP(ctx, hipGetLastError());
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PVLM_OK;
}

pvlm_status pvlm_reserve_staging(pvlm_ctx* ctx, int64_t bytes) {
  if (!ctx || bytes < 0) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  // the arena of the small copies as well (32 MB of pinned memory, allocated at its first use otherwise: 8 ms inside whichever call comes first)
  if (bytes > 0 && !ctx->stage.base && !ctx->capturing) {
    if (hipHostMalloc((void**)&ctx->stage.base, kStageBytes, hipHostMallocDefault) == hipSuccess) { ctx->stage.size = kStageBytes; std::memset(ctx->stage.base, 0, kStageBytes); }
    else { ctx->stage.base = nullptr; (void)hipGetLastError(); }          // stage_take tries again (and has its own fall-back)
  }
  if ((size_t)bytes <= ctx->up_bytes) return PVLM_OK;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->h_up) (void)hipHostFree(ctx->h_up);
  ctx->h_up = nullptr; ctx->up_bytes = 0;
  if (hipHostMalloc(&ctx->h_up, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_reserve_staging: %lld bytes of pinned memory unavailable", (long long)bytes); return PVLM_ERR_NOMEM; }
  ctx->up_bytes = (size_t)bytes;
  std::memset(ctx->h_up, 0, (size_t)bytes);         // first touch here, not inside the first upload
  return PVLM_OK;
}

pvlm_status pvlm_trim(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pvlm_i_assoc_ws_free(ctx);
  pvlm_i_free(ctx, ctx->d_neq_tmp); ctx->d_neq_tmp = nullptr; ctx->neq_tmp_count = 0;
  pvlm_i_pool_release(ctx, false);
  return PVLM_OK;
}

pvlm_status pvlm_mem_info(const pvlm_ctx* ctx, int64_t* reserved, int64_t* in_use, int64_t* peak, int64_t* device_allocs) {
  if (!ctx) return PVLM_ERR_ARG;
  if (reserved) *reserved = (int64_t)ctx->pool.reserved;
  if (in_use) *in_use = (int64_t)ctx->pool.in_use;
  if (peak) *peak = (int64_t)ctx->pool.peak;
  if (device_allocs) *device_allocs = (int64_t)ctx->pool.device_allocs;
  return PVLM_OK;
}

// ---- HIP graph of a step ------------------------------------------------------------------------
struct pvlm_graph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };

pvlm_status pvlm_graph_begin(pvlm_ctx* ctx) {
  if (!ctx) return PVLM_ERR_ARG;
  if (ctx->capturing) { PVLM_SET_ERR(ctx, "pvlm_graph_begin: a capture is already open"); return PVLM_ERR_STATE; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (ctx->stream == nullptr) { PVLM_SET_ERR(ctx, "pvlm_graph_begin: the legacy default stream cannot be captured; use pvlm_use_own_stream or a created stream"); return PVLM_ERR_STATE; }
  hipStreamCaptureMode mode = hipStreamCaptureModeThreadLocal;
  if (const char* e = getenv("PVLM_CAPTURE_MODE")) mode = e[0] == 'g' ? hipStreamCaptureModeGlobal : e[0] == 'r' ? hipStreamCaptureModeRelaxed : mode;
  PVLM_HIP(ctx, hipStreamBeginCapture(ctx->stream, mode));
  ctx->capturing = true;
  return PVLM_OK;
}

pvlm_status pvlm_graph_end(pvlm_ctx* ctx, pvlm_graph** out) {
  if (!ctx || !out) return PVLM_ERR_ARG;
  *out = nullptr;
  if (!ctx->capturing) { PVLM_SET_ERR(ctx, "pvlm_graph_end without pvlm_graph_begin"); return PVLM_ERR_STATE; }
  ctx->capturing = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  if (e != hipSuccess || !g) { (void)hipGetLastError(); PVLM_SET_ERR(ctx, "hipStreamEndCapture: %s (a call inside the capture was not capturable)", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { hipGraphDestroy(g); PVLM_SET_ERR(ctx, "hipGraphInstantiate: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  pvlm_graph* pg = new (std::nothrow) pvlm_graph();
  if (!pg) { hipGraphExecDestroy(x); hipGraphDestroy(g); return PVLM_ERR_NOMEM; }
  pg->graph = g; pg->exec = x;
  *out = pg;
  return PVLM_OK;
}

pvlm_status pvlm_graph_launch(pvlm_ctx* ctx, pvlm_graph* g) {
  if (!ctx || !g || !g->exec) return PVLM_ERR_ARG;
  if (ctx->capturing) { PVLM_SET_ERR(ctx, "pvlm_graph_launch inside a capture"); return PVLM_ERR_STATE; }
  PVLM_HIP(ctx, hipGraphLaunch(g->exec, ctx->stream));
  return PVLM_OK;
}

pvlm_status pvlm_graph_destroy(pvlm_ctx* ctx, pvlm_graph* g) {
  if (!ctx) return PVLM_ERR_ARG;
  if (!g) return PVLM_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (g->exec) hipGraphExecDestroy(g->exec);
  if (g->graph) hipGraphDestroy(g->graph);
  delete g;
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
  for (int w = 0; w < 4; ++w) {
    for (auto& pr : ctx->prof_pending[w]) { ctx->prof_pool.push_back(pr.first); ctx->prof_pool.push_back(pr.second); }
    ctx->prof_pending[w].clear();
    ctx->prof_ms[w] = 0; ctx->prof_n[w] = 0;
  }
  ctx->profiling = on != 0;
  return PVLM_OK;
}

pvlm_status pvlm_profile_read(pvlm_ctx* ctx, int which, double* total_ms, int64_t* launches) {
  if (!ctx || which < 0 || which > 3) return PVLM_ERR_ARG;
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
pvlm_status pvlm_resset_set_pose_ids(pvlm_ctx* ctx, pvlm_resset* rs, const int* pair_ref, const int* pair_nei) {
  if (!ctx || !rs || (rs->n_pairs > 0 && (!pair_ref || !pair_nei))) return PVLM_ERR_ARG;
  for (int p = 0; p < rs->n_pairs; ++p)
    if (pair_ref[p] < 0 || pair_nei[p] < 0) { PVLM_SET_ERR(ctx, "pvlm_resset_set_pose_ids: negative pose id in segment %d", p); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (rs->n_pairs == 0) return PVLM_OK;
  rs->h_ref.assign(pair_ref, pair_ref + rs->n_pairs);
  rs->h_nei.assign(pair_nei, pair_nei + rs->n_pairs);
  pvlm_status st = pvlm_i_h2d_q(ctx, rs->d_ref, rs->h_ref.data(), (size_t)rs->n_pairs * sizeof(int));
  if (!st) st = pvlm_i_h2d_q(ctx, rs->d_nei, rs->h_nei.data(), (size_t)rs->n_pairs * sizeof(int));
  rs->pair_tab_epoch = ~0ull;                  // the pair table is rebuilt from the new ids at the next evaluation (ensure_pair_table,
                                               // csrc/pvlm_eval.hip: every path that reads the pose table checks the ids against n_poses there)
  rs->serial = ++ctx->resset_serial;           // ... and every pvlm_neq bound to the set binds again
  return st;
}

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
  // stream-ordered pool: kernels still reading the set are ahead of any later user of its memory on the same stream
  for (double* b : rs->col_blocks) pvlm_i_free(ctx, b);
  for (int32_t* b : rs->d_qidx) pvlm_i_free(ctx, b);
  for (int32_t* b : rs->d_nn) pvlm_i_free(ctx, b);
  pvlm_i_free(ctx, rs->d_pair_cols); pvlm_i_free(ctx, rs->d_pair_stride); pvlm_i_free(ctx, rs->d_out_start); pvlm_i_free(ctx, rs->d_ref);
  pvlm_i_free(ctx, rs->d_nei); pvlm_i_free(ctx, rs->d_blk_pair); pvlm_i_free(ctx, rs->d_blk_chunk); pvlm_i_free(ctx, rs->d_pair_blk_start);
  pvlm_i_free(ctx, rs->d_pair_tab); pvlm_i_free(ctx, rs->d_partials); pvlm_i_free(ctx, rs->d_pair_blocks); pvlm_i_free(ctx, rs->d_stage);
  delete rs;
  return PVLM_OK;
}

// Needs: kind, n_pairs, h_seg_start (row inside the block), h_pair_block, h_out_start, h_ref, h_nei filled; col_blocks /
// block_rows allocated & filled (or being filled on the stream).  Uploads the segment table, builds the block work list
// and the scratch.  One host synchronisation at the end (the staging vectors go out of scope).
pvlm_status pvlm_i_resset_finalize(pvlm_ctx* ctx, pvlm_resset* rs) {
  const int P = rs->n_pairs;
  rs->serial = ++ctx->resset_serial;
  // the rows the evaluation kernels will read: the accepted rows of every segment, padded — not n_dev, which for a set made by the association is the
  // CAPACITY of its column blocks (every query has a slot)
  int64_t total = 0;
  for (int p = 0; p < P; ++p) total += pvlm_i_seg_rows(rs->h_out_start[(size_t)p + 1] - rs->h_out_start[(size_t)p]);
  int64_t chunk = ((total / 4096 + 511) / 512) * 512;
  chunk = std::max<int64_t>(512, std::min<int64_t>(16384, chunk));
  // Sets of many short segments (Room / Floor odometry: thousands of pairs of a few hundred blocks) are evaluated with a wave
  // per (pair, chunk) — k_eval_fused_wave — in chunks of 2048 rows (raw-target workload, 3 946 rows per pair: 81.8 / 85.6 / 87.3 G eval/s
  // with 1024 / 2048 / 4096, round 3); PVLM_WAVE_UNITS=0 / 1 forces the choice, PVLM_WAVE_CHUNK the chunk (A/B runs).
  rs->wave_units = P > 0 && rs->n / P < 4096;
  if (const char* env = getenv("PVLM_WAVE_UNITS")) rs->wave_units = atoi(env) != 0;
  if (rs->wave_units) chunk = 2048;
  if (const char* env = getenv("PVLM_WAVE_CHUNK")) if (rs->wave_units && atoi(env) >= 128) chunk = atoi(env) / 128 * 128;
  rs->chunk_rows = (int)chunk;
  std::vector<int> blk_pair, blk_chunk, pair_blk_start(P + 1, 0);
  std::vector<const double*> pair_cols(std::max(P, 1), nullptr);
  std::vector<int64_t> pair_stride(std::max(P, 1), 0);
  for (int p = 0; p < P; ++p) {
    pair_blk_start[p] = (int)blk_pair.size();
    const int64_t len = rs->h_out_start[p + 1] - rs->h_out_start[p];
    const int nch = (int)((len + chunk - 1) / chunk);
    for (int c = 0; c < nch; ++c) { blk_pair.push_back(p); blk_chunk.push_back(c); }
    const int b = rs->h_pair_block[p];
    pair_cols[p] = rs->col_blocks[b] + rs->h_seg_start[p];
    pair_stride[p] = rs->block_rows[b];
  }
  pair_blk_start[P] = (int)blk_pair.size();
  rs->n_blocks = (int)blk_pair.size();
  rs->h_pair_blk_start = pair_blk_start;
  pvlm_status st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_cols, pair_cols.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_stride, pair_stride.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_out_start, (size_t)P + 1))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_ref, (size_t)P))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_nei, (size_t)P))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_blk_pair, blk_pair.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_blk_chunk, blk_chunk.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_blk_start, pair_blk_start.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_tab, (size_t)std::max(P, 1) * PVLM_PAIR_TAB))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_partials, (size_t)std::max(rs->n_blocks, 1) * PVLM_PARTIAL))) return st;
  if ((st = pvlm_i_alloc(ctx, &rs->d_pair_blocks, (size_t)std::max(P, 1) * PVLM_PAIR_BLOCK))) return st;
  auto cp = [&](void* d, const void* h, size_t bytes) -> pvlm_status {
    return bytes ? pvlm_i_h2d_q(ctx, d, h, bytes) : PVLM_OK;
  };
  if ((st = cp(rs->d_pair_cols, pair_cols.data(), (size_t)P * sizeof(double*)))) return st;
  if ((st = cp(rs->d_pair_stride, pair_stride.data(), (size_t)P * sizeof(int64_t)))) return st;
  if ((st = cp(rs->d_out_start, rs->h_out_start.data(), rs->h_out_start.size() * sizeof(int64_t)))) return st;
  if ((st = cp(rs->d_ref, rs->h_ref.data(), (size_t)P * sizeof(int)))) return st;
  if ((st = cp(rs->d_nei, rs->h_nei.data(), (size_t)P * sizeof(int)))) return st;
  if ((st = cp(rs->d_blk_pair, blk_pair.data(), blk_pair.size() * sizeof(int)))) return st;
  if ((st = cp(rs->d_blk_chunk, blk_chunk.data(), blk_chunk.size() * sizeof(int)))) return st;
  if ((st = cp(rs->d_pair_blk_start, pair_blk_start.data(), pair_blk_start.size() * sizeof(int)))) return st;
  if ((st = pvlm_i_sync(ctx))) return st;            // one wait for everything the caller and this function queued
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
  for (int p = 0; p < n_pairs; ++p) { rs->h_seg_start[p] = o; o += pvlm_i_seg_rows(rs->h_out_start[p + 1] - rs->h_out_start[p]); }
  rs->h_seg_start[n_pairs] = o;
  rs->n_dev = o;
  rs->h_pair_block.assign(n_pairs, 0);   // an uploaded set is one column block

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
  double* d_block = nullptr;
  TRY(pvlm_i_alloc(ctx, &d_block, cols.size()));
  rs->col_blocks.push_back(d_block);
  rs->block_rows.push_back(std::max<int64_t>(rs->n_dev, 1));
  TRY(pvlm_i_h2d(ctx, d_block, cols.data(), cols.size() * sizeof(double)));
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
    const int stride = pvlm_i_stride(rs->kind);
    std::vector<double> cols;
    for (size_t b = 0; b < rs->col_blocks.size(); ++b) {
      const int64_t br = rs->block_rows[b];
      cols.resize((size_t)rs->ncols * br);
      pvlm_status st = pvlm_i_d2h(ctx, cols.data(), rs->col_blocks[b], cols.size() * sizeof(double));
      if (st) return st;
      for (int p = 0; p < rs->n_pairs; ++p) {
        if (rs->h_pair_block[p] != (int)b) continue;
        for (int64_t i = rs->h_out_start[p]; i < rs->h_out_start[p + 1]; ++i) {
          const int64_t d = rs->h_seg_start[p] + (i - rs->h_out_start[p]);
          double* r = rows + (size_t)i * stride;
          for (int k = 0; k < rs->ncols; ++k) r[k] = cols[(size_t)k * br + d];
          if (rs->kind == PVLM_POINT2LINE_METER || rs->kind == PVLM_POINT2LINE_ANGLE) {
            // device keeps (A, unit direction); hand back B' = A - direction (same line)
            for (int k = 0; k < 3; ++k) r[6 + k] = r[3 + k] - r[6 + k];
          }
        }
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
  pvlm_i_free(ctx, q->d_diag_off); pvlm_i_free(ctx, q->d_diag_items); pvlm_i_free(ctx, q->d_off_off); pvlm_i_free(ctx, q->d_off_items);
  // a queued copy out of d_packed (pvlm_neq_accumulate_async) may still be in flight: the block goes back to the pool, whose reuse
  // is ordered by the same stream, so the copy reads it before any later writer touches it
  pvlm_i_free(ctx, q->d_packed);
  delete q;
  return PVLM_OK;
}

int64_t pvlm_neq_size(const pvlm_neq* q) { return q ? (int64_t)q->n_poses * 36 + (int64_t)q->n_upairs * 36 + (int64_t)q->n_poses * 6 + 1 : -1; }

}  // extern "C"
