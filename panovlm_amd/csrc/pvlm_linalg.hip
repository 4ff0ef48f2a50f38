// Dense symmetric positive definite solve on the GPU for the LM driver of the host mirror (the reduced pose system of a
// Room/Floor-sized joint optimisation has 5e3..1e4 unknowns: a host Cholesky of it costs seconds per LM iteration and
// dwarfs everything the hot path does).  This is NOT part of the hot path and not a hand-written kernel: it is the one
// place where a vendor library is the right tool — rocSOLVER's potrf/potrs, dlopen-ed at first use like RCCL so that
// libpvlm.so has no link-time dependency on it and hosts that never solve on the GPU never load it.
// Upstream this step belongs to Ceres (SPARSE_SCHUR + SuiteSparse, util/Optimization.cpp:608-666).
#include <dlfcn.h>

#include "pvlm_internal.h"

namespace {
typedef void* rb_handle;
typedef int (*fn_create)(rb_handle*);
typedef int (*fn_destroy)(rb_handle);
typedef int (*fn_set_stream)(rb_handle, hipStream_t);
typedef int (*fn_potrf)(rb_handle, int, int, double*, int, int*);
typedef int (*fn_potrs)(rb_handle, int, int, int, double*, int, double*, int);
const int kFillLower = 122;   // rocblas_fill_lower (rocblas-types.h)

struct Solver {
  void *hb = nullptr, *hs = nullptr;
  fn_create create = nullptr; fn_destroy destroy = nullptr; fn_set_stream set_stream = nullptr; fn_potrf potrf = nullptr; fn_potrs potrs = nullptr;
  rb_handle handle = nullptr;
  bool Load(pvlm_ctx* ctx) {
    if (handle) return true;
    const char* blas[] = {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"};
    const char* solv[] = {"librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so"};
    for (const char* n : blas) { hb = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (hb) break; }
    for (const char* n : solv) { hs = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (hs) break; }
    if (!hb || !hs) { PVLM_SET_ERR(ctx, "rocBLAS / rocSOLVER not found: %s", dlerror()); return false; }
    create = (fn_create)dlsym(hb, "rocblas_create_handle"); destroy = (fn_destroy)dlsym(hb, "rocblas_destroy_handle");
    set_stream = (fn_set_stream)dlsym(hb, "rocblas_set_stream");
    potrf = (fn_potrf)dlsym(hs, "rocsolver_dpotrf"); potrs = (fn_potrs)dlsym(hs, "rocsolver_dpotrs");
    if (!create || !destroy || !set_stream || !potrf || !potrs) { PVLM_SET_ERR(ctx, "rocBLAS / rocSOLVER symbols missing"); return false; }
    if (create(&handle) != 0 || !handle) { PVLM_SET_ERR(ctx, "rocblas_create_handle failed"); handle = nullptr; return false; }
    return true;
  }
};
Solver g_solver;
}  // namespace

// dense M (n x n, symmetric) += scatter of 6x6 blocks: entry (r, c) of block b goes to (row_idx[6b + r], col_idx[6b + c])
// scaled by scale[i] * scale[j]; blocks flagged `mirror` (two different poses) are also added to the other triangle.
__global__ void k_scatter_blocks(int n, int n_blocks, const int* __restrict__ row_idx, const int* __restrict__ col_idx, const int* __restrict__ mirror,
                                 const double* __restrict__ blocks, const double* __restrict__ scale, double* __restrict__ M) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)n_blocks * 36) return;
  const int b = (int)(g / 36), e = (int)(g - (long long)b * 36), r = e / 6, c = e - r * 6;
  const int i = row_idx[6 * b + r], j = col_idx[6 * b + c];
  if (i < 0 || j < 0) return;
  const double v = blocks[g] * scale[i] * scale[j];
  unsafeAtomicAdd(&M[(size_t)i * n + j], v);
  if (mirror[b]) unsafeAtomicAdd(&M[(size_t)j * n + i], v);   // off-diagonal pose pair
}
__global__ void k_add_diag(int n, const double* __restrict__ d, double* __restrict__ M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) M[(size_t)i * n + i] += d[i];
}

extern "C" {

// Block-sparse form for the LM driver: assembles M = D (sum of blocks) D + diag(diag_add) on the device (D = diag(scale)),
// factorises it and solves M x = rhs in place.  Blocks are 6x6 row-major; row_idx / col_idx give the scalar index of each
// of their 6 rows / columns (-1 = constant parameter, dropped).  mirror[b] != 0 (a block between two different poses)
// also adds the transposed block to the other triangle; a block of one pose with itself is given in full, mirror = 0.
pvlm_status pvlm_spd_solve_blocks(pvlm_ctx* ctx, int n, int n_blocks, const int* row_idx, const int* col_idx, const int* mirror, const double* blocks,
                                  const double* scale, const double* diag_add, double* rhs, int* info_out) {
  if (!ctx || n < 0 || n_blocks < 0 || !info_out || (n > 0 && (!scale || !diag_add || !rhs)) || (n_blocks > 0 && (!row_idx || !col_idx || !mirror || !blocks)))
    return PVLM_ERR_ARG;
  *info_out = 0;
  if (n == 0) return PVLM_OK;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (!g_solver.Load(ctx)) return PVLM_ERR_STATE;
  double *d_M = nullptr, *d_blocks = nullptr, *d_scale = nullptr, *d_diag = nullptr, *d_rhs = nullptr; int *d_row = nullptr, *d_col = nullptr, *d_mir = nullptr, *d_info = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_M, (size_t)n * n);
  if (!st) st = pvlm_i_alloc(ctx, &d_blocks, (size_t)n_blocks * 36);
  if (!st) st = pvlm_i_alloc(ctx, &d_row, (size_t)n_blocks * 6);
  if (!st) st = pvlm_i_alloc(ctx, &d_col, (size_t)n_blocks * 6);
  if (!st) st = pvlm_i_alloc(ctx, &d_mir, (size_t)n_blocks);
  if (!st) st = pvlm_i_alloc(ctx, &d_scale, (size_t)n);
  if (!st) st = pvlm_i_alloc(ctx, &d_diag, (size_t)n);
  if (!st) st = pvlm_i_alloc(ctx, &d_rhs, (size_t)n);
  if (!st) st = pvlm_i_alloc(ctx, &d_info, (size_t)1);
  if (!st) {
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemsetAsync(d_M, 0, (size_t)n * n * sizeof(double), s);
    if (e == hipSuccess && n_blocks) e = hipMemcpyAsync(d_blocks, blocks, (size_t)n_blocks * 36 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && n_blocks) e = hipMemcpyAsync(d_row, row_idx, (size_t)n_blocks * 6 * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && n_blocks) e = hipMemcpyAsync(d_col, col_idx, (size_t)n_blocks * 6 * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && n_blocks) e = hipMemcpyAsync(d_mir, mirror, (size_t)n_blocks * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_scale, scale, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_diag, diag_add, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rhs, rhs, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s);
    int rc = 0, info = 0;
    if (e == hipSuccess) {
      if (n_blocks) hipLaunchKernelGGL(k_scatter_blocks, dim3((unsigned)(((long long)n_blocks * 36 + 255) / 256)), dim3(256), 0, s, n, n_blocks, d_row, d_col, d_mir, d_blocks, d_scale, d_M);
      hipLaunchKernelGGL(k_add_diag, dim3((n + 255) / 256), dim3(256), 0, s, n, d_diag, d_M);
      e = hipGetLastError();
    }
    if (e == hipSuccess) {
      rc = g_solver.set_stream(g_solver.handle, s);
      if (rc == 0) rc = g_solver.potrf(g_solver.handle, kFillLower, n, d_M, n, d_info);
      if (rc == 0) e = hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s);
      if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(s);
      if (rc == 0 && e == hipSuccess && info == 0) {
        rc = g_solver.potrs(g_solver.handle, kFillLower, n, 1, d_M, n, d_rhs, n);
        if (rc == 0) e = hipMemcpyAsync(rhs, d_rhs, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s);
        if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(s);
      }
    }
    *info_out = info;
    if (rc != 0) { PVLM_SET_ERR(ctx, "rocSOLVER potrf/potrs failed with rocblas_status %d", rc); st = PVLM_ERR_HIP; }
    else if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_spd_solve_blocks: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  hipStreamSynchronize(ctx->stream);
  hipFree(d_M); hipFree(d_blocks); hipFree(d_row); hipFree(d_col); hipFree(d_mir); hipFree(d_scale); hipFree(d_diag); hipFree(d_rhs); hipFree(d_info);
  return st;
}

// Solves A X = B for symmetric positive definite A (n x n, dense, both triangles or at least the lower one of the
// column-major view filled — for a symmetric matrix row- and column-major coincide) and B = n x nrhs (column-major).
// A and B are host buffers; B is overwritten by the solution.  *info_out = 0 on success, k > 0 when the leading minor of
// order k is not positive definite (the LM driver then treats the step as failed, like a failed host factorisation).
pvlm_status pvlm_spd_solve(pvlm_ctx* ctx, int n, int nrhs, const double* A, double* B, int* info_out) {
  if (!ctx || n < 0 || nrhs < 0 || !info_out || (n > 0 && (!A || (nrhs > 0 && !B)))) return PVLM_ERR_ARG;
  *info_out = 0;
  if (n == 0 || nrhs == 0) return PVLM_OK;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  if (!g_solver.Load(ctx)) return PVLM_ERR_STATE;
  double *d_A = nullptr, *d_B = nullptr; int* d_info = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_A, (size_t)n * n);
  if (!st) st = pvlm_i_alloc(ctx, &d_B, (size_t)n * nrhs);
  if (!st) st = pvlm_i_alloc(ctx, &d_info, (size_t)1);
  if (!st) {
    hipError_t e = hipMemcpyAsync(d_A, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_B, B, (size_t)n * nrhs * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    int rc = 0, info = 0;
    if (e == hipSuccess) {
      rc = g_solver.set_stream(g_solver.handle, ctx->stream);
      if (rc == 0) rc = g_solver.potrf(g_solver.handle, kFillLower, n, d_A, n, d_info);
      if (rc == 0) e = hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
      if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (rc == 0 && e == hipSuccess && info == 0) {
        rc = g_solver.potrs(g_solver.handle, kFillLower, n, nrhs, d_A, n, d_B, n);
        if (rc == 0) e = hipMemcpyAsync(B, d_B, (size_t)n * nrhs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      }
    }
    *info_out = info;
    if (rc != 0) { PVLM_SET_ERR(ctx, "rocSOLVER potrf/potrs failed with rocblas_status %d", rc); st = PVLM_ERR_HIP; }
    else if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_spd_solve: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  hipStreamSynchronize(ctx->stream);
  hipFree(d_A); hipFree(d_B); hipFree(d_info);
  return st;
}

}  // extern "C"
