// K5 / K6 — residual + Jacobian evaluation kernels (materialise and fused normal-equation modes)
// and the small pose / pair table + epilogue kernels around them.  gfx950 (wave64) only.
//
// Data layout in HBM: a residual set is SoA, `ncols` fp64 columns of n_dev rows; every pair
// segment starts on an even row so each lane streams two consecutive rows with one 16-byte load
// per column (1 KiB per wave instruction).  The per-pair constants (R_rn, t_rn, t_rw, J_l blocks)
// are read through wave-uniform (scalar) loads.  Algorithmic HBM traffic per evaluation in fused
// mode = 8*ncols bytes (56 B point-to-plane); materialise mode adds 8 + 96 B of stores.
#include <algorithm>
#include <cmath>
#include <limits>

#include "pvlm_functors.h"
#include "pvlm_internal.h"

using namespace pvlm_dev;

// 16-byte load of two consecutive rows of a column.  The column base comes out of a pointer table in memory, so the compiler
// cannot prove the address space and emits FLAT loads (aperture check, lgkmcnt + vmcnt); PVLM_GLOBAL_LOADS = 1 states that the
// columns live in global memory.  PVLM_NT_LOADS = 1: non-temporal (the 45 GB of a launch are read exactly once; keeps them
// out of the way of the pair table and the partials in L2 / MALL).  Measured on the benched kernel (tools/ab_eval_loads.sh,
// profiles/r2_ab_eval_loads.txt): flat 7.61 ms, global 7.60 ms, flat + nt 7.52 ms, global + nt 7.50 ms per launch.
#ifndef PVLM_NT_LOADS
#define PVLM_NT_LOADS 1
#endif
#ifndef PVLM_GLOBAL_LOADS
#define PVLM_GLOBAL_LOADS 1
#endif
typedef double pvlm_dbl2 __attribute__((ext_vector_type(2)));
#if PVLM_GLOBAL_LOADS
typedef const __attribute__((address_space(1))) pvlm_dbl2* pvlm_col_ptr;
#else
typedef const pvlm_dbl2* pvlm_col_ptr;
#endif
#ifndef PVLM_NT_LOADS_MATERIALISE   // the same hint in the kernels that also WRITE rows (k_eval_materialise, k_eval_wrench)
#define PVLM_NT_LOADS_MATERIALISE 0
#endif
// base + a 32-bit BYTE offset: the form the backend matches to `global_load ... v_off, s[base]` when the base is wave-uniform
__device__ __forceinline__ const double* at_byte(const double* base, unsigned bytes) {
  return reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + bytes);
}
template <bool NT>
__device__ __forceinline__ double2 stream_load2(const double* p) {
  pvlm_col_ptr q = (pvlm_col_ptr)(p);
  const pvlm_dbl2 v = NT ? __builtin_nontemporal_load(q) : *q;
  return make_double2(v.x, v.y);
}
// ... and the stores of the materialising kernels (r, the 1 x 12 rows, the wrench rows).  Non-temporal STORES were measured and lose:
// k_eval_materialise 4.80 TB/s against 5.04 TB/s with ordinary stores (profiles/r2_ab_eval_loads.txt) — off.
#ifndef PVLM_NT_STORES
#define PVLM_NT_STORES 0
#endif
__device__ __forceinline__ void stream_store2(double2* p, double2 v) {
#if PVLM_NT_STORES
  pvlm_dbl2 w; w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<pvlm_dbl2*>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ void stream_store1(double* p, double v) {
#if PVLM_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}


#ifndef PVLM_PREFETCH
#define PVLM_PREFETCH -1  // -1 = per-functor default, 0 / 1 force
#endif
#ifndef PVLM_FUSED_WAVES
#define PVLM_FUSED_WAVES 2  // waves per SIMD the fused kernel must fit (no spilling at 3-4: measured 2x slower)
#endif
#ifndef PVLM_FUSED_WAVES_7COL      // ... the block-per-chunk kernel of the 7-column (point-to-plane) functors: 128 VGPRs without scratch since the
#define PVLM_FUSED_WAVES_7COL 4    // saddr loads + scalar polynomial coefficients of round 3; 118.2 vs 117.2 G eval/s at 3 waves (profiles/r3_eval_diet_ab.txt)
#endif
#ifndef PVLM_FUSED_WAVES_WAVEFORM  // ... and the wave-per-(pair, chunk) kernel: 168 VGPRs fit three waves; left to itself the allocator takes 178 (two waves, -10 %)
#define PVLM_FUSED_WAVES_WAVEFORM 3
#endif

// ---------------------------------------------------------------------------------------------
// pose table: R_lw = exp([aa]x) with the same small-angle branch as ceres::AngleAxisToRotationMatrix
// (theta^2 <= DBL_EPSILON -> first order), and the SO(3) left Jacobian J_l(aa).
// ---------------------------------------------------------------------------------------------
__global__ void k_pose_table(int n, const double* __restrict__ aa, const double* __restrict__ t, double* __restrict__ tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = aa[3 * i], y = aa[3 * i + 1], z = aa[3 * i + 2];
  const double th2 = x * x + y * y + z * z;
  double* o = tab + (size_t)i * PVLM_POSE_TAB;
  double R[9];
  if (th2 > 2.220446049250313e-16) {
    const double th = sqrt(th2);
    const double wx = x / th, wy = y / th, wz = z / th;
    const double c = cos(th), s = sin(th), k = 1.0 - c;
    R[0] = c + wx * wx * k;      R[1] = wx * wy * k - wz * s; R[2] = wy * s + wx * wz * k;
    R[3] = wz * s + wx * wy * k; R[4] = c + wy * wy * k;      R[5] = -wx * s + wy * wz * k;
    R[6] = -wy * s + wx * wz * k; R[7] = wx * s + wy * wz * k; R[8] = c + wz * wz * k;
  } else {
    R[0] = 1; R[1] = -z; R[2] = y; R[3] = z; R[4] = 1; R[5] = -x; R[6] = -y; R[7] = x; R[8] = 1;
  }
  double A, B;  // J_l = I + A [w]x + B [w]x^2
  if (th2 > 1e-6) {
    const double th = sqrt(th2);
    A = (1.0 - cos(th)) / th2;
    B = (th - sin(th)) / (th2 * th);
  } else {
    A = 0.5 - th2 * (1.0 / 24.0) + th2 * th2 * (1.0 / 720.0);
    B = (1.0 / 6.0) - th2 * (1.0 / 120.0) + th2 * th2 * (1.0 / 5040.0);
  }
  // [w]x^2 = w w^T - th2 I
  const double J[9] = {1.0 + B * (x * x - th2), -A * z + B * x * y,       A * y + B * x * z,
                       A * z + B * x * y,       1.0 + B * (y * y - th2), -A * x + B * y * z,
                       -A * y + B * x * z,      A * x + B * y * z,        1.0 + B * (z * z - th2)};
#pragma unroll
  for (int k = 0; k < 9; ++k) { o[k] = R[k]; o[9 + k] = J[k]; }
  o[18] = t[3 * i]; o[19] = t[3 * i + 1]; o[20] = t[3 * i + 2];
}

// pair table: R_rn = R_r R_n^T, t_rn = t_r - R_rn t_n, t_rw = t_r, Jl_r, M_n = -R_rn Jl_n
__global__ void k_pair_table(int P, const int* __restrict__ ref, const int* __restrict__ nei, int n_poses,
                             const double* __restrict__ pose, double* __restrict__ tab) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int ir = ref[p], in = nei[p];
  double* o = tab + (size_t)p * PVLM_PAIR_TAB;
  if (ir >= n_poses || in >= n_poses) {  // out-of-table ids poison the segment instead of faulting
    for (int k = 0; k < PVLM_PAIR_TAB; ++k) o[k] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  const double* a = pose + (size_t)ir * PVLM_POSE_TAB;
  const double* b = pose + (size_t)in * PVLM_POSE_TAB;
  double R[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = a[i * 3] * b[j * 3] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = R[k];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[9 + i] = a[18 + i] - (R[i * 3] * b[18] + R[i * 3 + 1] * b[19] + R[i * 3 + 2] * b[20]);
    o[12 + i] = a[18 + i];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) o[15 + k] = a[9 + k];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o[24 + i * 3 + j] = -(R[i * 3] * b[9 + j] + R[i * 3 + 1] * b[9 + 3 + j] + R[i * 3 + 2] * b[9 + 6 + j]);
}

// ---------------------------------------------------------------------------------------------
// materialise: r[n], J[n x 12] in compact (host) order.
// ---------------------------------------------------------------------------------------------
template <int KIND, bool NORM, int NCOLS>
__global__ __launch_bounds__(256) void k_eval_materialise(const double* const* __restrict__ pair_cols,
                                                          const int64_t* __restrict__ pair_stride,
                                                          const int64_t* __restrict__ out_start,
                                                          const int* __restrict__ blk_pair, const int* __restrict__ blk_chunk,
                                                          int chunk_rows, const double* __restrict__ pair_tab, double weight,
                                                          double* __restrict__ r_out, double* __restrict__ J_out, int blk0, int64_t row0) {
  // blk0 / row0: the launch covers work-list blocks [blk0, blk0 + gridDim.x) whose first compact row is row0; outputs are
  // indexed relative to row0 (a bounded staging buffer filled slice by slice, pvlm_eval_host_async)
  const int bi = blockIdx.x + blk0;
  const int p = blk_pair[bi];
  const double* __restrict__ cols = pair_cols[p];   // first row of the pair's segment, column 0
  const int64_t n_dev = pair_stride[p];             // column stride of the pair's block
  const int64_t o0 = out_start[p] - row0;
  const int64_t len = out_start[p + 1] - out_start[p];
  const int64_t lo = (int64_t)blk_chunk[bi] * chunk_rows;
  const int64_t hi = min(len, lo + (int64_t)chunk_rows);
  double T[PVLM_PAIR_TAB];
#pragma unroll
  for (int k = 0; k < PVLM_PAIR_TAB; ++k) T[k] = pair_tab[(size_t)p * PVLM_PAIR_TAB + k];
  // Jacobian rows are staged through LDS so that every store instruction of a wave writes one
  // contiguous KiB (a lane's own 2 x 96 B would scatter 16-byte pieces over 64 cache lines).
  // Tile of one wave-iteration: 128 rows x 12 doubles = 12 KiB, private to the wave.
  __shared__ __attribute__((aligned(16))) double stage[4][128 * 12];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double* Jl = T + 15; const double* Mn = T + 24; const double* R = T;
  const int64_t n_it = (hi - lo + 511) / 512;
  for (int64_t it = 0; it < n_it; ++it) {
    const int64_t j = lo + it * 512 + 2 * (int64_t)threadIdx.x;
    if (j < hi) {
      double2 v[NCOLS];
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) v[c] = stream_load2<PVLM_NT_LOADS_MATERIALISE != 0>(cols + (size_t)c * n_dev + j);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (j + h >= hi) break;
        double rec[NCOLS];
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) rec[c] = h ? v[c].y : v[c].x;
        Wrench w;
        eval_wrench<KIND, NORM>(rec, T, weight, w);
        stream_store1(r_out + o0 + j + h, w.r);
        if (J_out) {
          double Jr[12];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            Jr[k] = w.c[0] * Jl[k] + w.c[1] * Jl[3 + k] + w.c[2] * Jl[6 + k];
            Jr[3 + k] = w.g[k];
            Jr[6 + k] = w.c[0] * Mn[k] + w.c[1] * Mn[3 + k] + w.c[2] * Mn[6 + k];
            Jr[9 + k] = -(w.g[0] * R[k] + w.g[1] * R[3 + k] + w.g[2] * R[6 + k]);
          }
          double2* dst = reinterpret_cast<double2*>(&stage[wv][(2 * lane + h) * 12]);
#pragma unroll
          for (int k = 0; k < 6; ++k) dst[k] = make_double2(Jr[2 * k], Jr[2 * k + 1]);
        }
      }
    }
    if (J_out) {
      __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order; keep the compiler from reordering
      const int64_t row0 = lo + it * 512 + (int64_t)wv * 128;         // first row of this wave's tile
      const int64_t rows = min((int64_t)128, hi - row0);               // may be <= 0 for idle waves
      if (rows > 0) {
        double2* gdst = reinterpret_cast<double2*>(J_out + (size_t)(o0 + row0) * 12);
        const double2* src = reinterpret_cast<const double2*>(&stage[wv][0]);
        const int n2 = (int)rows * 6;                                   // double2 elements in the tile
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          const int e = t * 64 + lane;
          if (e < n2) stream_store2(gdst + e, src[e]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wrench rows: [r | c(3) | g(3)] per block, compact order — everything a host needs to form the 1 x 12 Jacobian row with
// the per-pair table (J = [c^T J_l(aa_r) | g^T | c^T M_n | -g^T R_rn]) in 56 B instead of 104 B: the Ceres-feeding
// boundary is a PCIe link (30 GB/s measured), not HBM.  Rows staged through LDS so that every store instruction of a
// wave writes contiguous memory.
// ---------------------------------------------------------------------------------------------
// W = 7: [r | c | g].  W = 4 (point functors only): [r | g] — the moment c = (P_r - t_rw) x g is rebuilt by the host from the point it
// already holds and the pair table, so that 32 B instead of 56 B cross the PCIe link per block (pvlm_eval_force_host_async).
template <int KIND, bool NORM, int NCOLS, int W>
__global__ __launch_bounds__(256) void k_eval_wrench(const double* const* __restrict__ pair_cols, const int64_t* __restrict__ pair_stride,
                                                     const int64_t* __restrict__ out_start, const int* __restrict__ blk_pair,
                                                     const int* __restrict__ blk_chunk, int chunk_rows, const double* __restrict__ pair_tab,
                                                     double weight, double* __restrict__ w_out, int blk0, int64_t row0) {
  const int bi = blockIdx.x + blk0;
  const int p = blk_pair[bi];
  const double* __restrict__ cols = pair_cols[p];
  const int64_t n_dev = pair_stride[p];
  const int64_t o0 = out_start[p] - row0;
  const int64_t len = out_start[p + 1] - out_start[p];
  const int64_t lo = (int64_t)blk_chunk[bi] * chunk_rows;
  const int64_t hi = min(len, lo + (int64_t)chunk_rows);
  double T[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) T[k] = pair_tab[(size_t)p * PVLM_PAIR_TAB + k];
  __shared__ double stage[4][128 * W];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t n_it = (hi - lo + 511) / 512;
  for (int64_t it = 0; it < n_it; ++it) {
    const int64_t j = lo + it * 512 + 2 * (int64_t)threadIdx.x;
    if (j < hi) {
      double2 v[NCOLS];
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) v[c] = stream_load2<PVLM_NT_LOADS_MATERIALISE != 0>(cols + (size_t)c * n_dev + j);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (j + h >= hi) break;
        double rec[NCOLS];
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) rec[c] = h ? v[c].y : v[c].x;
        Wrench w;
        eval_wrench<KIND, NORM>(rec, T, weight, w);
        double* dst = &stage[wv][(2 * lane + h) * W];
        if (W == 7) { dst[0] = w.r; dst[1] = w.c[0]; dst[2] = w.c[1]; dst[3] = w.c[2]; dst[4] = w.g[0]; dst[5] = w.g[1]; dst[6] = w.g[2]; }
        else { dst[0] = w.r; dst[1] = w.g[0]; dst[2] = w.g[1]; dst[3] = w.g[2]; }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t r0 = lo + it * 512 + (int64_t)wv * 128;
    const int64_t rows = min((int64_t)128, hi - r0);
    if (rows > 0) {
      double* g = w_out + (size_t)(o0 + r0) * W;
      const int n = (int)rows * W;
#pragma unroll
      for (int t = 0; t < 2 * W; ++t) {
        const int e = t * 64 + lane;
        if (e < n) stream_store1(g + e, stage[wv][e]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// fused: per block partial [S upper(21) | gv(6) | cost(1)],  S = sum rho' v v^T, gv = sum rho' v r,
// cost = sum 1/2 rho(r^2).  v = [c ; g] (the wrench).  Deterministic: fixed tree inside the block,
// chunks summed in order by k_pair_epilogue.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// one residual block: evaluate, robustify, accumulate  S += rho' v v^T, gv += rho' v r, cost += rho/2
template <int KIND, bool NORM, int NCOLS, int LOSS>
__device__ __forceinline__ void accumulate_row(const double* rec, const double* T, double weight, double loss_a, double a2, double* acc) {
  Wrench w;
  eval_wrench<KIND, NORM>(rec, T, weight, w);
  const double s = w.r * w.r;
  double rho1 = 1.0, half_rho = 0.5 * s;
  if (LOSS == PVLM_LOSS_HUBER) {
    // ceres::HuberLoss(a): s > a^2 -> rho = 2 a sqrt(s) - a^2, rho' = max(DBL_MIN, a / sqrt(s))
    const double rr = fabs(w.r);
    const bool out = s > a2;
    const double inv = fast_rcp(out ? rr : 1.0);
    rho1 = out ? fmax(std::numeric_limits<double>::min(), loss_a * inv) : 1.0;
    half_rho = out ? fma(loss_a, rr, -0.5 * a2) : half_rho;
  }
  const double vv[6] = {w.c[0], w.c[1], w.c[2], w.g[0], w.g[1], w.g[2]};
  int q = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double wa = rho1 * vv[a];
#pragma unroll
    for (int b = a; b < 6; ++b) acc[q++] += wa * vv[b];
    acc[21 + a] += wa * w.r;
  }
  acc[27] += half_rho;
}

// Sums of N <= 32 per-lane values over the wave with a fixed tree: at offset 32 every lane keeps one half of the values and trades
// the other half with its partner (16 exchanges), at 16 a quarter, ... — after offset 2 a lane holds one value, its index = lane >> 1,
// and the last exchange adds the two partial sums.  The order of the additions is the same on every run.
template <int N>
__device__ __forceinline__ double wave_transpose_sum(const double (&a)[N], int lane) {
  static_assert(N <= 32, "at most 32 values");
  double v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) v[k] = k < N ? a[k] : 0.0;
#pragma unroll
  for (int half = 16; half >= 1; half >>= 1) {             // exchange distance 2 * half lanes
    const bool up = (lane & (2 * half)) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      const double keep = up ? v[half + k] : v[k];
      const double send = up ? v[k] : v[half + k];
      v[k] = keep + __shfl_xor(send, 2 * half, 64);
    }
  }
  return v[0] + __shfl_xor(v[0], 1, 64);
}

template <int KIND, bool NORM, int NCOLS, int LOSS>
__global__ __launch_bounds__(256, (NCOLS == 7 ? PVLM_FUSED_WAVES_7COL : PVLM_FUSED_WAVES)) void k_eval_fused(const double* const* __restrict__ pair_cols,
                                                    const int64_t* __restrict__ pair_stride,
                                                    const int64_t* __restrict__ out_start,
                                                    const int* __restrict__ blk_pair, const int* __restrict__ blk_chunk,
                                                    int chunk_rows, const double* __restrict__ pair_tab, double weight,
                                                    double loss_a, double* __restrict__ partials) {
  const int p = blk_pair[blockIdx.x];
  const double* __restrict__ cols = pair_cols[p];   // first row of the pair's segment, column 0
  const int64_t n_dev = pair_stride[p];             // column stride of the pair's block
  const int64_t len = out_start[p + 1] - out_start[p];
  const int64_t lo = (int64_t)blk_chunk[blockIdx.x] * chunk_rows;
  const int64_t hi = min(len, lo + (int64_t)chunk_rows);
  double T[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) T[k] = pair_tab[(size_t)p * PVLM_PAIR_TAB + k];
  double acc[PVLM_PARTIAL];
#pragma unroll
  for (int k = 0; k < PVLM_PARTIAL; ++k) acc[k] = 0.0;
  const double a2 = loss_a * loss_a;
  // Each lane streams two consecutive rows per column with one 16-byte load.  The cheap (Meter)
  // functors are latency-bound: their next tile is prefetched into registers while the current one
  // is evaluated (+7 % on MI355X); the Angle functors are VALU-bound at 3 waves/SIMD and lose
  // occupancy to the extra 28 VGPRs, so they load in place.
  constexpr bool kPrefetch = PVLM_PREFETCH >= 0 ? (PVLM_PREFETCH != 0) : (KIND == PVLM_POINT2PLANE_METER || KIND == PVLM_POINT2LINE_METER);
  // addresses = a wave-uniform column base (scalar registers) + one 32-bit row offset shared by the columns: the loads take the
  // saddr + voffset form and an iteration advances ONE register instead of a 64-bit pointer per column
  const double* colb[NCOLS];
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) colb[c] = cols + (size_t)c * n_dev + lo;
  const unsigned span = hi > lo ? (unsigned)(hi - lo) : 0u;
  unsigned j = 2u * threadIdx.x;
  double2 nx[NCOLS];
  if (kPrefetch && j < span) {
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) nx[c] = stream_load2<PVLM_NT_LOADS != 0>(at_byte(colb[c], 8u * j));
  }
  for (; j < span; j += 512) {
    double2 v[NCOLS];
    if (kPrefetch) {
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) v[c] = nx[c];
      if (j + 512 < span) {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) nx[c] = stream_load2<PVLM_NT_LOADS != 0>(at_byte(colb[c], 8u * (j + 512)));
      }
    } else {
#ifdef PVLM_EXP_SKIP   // timing experiment only (wrong results): how does the rate respond to fewer bytes per evaluation?
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) v[c] = *reinterpret_cast<const double2*>(colb[c < NCOLS - PVLM_EXP_SKIP ? c : 0] + j);
#else
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) v[c] = stream_load2<PVLM_NT_LOADS != 0>(at_byte(colb[c], 8u * j));
#endif
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (j + h >= span) break;
      double rec[NCOLS];
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) rec[c] = h ? v[c].y : v[c].x;
      accumulate_row<KIND, NORM, NCOLS, LOSS>(rec, T, weight, loss_a, a2, acc);
    }
  }
  __shared__ double red[4][PVLM_PARTIAL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    const double s = wave_transpose_sum<PVLM_PARTIAL>(acc, lane);       // lane 2k: total k of this wave
    if (!(lane & 1) && (lane >> 1) < PVLM_PARTIAL) red[wv][lane >> 1] = s;
  }
  __syncthreads();
  if (threadIdx.x < PVLM_PARTIAL)
    partials[(size_t)blockIdx.x * PVLM_PARTIAL + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// The same evaluation with a WAVE as the unit of work (work-list entry = (pair, chunk) per wave, four independent waves per
// workgroup): for residual sets made of many short segments — Room / Floor odometry: thousands of scan pairs of a few hundred
// blocks each — where a 256-thread workgroup per segment leaves most lanes idle and pays a 28-value cross-wave reduction through
// LDS + a barrier per segment.  A wave streams 128 rows per iteration and reduces with shuffles only.  Selected per residual set
// at finalize (pvlm_resset::wave_units: mean segment < 4096 rows); the headline's long segments keep k_eval_fused.
template <int KIND, bool NORM, int NCOLS, int LOSS>
__global__ __launch_bounds__(256, PVLM_FUSED_WAVES_WAVEFORM) void k_eval_fused_wave(const double* const* __restrict__ pair_cols,
                                                    const int64_t* __restrict__ pair_stride,
                                                    const int64_t* __restrict__ out_start,
                                                    const int* __restrict__ blk_pair, const int* __restrict__ blk_chunk,
                                                    int chunk_rows, int n_units, const double* __restrict__ pair_tab, double weight,
                                                    double loss_a, double* __restrict__ partials) {
  // the unit is the same for the 64 lanes: said explicitly, the pair's pointers, bounds and the 15 pose constants are scalar loads
  // into scalar registers (30 VGPRs less) instead of 64 identical vector loads
  const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (unit >= n_units) return;
  const int p = blk_pair[unit];
  const double* __restrict__ cols = pair_cols[p];
  const int64_t n_dev = pair_stride[p];
  const int64_t len = out_start[p + 1] - out_start[p];
  const int64_t lo = (int64_t)blk_chunk[unit] * chunk_rows;
  const int64_t hi = min(len, lo + (int64_t)chunk_rows);
  double T[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) T[k] = pair_tab[(size_t)p * PVLM_PAIR_TAB + k];
  double acc[PVLM_PARTIAL];
#pragma unroll
  for (int k = 0; k < PVLM_PARTIAL; ++k) acc[k] = 0.0;
  const double a2 = loss_a * loss_a;
  const double* colb[NCOLS];       // uniform column bases + one 32-bit row offset, as in k_eval_fused
#pragma unroll
  for (int c = 0; c < NCOLS; ++c) colb[c] = cols + (size_t)c * n_dev + lo;
  const unsigned span = hi > lo ? (unsigned)(hi - lo) : 0u;
  for (unsigned j = 2u * lane; j < span; j += 128) {
    double2 v[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; ++c) v[c] = stream_load2<PVLM_NT_LOADS != 0>(at_byte(colb[c], 8u * j));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (j + h >= span) break;
      double rec[NCOLS];
#pragma unroll
      for (int c = 0; c < NCOLS; ++c) rec[c] = h ? v[c].y : v[c].x;
      accumulate_row<KIND, NORM, NCOLS, LOSS>(rec, T, weight, loss_a, a2, acc);
    }
  }
  // the 28 totals in one transposing butterfly (32 exchanges instead of 28 x 6): lanes 2k and 2k + 1 end with total k
  const double mine = wave_transpose_sum<PVLM_PARTIAL>(acc, lane);
  if (!(lane & 1) && (lane >> 1) < PVLM_PARTIAL) partials[(size_t)unit * PVLM_PARTIAL + (lane >> 1)] = mine;
}

// one block (128 threads) per pair: sum chunk partials in order, expand S, apply
// D_r = blockdiag(Jl_r, I), D_n = blockdiag(M_n, -R_rn) and write the 121-double pair block.
__global__ __launch_bounds__(128) void k_pair_epilogue(int P, const int* __restrict__ pair_blk_start,
                                                       const double* __restrict__ partials,
                                                       const double* __restrict__ pair_tab, double* __restrict__ out) {
  const int p = blockIdx.x;
  __shared__ double S[6][6], gv[6], Dr[6][6], Dn[6][6], cost;
  __shared__ double tot[PVLM_PARTIAL];
  const int t = threadIdx.x;
  if (t < PVLM_PARTIAL) {
    double s = 0.0;
    for (int b = pair_blk_start[p]; b < pair_blk_start[p + 1]; ++b) s += partials[(size_t)b * PVLM_PARTIAL + t];
    tot[t] = s;
  }
  if (t < 36) {
    const int i = t / 6, j = t % 6;
    const double* T = pair_tab + (size_t)p * PVLM_PAIR_TAB;
    double dr = 0.0, dn = 0.0;
    if (i < 3 && j < 3) { dr = T[15 + i * 3 + j]; dn = T[24 + i * 3 + j]; }
    else if (i >= 3 && j >= 3) { dr = (i == j) ? 1.0 : 0.0; dn = -T[(i - 3) * 3 + (j - 3)]; }
    Dr[i][j] = dr; Dn[i][j] = dn;
  }
  __syncthreads();
  if (t < 36) {
    const int i = t / 6, j = t % 6;
    const int a = i < j ? i : j, b = i < j ? j : i;
    S[i][j] = tot[a * 6 - a * (a - 1) / 2 + (b - a)];
  }
  if (t < 6) gv[t] = tot[21 + t];
  if (t == 0) cost = tot[27];
  __syncthreads();
  double* o = out + (size_t)p * PVLM_PAIR_BLOCK;
  if (t < 108) {
    const int blk = t / 36, i = (t % 36) / 6, j = t % 6;
    const double(*L)[6] = (blk == 2) ? Dn : Dr;   // left factor (transposed)
    const double(*Rm)[6] = (blk == 0) ? Dr : Dn;  // right factor
    double s = 0.0;
    for (int a = 0; a < 6; ++a) {
      double u = 0.0;
      for (int b = 0; b < 6; ++b) u += S[a][b] * Rm[b][j];
      s += L[a][i] * u;
    }
    o[t] = s;
  } else if (t < 120) {
    const int i = (t - 108) % 6;
    const double(*L)[6] = (t < 114) ? Dr : Dn;
    double s = 0.0;
    for (int a = 0; a < 6; ++a) s += L[a][i] * gv[a];
    o[t] = s;
  } else if (t == 120) {
    o[120] = cost;
  }
}

// packed normal equations: deterministic gather over the CSR lists built at bind time.
__global__ void k_neq_gather(int n_poses, int n_upairs, const int* __restrict__ diag_off, const int* __restrict__ diag_items,
                             const int* __restrict__ off_off, const int* __restrict__ off_items,
                             const double* __restrict__ pair_blocks, int P, int zero_first, double* __restrict__ packed) {
  const int b = blockIdx.x, t = threadIdx.x;
  double* Hd = packed;
  double* Ho = packed + (size_t)n_poses * 36;
  double* g = Ho + (size_t)n_upairs * 36;
  double* cost = g + (size_t)n_poses * 6;
  if (b < n_poses) {
    if (t < 42) {
      double s = 0.0;
      for (int k = diag_off[b]; k < diag_off[b + 1]; ++k) {
        const int it = diag_items[k];
        const double* pb = pair_blocks + (size_t)(it >> 1) * PVLM_PAIR_BLOCK;
        const int role = it & 1;
        s += (t < 36) ? pb[(role ? 72 : 0) + t] : pb[108 + (role ? 6 : 0) + (t - 36)];
      }
      double* dst = (t < 36) ? &Hd[(size_t)b * 36 + t] : &g[(size_t)b * 6 + (t - 36)];
      *dst = zero_first ? s : (*dst + s);
    }
  } else if (b < n_poses + n_upairs) {
    const int u = b - n_poses;
    if (t < 36) {
      const int i = t / 6, j = t % 6;
      double s = 0.0;
      for (int k = off_off[u]; k < off_off[u + 1]; ++k) {
        const int it = off_items[k];
        const double* pb = pair_blocks + (size_t)(it >> 1) * PVLM_PAIR_BLOCK + 36;  // H_rn
        s += (it & 1) ? pb[j * 6 + i] : pb[t];
      }
      double* dst = &Ho[(size_t)u * 36 + t];
      *dst = zero_first ? s : (*dst + s);
    }
  } else {
    // cost: single block, fixed-order tree over pairs
    __shared__ double red[256];
    double s = 0.0;
    for (int p = t; p < P; p += 256) s += pair_blocks[(size_t)p * PVLM_PAIR_BLOCK + 120];
    red[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (t < w) red[t] += red[t + w]; __syncthreads(); }
    if (t == 0) *cost = zero_first ? red[0] : (*cost + red[0]);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static pvlm_status ensure_pose_cap(pvlm_ctx* ctx, int n) {
  if (n <= ctx->cap_poses) return PVLM_OK;
  if (ctx->capturing) { PVLM_SET_ERR(ctx, "pose table would grow inside a graph capture (run the step once before pvlm_graph_begin)"); return PVLM_ERR_STATE; }
  PVLM_TRY_SYNC(ctx);
  pvlm_i_free(ctx, ctx->d_aa); pvlm_i_free(ctx, ctx->d_t); pvlm_i_free(ctx, ctx->d_pose_tab);
  ctx->d_aa = ctx->d_t = ctx->d_pose_tab = nullptr;
  ctx->cap_poses = 0;
  pvlm_status st;
  if ((st = pvlm_i_alloc(ctx, &ctx->d_aa, (size_t)n * 3))) return st;
  if ((st = pvlm_i_alloc(ctx, &ctx->d_t, (size_t)n * 3))) return st;
  if ((st = pvlm_i_alloc(ctx, &ctx->d_pose_tab, (size_t)n * PVLM_POSE_TAB))) return st;
  ctx->cap_poses = n;
  return PVLM_OK;
}

static pvlm_status pose_table_launch(pvlm_ctx* ctx, int n, const double* d_aa, const double* d_t) {
  if (n > 0) {
    hipLaunchKernelGGL(k_pose_table, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, n, d_aa, d_t, ctx->d_pose_tab);
    PVLM_HIP(ctx, hipGetLastError());
  }
  ctx->n_poses = n;
  ctx->poses_set = true;
  ctx->pose_epoch++;
  return PVLM_OK;
}

static pvlm_status ensure_pair_table(pvlm_ctx* ctx, const pvlm_resset* crs) {
  pvlm_resset* rs = const_cast<pvlm_resset*>(crs);
  if (!ctx->poses_set) { PVLM_SET_ERR(ctx, "pvlm_set_poses must be called before evaluation"); return PVLM_ERR_STATE; }
  for (int p = 0; p < rs->n_pairs; ++p)
    if (rs->h_ref[p] >= ctx->n_poses || rs->h_nei[p] >= ctx->n_poses) {
      PVLM_SET_ERR(ctx, "segment %d references pose %d/%d but only %d poses are set", p, rs->h_ref[p], rs->h_nei[p], ctx->n_poses);
      return PVLM_ERR_STATE;
    }
  if (rs->pair_tab_epoch == ctx->pose_epoch || rs->n_pairs == 0) return PVLM_OK;
  hipLaunchKernelGGL(k_pair_table, dim3((rs->n_pairs + 63) / 64), dim3(64), 0, ctx->stream, rs->n_pairs, rs->d_ref, rs->d_nei,
                     ctx->n_poses, ctx->d_pose_tab, rs->d_pair_tab);
  PVLM_HIP(ctx, hipGetLastError());
  rs->pair_tab_epoch = ctx->pose_epoch;
  return PVLM_OK;
}

template <int KIND, bool NORM, int NCOLS>
static void launch_materialise(pvlm_ctx* ctx, const pvlm_resset* rs, double* d_r, double* d_J, int blk0 = 0, int nblk = -1, int64_t row0 = 0) {
  hipLaunchKernelGGL((k_eval_materialise<KIND, NORM, NCOLS>), dim3(nblk < 0 ? rs->n_blocks : nblk), dim3(256), 0, ctx->stream, rs->d_pair_cols,
                     rs->d_pair_stride, rs->d_out_start, rs->d_blk_pair, rs->d_blk_chunk, rs->chunk_rows, rs->d_pair_tab, rs->weight, d_r, d_J,
                     blk0, row0);
}
template <int KIND, bool NORM, int NCOLS>
static void launch_wrench(pvlm_ctx* ctx, const pvlm_resset* rs, double* d_w, int blk0, int nblk, int64_t row0, bool force_only) {
  if (force_only)
    hipLaunchKernelGGL((k_eval_wrench<KIND, NORM, NCOLS, 4>), dim3(nblk), dim3(256), 0, ctx->stream, rs->d_pair_cols, rs->d_pair_stride, rs->d_out_start,
                       rs->d_blk_pair, rs->d_blk_chunk, rs->chunk_rows, rs->d_pair_tab, rs->weight, d_w, blk0, row0);
  else
    hipLaunchKernelGGL((k_eval_wrench<KIND, NORM, NCOLS, 7>), dim3(nblk), dim3(256), 0, ctx->stream, rs->d_pair_cols, rs->d_pair_stride, rs->d_out_start,
                       rs->d_blk_pair, rs->d_blk_chunk, rs->chunk_rows, rs->d_pair_tab, rs->weight, d_w, blk0, row0);
}

template <int KIND, bool NORM, int NCOLS>
static void launch_fused(pvlm_ctx* ctx, const pvlm_resset* rs, int loss, double a) {
  if (rs->wave_units) {          // many short segments: one wave per (pair, chunk)
    const dim3 grid((unsigned)((rs->n_blocks + 3) / 4));
    if (loss == PVLM_LOSS_HUBER)
      hipLaunchKernelGGL((k_eval_fused_wave<KIND, NORM, NCOLS, PVLM_LOSS_HUBER>), grid, dim3(256), 0, ctx->stream, rs->d_pair_cols, rs->d_pair_stride, rs->d_out_start,
                         rs->d_blk_pair, rs->d_blk_chunk, rs->chunk_rows, rs->n_blocks, rs->d_pair_tab, rs->weight, a, rs->d_partials);
    else
      hipLaunchKernelGGL((k_eval_fused_wave<KIND, NORM, NCOLS, PVLM_LOSS_NONE>), grid, dim3(256), 0, ctx->stream, rs->d_pair_cols, rs->d_pair_stride, rs->d_out_start,
                         rs->d_blk_pair, rs->d_blk_chunk, rs->chunk_rows, rs->n_blocks, rs->d_pair_tab, rs->weight, a, rs->d_partials);
    return;
  }
  if (loss == PVLM_LOSS_HUBER)
    hipLaunchKernelGGL((k_eval_fused<KIND, NORM, NCOLS, PVLM_LOSS_HUBER>), dim3(rs->n_blocks), dim3(256), 0, ctx->stream, rs->d_pair_cols,
                       rs->d_pair_stride, rs->d_out_start, rs->d_blk_pair, rs->d_blk_chunk, rs->chunk_rows, rs->d_pair_tab,
                       rs->weight, a, rs->d_partials);
  else
    hipLaunchKernelGGL((k_eval_fused<KIND, NORM, NCOLS, PVLM_LOSS_NONE>), dim3(rs->n_blocks), dim3(256), 0, ctx->stream, rs->d_pair_cols,
                       rs->d_pair_stride, rs->d_out_start, rs->d_blk_pair, rs->d_blk_chunk, rs->chunk_rows, rs->d_pair_tab,
                       rs->weight, a, rs->d_partials);
}

#define PVLM_DISPATCH(FN, ...)                                                                         \
  do {                                                                                                 \
    const bool nz = (rs->flags & PVLM_FLAG_NORMALIZE_DISTANCE) != 0;                                   \
    switch (rs->kind) {                                                                                \
      case PVLM_POINT2PLANE_METER: FN<PVLM_POINT2PLANE_METER, false, 7>(__VA_ARGS__); break;           \
      case PVLM_POINT2PLANE_ANGLE:                                                                     \
        if (nz) FN<PVLM_POINT2PLANE_ANGLE, true, 7>(__VA_ARGS__);                                      \
        else FN<PVLM_POINT2PLANE_ANGLE, false, 7>(__VA_ARGS__);                                        \
        break;                                                                                         \
      case PVLM_POINT2LINE_METER: FN<PVLM_POINT2LINE_METER, false, 9>(__VA_ARGS__); break;             \
      case PVLM_POINT2LINE_ANGLE:                                                                      \
        if (nz) FN<PVLM_POINT2LINE_ANGLE, true, 9>(__VA_ARGS__);                                       \
        else FN<PVLM_POINT2LINE_ANGLE, false, 9>(__VA_ARGS__);                                         \
        break;                                                                                         \
      case PVLM_PLANE2PLANE_GLOBAL: FN<PVLM_PLANE2PLANE_GLOBAL, false, 10>(__VA_ARGS__); break;        \
      case PVLM_PLANE_IOU: FN<PVLM_PLANE_IOU, false, 12>(__VA_ARGS__); break;                          \
    }                                                                                                  \
  } while (0)

extern "C" {

pvlm_status pvlm_set_poses(pvlm_ctx* ctx, int n, const double* aa, const double* t) {
  if (!ctx || n < 0 || (n > 0 && (!aa || !t))) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_status st = ensure_pose_cap(ctx, n);
  if (st) return st;
  if (n > 0) {
    // through the pinned arena: the caller may reuse aa / t at once, and nothing waits for the device here
    if ((st = pvlm_i_h2d_q(ctx, ctx->d_aa, aa, (size_t)n * 3 * sizeof(double)))) return st;
    if ((st = pvlm_i_h2d_q(ctx, ctx->d_t, t, (size_t)n * 3 * sizeof(double)))) return st;
  }
  return pose_table_launch(ctx, n, ctx->d_aa, ctx->d_t);
}

pvlm_status pvlm_set_poses_dev(pvlm_ctx* ctx, int n, const double* d_aa, const double* d_t) {
  if (!ctx || n < 0 || (n > 0 && (!d_aa || !d_t))) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_status st = ensure_pose_cap(ctx, n);
  if (st) return st;
  return pose_table_launch(ctx, n, d_aa, d_t);
}

pvlm_status pvlm_eval_dev(pvlm_ctx* ctx, const pvlm_resset* rs, double* d_r, double* d_J) {
  if (!ctx || !rs || !d_r) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_status st = ensure_pair_table(ctx, rs);
  if (st) return st;
  if (rs->n_blocks == 0) return PVLM_OK;
  {
    pvlm_prof_scope prof(ctx, 1);
    PVLM_DISPATCH(launch_materialise, ctx, rs, d_r, d_J);
  }
  PVLM_HIP(ctx, hipGetLastError());
  return PVLM_OK;
}

pvlm_status pvlm_eval(pvlm_ctx* ctx, const pvlm_resset* rs, double* r, double* J) {
  if (!ctx || !rs || (rs->n > 0 && !r)) return PVLM_ERR_ARG;
  if (rs->n == 0) return ensure_pair_table(ctx, rs);
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  double *d_r = nullptr, *d_J = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_r, (size_t)rs->n);
  if (!st && J) st = pvlm_i_alloc(ctx, &d_J, (size_t)rs->n * 12);
  if (!st) st = pvlm_eval_dev(ctx, rs, d_r, d_J);
  // the caller's r / J are ordinary (pageable) arrays: staged through the pinned arena in 16 MB pieces
  if (!st) st = pvlm_i_d2h_q(ctx, r, d_r, (size_t)rs->n * sizeof(double));
  if (!st && J) st = pvlm_i_d2h_q(ctx, J, d_J, (size_t)rs->n * 12 * sizeof(double));
  { const pvlm_status s2 = pvlm_i_sync(ctx); if (!st) st = s2; }
  pvlm_i_free(ctx, d_r); pvlm_i_free(ctx, d_J);
  return st;
}

// Evaluation delivered to HOST memory, slice by slice through a bounded device staging buffer (PVLM_STAGE_ROWS rows,
// 32 M by default): kernel over a run of whole pairs -> asynchronous copy of that slice; everything on the context stream.
// mode 0: residuals[n] (+ jacobians[n x 12]); mode 1: wrench rows [n x 7] + the pair tables; mode 2: force rows [n x 4] + the pair tables.
static pvlm_status eval_to_host(pvlm_ctx* ctx, const pvlm_resset* crs, int mode, double* h_a, double* h_b) {
  pvlm_resset* rs = const_cast<pvlm_resset*>(crs);
  pvlm_status st = ensure_pair_table(ctx, rs);
  if (st) return st;
  if (mode >= 1 && h_b && rs->n_pairs > 0)
    PVLM_HIP(ctx, hipMemcpyAsync(h_b, rs->d_pair_tab, (size_t)rs->n_pairs * PVLM_PAIR_TAB * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (rs->n == 0) return PVLM_OK;
  int64_t stage_rows = 32ll << 20;
  if (const char* e = getenv("PVLM_STAGE_ROWS")) { const long long v = atoll(e); if (v > 0) stage_rows = v; }
  const int width = mode == 1 ? 7 : (mode == 2 ? 4 : (h_b ? 13 : 1));
  int64_t need = 0;      // the staging buffer holds the largest run of pairs that fits stage_rows (at least one pair)
  for (int p = 0; p < rs->n_pairs;) {
    int q = p; int64_t rows = 0;
    while (q < rs->n_pairs && (q == p || rows + (rs->h_out_start[q + 1] - rs->h_out_start[q]) <= stage_rows)) { rows += rs->h_out_start[q + 1] - rs->h_out_start[q]; ++q; }
    need = std::max(need, rows);
    p = q;
  }
  if (rs->stage_doubles < (size_t)need * width) {
    if (ctx->capturing) { PVLM_SET_ERR(ctx, "staging buffer would grow inside a graph capture"); return PVLM_ERR_STATE; }
    pvlm_i_free(ctx, rs->d_stage); rs->d_stage = nullptr; rs->stage_doubles = 0;
    if ((st = pvlm_i_alloc(ctx, &rs->d_stage, (size_t)need * width))) return st;
    rs->stage_doubles = (size_t)need * width;
  }
  for (int p = 0; p < rs->n_pairs;) {
    int q = p; int64_t rows = 0;
    while (q < rs->n_pairs && (q == p || rows + (rs->h_out_start[q + 1] - rs->h_out_start[q]) <= stage_rows)) { rows += rs->h_out_start[q + 1] - rs->h_out_start[q]; ++q; }
    const int blk0 = rs->h_pair_blk_start[p], nblk = rs->h_pair_blk_start[q] - blk0;
    const int64_t row0 = rs->h_out_start[p];
    if (nblk > 0) {
      double* d_r = rs->d_stage;
      double* d_J = rs->d_stage + rows;
      {
        pvlm_prof_scope prof(ctx, 1);
        if (mode >= 1) PVLM_DISPATCH(launch_wrench, ctx, rs, rs->d_stage, blk0, nblk, row0, mode == 2);
        else PVLM_DISPATCH(launch_materialise, ctx, rs, d_r, h_b ? d_J : nullptr, blk0, nblk, row0);
      }
      PVLM_HIP(ctx, hipGetLastError());
      if (mode >= 1) PVLM_HIP(ctx, hipMemcpyAsync(h_a + (size_t)row0 * width, rs->d_stage, (size_t)rows * width * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      else {
        PVLM_HIP(ctx, hipMemcpyAsync(h_a + row0, d_r, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (h_b) PVLM_HIP(ctx, hipMemcpyAsync(h_b + (size_t)row0 * 12, d_J, (size_t)rows * 12 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      }
    }
    p = q;
  }
  return PVLM_OK;
}

pvlm_status pvlm_eval_host_async(pvlm_ctx* ctx, const pvlm_resset* rs, double* residuals, double* jacobians) {
  if (!ctx || !rs || (rs->n > 0 && !residuals)) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  return eval_to_host(ctx, rs, 0, residuals, jacobians);
}

pvlm_status pvlm_eval_wrench_host_async(pvlm_ctx* ctx, const pvlm_resset* rs, double* wrench_rows, double* pair_tables) {
  if (!ctx || !rs || (rs->n > 0 && !wrench_rows) || (rs->n_pairs > 0 && !pair_tables)) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  return eval_to_host(ctx, rs, 1, wrench_rows, pair_tables);
}

pvlm_status pvlm_eval_force_host_async(pvlm_ctx* ctx, const pvlm_resset* rs, double* force_rows, double* pair_tables) {
  if (!ctx || !rs || (rs->n > 0 && !force_rows) || (rs->n_pairs > 0 && !pair_tables)) return PVLM_ERR_ARG;
  if (rs->kind > PVLM_POINT2LINE_ANGLE) { PVLM_SET_ERR(ctx, "pvlm_eval_force_host_async: point functors only (kinds 0..3): the moment of a plane / IOU block is not (P_r - t_rw) x g"); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  return eval_to_host(ctx, rs, 2, force_rows, pair_tables);
}

pvlm_status pvlm_host_alloc(pvlm_ctx* ctx, int64_t bytes, void** out) {
  if (!ctx || !out || bytes < 0) return PVLM_ERR_ARG;
  *out = nullptr;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t want = (size_t)std::max<int64_t>(bytes, 1);
  try {
    if (ctx->host_spare && ctx->host_spare_bytes >= want && ctx->host_spare_bytes <= 2 * want + 4096) {       // the block the last pvlm_host_free left behind
      *out = ctx->host_spare;
      ctx->host_live.push_back({ctx->host_spare, ctx->host_spare_bytes});
      ctx->host_spare = nullptr; ctx->host_spare_bytes = 0;
      return PVLM_OK;
    }
    ctx->host_live.reserve(ctx->host_live.size() + 1);
  } catch (const std::bad_alloc&) { return PVLM_ERR_NOMEM; }
  PVLM_HIP(ctx, hipHostMalloc(out, want, hipHostMallocDefault));
  ctx->host_live.push_back({*out, want});
  return PVLM_OK;
}

pvlm_status pvlm_host_free(pvlm_ctx* ctx, void* p) {
  if (!ctx) return PVLM_ERR_ARG;
  if (!p) return PVLM_OK;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  PVLM_TRY_SYNC(ctx);
  size_t bytes = 0;
  for (size_t k = 0; k < ctx->host_live.size(); ++k)
    if (ctx->host_live[k].first == p) { bytes = ctx->host_live[k].second; ctx->host_live[k] = ctx->host_live.back(); ctx->host_live.pop_back(); break; }
  if (bytes > 0 && bytes <= ((size_t)64 << 20)) {                  // kept for the next allocation of about this size; the previous spare goes
    if (ctx->host_spare) PVLM_HIP(ctx, hipHostFree(ctx->host_spare));
    ctx->host_spare = p; ctx->host_spare_bytes = bytes;
    return PVLM_OK;
  }
  PVLM_HIP(ctx, hipHostFree(p));
  return PVLM_OK;
}

pvlm_status pvlm_eval_pair_blocks_dev(pvlm_ctx* ctx, const pvlm_resset* rs, pvlm_loss loss, double a, double* d_out) {
  if (!ctx || !rs || !d_out || (loss != PVLM_LOSS_NONE && loss != PVLM_LOSS_HUBER)) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_status st = ensure_pair_table(ctx, rs);
  if (st) return st;
  if (rs->n_pairs == 0) return PVLM_OK;
  if (rs->n_blocks > 0) {
    pvlm_prof_scope prof(ctx, 0);
    PVLM_DISPATCH(launch_fused, ctx, rs, (int)loss, a);
    PVLM_HIP(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(k_pair_epilogue, dim3(rs->n_pairs), dim3(128), 0, ctx->stream, rs->n_pairs, rs->d_pair_blk_start, rs->d_partials,
                     rs->d_pair_tab, d_out);
  PVLM_HIP(ctx, hipGetLastError());
  return PVLM_OK;
}

pvlm_status pvlm_eval_pair_blocks(pvlm_ctx* ctx, const pvlm_resset* rs, pvlm_loss loss, double a, double* out) {
  if (!ctx || !rs || (rs->n_pairs > 0 && !out)) return PVLM_ERR_ARG;
  pvlm_status st = pvlm_eval_pair_blocks_dev(ctx, rs, loss, a, rs->d_pair_blocks);
  if (st) return st;
  if (rs->n_pairs > 0) {
    if ((st = pvlm_i_d2h(ctx, out, rs->d_pair_blocks, (size_t)rs->n_pairs * PVLM_PAIR_BLOCK * sizeof(double)))) return st;
  }
  return PVLM_OK;
}

static pvlm_status neq_bind(pvlm_ctx* ctx, pvlm_neq* q, const pvlm_resset* rs) {
  if (q->bound == rs && q->bound_serial == rs->serial) return PVLM_OK;
  if (ctx->capturing) { PVLM_SET_ERR(ctx, "normal equations not bound to this residual set yet (run the step once before pvlm_graph_begin)"); return PVLM_ERR_STATE; }
  q->bound = nullptr;
  const int P = rs->n_pairs;
  std::vector<std::vector<int>> diag(q->n_poses), off(q->n_upairs);
  // unordered pair lookup
  std::vector<std::pair<long long, int>> key(q->n_upairs);
  for (int u = 0; u < q->n_upairs; ++u) key[u] = {(long long)q->ui[u] * q->n_poses + q->uj[u], u};
  std::sort(key.begin(), key.end());
  for (int p = 0; p < P; ++p) {
    const int r = rs->h_ref[p], n = rs->h_nei[p];
    if (r >= q->n_poses || n >= q->n_poses) { PVLM_SET_ERR(ctx, "segment %d pose id outside the normal-equation structure", p); return PVLM_ERR_ARG; }
    if (r == n) { PVLM_SET_ERR(ctx, "segment %d has ref == nei", p); return PVLM_ERR_ARG; }
    diag[r].push_back(p * 2 + 0);
    diag[n].push_back(p * 2 + 1);
    const int i = std::min(r, n), j = std::max(r, n);
    const long long k = (long long)i * q->n_poses + j;
    auto it = std::lower_bound(key.begin(), key.end(), std::make_pair(k, -1));
    if (it == key.end() || it->first != k) { PVLM_SET_ERR(ctx, "pose pair (%d,%d) of segment %d missing from the upair list", i, j, p); return PVLM_ERR_ARG; }
    off[it->second].push_back(p * 2 + (r < n ? 0 : 1));  // H_rn is d2/dx_r dx_n; transposed when r is the larger index
  }
  std::vector<int> doff(q->n_poses + 1, 0), ditems, ooff(q->n_upairs + 1, 0), oitems;
  for (int i = 0; i < q->n_poses; ++i) { doff[i] = (int)ditems.size(); ditems.insert(ditems.end(), diag[i].begin(), diag[i].end()); }
  doff[q->n_poses] = (int)ditems.size();
  for (int u = 0; u < q->n_upairs; ++u) { ooff[u] = (int)oitems.size(); oitems.insert(oitems.end(), off[u].begin(), off[u].end()); }
  ooff[q->n_upairs] = (int)oitems.size();
  pvlm_i_free(ctx, q->d_diag_off); pvlm_i_free(ctx, q->d_diag_items); pvlm_i_free(ctx, q->d_off_off); pvlm_i_free(ctx, q->d_off_items);
  q->d_diag_off = q->d_diag_items = q->d_off_off = q->d_off_items = nullptr;
  pvlm_status st;
  if ((st = pvlm_i_alloc(ctx, &q->d_diag_off, doff.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &q->d_diag_items, ditems.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &q->d_off_off, ooff.size()))) return st;
  if ((st = pvlm_i_alloc(ctx, &q->d_off_items, oitems.size()))) return st;
  if ((st = pvlm_i_h2d_q(ctx, q->d_diag_off, doff.data(), doff.size() * sizeof(int)))) return st;
  if (!ditems.empty() && (st = pvlm_i_h2d_q(ctx, q->d_diag_items, ditems.data(), ditems.size() * sizeof(int)))) return st;
  if ((st = pvlm_i_h2d_q(ctx, q->d_off_off, ooff.data(), ooff.size() * sizeof(int)))) return st;
  if (!oitems.empty() && (st = pvlm_i_h2d_q(ctx, q->d_off_items, oitems.data(), oitems.size() * sizeof(int)))) return st;
  q->bound = rs;
  q->bound_serial = rs->serial;
  return PVLM_OK;
}

pvlm_status pvlm_neq_accumulate_dev(pvlm_ctx* ctx, pvlm_neq* q, const pvlm_resset* rs, pvlm_loss loss, double a, int zero_first,
                                    double* d_packed) {
  if (!ctx || !q || !rs || !d_packed) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_status st = neq_bind(ctx, q, rs);
  if (st) return st;
  st = pvlm_eval_pair_blocks_dev(ctx, rs, loss, a, rs->d_pair_blocks);
  if (st) return st;
  hipLaunchKernelGGL(k_neq_gather, dim3(q->n_poses + q->n_upairs + 1), dim3(256), 0, ctx->stream, q->n_poses, q->n_upairs, q->d_diag_off,
                     q->d_diag_items, q->d_off_off, q->d_off_items, rs->d_pair_blocks, rs->n_pairs, zero_first, d_packed);
  PVLM_HIP(ctx, hipGetLastError());
  return PVLM_OK;
}

pvlm_status pvlm_neq_accumulate(pvlm_ctx* ctx, pvlm_neq* q, const pvlm_resset* rs, pvlm_loss loss, double a, int zero_first, double* packed) {
  if (!ctx || !q || !rs || !packed) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t cnt = (size_t)pvlm_neq_size(q);
  const size_t nb = cnt * sizeof(double);
  // the device copy of the packed buffer lives as long as the context (an LM driver calls this every evaluation)
  if (ctx->neq_tmp_count < cnt) {
    pvlm_i_free(ctx, ctx->d_neq_tmp); ctx->d_neq_tmp = nullptr; ctx->neq_tmp_count = 0;
    pvlm_status sa = pvlm_i_alloc(ctx, &ctx->d_neq_tmp, cnt);
    if (sa) return sa;
    ctx->neq_tmp_count = cnt;
  }
  double* d = ctx->d_neq_tmp;
  pvlm_status st = PVLM_OK;
  if (!zero_first && (st = pvlm_i_h2d_q(ctx, d, packed, nb))) return st;
  st = pvlm_neq_accumulate_dev(ctx, q, rs, loss, a, zero_first, d);
  if (st) return st;
  return pvlm_i_d2h(ctx, packed, d, nb);
}

pvlm_status pvlm_neq_accumulate_async(pvlm_ctx* ctx, pvlm_neq* q, const pvlm_resset* rs, pvlm_loss loss, double a, double* packed) {
  if (!ctx || !q || !rs || !packed) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t cnt = (size_t)pvlm_neq_size(q);
  if (!q->d_packed) {
    const pvlm_status sa = pvlm_i_alloc(ctx, &q->d_packed, cnt);
    if (sa) return sa;
  }
  const pvlm_status st = pvlm_neq_accumulate_dev(ctx, q, rs, loss, a, 1, q->d_packed);
  if (st) return st;
  return pvlm_i_d2h_q(ctx, packed, q->d_packed, cnt * sizeof(double));   // complete at the next pvlm_synchronize
}

pvlm_status pvlm_neq_accumulate_sets(pvlm_ctx* ctx, int n, pvlm_neq* const* neq, const pvlm_resset* const* rs, const pvlm_loss* loss,
                                     const double* loss_a, double* packed) {
  if (!ctx || n <= 0 || !neq || !rs || !loss || !loss_a || !packed) return PVLM_ERR_ARG;
  for (int k = 0; k < n; ++k) {
    if (!neq[k] || !rs[k]) return PVLM_ERR_ARG;
    if (neq[k]->n_poses != neq[0]->n_poses || neq[k]->n_upairs != neq[0]->n_upairs || neq[k]->ui != neq[0]->ui || neq[k]->uj != neq[0]->uj) {
      PVLM_SET_ERR(ctx, "pvlm_neq_accumulate_sets: structure %d differs from structure 0 (same n_poses and pair list required)", k);
      return PVLM_ERR_ARG;
    }
  }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_neq* q0 = neq[0];
  const size_t cnt = (size_t)pvlm_neq_size(q0);
  if (!q0->d_packed) {
    const pvlm_status sa = pvlm_i_alloc(ctx, &q0->d_packed, cnt);
    if (sa) return sa;
  }
  for (int k = 0; k < n; ++k) {
    const pvlm_status st = pvlm_neq_accumulate_dev(ctx, neq[k], rs[k], loss[k], loss_a[k], k == 0 ? 1 : 0, q0->d_packed);
    if (st) return st;
  }
  return pvlm_i_d2h_q(ctx, packed, q0->d_packed, cnt * sizeof(double));   // complete at the next pvlm_synchronize
}

}  // extern "C"

// pvlm_preload: HIP loads the code object of a translation unit at the first launch of one of its kernels (15 ms for the larger ones) — an empty launch from here
// moves that out of the first call that needs this file's kernels
__global__ void k_preload_eval() {}
void pvlm_i_preload_eval(hipStream_t s) { hipLaunchKernelGGL(k_preload_eval, dim3(1), dim3(1), 0, s); }
