// Dense symmetric positive definite solve on the GPU for the LM driver of the host mirror (the reduced pose system of a
// Room/Floor-sized joint optimisation has 5e3..1e4 unknowns: a host Cholesky of it costs seconds per LM iteration and
// dwarfs everything the hot path does — measured 158 s of a 164 s JointOptimize at 454 frames).  Not part of the hot
// path (upstream this step is inside ceres::Solve: SPARSE_SCHUR + SuiteSparse, util/Optimization.cpp:608-666), but it
// has to exist for the mirrored call surface to be usable at Room scale.  Hand-written blocked Cholesky: rocSOLVER's
// potrf was tried first and works, but the first rocblas_create_handle of a process pages the whole rocBLAS kernel
// library in (measured 140..510 s on a cold box) — unacceptable inside a library call.
#include <algorithm>
#include <cstdlib>
#include <iterator>
#include <new>
#include <chrono>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

#include "pvlm_internal.h"
#include "pvlm_spd_plan.h"

// ---- blocked right-looking Cholesky, fp64, lower triangle of a row-major dense matrix ------------------------------
// Step k (block column of NB = 32): (1) one workgroup factorises the diagonal block in LDS and inverts it, (2) the panel
// below is multiplied by that inverse, (3) the trailing lower triangle gets its rank-32 update in 64 x 64 tiles on the
// f64 matrix core (both panel slices staged in LDS).  *info != 0 (set by a non-positive pivot, 1-based
// like LAPACK) makes every later kernel of the sequence return at once.
#define PVLM_CHOL_NB 32

// Measured variants that lost (round 1's separate diagonal / panel launches, the register-tiled VALU trailing update, the two-stream
// look-ahead) are compiled only into a library built with -DPVLM_MEASURED_VARIANTS=1 (python -m panovlm_amd.build --variant measured
// -DPVLM_MEASURED_VARIANTS=1), where PVLM_CHOL_SPLIT / PVLM_CHOL_VALU / PVLM_CHOL_LOOKAHEAD select them; their numbers are in
// profiles/r1_chol_bench.jsonl and DESIGN.md §3 K10.
#ifndef PVLM_MEASURED_VARIANTS
#define PVLM_MEASURED_VARIANTS 0
#endif
#if PVLM_MEASURED_VARIANTS
__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ M, int n, int k0, int kb, double* __restrict__ Linv, int* __restrict__ info) {
  __shared__ double a[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ double inv[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ int fail;
  if (*info != 0) return;
  const int t = threadIdx.x;
  if (t == 0) fail = 0;
  for (int e = t; e < PVLM_CHOL_NB * PVLM_CHOL_NB; e += 256) {
    const int i = e / PVLM_CHOL_NB, j = e % PVLM_CHOL_NB;
    a[i][j] = (i < kb && j <= i) ? M[(size_t)(k0 + i) * n + k0 + j] : 0.0;
    inv[i][j] = (i == j && i < kb) ? 1.0 : 0.0;
  }
  __syncthreads();
  // right-looking factorisation of the block; the same eliminations applied to an identity give L^-1 on the way
  // (Y = I; row j of Y is divided by the pivot, rows below get Y[i] -= L[i][j] Y[j]): no extra barriers.
  for (int j = 0; j < kb; ++j) {
    if (t == 0) { const double d = a[j][j]; if (!(d > 0.0)) fail = j + 1; else a[j][j] = sqrt(d); }
    __syncthreads();
    if (fail) break;
    const double piv = a[j][j];
    if (t > j && t < kb) a[t][j] /= piv;
    if (t >= 64 && t - 64 <= j) inv[j][t - 64] /= piv;          // second wave: row j of the inverse (columns <= j)
    __syncthreads();
    for (int e = t; e < PVLM_CHOL_NB * PVLM_CHOL_NB; e += 256) {
      const int i = e / PVLM_CHOL_NB, c = e % PVLM_CHOL_NB;
      if (i > j && i < kb) {
        if (c > j && c <= i) a[i][c] -= a[i][j] * a[c][j];
        else if (c <= j) inv[i][c] -= a[i][j] * inv[j][c];
      }
    }
    __syncthreads();
  }
  if (fail) { if (t == 0) *info = k0 + fail; return; }
  double* Lk = Linv + (size_t)(k0 / PVLM_CHOL_NB) * PVLM_CHOL_NB * PVLM_CHOL_NB;
  for (int e = t; e < PVLM_CHOL_NB * PVLM_CHOL_NB; e += 256) {
    const int i = e / PVLM_CHOL_NB, j = e % PVLM_CHOL_NB;
    if (i < kb && j <= i) M[(size_t)(k0 + i) * n + k0 + j] = a[i][j];
    Lk[e] = inv[i][j];
  }
}

// panel below the diagonal block: X = A L_kk^-T, i.e. X[row][c] = sum_{d <= c} A[row][d] Linv[c][d] — a small product,
// 8 rows x 32 columns per workgroup, no dependency chain (the first version solved the triangular system per row:
// 25 us per block column; a fully unrolled register version of that needed 512 registers + scratch and mis-behaved).
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ M, int n, int k0, int kb, const double* __restrict__ Linv,
                                                    const int* __restrict__ info) {
  __shared__ double inv[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ double As[8][PVLM_CHOL_NB + 1];
  if (*info != 0) return;
  const int t = threadIdx.x;
  const double* Lk = Linv + (size_t)(k0 / PVLM_CHOL_NB) * PVLM_CHOL_NB * PVLM_CHOL_NB;
  for (int e = t; e < PVLM_CHOL_NB * PVLM_CHOL_NB; e += 256) inv[e / PVLM_CHOL_NB][e % PVLM_CHOL_NB] = Lk[e];
  const int lr = t / PVLM_CHOL_NB, c = t % PVLM_CHOL_NB;
  const int row = k0 + kb + blockIdx.x * 8 + lr;
  const bool live = row < n && c < kb;
  As[lr][c] = live ? M[(size_t)row * n + k0 + c] : 0.0;
  __syncthreads();
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {
    double x = 0.0;
    for (int d = 0; d <= c; ++d) x += As[8 * sub + lr][d] * inv[c][d];
    if (lives[sub]) M[(size_t)rows[sub] * n + k0 + c] = x;
  }
}
#endif  // PVLM_MEASURED_VARIANTS

// MEASURED VARIANT (-DPVLM_CHOL_CLOCK=1): where workgroup 0 of k_chol_diag_panel spends its time, 100 MHz wall clock, summed over the launches
// of a solve and printed by chol_factor_solve: [0] entry -> block + rows in LDS, [1] the 32-pivot chain + inverse, [2] stores + panel rows, [3] launches.
#ifndef PVLM_CHOL_CLOCK
#define PVLM_CHOL_CLOCK 0
#endif
#if PVLM_CHOL_CLOCK
__device__ unsigned long long g_chol_clock[4];
#define CHOL_NOW() wall_clock64()
#define CHOL_ADD(i, v) do { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_chol_clock[i], (unsigned long long)(v)); } while (0)
#else
#define CHOL_NOW() 0ull
#define CHOL_ADD(i, v) do { (void)(v); } while (0)
#endif
__device__ __forceinline__ double bcast_f64(double v, int src_lane) {     // v of lane src_lane (wave-uniform source) for every lane
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// Diagonal block + panel in ONE launch (round 2): every workgroup of the panel factorises the 32 x 32 diagonal block itself —
// redundantly, from the same input — and then multiplies its eight panel rows by the inverse it has just computed.  The
// factorisation is a dependent chain (32 pivots) that takes as long in 600 workgroups side by side as in one, so the
// separate k_chol_diag launch (24 us + a launch gap per block column, 171 block columns at n = 5442) disappears.  The chain
// itself is run by ONE wave out of registers: lane i owns row i, pivots and column entries cross lanes by v_readlane, then
// L^-1 column by column (lane c solves L x = e_c).  (A first version kept the block in LDS and paid two LDS round trips per
// update element: 50 us per block column, slower than the two launches it replaced.)  Workgroup 0 stores the inverse for the triangular solves.  M's diagonal block keeps A's
// values (nothing downstream reads L_kk: the update uses the panel, the solves use the inverses).
// fwd_b != nullptr: the forward substitution of ONE right-hand side rides along — workgroup 0 also forms y_k = L_kk^-1 b_k
// (b_k is final: every earlier block column's update has been applied), and the trailing-update launch of this block column
// subtracts L[j, k] y_k from the rows below (k_chol_fwd_rows): the 171 k_fwd_step launches of round 1 disappear.
// row_tiles != nullptr (the tile-sparse factorisation below): workgroup g handles rows 8 (g & 7) .. + 8 of the 64-row tile
// row_tiles[g >> 3] — only the tiles that hold a nonzero of this block column — instead of the g-th group of 8 rows below the block.
// SUBS: groups of 8 panel rows a workgroup multiplies by the inverse (1: eight rows — the fewest rows behind the chain, right for a handful of workgroups; 8: a whole
// 64-row tile — an eighth of the workgroups, each of which repeats the 32-pivot chain: a level of forty leaf columns has 13 000 eight-row groups)
template <int SUBS>
__device__ __forceinline__ void chol_diag_panel_body(double* __restrict__ M, int n, int k0, int kb, double* __restrict__ Linv, int* __restrict__ info,
                                                     int* __restrict__ fail_out, const double* __restrict__ fwd_b, double* __restrict__ fwd_y,
                                                     const int* __restrict__ row_tiles, int n_row_tiles, const unsigned wg) {
  __shared__ double a[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ double inv[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ double As[8 * SUBS][PVLM_CHOL_NB + 1];
  __shared__ int fail;
  if (*info != 0) return;
  const unsigned long long ck0 = CHOL_NOW();
  const int t = threadIdx.x;
  if (t == 0) fail = 0;
  {
    constexpr int kPer = PVLM_CHOL_NB * PVLM_CHOL_NB / 256;           // reads first (clamped addresses), LDS writes after: see k_chol_update_mfma
    double av[kPer];
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int e = t + 256 * it, i = min(e / PVLM_CHOL_NB, kb - 1), j = min(e % PVLM_CHOL_NB, kb - 1);
      av[it] = M[(size_t)(k0 + i) * n + k0 + j];
    }
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int e = t + 256 * it, i = e / PVLM_CHOL_NB, j = e % PVLM_CHOL_NB;
      a[i][j] = (i < kb && j <= i) ? av[it] : 0.0;
      inv[i][j] = 0.0;
    }
  }
  const int lr = t / PVLM_CHOL_NB, c = t % PVLM_CHOL_NB;
  int rows[SUBS]; bool lives[SUBS];
  {
    double a_row[SUBS];
#pragma unroll
    for (int sub = 0; sub < SUBS; ++sub) {
      const unsigned unit = wg * SUBS + sub;                 // which group of 8 rows
      int row = k0 + kb + unit * 8 + lr;
      if (row_tiles) row = (int)(unit >> 3) < n_row_tiles ? row_tiles[unit >> 3] * 64 + (int)(unit & 7) * 8 + lr : n;
      rows[sub] = row; lives[sub] = row >= k0 + kb && row < n && c < kb;
      a_row[sub] = M[(size_t)min(max(row, 0), n - 1) * n + k0 + min(c, kb - 1)];
    }
#pragma unroll
    for (int sub = 0; sub < SUBS; ++sub) As[8 * sub + lr][c] = lives[sub] ? a_row[sub] : 0.0;
  }
  __syncthreads();
  const unsigned long long ck1 = CHOL_NOW();
  if (t < 64) {
    // One wave, no LDS and no barrier inside the chain: lane i keeps ROW i of the block in registers (fully unrolled, so
    // every register index is a compile-time constant); the pivot and the column entries l_cj a lane needs from another
    // lane travel by v_readlane with a constant source lane.
    const int i = t & (PVLM_CHOL_NB - 1);
    double r[PVLM_CHOL_NB], x[PVLM_CHOL_NB];
#pragma unroll
    for (int cc = 0; cc < PVLM_CHOL_NB; ++cc) { r[cc] = a[i][cc]; x[cc] = 0.0; }
    int bad = 0;
    double rdiag[PVLM_CHOL_NB];                               // 1 / l_jj (wave-uniform), reused by the inverse below
#pragma unroll
    for (int j = 0; j < PVLM_CHOL_NB; ++j) rdiag[j] = 0.0;
#pragma unroll
    for (int j = 0; j < PVLM_CHOL_NB; ++j) {
      if (j < kb && !bad) {                                   // wave-uniform
        const double d = bcast_f64(r[j], j);
        if (!(d > 0.0)) bad = j + 1;
        else {
          // 1 / sqrt(d) by v_rsq_f64 + three Newton steps (12 dependent instructions) instead of sqrt and a division per pivot (~45, a fifth
          // of the chain's time: the chain is ONE wave's dependent instructions, 23.6 us per block column measured with -DPVLM_CHOL_CLOCK=1);
          // l_jj = d y, l_ij = a_ij y: last-bit differences against sqrt-and-divide, as any other summation order gives
          double y = __builtin_amdgcn_rsq(d);
          y = y * (1.5 - 0.5 * d * y * y);
          y = y * (1.5 - 0.5 * d * y * y);
          y = y * (1.5 - 0.5 * d * y * y);
          rdiag[j] = y;
          const double sd = d * y;
          const double lij = r[j] * y;                        // meaningful in the lanes below the pivot
#pragma unroll
          for (int cc = j + 1; cc < PVLM_CHOL_NB; ++cc) {
            const double lcj = bcast_f64(lij, cc);
            r[cc] -= lij * lcj;       // every lane, no mask: entries above the diagonal (cc > i, and all of the lanes i <= j) turn into
          }                           // values nobody reads — a per-step exec mask cost more than the update itself (50 cycles per step)
          r[j] = lij;                 // lane j: d y = l_jj; lanes above the pivot: unused
          (void)sd;
        }
      }
    }
    if (bad) { if (t == 0) fail = bad; }
    else {
      // L^-1 column by column: lane c solves L x = e_c, x_q = -(sum_{k = c}^{q-1} l_qk x_k) / l_qq; l_qk lives in lane q
#pragma unroll
      for (int q = 0; q < PVLM_CHOL_NB; ++q) {
        if (q < kb) {                                         // wave-uniform
          double sacc = 0.0;
#pragma unroll
          for (int k = 0; k < q; ++k) {
            const double lqk = bcast_f64(r[k], q);
            sacc += lqk * x[k];       // x[k] = 0 for k < i: no mask needed
          }
          const double rq = rdiag[q];
          x[q] = q == i ? rq : (q > i ? -sacc * rq : 0.0);
        }
      }
      if (t < PVLM_CHOL_NB && i < kb) {
#pragma unroll
        for (int q = 0; q < PVLM_CHOL_NB; ++q) inv[q][i] = x[q];
      }
    }
  }
  __syncthreads();
  const unsigned long long ck2 = CHOL_NOW();
  if (fail) { if (t == 0 && wg == 0) { *fail_out = k0 + fail; } return; }
  if (wg == 0) {
    double* Lk = Linv + (size_t)(k0 / PVLM_CHOL_NB) * PVLM_CHOL_NB * PVLM_CHOL_NB;
    for (int e = t; e < PVLM_CHOL_NB * PVLM_CHOL_NB; e += 256) Lk[e] = inv[e / PVLM_CHOL_NB][e % PVLM_CHOL_NB];
    if (fwd_b && t < kb) {
      double sacc = 0.0;
      for (int d = 0; d <= t; ++d) sacc += inv[t][d] * fwd_b[k0 + d];
      fwd_y[k0 + t] = sacc;
    }
  }
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {
    double x = 0.0;
    for (int d = 0; d <= c; ++d) x += As[8 * sub + lr][d] * inv[c][d];
    if (lives[sub]) M[(size_t)rows[sub] * n + k0 + c] = x;
  }
  CHOL_ADD(0, ck1 - ck0); CHOL_ADD(1, ck2 - ck1); CHOL_ADD(2, CHOL_NOW() - ck2); CHOL_ADD(3, 1);
}

__global__ __launch_bounds__(256) void k_chol_diag_panel(double* __restrict__ M, int n, int k0, int kb, double* __restrict__ Linv, int* __restrict__ info,
                                                         int* __restrict__ fail_out, const double* __restrict__ fwd_b, double* __restrict__ fwd_y,
                                                         const int* __restrict__ row_tiles, int n_row_tiles) {
  chol_diag_panel_body<1>(M, n, k0, kb, Linv, info, fail_out, fwd_b, fwd_y, row_tiles, n_row_tiles, blockIdx.x);
}
// Level schedule (csrc/pvlm_spd_plan.h: plan_levels): every block column of a level in ONE launch — workgroup g works for block column groups[g].x as its
// groups[g].y-th workgroup.  A failed pivot is recorded by the first workgroup of its column (several columns may fail: any of them is a valid report).
template <int SUBS>
__global__ __launch_bounds__(256) void k_nd_panel(double* __restrict__ M, int n, double* __restrict__ Linv, int* __restrict__ info, const double* __restrict__ fwd_b,
                                                  double* __restrict__ fwd_y, const int2* __restrict__ groups, const int* __restrict__ row_off, const int* __restrict__ row_tiles) {
  const int2 g = groups[blockIdx.x];
  const int k0 = g.x * PVLM_CHOL_NB;
  chol_diag_panel_body<SUBS>(M, n, k0, min(PVLM_CHOL_NB, n - k0), Linv, info, info, fwd_b, fwd_y, row_tiles + row_off[g.x], row_off[g.x + 1] - row_off[g.x], (unsigned)g.y);
}

// Which 64 x 64 tile (ti >= tj) of the trailing lower triangle a workgroup updates.  part 0: all tiles, linear id -> lower
// triangle; part 1: the first tile column only (tj = 0: the columns the NEXT block column's factorisation needs);
// part 2: the rest (tj >= 1) — parts 1 and 2 run on two streams (look-ahead, chol_factor_solve).
__device__ __forceinline__ bool chol_tile_of(int id, int tiles, int part, int* ti_out, int* tj_out) {
  if (part == 1) { if (id >= tiles) return false; *ti_out = id; *tj_out = 0; return true; }
  const int shift = part == 2 ? 1 : 0, t = tiles - shift;
  int ti = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
  while ((long long)(ti + 1) * (ti + 2) / 2 <= (long long)id) ++ti;
  while ((long long)ti * (ti + 1) / 2 > (long long)id) --ti;
  const int tj = id - ti * (ti + 1) / 2;
  if (ti >= t) return false;
  *ti_out = ti + shift; *tj_out = tj + shift;
  return true;
}

// b[i] -= L[i, k-block] . y_k for the rows below block column k (the second half of round 1's k_fwd_step), run by the
// workgroups past the tile list of the trailing-update launches
__device__ __forceinline__ void chol_fwd_rows(const double* __restrict__ M, int n, int k0, int kb, int chunk, double* __restrict__ b, const double* __restrict__ yv) {
  __shared__ double y[PVLM_CHOL_NB];
  const int t = threadIdx.x;
  if (t < PVLM_CHOL_NB) y[t] = t < kb ? yv[k0 + t] : 0.0;
  __syncthreads();
  const int i = k0 + kb + chunk * 256 + t;
  if (i >= n) return;
  const double* r = M + (size_t)i * n + k0;
  double sacc = 0.0;
  for (int c = 0; c < kb; ++c) sacc += r[c] * y[c];
  b[i] -= sacc;
}

#if PVLM_MEASURED_VARIANTS
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ M, int n, int k0, int kb, int tiles, const int* __restrict__ info,
                                                     int n_tile_blocks, double* __restrict__ fwd_b, const double* __restrict__ fwd_y, int part) {
  if ((int)blockIdx.x >= n_tile_blocks) { if (*info == 0) chol_fwd_rows(M, n, k0, kb, (int)blockIdx.x - n_tile_blocks, fwd_b, fwd_y); return; }
  __shared__ double As[64][PVLM_CHOL_NB + 1];
  __shared__ double Bs[64][PVLM_CHOL_NB + 1];
  if (*info != 0) return;
  // linear block id -> (ti >= tj) of the lower triangle of the tile grid
  int ti, tj;
  if (!chol_tile_of((int)blockIdx.x, tiles, part, &ti, &tj)) return;
  const int base = k0 + kb, r0 = base + ti * 64, c0 = base + tj * 64;
  for (int e = threadIdx.x; e < 64 * PVLM_CHOL_NB; e += 256) {
    const int i = e / PVLM_CHOL_NB, c = e % PVLM_CHOL_NB;
    As[i][c] = (r0 + i < n && c < kb) ? M[(size_t)(r0 + i) * n + k0 + c] : 0.0;
    Bs[i][c] = (c0 + i < n && c < kb) ? M[(size_t)(c0 + i) * n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
  double acc[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 8
  for (int c = 0; c < PVLM_CHOL_NB; ++c) {
    double av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { av[q] = As[ty * 4 + q][c]; bv[q] = Bs[tx * 4 + q][c]; }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[p][q] += av[p] * bv[q];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = r0 + ty * 4 + p;
    if (row >= n) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = c0 + tx * 4 + q;
      if (col <= row) M[(size_t)row * n + col] -= acc[p][q];
    }
  }
}

#endif  // PVLM_MEASURED_VARIANTS

// The same rank-32 trailing update on the matrix core: v_mfma_f64_16x16x4_f64 (the one GEMM-shaped kernel of this
// library).  A 64 x 64 tile per workgroup, wave w owns the 16-row band [16w, 16w + 16) and all four 16-column tiles:
// per k-step of 4 one A fragment (lane l: A[l & 15][k + (l >> 4)]) and four B fragments (B[k + (l >> 4)][l & 15] =
// panel row of the column tile) from LDS, four MFMAs.  D layout of the f64 form: col = lane & 15,
// row = (lane >> 4) + 4 * reg (cdna_hip_programming.md §3) — not the f32 map.
typedef double pvlm_d4 __attribute__((ext_vector_type(4)));
// pairs != nullptr (tile-sparse factorisation): workgroup g updates the tile pairs[g] = (ti, tj) in ABSOLUTE 64-row tiles of the
// matrix — only the pairs whose two row tiles hold a nonzero of this block column; rows / columns above `base` are masked.
__global__ __launch_bounds__(256) void k_chol_update_mfma(double* __restrict__ M, int n, int k0, int kb, int tiles, const int* __restrict__ info,
                                                          int n_tile_blocks, double* __restrict__ fwd_b, const double* __restrict__ fwd_y, int part,
                                                          const int2* __restrict__ pairs) {
  __shared__ double As[64][PVLM_CHOL_NB + 1];
  __shared__ double Bs[64][PVLM_CHOL_NB + 1];
  if (*info != 0) return;
  if ((int)blockIdx.x >= n_tile_blocks) { chol_fwd_rows(M, n, k0, kb, (int)blockIdx.x - n_tile_blocks, fwd_b, fwd_y); return; }
  int ti, tj;
  const int base = k0 + kb;
  int r0, c0;
  if (pairs) { const int2 pr = pairs[blockIdx.x]; r0 = pr.x * 64; c0 = pr.y * 64; }
  else { if (!chol_tile_of((int)blockIdx.x, tiles, part, &ti, &tj)) return; r0 = base + ti * 64; c0 = base + tj * 64; }
  {
    // the panel slices: every global read first, from clamped (always valid) addresses, then the LDS writes — `cond ? M[..] : 0` per element
    // compiled into one load -> wait -> LDS write after the other (sixteen dependent round trips per thread)
    constexpr int kPer = 64 * PVLM_CHOL_NB / 256;
    double av[kPer], bv[kPer];
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int e = threadIdx.x + 256 * it, i = e / PVLM_CHOL_NB, c = min(e % PVLM_CHOL_NB, kb - 1);
      av[it] = M[(size_t)min(max(r0 + i, 0), n - 1) * n + k0 + c];
      bv[it] = M[(size_t)min(max(c0 + i, 0), n - 1) * n + k0 + c];
    }
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int e = threadIdx.x + 256 * it, i = e / PVLM_CHOL_NB, c = e % PVLM_CHOL_NB;
      As[i][c] = (r0 + i >= base && r0 + i < n && c < kb) ? av[it] : 0.0;
      Bs[i][c] = (c0 + i >= base && c0 + i < n && c < kb) ? bv[it] : 0.0;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  pvlm_d4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < PVLM_CHOL_NB; k += 4) {
    const double a = As[16 * w + li][k + lk];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[16 * t + li][k + lk], acc[t], 0, 0, 0);
  }
  // C -= acc: the 16 reads first, all in flight together, from clamped (always valid) addresses, then the guarded stores.  Written as
  // `if (...) M[..] -= acc` the compiler emitted sixteen load -> wait -> subtract -> store sequences, each under its own exec mask:
  // sixteen dependent memory round trips per workgroup — most of the kernel's 21 us average on the tile-sparse Floor system.
  double cv[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = min(c0 + 16 * t + li, n - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(r0 + 16 * w + lk + 4 * r, n - 1);
      cv[t][r] = M[(size_t)row * n + col];
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = c0 + 16 * t + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 16 * w + lk + 4 * r;
      if (row < n && col <= row && col >= base) M[(size_t)row * n + col] = cv[t][r] - acc[t][r];
    }
  }
}

// Trailing updates of ONE LEVEL of the schedule, target-centric: workgroup g owns the tile pair targets[g] and subtracts the rank-32 products of every
// source block column of the level that reaches it, in list order, from registers — one read-modify-write of the tile whatever the number of sources, no two
// workgroups on the same tile, no atomics: the factor is bit-reproducible.  Workgroups past the tile list do the same for 64 rows of the right-hand side
// (the forward substitution rides along: y of a level's columns comes out of its panel launch).
struct NdTarget { int ti, tj, src_off, n_src, col_min, pad; };
struct NdRowTarget { int tile, src_off, n_src, pad; };
__global__ __launch_bounds__(256) void k_nd_update(double* __restrict__ M, int n, const int* __restrict__ info, const NdTarget* __restrict__ targets, int n_targets,
                                                   const int* __restrict__ sources, const NdRowTarget* __restrict__ row_targets, const int* __restrict__ row_sources,
                                                   double* __restrict__ fwd_b, const double* __restrict__ fwd_y) {
  __shared__ double As[64][PVLM_CHOL_NB + 1];
  __shared__ double Bs[64][PVLM_CHOL_NB + 1];
  if (*info != 0) return;
  if ((int)blockIdx.x >= n_targets) {
    // b[rows of the tile] -= L[rows, block column k] y_k for the sources of the row tile: four threads per row, eight columns each, in source order
    const NdRowTarget rt = row_targets[(int)blockIdx.x - n_targets];
    double* ys = &As[0][0];
    const int r = threadIdx.x >> 2, part = threadIdx.x & 3, row = rt.tile * 64 + r;
    double acc = 0.0;
    for (int q = 0; q < rt.n_src; ++q) {
      const int k0 = row_sources[rt.src_off + q] * PVLM_CHOL_NB, kb = min(PVLM_CHOL_NB, n - k0);
      __syncthreads();
      if (threadIdx.x < PVLM_CHOL_NB) ys[threadIdx.x] = (int)threadIdx.x < kb ? fwd_y[k0 + threadIdx.x] : 0.0;
      __syncthreads();
      if (row < n && row >= k0 + kb) {
        const double* m = M + (size_t)row * n + k0 + 8 * part;
        double sacc = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) sacc += (8 * part + c < kb ? m[c] : 0.0) * ys[8 * part + c];
        acc += sacc;
      }
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
    if (part == 0 && row < n) fwd_b[row] -= acc;
    return;
  }
  const NdTarget tg = targets[blockIdx.x];
  const int r0 = tg.ti * 64, c0 = tg.tj * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  pvlm_d4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
  for (int q = 0; q < tg.n_src; ++q) {
    const int k0 = sources[tg.src_off + q] * PVLM_CHOL_NB, kb = min(PVLM_CHOL_NB, n - k0), base = k0 + kb;
    constexpr int kPer = 64 * PVLM_CHOL_NB / 256;
    double av[kPer], bv[kPer];
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int e = threadIdx.x + 256 * it, i = e / PVLM_CHOL_NB, c = min(e % PVLM_CHOL_NB, kb - 1);
      av[it] = M[(size_t)min(r0 + i, n - 1) * n + k0 + c];
      bv[it] = M[(size_t)min(c0 + i, n - 1) * n + k0 + c];
    }
    if (q) __syncthreads();                                   // the previous source's slices have been consumed
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int e = threadIdx.x + 256 * it, i = e / PVLM_CHOL_NB, c = e % PVLM_CHOL_NB;
      As[i][c] = (r0 + i >= base && r0 + i < n && c < kb) ? av[it] : 0.0;
      Bs[i][c] = (c0 + i >= base && c0 + i < n && c < kb) ? bv[it] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PVLM_CHOL_NB; k += 4) {
      const double a = As[16 * w + li][k + lk];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[16 * t + li][k + lk], acc[t], 0, 0, 0);
    }
  }
  double cv[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = min(c0 + 16 * t + li, n - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(r0 + 16 * w + lk + 4 * r, n - 1);
      cv[t][r] = M[(size_t)row * n + col];
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = c0 + 16 * t + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 16 * w + lk + 4 * r;
      if (row < n && col <= row && col >= tg.col_min) M[(size_t)row * n + col] = cv[t][r] - acc[t][r];
    }
  }
}

// Backward substitution of ONE LEVEL (levels in descending order): workgroup g solves block column cols[g] in gather form —
//   x_k = L_kk^-T (y_k - sum over the rows i of the column's panel tiles of L[i, k]^T x_i);
// every such row belongs to a block column of a higher level, whose x is final.  1024 threads: thread (rg, c) takes rows rg and rg + 32 of every tile for column
// c, eight tiles' loads in flight together (a leaf column has a hundred tiles below it — its own group and every separator above: eight rows per round trip took 35 us
// per level).  (Measured and removed: a workgroup per chunk of eight tiles with the last one to arrive adding the chunks up in order — the fences and the counter cost
// the 8 us the shorter loop saves: 1.83 against 1.75 ms for the 108 levels of the Floor system.)
__global__ __launch_bounds__(1024) void k_nd_bwd(const double* __restrict__ M, int n, const double* __restrict__ Linv, double* __restrict__ b, const double* __restrict__ yv,
                                                 const int* __restrict__ info, const int* __restrict__ cols, const int* __restrict__ row_off, const int* __restrict__ row_tiles) {
  __shared__ double part[32][PVLM_CHOL_NB + 1];
  __shared__ double inv[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ double v[PVLM_CHOL_NB];
  if (*info != 0) return;
  const int kc = cols[blockIdx.x], k0 = kc * PVLM_CHOL_NB, kb = min(PVLM_CHOL_NB, n - k0), base = k0 + kb;
  const int t = threadIdx.x, c = t % PVLM_CHOL_NB, rg = t / PVLM_CHOL_NB;
  const double* Lk = Linv + (size_t)kc * PVLM_CHOL_NB * PVLM_CHOL_NB;
  inv[t / PVLM_CHOL_NB][t % PVLM_CHOL_NB] = Lk[t];
  double acc = 0.0;
  const int* rt = row_tiles + row_off[kc]; const int nrt = row_off[kc + 1] - row_off[kc];
  const int cc = k0 + min(c, kb - 1);
  for (int a = 0; a < nrt; a += 8) {
    double m[16], xr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ai = a + (q >> 1);
      const int row = ai < nrt ? rt[ai] * 64 + rg + 32 * (q & 1) : n, rc = min(row, n - 1);
      m[q] = M[(size_t)rc * n + cc];
      xr[q] = (row >= base && row < n) ? b[rc] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += m[q] * xr[q];
  }
  part[rg][c] = c < kb ? acc : 0.0;
  __syncthreads();
  if (t < PVLM_CHOL_NB) {
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) sacc += part[q][t];
    v[t] = t < kb ? yv[k0 + t] - sacc : 0.0;
  }
  __syncthreads();
  if (t < kb) {
    double sacc = 0.0;
    for (int d = t; d < PVLM_CHOL_NB; ++d) sacc += inv[d][t] * v[d];
    b[k0 + t] = sacc;
  }
}

// ---- the dense tail of a level schedule: ONE launch for the top separator ---------------------------------------------------------------------
// The last group of a nested dissection (the top separator: 247 poses = 1 482 rows of the Floor graph) is dense once everything below it has been
// eliminated, and every one of its block columns is a level of its own: 46 of the 108 levels, 46 x (panel + update + backward) launches that each wait for the
// one before (2.1 of the 6.3 ms of kernel time of a solve, profiles/r6_spd_levels_kernel_stats.csv).  k_nd_tail factorises that trailing block as a TILE
// Cholesky in one launch: a workgroup per 64 x 64 tile (i, j), handed out by a ticket in column-major order (so the lowest unfinished tile always runs — no
// residency assumption); it subtracts L(i,k) L(j,k)^T for k < j as the tiles of the row panels are PUBLISHED, then
//   i == j: factorises the 64 x 64 block (two passes of the 32-pivot register chain of chol_diag_panel_body with the rank-32 update between them), inverts it,
//           publishes the inverse, and forms y_j = L_jj^-1 (b_j - sum_k L(j,k) y_k) — the forward substitution rides along;
//   i >  j: waits for that inverse, multiplies, publishes L(i,j).
// The dependent chain is 64 pivots + two hand-offs per 64 columns instead of six launches.  Hand-offs follow cdna_hip_programming.md Guideline 16: payload stored
// write-through by agent-scope stores, every storing wave drains, ONE lane stores the flag; consumers poll the flag relaxed and read the payload with agent-scope loads.
// Every sum has a fixed order: the factor is bit-reproducible.  k_nd_tail_bwd is the backward substitution of the same block, a workgroup per tile column.
typedef __attribute__((address_space(1))) unsigned pvlm_gu32;
typedef __attribute__((address_space(1))) unsigned long long pvlm_gu64;
__device__ __forceinline__ double tail_ld(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((pvlm_gu64*)(unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void tail_st(double* p, double v) {
  __hip_atomic_store((pvlm_gu64*)(unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// thread 0 of the workgroup: waits until *flag != 0; false when the solve has failed elsewhere (info != 0) or nothing came for 2 s of wall clock (info := -1: the host redoes the solve with the level launches).
// A launch that has the GPU to itself waits for milliseconds at most (the work of the tasks before this one).  With 8 processes factorising on ONE GPU at the same
// time about 1 solve in 300 did run into the limit on some boxes of the pool and on others never (limits of 1 s and of 20 s alike: a task that had its ticket never
// published; not reproduced with one process) — workgroups that wait hold the compute units the ones that work need, and the one-launch form is not meant for a
// shared GPU: pvlm_spd_one_launch(ctx, 0) selects the level launches, and a solve that meets the limit is redone with them by itself.
// Long waits back off (s_sleep up to ~3 us) so that the workgroups that wait leave the memory system to the ones that work.
__device__ unsigned g_wg_state[1024 * 2];     // debug: per workgroup of k_nd_flow {task, phase | source index << 8}
#define WG_STATE(task, phase) do { if (LISTS && threadIdx.x == 0 && blockIdx.x < 1024) { g_wg_state[2 * blockIdx.x] = (unsigned)(task); g_wg_state[2 * blockIdx.x + 1] = (unsigned)(phase); } } while (0)
__device__ unsigned g_tail_timeout[8];        // the first wait of a process that ran into the limit: [0] marker, [1] site, [2] own task / ticket, [3] awaited flag word, [4..5] the two ticket counters, [6] workgroup
__device__ __forceinline__ bool tail_wait(unsigned* flag, int* info, const unsigned* flags = nullptr, int site = 0, int own = 0) {
  unsigned spins = 0;
  unsigned long long t0 = 0;
  while (__hip_atomic_load((pvlm_gu32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    if (spins < 256u) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(127);
    if ((++spins & 63u) == 0u) {
      if (__hip_atomic_load((pvlm_gu32*)(unsigned*)info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      const unsigned long long now = wall_clock64();              // 100 MHz
      if (t0 == 0) t0 = now;
      else if (now - t0 > 200000000ull) {
        if (flags && atomicCAS(&g_tail_timeout[0], 0u, 1u) == 0u) {
          g_tail_timeout[1] = (unsigned)site; g_tail_timeout[2] = (unsigned)own; g_tail_timeout[3] = (unsigned)(flag - flags);
          g_tail_timeout[4] = __hip_atomic_load((pvlm_gu32*)flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g_tail_timeout[5] = __hip_atomic_load((pvlm_gu32*)(flags + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g_tail_timeout[6] = blockIdx.x;
        }
        __hip_atomic_store((pvlm_gu32*)(unsigned*)info, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  return true;
}
// every storing wave has drained its stores (the caller's __syncthreads() follows the drain); ONE lane raises the flag
__device__ __forceinline__ void tail_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tail_raise(unsigned* flag) { __hip_atomic_store((pvlm_gu32*)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static size_t tail_flag_words(int T) { return (size_t)2 + (size_t)T * T + 3 * (size_t)T; }
// lane q of every row of 16 lanes, for all lanes of that row (DPP row_newbcast: one v_mov_b64_dpp)
template <int Q> __device__ __forceinline__ double row_bcast_f64_c(double v) {
  return __longlong_as_double(__builtin_amdgcn_update_dpp(0ll, __double_as_longlong(v), 0x150 + Q, 0xf, 0xf, false));
}
__device__ __forceinline__ double row_bcast_f64(double v, int q) {      // q: a constant after unrolling
  switch (q) {
    case 0: return row_bcast_f64_c<0>(v); case 1: return row_bcast_f64_c<1>(v); case 2: return row_bcast_f64_c<2>(v); case 3: return row_bcast_f64_c<3>(v);
    case 4: return row_bcast_f64_c<4>(v); case 5: return row_bcast_f64_c<5>(v); case 6: return row_bcast_f64_c<6>(v); case 7: return row_bcast_f64_c<7>(v);
    case 8: return row_bcast_f64_c<8>(v); case 9: return row_bcast_f64_c<9>(v); case 10: return row_bcast_f64_c<10>(v); case 11: return row_bcast_f64_c<11>(v);
    case 12: return row_bcast_f64_c<12>(v); case 13: return row_bcast_f64_c<13>(v); case 14: return row_bcast_f64_c<14>(v); default: return row_bcast_f64_c<15>(v);
  }
}
#define PVLM_TAIL_LD 65     // row stride (doubles) of the 64 x 64 tiles in LDS
// flags: [0] ticket, [1] ticket of the backward launch, [2 + i * T + j] tile (i, j) published, then T words each: inverse of column j, y_j, x_j
// LISTS (k_nd_flow): the WHOLE factorisation in this form — the tiles are the tasks of pvlm_spd::plan_flow (tile row, tile column, sources = the earlier tile columns
// that hold both tiles, with the tasks that publish them), r0 = 0, T = tile columns of the padded system; tile flags are indexed by task.
// Two workgroups of the one-launch factorisation per CU: the inverse of a diagonal tile lives where the second source tile was (72 KB of LDS instead of 105) and the
// register budget is that of __launch_bounds__(256, 2) (256 registers, 68 bytes of scratch).  The first half of the launch is bound by the traffic of its source tiles
// and twice the workgroups keep twice the loads in flight: 3.52 -> 3.28 ms per Floor solve.  -DPVLM_FLOW_OCC2=0 (python -m panovlm_amd.build --variant occ1
// -DPVLM_FLOW_OCC2=0) is the one-workgroup form, kept for the A/B.
#ifndef PVLM_FLOW_OCC2
#define PVLM_FLOW_OCC2 1
#endif
struct NdFlowTask { int I, J, src_off, n_src, prev, final_, pad0, pad1; };   // pvlm_spd::FlowTask
struct NdFlowSource { int K, task_a, task_b, pad; };
template <bool LISTS>
__device__ __forceinline__ void nd_tile_flow(double* __restrict__ M, int n, int r0, int T, double* __restrict__ inv64, unsigned* __restrict__ flags, int* __restrict__ info,
                                             double* __restrict__ b, double* __restrict__ yv, unsigned long long* __restrict__ clk,
                                             const NdFlowTask* __restrict__ tasks, const NdFlowSource* __restrict__ sources, int n_tasks, int withhold = -1) {
  // clk != nullptr (PVLM_SPD_TAIL_CLOCK=1): 100 MHz wall-clock stamps of the dependent chain, twelve per tile column — the diagonal tile: [0] ticket taken, [1] last
  // dependency seen, [2] products done, [3] tile in LDS, [4] factor + inverse done, [5] inverse published, [8] first 32 pivots, [9] rank-32 update, [10] last 32
  // pivots; the tile below it: [6] inverse seen, [7] tile published
  __shared__ double lds[2 * 64 * PVLM_TAIL_LD];                        // the k loop: As | Bs (64 x 65 each); afterwards As is the tile itself
#if PVLM_FLOW_OCC2
  double* Is = lds + 64 * PVLM_TAIL_LD;                                // the inverse of the diagonal block where Bs was (free after the source loop): 72 KB, two workgroups per CU
#else
  __shared__ double Is[64 * PVLM_TAIL_LD];                             // the inverse of the diagonal block
#endif
  __shared__ double ys[64], vs[64], rds[64];
  __shared__ double Ts[2][16][17];
  __shared__ int s_id, s_ok, s_fail;
  double (*As)[PVLM_TAIL_LD] = reinterpret_cast<double (*)[PVLM_TAIL_LD]>(lds);
  double (*Bs)[PVLM_TAIL_LD] = reinterpret_cast<double (*)[PVLM_TAIL_LD]>(lds + 64 * PVLM_TAIL_LD);
  double (*Cs)[PVLM_TAIL_LD] = As;
  double (*Iv)[PVLM_TAIL_LD] = reinterpret_cast<double (*)[PVLM_TAIL_LD]>(Is);
  unsigned* tile_flag = flags + 2; unsigned* inv_flag = tile_flag + (LISTS ? n_tasks : T * T); unsigned* y_flag = inv_flag + T;
  const int n_tiles = LISTS ? n_tasks : T * (T + 1) / 2;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 15, lk = lane >> 4;
  if (*info != 0) return;                                            // a level below has failed
  for (;;) {
    __syncthreads();
    if (t == 0) { s_id = (int)atomicAdd(flags, 1u); s_fail = 0; }
    __syncthreads();
    int id = s_id;
    if (id >= n_tiles) return;
    int i, j, n_src, src_off = 0, prev = -1;
    bool final_task = true;
    if (LISTS) { const NdFlowTask tk = tasks[id]; i = tk.I; j = tk.J; n_src = tk.n_src; src_off = tk.src_off; prev = tk.prev; final_task = tk.final_ != 0; }
    else {
      j = 0;
      while (id >= T - j) { id -= T - j; ++j; }                      // column-major: column j holds the tiles i = j .. T - 1
      i = j + id; n_src = j;
    }
    const int my_flag = LISTS ? s_id : i * T + j;
    WG_STATE(s_id, 1);
    const bool diag = i == j;
    const size_t row_i = (size_t)(r0 + 64 * i), row_j = (size_t)(r0 + 64 * j);
    const bool stamp = clk && t == 0;
    if (stamp && diag && final_task) clk[12 * j + 0] = wall_clock64();
    pvlm_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
    double bacc = 0.0;                                               // diagonal tiles, threads 0..63: (sum_k L(j,k) y_k)[t]
    // thread 0 looks at the flags of source q + 1 while source q is loaded and multiplied (two dependent round trips of ~1 us per source otherwise — a third of a
    // source step, and the launch is throughput-bound for its first half); a flag seen raised is final, one not yet raised is waited for as before
    unsigned seen_a = 0u, seen_b = 0u;
    for (int q_src = 0; q_src < n_src; ++q_src) {
      int k = q_src, flag_a = i * T + q_src, flag_b = j * T + q_src;
      if (LISTS) { const NdFlowSource sr = sources[src_off + q_src]; k = sr.K; flag_a = sr.task_a; flag_b = sr.task_b; }
      WG_STATE(s_id, 1 | (q_src << 8));
      if (t == 0) {
        bool ok = seen_a != 0u || tail_wait(tile_flag + flag_a, info, flags, 1, s_id);
        if (ok && seen_b == 0u) ok = tail_wait(diag ? y_flag + k : tile_flag + flag_b, info, flags, diag ? 2 : 3, s_id);
        s_ok = ok ? 1 : 0;
        if (stamp && diag && final_task && q_src == n_src - 1) clk[12 * j + 1] = wall_clock64();
        seen_a = seen_b = 0u;
        if (ok && q_src + 1 < n_src) {
          int k1 = q_src + 1, fa1 = i * T + q_src + 1, fb1 = j * T + q_src + 1;
          if (LISTS) { const NdFlowSource s1 = sources[src_off + q_src + 1]; k1 = s1.K; fa1 = s1.task_a; fb1 = s1.task_b; }
          seen_a = __hip_atomic_load((pvlm_gu32*)(tile_flag + fa1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          seen_b = __hip_atomic_load((pvlm_gu32*)(diag ? y_flag + k1 : tile_flag + fb1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      if (!s_ok) return;
      if (diag && t < 64) ys[t] = tail_ld(yv + r0 + 64 * k + t);
      {
        // both tiles whole (64 x 64 each), every load of the step in flight together: the last step of a diagonal tile is on the chain of dependent steps
        double av[16], bv[16];
        const size_t col = (size_t)(r0 + 64 * k);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int e = t + 256 * it, r = e >> 6, c = e & 63;
          av[it] = tail_ld(M + (row_i + r) * n + col + c);
          if (!diag) bv[it] = tail_ld(M + (row_j + r) * n + col + c);
        }
        __syncthreads();                                             // the previous tiles have been consumed (and ys is complete)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int e = t + 256 * it, r = e >> 6, c = e & 63;
          As[r][c] = av[it];
          if (!diag) Bs[r][c] = bv[it];
        }
        __syncthreads();
        double (*Bp)[PVLM_TAIL_LD] = diag ? As : Bs;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 4) {
          const double a = As[16 * w + li][kk + lk];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bp[16 * q + li][kk + lk], acc[q], 0, 0, 0);
        }
        if (diag && t < 64) {
          double sacc = 0.0;
#pragma unroll
          for (int c = 0; c < 64; ++c) sacc += As[t][c] * ys[c];
          bacc += sacc;
        }
      }
    }
    // the tile's own values (written by the launches before this one, or zero) minus the products
    if (stamp && diag && final_task) clk[12 * j + 2] = wall_clock64();
    double cv[4][4];
    if (LISTS) {
      // the tile may hold the partial sums of a chunk task (another workgroup, this launch): wait for the last of them, read through the agent-scope path
      WG_STATE(s_id, 2);
      if (prev >= 0) {
        if (t == 0) s_ok = tail_wait(tile_flag + prev, info, flags, 4, s_id) ? 1 : 0;
        __syncthreads();
        if (!s_ok) return;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[q][r] = tail_ld(M + (row_i + 16 * w + lk + 4 * r) * n + row_j + 16 * q + li);
      WG_STATE(s_id, 3);
      if (!final_task) {
        // a chunk: the tile in memory minus this chunk's products (and b_j minus its share of the forward substitution), published for the next task of the tile
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) tail_st(M + (row_i + 16 * w + lk + 4 * r) * n + row_j + 16 * q + li, cv[q][r] - acc[q][r]);
        if (diag && t < 64) tail_st(b + r0 + 64 * j + t, tail_ld(b + r0 + 64 * j + t) - bacc);
        tail_drain();
        __syncthreads();
        if (t == 0 && !(LISTS && s_id == withhold)) tail_raise(tile_flag + my_flag);
        continue;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[q][r] = M[(row_i + 16 * w + lk + 4 * r) * n + row_j + 16 * q + li];
    }
    __syncthreads();                                                 // the last slices have been consumed: lds becomes the tile
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * w + lk + 4 * r, cc = 16 * q + li;
        Cs[rr][cc] = (!diag || cc <= rr) ? cv[q][r] - acc[q][r] : 0.0;
      }
    if (!diag) {
      WG_STATE(s_id, 4);
      if (t == 0) s_ok = tail_wait(inv_flag + j, info, flags, 5, s_id) ? 1 : 0;
      if (stamp && (LISTS ? (s_id > 0 && tasks[s_id - 1].I == j && tasks[s_id - 1].J == j) : i == j + 1)) clk[12 * j + 6] = wall_clock64();
      __syncthreads();
      if (!s_ok) return;
      double iv[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) iv[it] = tail_ld(inv64 + (size_t)j * 4096 + t + 256 * it);
#pragma unroll
      for (int it = 0; it < 16; ++it) { const int e = t + 256 * it; Iv[e >> 6][e & 63] = iv[it]; }
      __syncthreads();
      // X = C L_jj^-T: X[r][c] = sum_d C[r][d] inv[c][d]
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 64; kk += 4) {
        const double a = Cs[16 * w + li][kk + lk];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Iv[16 * q + li][kk + lk], acc[q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) tail_st(M + (row_i + 16 * w + lk + 4 * r) * n + row_j + 16 * q + li, acc[q][r]);
      tail_drain();
      __syncthreads();
      if (t == 0 && !(LISTS && s_id == withhold)) tail_raise(tile_flag + my_flag);       // withhold: the test of the recovery path — this task never publishes
      WG_STATE(s_id, 9);
      if (stamp && (LISTS ? (s_id > 0 && tasks[s_id - 1].I == j && tasks[s_id - 1].J == j) : i == j + 1)) clk[12 * j + 7] = wall_clock64();
      continue;
    }
    // ---- the diagonal tile: Cs (lower triangle, zeros above) -> L and L^-1, 32 columns at a time
    WG_STATE(s_id, 6);
    __syncthreads();
    if (stamp) clk[12 * j + 3] = wall_clock64();
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
      if (t < 64) {
        // lane l keeps row l (blk 0: rows 0..63 — the lanes 32..63 come out as L21, the rows below the block) resp. row 32 + (l & 31) (blk 1) of the
        // 32 columns of this pass in registers; pivots and column entries cross lanes by v_readlane (see chol_diag_panel_body)
        const int i32 = t & (PVLM_CHOL_NB - 1);
        const int row = blk == 0 ? t : PVLM_CHOL_NB + i32;
        double r[PVLM_CHOL_NB];
#pragma unroll
        for (int cc = 0; cc < PVLM_CHOL_NB; ++cc) r[cc] = Cs[row][PVLM_CHOL_NB * blk + cc];
        int bad = 0;
        // kb = 32, but not to the compiler: the branch per step keeps the steps apart (without it the scheduler hoists the v_readlane of later steps ahead of their
        // sums, runs out of scalar registers and spills them to vector lanes: 25 us per pass instead of 9 when the inverse was still computed here)
        int kb = PVLM_CHOL_NB;
        asm volatile("" : "+s"(kb));
#pragma unroll
        for (int jj = 0; jj < PVLM_CHOL_NB; ++jj) {
          if (jj < kb && !bad) {
            const double d = bcast_f64(r[jj], jj);
            if (!(d > 0.0)) bad = jj + 1;
            else {
              double y = __builtin_amdgcn_rsq(d);
              y = y * (1.5 - 0.5 * d * y * y);
              y = y * (1.5 - 0.5 * d * y * y);
              y = y * (1.5 - 0.5 * d * y * y);
              if (t == 0) rds[PVLM_CHOL_NB * blk + jj] = y;     // 1 / l_jj for the inverse below
              const double lij = r[jj] * y;
#pragma unroll
              for (int cc = jj + 1; cc < PVLM_CHOL_NB; ++cc) r[cc] -= lij * bcast_f64(lij, cc);
              r[jj] = lij;
            }
          }
        }
        if (bad) { if (t == 0) s_fail = PVLM_CHOL_NB * blk + bad; }
        else if (blk == 0 || t < PVLM_CHOL_NB) {
          // L back into the tile (blk 0: rows 0..63 of the left half; blk 1: rows 32..63 of the right half)
#pragma unroll
          for (int cc = 0; cc < PVLM_CHOL_NB; ++cc) Cs[row][PVLM_CHOL_NB * blk + cc] = (blk == 0 ? (t >= PVLM_CHOL_NB || cc <= t) : cc <= i32) ? r[cc] : 0.0;
        }
      }
      __syncthreads();
      if (stamp) clk[12 * j + 8 + 2 * blk] = wall_clock64();
      if (s_fail) break;
      if (blk == 0) {
        // A22 -= L21 L21^T (lower triangle): a 16 x 16 block per wave on the matrix core
        const int bm = w >> 1, bn = w & 1;
        pvlm_d4 c4 = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < PVLM_CHOL_NB; kk += 4)
          c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cs[PVLM_CHOL_NB + 16 * bm + li][kk + lk], Cs[PVLM_CHOL_NB + 16 * bn + li][kk + lk], c4, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 16 * bm + lk + 4 * r, bcol = 16 * bn + li;
          if (bcol <= a) Cs[PVLM_CHOL_NB + a][PVLM_CHOL_NB + bcol] -= c4[r];
        }
        __syncthreads();
        if (stamp) clk[12 * j + 9] = wall_clock64();
      }
    }
    if (s_fail) { if (t == 0) atomicCAS(info, 0, r0 + 64 * j + s_fail); return; }
    // ---- L^-1 of the 64 x 64 tile from L (in Cs), by halves: the four 16 x 16 diagonal blocks at once in ONE wave — lane 16 g + c solves L_g x = e_c, the entries of
    // L_g reach the sixteen lanes of their group by DPP row_newbcast (one v_mov_b64_dpp per term where the 32-wide inverse of the pivot chain needed two v_readlane
    // and could do one block at a time: 2 x 4.8 us of a diagonal tile's 24) —, then the blocks below them by products on the matrix core, 16 -> 32 -> 64.
    if (t < 64) {
      const int g16 = 16 * (t >> 4), c = t & 15;
      double lrow[16], x[16];
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) lrow[k2] = Cs[g16 + c][g16 + k2];        // row c of the group's block (zeros above the diagonal)
      const double rd_own = rds[g16 + c];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        double sacc = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < q; ++k2) sacc += row_bcast_f64(lrow[k2], q) * x[k2];   // x[k2] = 0 for k2 < c
        const double rq = row_bcast_f64(rd_own, q);
        x[q] = q == c ? rq : (q > c ? -sacc * rq : 0.0);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) Iv[g16 + q][g16 + c] = x[q];
    }
    __syncthreads();
    if (w < 2) {
      // inside each 32 x 32 diagonal block h = w: X21 = -X22 (L21 X11), all 16 x 16
      const int h0 = PVLM_CHOL_NB * w;
      pvlm_d4 c4 = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cs[h0 + 16 + li][h0 + kk + lk], Iv[h0 + kk + lk][h0 + li], c4, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ts[w][lk + 4 * r][li] = c4[r];
    }
    __syncthreads();
    if (w < 2) {
      const int h0 = PVLM_CHOL_NB * w;
      pvlm_d4 c4 = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Iv[h0 + 16 + li][h0 + 16 + kk + lk], Ts[w][kk + lk][li], c4, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) { Iv[h0 + 16 + lk + 4 * r][h0 + li] = -c4[r]; Iv[h0 + lk + 4 * r][h0 + 16 + li] = 0.0; }
    }
    __syncthreads();
    {
      // inv21 = -inv22 (L21 inv11), a 16 x 16 block per wave on the matrix core: tmp = L21 inv11 into the upper right quarter of Cs (free: zeros above the diagonal),
      // then the product into Iv's lower left quarter.  (As loops of 32 over LDS per thread the two products took 9.7 us of the 34 of a diagonal tile.)
      const int bm = w >> 1, bn = w & 1;
      pvlm_d4 c4 = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < PVLM_CHOL_NB; kk += 4)
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Cs[PVLM_CHOL_NB + 16 * bm + li][kk + lk], Iv[kk + lk][16 * bn + li], c4, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[16 * bm + lk + 4 * r][PVLM_CHOL_NB + 16 * bn + li] = c4[r];
      __syncthreads();
      c4 = (pvlm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < PVLM_CHOL_NB; kk += 4)
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Iv[PVLM_CHOL_NB + 16 * bm + li][PVLM_CHOL_NB + kk + lk], Cs[kk + lk][PVLM_CHOL_NB + 16 * bn + li], c4, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) { Iv[PVLM_CHOL_NB + 16 * bm + lk + 4 * r][16 * bn + li] = -c4[r]; Iv[16 * bm + lk + 4 * r][PVLM_CHOL_NB + 16 * bn + li] = 0.0; }
      // the upper triangles of the 16 x 16 diagonal blocks of Iv: lane c wrote x[q] = 0 for q < c — complete
      __syncthreads();
    }
    if (stamp) clk[12 * j + 4] = wall_clock64();
    // publish the inverse
#pragma unroll
    for (int it = 0; it < 16; ++it) { const int e = t + 256 * it; tail_st(inv64 + (size_t)j * 4096 + e, Iv[e >> 6][e & 63]); }
    tail_drain();
    __syncthreads();
    if (t == 0 && !(LISTS && s_id == withhold)) tail_raise(inv_flag + j);
    if (stamp) clk[12 * j + 5] = wall_clock64();
    // y_j = L_jj^-1 (b_j - sum_k L(j,k) y_k)
    if (t < 64) vs[t] = (LISTS ? tail_ld(b + r0 + 64 * j + t) : b[r0 + 64 * j + t]) - bacc;
    __syncthreads();
    if (t < 64) {
      double sacc = 0.0;
      for (int d = 0; d <= t; ++d) sacc += Iv[t][d] * vs[d];
      tail_st(yv + r0 + 64 * j + t, sacc);
    }
    tail_drain();
    __syncthreads();
    if (t == 0) tail_raise(y_flag + j);
    WG_STATE(s_id, 10);
  }
}
__global__ __launch_bounds__(256) void k_nd_tail(double* __restrict__ M, int n, int r0, int T, double* __restrict__ inv64, unsigned* __restrict__ flags, int* __restrict__ info,
                                                 double* __restrict__ b, double* __restrict__ yv, unsigned long long* __restrict__ clk) {
  nd_tile_flow<false>(M, n, r0, T, inv64, flags, info, b, yv, clk, nullptr, nullptr, 0);
}
#if PVLM_FLOW_OCC2
__global__ __launch_bounds__(256, 2) void k_nd_flow(
#else
__global__ __launch_bounds__(256) void k_nd_flow(
#endif
                                                 double* __restrict__ M, int n, int T, double* __restrict__ inv64, unsigned* __restrict__ flags, int* __restrict__ info,
                                                 double* __restrict__ b, double* __restrict__ yv, const NdFlowTask* __restrict__ tasks,
                                                 const NdFlowSource* __restrict__ sources, int n_tasks, unsigned long long* __restrict__ clk, int withhold) {
  nd_tile_flow<true>(M, n, 0, T, inv64, flags, info, b, yv, clk, tasks, sources, n_tasks, withhold);
}

// Backward substitution of the tail: x_j = L_jj^-T (y_j - sum_{i > j} L(i,j)^T x_i), a workgroup per tile column (ticket order: j descending), the tile of the next
// i on its way while the workgroup waits for x_i.  x goes into b (where k_nd_bwd of the levels below gathers it).
template <bool LISTS>
__device__ __forceinline__ void nd_tile_bwd(const double* __restrict__ M, int n, int r0, int T, const double* __restrict__ inv64, unsigned* __restrict__ flags, int n_tile_flags,
                                            int* __restrict__ info, double* __restrict__ b, const double* __restrict__ yv, const int* __restrict__ col_order,
                                            const int* __restrict__ below_off, const int* __restrict__ below) {
  __shared__ double Is[64 * PVLM_TAIL_LD];
  __shared__ double xs[64], part[4][64], vs[64];
  __shared__ int s_id, s_ok;
  double (*Iv)[PVLM_TAIL_LD] = reinterpret_cast<double (*)[PVLM_TAIL_LD]>(Is);
  unsigned* x_flag = flags + 2 + n_tile_flags + 2 * T;
  const int t = threadIdx.x, c = t & 63, rg = t >> 6;
  if (*info != 0) return;
  if (t == 0) s_id = (int)atomicAdd(flags + 1, 1u);
  __syncthreads();
  if (s_id >= T) return;
  const int j = LISTS ? col_order[T - 1 - s_id] : T - 1 - s_id;
  // the tile rows below the diagonal, walked from the last to the first: rows[q], q = n_below - 1 .. 0
  const int* rows = LISTS ? below + below_off[j] : nullptr;
  const int n_below = LISTS ? below_off[j + 1] - below_off[j] : T - 1 - j;
  auto row_at = [&](int q) { return LISTS ? rows[q] : j + 1 + q; };
  const size_t col_j = (size_t)(r0 + 64 * j);
#pragma unroll
  for (int it = 0; it < 16; ++it) { const int e = t + 256 * it; Iv[e >> 6][e & 63] = inv64[(size_t)j * 4096 + e]; }
  double acc = 0.0;
  double m[16], mn[16];
  if (n_below > 0) {
    const int i0 = row_at(n_below - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) mn[q] = M[((size_t)(r0 + 64 * i0) + 16 * rg + q) * n + col_j + c];
  }
  for (int qi = n_below - 1; qi >= 0; --qi) {
    const int i = row_at(qi);
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] = mn[q];
    if (qi > 0) {
      const int i1 = row_at(qi - 1);
#pragma unroll
      for (int q = 0; q < 16; ++q) mn[q] = M[((size_t)(r0 + 64 * i1) + 16 * rg + q) * n + col_j + c];
    }
    if (t == 0) s_ok = tail_wait(x_flag + i, info, flags, 6, s_id) ? 1 : 0;
    __syncthreads();
    if (!s_ok) return;
    if (t < 64) xs[t] = tail_ld(b + r0 + 64 * i + t);
    __syncthreads();
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc += m[q] * xs[16 * rg + q];
    acc += sacc;
    __syncthreads();                                                 // xs is read: the next round may overwrite it
  }
  part[rg][c] = acc;
  __syncthreads();
  if (t < 64) vs[t] = yv[col_j + t] - (((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]);
  __syncthreads();
  if (t < 64) {
    double sacc = 0.0;
    for (int d = t; d < 64; ++d) sacc += Iv[d][t] * vs[d];
    tail_st(b + col_j + t, sacc);
  }
  tail_drain();
  __syncthreads();
  if (t == 0) tail_raise(x_flag + j);
}
__global__ __launch_bounds__(256) void k_nd_tail_bwd(const double* __restrict__ M, int n, int r0, int T, const double* __restrict__ inv64, unsigned* __restrict__ flags,
                                                     int* __restrict__ info, double* __restrict__ b, const double* __restrict__ yv) {
  nd_tile_bwd<false>(M, n, r0, T, inv64, flags, T * T, info, b, yv, nullptr, nullptr, nullptr);
}
// the whole backward substitution (k_nd_flow's factor): tile columns in the reverse of the task order, the tile rows of a column from its list
__global__ __launch_bounds__(256) void k_nd_flow_bwd(const double* __restrict__ M, int n, int T, const double* __restrict__ inv64, unsigned* __restrict__ flags, int n_tasks,
                                                     int* __restrict__ info, double* __restrict__ b, const double* __restrict__ yv, const int* __restrict__ col_order,
                                                     const int* __restrict__ below_off, const int* __restrict__ below) {
  nd_tile_bwd<true>(M, n, 0, T, inv64, flags, n_tasks, info, b, yv, col_order, below_off, below);
}

// ---- triangular solves with the factor (single right-hand side), one launch per block column ------------------------
// forward step k: y_k = Linv_k b_k (every workgroup recomputes the 32-vector, workgroup 0 stores it in yv), then
// b[i] -= L[i, k] . y_k for the rows below.  yv is a separate vector so that no workgroup reads what another one writes.
__global__ __launch_bounds__(256) void k_fwd_step(const double* __restrict__ M, int n, int k0, int kb, const double* __restrict__ Linv,
                                                  double* __restrict__ b, double* __restrict__ yv, const int* __restrict__ info) {
  __shared__ double inv[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ double v[PVLM_CHOL_NB], y[PVLM_CHOL_NB];
  if (*info != 0) return;
  const int t = threadIdx.x;
  const double* Lk = Linv + (size_t)(k0 / PVLM_CHOL_NB) * PVLM_CHOL_NB * PVLM_CHOL_NB;
  for (int e = t; e < PVLM_CHOL_NB * PVLM_CHOL_NB; e += 256) inv[e / PVLM_CHOL_NB][e % PVLM_CHOL_NB] = Lk[e];
  if (t < PVLM_CHOL_NB) v[t] = t < kb ? b[k0 + t] : 0.0;
  __syncthreads();
  if (t < PVLM_CHOL_NB) {
    double sacc = 0.0;
    for (int d = 0; d <= t; ++d) sacc += inv[t][d] * v[d];
    y[t] = sacc;
    if (blockIdx.x == 0 && t < kb) yv[k0 + t] = sacc;
  }
  __syncthreads();
  const int i = k0 + kb + blockIdx.x * 256 + t;
  if (i >= n) return;
  const double* r = M + (size_t)i * n + k0;
  double sacc = 0.0;
  for (int c = 0; c < kb; ++c) sacc += r[c] * y[c];
  b[i] -= sacc;
}
// backward step k: x_k = Linv_k^T yv_k (stored into b by workgroup 0), then yv[j] -= L[k, j]^T . x_k for the columns j < k0
__global__ __launch_bounds__(256) void k_bwd_step(const double* __restrict__ M, int n, int k0, int kb, const double* __restrict__ Linv,
                                                  double* __restrict__ b, double* __restrict__ yv, const int* __restrict__ info) {
  __shared__ double inv[PVLM_CHOL_NB][PVLM_CHOL_NB + 1];
  __shared__ double v[PVLM_CHOL_NB], x[PVLM_CHOL_NB];
  if (*info != 0) return;
  const int t = threadIdx.x;
  const double* Lk = Linv + (size_t)(k0 / PVLM_CHOL_NB) * PVLM_CHOL_NB * PVLM_CHOL_NB;
  for (int e = t; e < PVLM_CHOL_NB * PVLM_CHOL_NB; e += 256) inv[e / PVLM_CHOL_NB][e % PVLM_CHOL_NB] = Lk[e];
  if (t < PVLM_CHOL_NB) v[t] = t < kb ? yv[k0 + t] : 0.0;
  __syncthreads();
  if (t < PVLM_CHOL_NB) {
    double sacc = 0.0;
    for (int d = t; d < PVLM_CHOL_NB; ++d) sacc += inv[d][t] * v[d];
    x[t] = sacc;
    if (blockIdx.x == 0 && t < kb) b[k0 + t] = sacc;
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + t;
  if (j >= k0) return;
  // all 32 rows of the block in flight together (rows past a short last block are clamped; their x is zero)
  double m[PVLM_CHOL_NB];
#pragma unroll
  for (int c = 0; c < PVLM_CHOL_NB; ++c) m[c] = M[(size_t)min(k0 + c, n - 1) * n + j];
  double sacc = 0.0;
#pragma unroll
  for (int c = 0; c < PVLM_CHOL_NB; ++c) sacc += m[c] * x[c];
  yv[j] -= sacc;
}

// factorises d_M (n x n row-major, lower triangle used and overwritten by L) and solves for nrhs right-hand sides stored
// one after the other in d_B; all on ctx->stream.
// d_Linv: ceil(n / 32) x 32 x 32 doubles (inverses of the diagonal blocks), d_y: n doubles (forward-solve result).
// Tile-sparse form (plan != nullptr): block column k only touches the 64-row tiles that hold one of its nonzeros — lists made by the
// symbolic factorisation of SpdPlan, below — instead of every row below it.
struct SpdTileLists {
  const int* d_row_tiles; const int2* d_pairs;          // concatenated over the block columns
  const int* row_off; const int* pair_off;              // host: steps + 1 offsets
};
static void chol_factor_solve(pvlm_ctx* ctx, int n, double* d_M, double* d_B, int nrhs, double* d_Linv, double* d_y, int* d_info, const SpdTileLists* plan = nullptr) {
  hipStream_t s = ctx->stream;
#if PVLM_MEASURED_VARIANTS
  static const bool use_mfma = getenv("PVLM_CHOL_VALU") == nullptr;   // PVLM_CHOL_VALU=1: the register-tiled VALU update (measured variant)
  static const bool fused = getenv("PVLM_CHOL_SPLIT") == nullptr;      // PVLM_CHOL_SPLIT=1: round 1's launch structure (k_chol_diag, k_chol_panel, k_fwd_step)
  static const bool want_ahead = getenv("PVLM_CHOL_LOOKAHEAD") != nullptr;   // opt-in: measured SLOWER (below)
#else
  const bool fused = true, want_ahead = false;                        // the default library: fused diagonal + panel launch, MFMA-f64 update, one stream
#endif
  const bool ride = fused && nrhs >= 1;                                // the first right-hand side's forward substitution rides along
  // Look-ahead on a second stream (PVLM_CHOL_LOOKAHEAD=1; built, measured, not adopted: 5.45 ms against 4.56 ms at n = 2724 —
  // two event records and two stream waits per block column cost more on this runtime than the 20 us of overlap they buy):
  // block column k + 1 can be factorised as soon as the FIRST tile column of update k is done;
  // the rest of update k (part 2) runs beside it.  Stream A (the context stream): F(k), U1(k) [first tile column + rhs rows];
  // stream B: U2(k).  U2(k) waits for F(k) (it needs the panel); U1(k) waits for U2(k - 1) (they touch the same columns);
  // everything else is ordered by the streams themselves.
  bool ahead = fused && want_ahead && !ctx->capturing && n > 8 * PVLM_CHOL_NB;
  if (ahead && !ctx->aux_stream) {
    ahead = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) == hipSuccess;
    for (int q = 0; q < 8 && ahead; ++q) ahead = hipEventCreateWithFlags(&ctx->aux_ev[q], hipEventDisableTiming) == hipSuccess;
    if (!ahead) ctx->aux_stream = nullptr;
  }
  hipStream_t sb = ahead ? ctx->aux_stream : s;
  if (ahead) { hipEventRecord(ctx->aux_ev[0], s); hipStreamWaitEvent(sb, ctx->aux_ev[0], 0); }
  int step = 0, last_u2 = -1;
  auto launch_update = [&](hipStream_t st, int k0, int kb, int tiles, int part, int tile_blocks, int fwd_blocks) {
    if (tile_blocks + fwd_blocks <= 0) return;
#if PVLM_MEASURED_VARIANTS
    if (!use_mfma) { hipLaunchKernelGGL(k_chol_update, dim3((unsigned)(tile_blocks + fwd_blocks)), dim3(256), 0, st, d_M, n, k0, kb, tiles, d_info, tile_blocks, d_B, d_y, part); return; }
#endif
    hipLaunchKernelGGL(k_chol_update_mfma, dim3((unsigned)(tile_blocks + fwd_blocks)), dim3(256), 0, st, d_M, n, k0, kb, tiles, d_info, tile_blocks, d_B, d_y, part,
                       (const int2*)nullptr);
  };
  for (int k0 = 0; k0 < n; k0 += PVLM_CHOL_NB, ++step) {
    const int kb = std::min(PVLM_CHOL_NB, n - k0), rem = n - k0 - kb;
    if (plan) {
      const int nrt = plan->row_off[step + 1] - plan->row_off[step], npr = plan->pair_off[step + 1] - plan->pair_off[step];
      hipLaunchKernelGGL(k_chol_diag_panel, dim3((unsigned)std::max(1, nrt * 8)), dim3(256), 0, s, d_M, n, k0, kb, d_Linv, d_info, d_info,
                         ride ? (const double*)d_B : nullptr, d_y, plan->d_row_tiles + plan->row_off[step], nrt);
      const int fwd_blocks = (ride && rem > 0) ? (rem + 255) / 256 : 0;
      if (npr + fwd_blocks > 0)
        hipLaunchKernelGGL(k_chol_update_mfma, dim3((unsigned)(npr + fwd_blocks)), dim3(256), 0, s, d_M, n, k0, kb, 0, d_info, npr, d_B, d_y, 0,
                           plan->d_pairs + plan->pair_off[step]);
      continue;
    }
    if (fused) {
      // a failed pivot: every workgroup finds it itself (same arithmetic on the same block) and returns; workgroup 0 records it
      // in *info, which the later launches test on entry
      hipLaunchKernelGGL(k_chol_diag_panel, dim3(std::max(1, (rem + 7) / 8)), dim3(256), 0, s, d_M, n, k0, kb, d_Linv, d_info, d_info,
                         ride ? (const double*)d_B : nullptr, d_y, (const int*)nullptr, 0);
    }
#if PVLM_MEASURED_VARIANTS
    else {
      hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, s, d_M, n, k0, kb, d_Linv, d_info);
      if (rem > 0) hipLaunchKernelGGL(k_chol_panel, dim3((rem + 7) / 8), dim3(256), 0, s, d_M, n, k0, kb, d_Linv, d_info);
    }
#endif
    if (rem > 0) {
      const int tiles = (rem + 63) / 64;
      const int fwd_blocks = ride ? (rem + 255) / 256 : 0;
      if (ahead && tiles >= 2) {
        hipEvent_t ef = ctx->aux_ev[1 + (step & 1)], eu = ctx->aux_ev[3 + (step & 3)];
        hipEventRecord(ef, s);                                   // F(k) done: the panel exists
        hipStreamWaitEvent(sb, ef, 0);
        launch_update(sb, k0, kb, tiles, 2, (int)((long long)(tiles - 1) * tiles / 2), 0);
        hipEventRecord(eu, sb);
        if (last_u2 >= 0) hipStreamWaitEvent(s, ctx->aux_ev[3 + (last_u2 & 3)], 0);    // U1(k) after U2(k - 1)
        launch_update(s, k0, kb, tiles, 1, tiles, fwd_blocks);
        last_u2 = step;
      } else {
        if (ahead && last_u2 >= 0) { hipStreamWaitEvent(s, ctx->aux_ev[3 + (last_u2 & 3)], 0); last_u2 = -1; }
        launch_update(s, k0, kb, tiles, 0, (int)((long long)tiles * (tiles + 1) / 2), fwd_blocks);
      }
    }
  }
  if (ahead && last_u2 >= 0) hipStreamWaitEvent(s, ctx->aux_ev[3 + (last_u2 & 3)], 0);
  for (int r = 0; r < nrhs; ++r) {
    double* b = d_B + (size_t)r * n;
    if (!(ride && r == 0))
      for (int k0 = 0; k0 < n; k0 += PVLM_CHOL_NB) {
        const int kb = std::min(PVLM_CHOL_NB, n - k0), rem = n - k0 - kb;
        hipLaunchKernelGGL(k_fwd_step, dim3(std::max(1, (rem + 255) / 256)), dim3(256), 0, s, d_M, n, k0, kb, d_Linv, b, d_y, d_info);
      }
    for (int k0 = ((n - 1) / PVLM_CHOL_NB) * PVLM_CHOL_NB; k0 >= 0; k0 -= PVLM_CHOL_NB) {
      const int kb = std::min(PVLM_CHOL_NB, n - k0);
      hipLaunchKernelGGL(k_bwd_step, dim3(std::max(1, (k0 + 255) / 256)), dim3(256), 0, s, d_M, n, k0, kb, d_Linv, b, d_y, d_info);
    }
  }
#if PVLM_CHOL_CLOCK
  {
    unsigned long long h[4] = {}, z[4] = {};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chol_clock), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_clock), z, sizeof z);
    const double L = (double)std::max<unsigned long long>(h[3], 1);
    fprintf(stderr, "[chol clock] %llu panel launches | us per launch (workgroup 0): load %.2f  chain %.2f  stores + rows %.2f\n", h[3], h[0] * 0.01 / L, h[1] * 0.01 / L, h[2] * 0.01 / L);
  }
#endif
}

// The same factorisation and the two triangular solves by the level schedule: per level ONE panel launch (every block column of the level) and ONE update
// launch (every tile the level's columns reach + the right-hand side rows), then the backward substitution level by level in descending order.
struct SpdPlan;
static void chol_factor_solve_levels(pvlm_ctx* ctx, int n, double* d_M, double* d_b, double* d_Linv, double* d_y, int* d_info, const SpdPlan* P);

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

// carve aligned pieces out of the context's grow-only workspace
struct WsCarver {
  char* base; size_t used = 0;
  template <typename T> T* take(size_t count) { T* p = reinterpret_cast<T*>(base + used); used += (count * sizeof(T) + 255) & ~(size_t)255; return p; }
};
static pvlm_status ws_reserve(pvlm_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return PVLM_OK;
  PVLM_TRY_SYNC(ctx);
  pvlm_i_free(ctx, ctx->d_ws); ctx->d_ws = nullptr; ctx->ws_bytes = 0;
  pvlm_status st = pvlm_i_alloc_bytes(ctx, &ctx->d_ws, bytes);
  if (st) return st;
  ctx->ws_bytes = bytes;
  return PVLM_OK;
}
static size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// Only the tiles the one-launch factorisation reads have to start from zero — the tiles of its task list, 170 MB of the 1.16 GB of the padded Floor matrix (the
// assembly also adds mirrored blocks into tiles above the diagonal, which nobody reads).  A workgroup per task; chunk tasks name a tile a second time and return.
__global__ __launch_bounds__(256) void k_zero_flow_tiles(double* __restrict__ M, int n, const NdFlowTask* __restrict__ tasks) {
  const NdFlowTask tk = tasks[blockIdx.x];
  if (!tk.final_) return;
  double2* row0 = reinterpret_cast<double2*>(M + (size_t)(64 * tk.I) * n + 64 * tk.J);
  const int r = threadIdx.x >> 2, c = threadIdx.x & 3;                  // 64 rows x 4 threads, 8 double2 each (n is a multiple of 64: 16-byte aligned)
  double2* p = row0 + (size_t)r * (n / 2) + c;
#pragma unroll
  for (int q = 0; q < 8; ++q) p[4 * q] = make_double2(0.0, 0.0);
}

// ---- tile-sparse plan of pvlm_spd_solve_blocks ------------------------------------------------------------------------------------
// The reduced pose system of a Floor-sized run (1593 scans, 9558 unknowns) is block-sparse: 15 neighbours per scan.  Factorised as a
// dense matrix it cost 39.6 ms per LM step and 46 % of the whole EstimatePose call (profiles/r3_floor_like_1593.txt); the reference
// selects SPARSE_SCHUR there (util/Optimization.cpp:641-658).  The plan keeps the dense kernels and their 64 x 64 / 32-column tiling
// and makes them skip what is structurally zero: ordering + symbolic factorisation in csrc/pvlm_spd_plan.h.
// Unknowns are permuted on the host (indices, scale, damping, right-hand side); the device sees an ordinary symmetric system.
// One plan is cached per context, keyed by a hash of the index lists: the LM steps of a Solve share their structure.
struct SpdPlan {
  unsigned long long key = 0;
  int n = 0;
  bool sparse = false;
  std::vector<int> new_of_old;                 // permutation of the unknowns
  std::vector<int> row_off, pair_off;          // steps + 1
  std::vector<int> key_rows, key_cols, key_mirror;   // the structure the plan was built for: compared on every cache hit (a hash alone may collide)
  int* d_row_tiles = nullptr; int2* d_pairs = nullptr;
  double update_fraction = 1.0;                // tile updates / tile updates of the dense factorisation
  // level schedule (nested dissection, pvlm_spd::plan_levels): the system is padded to n_pad rows (groups aligned to 64-row tiles, dummy rows = identity)
  bool levels = false;
  int n_pad = 0, n_levels = 0;
  std::vector<int> col_off, pwg_off, pwt_off, upd_off, fwd_off;       // levels + 1 each (host)
  int* d_cols = nullptr; int* d_row_off = nullptr; int2* d_pwg = nullptr; int2* d_pwt = nullptr; NdTarget* d_targets = nullptr; int* d_sources = nullptr;
  NdRowTarget* d_ftargets = nullptr; int* d_fsources = nullptr;
  // the dense tail (k_nd_tail): levels n_main .. n_levels - 1 = the rows tail_r0 .. n_pad - 1 (tail_T 64-row tiles); 0 tiles: every level by its launches
  int n_main = 0, tail_r0 = 0, tail_T = 0;
  double* d_tail_inv = nullptr; unsigned* d_tail_flags = nullptr;
  // the whole factorisation as the tasks of one launch (k_nd_flow / k_nd_flow_bwd, pvlm_spd::plan_flow); flow_T = 0: levels (+ tail)
  int flow_T = 0, flow_tasks = 0, flow_depth = 0;
  bool flow_possible = false;                       // the host plan had a task list (whether or not this plan uses it: pvlm_spd_one_launch)
  // the permuted index lists of the structure on the device, made by the first solve with this plan: the LM steps of a Solve change the values only
  int* d_idx_rows = nullptr; int* d_idx_cols = nullptr; int* d_idx_mirror = nullptr; bool idx_on_device = false;
  NdFlowTask* d_flow_tasks = nullptr; NdFlowSource* d_flow_sources = nullptr; int* d_flow_cols = nullptr; int* d_flow_below_off = nullptr; int* d_flow_below = nullptr;
};

static unsigned long long spd_hash(int n, int n_blocks, const int* row_idx, const int* col_idx, const int* mirror) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](const int* p, size_t count) { for (size_t i = 0; i < count; ++i) { h ^= (unsigned)p[i]; h *= 1099511628211ull; } };
  mix(&n, 1); mix(&n_blocks, 1); mix(row_idx, (size_t)n_blocks * 6); mix(col_idx, (size_t)n_blocks * 6); mix(mirror, (size_t)n_blocks);
  return h;
}

// The HOST half of a plan: ordering + symbolic factorisation + schedule lists (csrc/pvlm_spd_plan.h), no device call — what pvlm_spd_plan_prefetch runs on a thread of
// its own while the caller's GPU work (structures, first linearisation of a Solve) goes on.  kind: 0 = natural order + dense kernels, 1 = level schedule, 2 = round-5 plan.
struct SpdHostPlan {
  int kind = 0;
  double update_fraction = 1.0;
  pvlm_spd::LevelPlan L;
  pvlm_spd::Symbolic S;
};
static void spd_plan_host(int n, int n_blocks, const int* row_idx, const int* col_idx, SpdHostPlan* H) {
  H->kind = 0; H->update_fraction = 1.0;
  static const int min_n = getenv("PVLM_SPD_SPARSE_MIN") ? atoi(getenv("PVLM_SPD_SPARSE_MIN")) : 1500;
  if (n < min_n || n_blocks == 0) return;
  static_assert(64 % PVLM_CHOL_NB == 0, "a 64-row tile must span a whole number of block columns (pvlm_spd_plan.h marks the fill per tile)");
  // nested dissection + level schedule (PVLM_SPD_LEVELS=0: the round-5 plan — minimum degree, one block column after the other)
  static const bool want_levels = !(getenv("PVLM_SPD_LEVELS") && atoi(getenv("PVLM_SPD_LEVELS")) == 0);
  if (want_levels) {
    static const int leaf = getenv("PVLM_SPD_LEAF") ? std::max(1, atoi(getenv("PVLM_SPD_LEAF"))) : 42;      // nodes per undissected group (42 poses = 252 rows = 4 tiles)
    static const int flow_chunk = getenv("PVLM_SPD_FLOW_CHUNK") ? std::max(0, atoi(getenv("PVLM_SPD_FLOW_CHUNK"))) : 12;      // sources per chunk task of the one launch (0: no chunks)
    pvlm_spd::plan_levels(n, n_blocks, row_idx, col_idx, PVLM_CHOL_NB, leaf, &H->L, flow_chunk);
    static const double max_pad = getenv("PVLM_SPD_MAX_PAD") ? atof(getenv("PVLM_SPD_MAX_PAD")) : 1.6;
    // adopted when it shortens the chain of dependent launches by a third at least and the padding stays bounded
    if (H->L.ordered && H->L.levels * 3 <= H->L.cols_total * 2 + 2 && (double)H->L.n_pad <= max_pad * (double)n + 256.0) { H->kind = 1; H->update_fraction = H->L.update_fraction; return; }
    H->L = pvlm_spd::LevelPlan();
  }
  pvlm_spd::plan_symbolic(n, n_blocks, row_idx, col_idx, PVLM_CHOL_NB, &H->S);       // csrc/pvlm_spd_plan.h (host only, checked on the CPU)
  H->update_fraction = H->S.update_fraction;
  static const double max_fraction = getenv("PVLM_SPD_SPARSE_FRACTION") ? atof(getenv("PVLM_SPD_SPARSE_FRACTION")) : 0.6;
  if (H->S.ordered && H->S.update_fraction <= max_fraction) H->kind = 2;             // else: not sparse enough to pay for the irregular tile lists
}

// A plan being made ahead of the solve that will need it (pvlm_spd_plan_prefetch): the structure it is for (compared entry by entry when a solve misses the
// cache) and the thread that computes its host half.  One per context; a newer prefetch or the context's end joins the thread.
struct SpdPrefetch {
  int n = 0, n_blocks = 0;
  std::vector<int> rows, cols, mirror;
  SpdHostPlan host;
  bool failed = false;
  double worker_ms = 0.0;             // how long the host half took on its thread (PVLM_TRACE)
  std::thread worker;
};
static void spd_prefetch_drop(pvlm_ctx* ctx) {
  SpdPrefetch* f = static_cast<SpdPrefetch*>(ctx->spd_prefetch);
  if (!f) return;
  if (f->worker.joinable()) f->worker.join();
  delete f;
  ctx->spd_prefetch = nullptr;
}

static void spd_plan_free(pvlm_ctx* ctx, SpdPlan* p) {
  if (!p) return;
  pvlm_i_free(ctx, p->d_row_tiles); pvlm_i_free(ctx, p->d_pairs);
  pvlm_i_free(ctx, p->d_cols); pvlm_i_free(ctx, p->d_row_off); pvlm_i_free(ctx, p->d_pwg); pvlm_i_free(ctx, p->d_pwt); pvlm_i_free(ctx, p->d_targets); pvlm_i_free(ctx, p->d_sources);
  pvlm_i_free(ctx, p->d_ftargets); pvlm_i_free(ctx, p->d_fsources);
  pvlm_i_free(ctx, p->d_tail_inv); pvlm_i_free(ctx, p->d_tail_flags);
  pvlm_i_free(ctx, p->d_idx_rows); pvlm_i_free(ctx, p->d_idx_cols); pvlm_i_free(ctx, p->d_idx_mirror);
  pvlm_i_free(ctx, p->d_flow_tasks); pvlm_i_free(ctx, p->d_flow_sources); pvlm_i_free(ctx, p->d_flow_cols); pvlm_i_free(ctx, p->d_flow_below_off); pvlm_i_free(ctx, p->d_flow_below);
  delete p;
}
void pvlm_i_spd_plan_release(pvlm_ctx* ctx) { spd_prefetch_drop(ctx); spd_plan_free(ctx, static_cast<SpdPlan*>(ctx->spd_plan)); ctx->spd_plan = nullptr; }

// The DEVICE half: the lists of a host plan uploaded, the host plan's vectors moved into *P.
static pvlm_status spd_plan_adopt(pvlm_ctx* ctx, int n, SpdHostPlan& H, SpdPlan* P) {
  P->n = n; P->sparse = false;
  P->update_fraction = H.update_fraction;
  if (H.kind == 0) {
    P->new_of_old.resize((size_t)n);
    for (int i = 0; i < n; ++i) P->new_of_old[(size_t)i] = i;
    return PVLM_OK;
  }
  static_assert(sizeof(pvlm_spd::Target) == sizeof(NdTarget) && sizeof(pvlm_spd::RowTarget) == sizeof(NdRowTarget) && sizeof(pvlm_spd::PanelGroup) == sizeof(int2), "level lists are uploaded as they are");
  static_assert(sizeof(pvlm_spd::TilePair) == sizeof(int2), "tile pairs are uploaded as int2");
  pvlm_status st = PVLM_OK;
  auto up = [&](void** d, const void* src, size_t bytes) {
    if (st) return;
    st = pvlm_i_alloc_bytes(ctx, d, std::max<size_t>(bytes, 256));
    if (!st && bytes) st = pvlm_i_h2d_q(ctx, *d, src, bytes);
  };
  if (H.kind == 1) {
    pvlm_spd::LevelPlan& L = H.L;
    up((void**)&P->d_cols, L.cols.data(), L.cols.size() * sizeof(int));
    up((void**)&P->d_row_off, L.row_off.data(), L.row_off.size() * sizeof(int));
    up((void**)&P->d_row_tiles, L.row_tiles.data(), L.row_tiles.size() * sizeof(int));
    up((void**)&P->d_pwg, L.pwg.data(), L.pwg.size() * sizeof(int2));
    up((void**)&P->d_pwt, L.pwt.data(), L.pwt.size() * sizeof(int2));
    up((void**)&P->d_targets, L.targets.data(), L.targets.size() * sizeof(NdTarget));
    up((void**)&P->d_sources, L.sources.data(), L.sources.size() * sizeof(int));
    up((void**)&P->d_ftargets, L.ftargets.data(), L.ftargets.size() * sizeof(NdRowTarget));
    up((void**)&P->d_fsources, L.fsources.data(), L.fsources.size() * sizeof(int));
    if (!st) st = pvlm_i_sync(ctx);
    if (st) return st;
    P->new_of_old.swap(L.new_of_old); P->col_off.swap(L.col_off); P->pwg_off.swap(L.pwg_off); P->pwt_off.swap(L.pwt_off); P->upd_off.swap(L.upd_off); P->fwd_off.swap(L.fwd_off);
    P->n_pad = L.n_pad; P->n_levels = L.levels;
    P->levels = true; P->sparse = true;
    P->n_main = L.levels; P->tail_T = 0;
    // one launch for the whole factorisation (PVLM_SPD_FLOW=0: the level launches, with the dense tail below)
    static const bool want_flow = !(getenv("PVLM_SPD_FLOW") && atoi(getenv("PVLM_SPD_FLOW")) == 0);
    P->flow_possible = want_flow && L.flow.ready && L.flow.tile_cols >= 2 && L.flow.tile_cols <= 4096;
    static_assert(sizeof(pvlm_spd::FlowTask) == sizeof(NdFlowTask) && sizeof(pvlm_spd::FlowSource) == sizeof(NdFlowSource), "flow lists are uploaded as they are");
    if (want_flow && ctx->spd_one_launch && L.flow.ready && L.flow.tile_cols >= 2 && L.flow.tile_cols <= 4096) {
      const pvlm_spd::FlowPlan& F = L.flow;
      up((void**)&P->d_flow_tasks, F.tasks.data(), F.tasks.size() * sizeof(NdFlowTask));
      up((void**)&P->d_flow_sources, F.sources.data(), F.sources.size() * sizeof(NdFlowSource));
      up((void**)&P->d_flow_cols, F.col_order.data(), F.col_order.size() * sizeof(int));
      up((void**)&P->d_flow_below_off, F.below_off.data(), F.below_off.size() * sizeof(int));
      up((void**)&P->d_flow_below, F.below.data(), F.below.size() * sizeof(int));
      if (!st) st = pvlm_i_alloc_bytes(ctx, (void**)&P->d_tail_inv, (size_t)F.tile_cols * 4096 * sizeof(double));
      if (!st) st = pvlm_i_alloc_bytes(ctx, (void**)&P->d_tail_flags, ((size_t)2 + F.tasks.size() + 3 * (size_t)F.tile_cols) * sizeof(unsigned));
      if (!st) st = pvlm_i_sync(ctx);
      if (st) return st;
      P->flow_T = F.tile_cols; P->flow_tasks = (int)F.tasks.size(); P->flow_depth = F.depth;
      return PVLM_OK;
    }
    static const bool want_tail = !(getenv("PVLM_SPD_TAIL") && atoi(getenv("PVLM_SPD_TAIL")) == 0);
    if (want_tail && L.tail_col0 < L.cols_total) {
      const int r0 = L.tail_col0 * PVLM_CHOL_NB, T = (L.n_pad - r0) / 64;
      if (T >= 2 && T <= 1024 && r0 % 64 == 0) {
        st = pvlm_i_alloc_bytes(ctx, (void**)&P->d_tail_inv, (size_t)T * 4096 * sizeof(double));
        if (!st) st = pvlm_i_alloc_bytes(ctx, (void**)&P->d_tail_flags, tail_flag_words(T) * sizeof(unsigned));
        if (st) return st;
        P->n_main = L.main_levels; P->tail_r0 = r0; P->tail_T = T;
      }
    }
    return PVLM_OK;
  }
  pvlm_spd::Symbolic& S = H.S;
  up((void**)&P->d_row_tiles, S.row_tiles.data(), S.row_tiles.size() * sizeof(int));
  up((void**)&P->d_pairs, S.pairs.data(), S.pairs.size() * sizeof(int2));
  if (!st) st = pvlm_i_sync(ctx);
  if (st) return st;
  P->new_of_old.swap(S.new_of_old); P->row_off.swap(S.row_off); P->pair_off.swap(S.pair_off);
  P->sparse = true;
  return PVLM_OK;
}

static void chol_factor_solve_levels(pvlm_ctx* ctx, int n, double* d_M, double* d_b, double* d_Linv, double* d_y, int* d_info, const SpdPlan* P) {
  hipStream_t s = ctx->stream;
  if (P->flow_T > 0) {
    const int T = P->flow_T;
    (void)hipMemsetAsync(P->d_tail_flags, 0, ((size_t)2 + (size_t)P->flow_tasks + 3 * (size_t)T) * sizeof(unsigned), s);
    static const bool want_clock = getenv("PVLM_SPD_TAIL_CLOCK") && atoi(getenv("PVLM_SPD_TAIL_CLOCK")) != 0;
    unsigned long long* d_clk = nullptr;
    if (want_clock && pvlm_i_alloc_bytes(ctx, (void**)&d_clk, (size_t)T * 12 * sizeof(unsigned long long)) != PVLM_OK) d_clk = nullptr;
    if (d_clk) (void)hipMemsetAsync(d_clk, 0, (size_t)T * 12 * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_nd_flow, dim3((unsigned)std::min(P->flow_tasks, 1024)), dim3(256), 0, s, d_M, n, T, P->d_tail_inv, P->d_tail_flags, d_info, d_b, d_y,
                       (const NdFlowTask*)P->d_flow_tasks, (const NdFlowSource*)P->d_flow_sources, P->flow_tasks, d_clk, ctx->spd_withhold_task);
    if (d_clk) {
      std::vector<unsigned long long> c((size_t)T * 12);
      if (hipMemcpyAsync(c.data(), d_clk, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
        unsigned long long t0 = ~0ull;
        for (int j = 0; j < T; ++j) if (c[(size_t)12 * j]) t0 = std::min(t0, c[(size_t)12 * j]);
        fprintf(stderr, "k_nd_flow clocks (us from the first ticket): tile column | ticket dep-seen products tile-in-LDS factored published | first tile below: inverse-seen published | 32 pivots, update, 32 pivots\n");
        for (int j = 0; j < T; ++j) {
          fprintf(stderr, "%3d |", j);
          for (int q = 0; q < 12; ++q) { const unsigned long long v = c[(size_t)12 * j + q]; if (q == 6) fprintf(stderr, " |"); if (v) fprintf(stderr, " %8.2f", (double)(long long)(v - t0) * 0.01); else fprintf(stderr, "        -"); }
          fprintf(stderr, "\n");
        }
      }
      pvlm_i_free(ctx, d_clk);
    }
    hipLaunchKernelGGL(k_nd_flow_bwd, dim3((unsigned)T), dim3(256), 0, s, (const double*)d_M, n, T, (const double*)P->d_tail_inv, P->d_tail_flags, P->flow_tasks, d_info, d_b,
                       (const double*)d_y, (const int*)P->d_flow_cols, (const int*)P->d_flow_below_off, (const int*)P->d_flow_below);
    return;
  }
  const int n_main = P->tail_T > 0 ? P->n_main : P->n_levels;
  if (P->tail_T > 0) (void)hipMemsetAsync(P->d_tail_flags, 0, tail_flag_words(P->tail_T) * sizeof(unsigned), s);     // ahead of the levels: not between two dependent launches
  for (int l = 0; l < n_main; ++l) {
    const int npw = P->pwg_off[(size_t)l + 1] - P->pwg_off[(size_t)l], ntg = P->upd_off[(size_t)l + 1] - P->upd_off[(size_t)l], nft = P->fwd_off[(size_t)l + 1] - P->fwd_off[(size_t)l];
    // eight-row workgroups while they fit the device in about one round (256 CUs x 8 resident workgroups), whole tiles beyond
    const int npt = P->pwt_off[(size_t)l + 1] - P->pwt_off[(size_t)l];
    if (npw > 3072)
      hipLaunchKernelGGL(k_nd_panel<8>, dim3((unsigned)npt), dim3(256), 0, s, d_M, n, d_Linv, d_info, (const double*)d_b, d_y, (const int2*)P->d_pwt + P->pwt_off[(size_t)l],
                         (const int*)P->d_row_off, (const int*)P->d_row_tiles);
    else if (npw > 0)
      hipLaunchKernelGGL(k_nd_panel<1>, dim3((unsigned)npw), dim3(256), 0, s, d_M, n, d_Linv, d_info, (const double*)d_b, d_y, (const int2*)P->d_pwg + P->pwg_off[(size_t)l],
                         (const int*)P->d_row_off, (const int*)P->d_row_tiles);
    if (ntg + nft > 0)
      hipLaunchKernelGGL(k_nd_update, dim3((unsigned)(ntg + nft)), dim3(256), 0, s, d_M, n, (const int*)d_info, (const NdTarget*)P->d_targets + P->upd_off[(size_t)l], ntg,
                         (const int*)P->d_sources, (const NdRowTarget*)P->d_ftargets + P->fwd_off[(size_t)l], (const int*)P->d_fsources, d_b, (const double*)d_y);
  }
  if (P->tail_T > 0) {
    const int T = P->tail_T, n_tiles = T * (T + 1) / 2;
    static const bool want_clock = getenv("PVLM_SPD_TAIL_CLOCK") && atoi(getenv("PVLM_SPD_TAIL_CLOCK")) != 0;
    unsigned long long* d_clk = nullptr;
    if (want_clock && pvlm_i_alloc_bytes(ctx, (void**)&d_clk, (size_t)T * 12 * sizeof(unsigned long long)) != PVLM_OK) d_clk = nullptr;
    if (d_clk) (void)hipMemsetAsync(d_clk, 0, (size_t)T * 12 * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_nd_tail, dim3((unsigned)std::min(n_tiles, 512)), dim3(256), 0, s, d_M, n, P->tail_r0, T, P->d_tail_inv, P->d_tail_flags, d_info, d_b, d_y, d_clk);
    hipLaunchKernelGGL(k_nd_tail_bwd, dim3((unsigned)T), dim3(256), 0, s, (const double*)d_M, n, P->tail_r0, T, (const double*)P->d_tail_inv, P->d_tail_flags, d_info, d_b, (const double*)d_y);
    if (d_clk) {
      std::vector<unsigned long long> c((size_t)T * 12);
      if (hipMemcpyAsync(c.data(), d_clk, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
        const unsigned long long t0 = c[0];
        fprintf(stderr, "k_nd_tail clocks (us from the first ticket): col | ticket dep-seen products tile-in-LDS factored published | below: inverse-seen published\n");
        for (int j = 0; j < T; ++j) {
          fprintf(stderr, "%3d |", j);
          for (int q = 0; q < 12; ++q) { const unsigned long long v = c[(size_t)12 * j + q]; if (q == 6) fprintf(stderr, " |"); if (v) fprintf(stderr, " %8.2f", (double)(long long)(v - t0) * 0.01); else fprintf(stderr, "        -"); }
          fprintf(stderr, "\n");
        }
      }
      pvlm_i_free(ctx, d_clk);
    }
  }
  for (int l = n_main - 1; l >= 0; --l) {
    const int nc = P->col_off[(size_t)l + 1] - P->col_off[(size_t)l];
    if (nc > 0)
      hipLaunchKernelGGL(k_nd_bwd, dim3((unsigned)nc), dim3(1024), 0, s, (const double*)d_M, n, (const double*)d_Linv, d_b, (const double*)d_y, (const int*)d_info,
                         (const int*)P->d_cols + P->col_off[(size_t)l], (const int*)P->d_row_off, (const int*)P->d_row_tiles);
  }
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
  const int* const user_rows = row_idx; const int* const user_cols = col_idx; const double* const user_scale = scale; const double* const user_diag = diag_add;
  // the plan of this structure (ordering + tile lists), built once and reused while the index lists stay the same
  SpdPlan* plan = static_cast<SpdPlan*>(ctx->spd_plan);
  std::vector<int> prow, pcol;
  std::vector<double> pscale, pdiag, prhs;
  try {
    const unsigned long long key = spd_hash(n, n_blocks, row_idx, col_idx, mirror);
    // a hit needs the SAME index lists, not only the same 64-bit hash: with a colliding structure the tile lists of the old one would skip the new
    // one's non-zeros and the solve would be wrong with info = 0 (three memcmp of ~100 KB at Floor size against a 10 ms solve)
    const bool same = plan && plan->key == key && plan->n == n && (!plan->levels || (plan->flow_T > 0) == (ctx->spd_one_launch != 0 && plan->flow_possible)) && plan->key_rows.size() == (size_t)n_blocks * 6 && plan->key_mirror.size() == (size_t)n_blocks &&
                      std::memcmp(plan->key_rows.data(), row_idx, (size_t)n_blocks * 6 * sizeof(int)) == 0 &&
                      std::memcmp(plan->key_cols.data(), col_idx, (size_t)n_blocks * 6 * sizeof(int)) == 0 &&
                      std::memcmp(plan->key_mirror.data(), mirror, (size_t)n_blocks * sizeof(int)) == 0;
    if (!same) {
      PVLM_TRY_SYNC(ctx);
      pvlm_i_trace("spd solve: before the plan");
      spd_plan_free(ctx, static_cast<SpdPlan*>(ctx->spd_plan)); ctx->spd_plan = nullptr;      // the old plan only: a prefetch in flight is looked at below
      plan = new SpdPlan();
      plan->key = key;
      plan->key_rows.assign(row_idx, row_idx + (size_t)n_blocks * 6); plan->key_cols.assign(col_idx, col_idx + (size_t)n_blocks * 6);
      plan->key_mirror.assign(mirror, mirror + n_blocks);
      ctx->spd_plan = plan;
      // the host half: taken from pvlm_spd_plan_prefetch when that was given exactly these lists (it has been running beside the caller's GPU work), else made now
      SpdPrefetch* pre = static_cast<SpdPrefetch*>(ctx->spd_prefetch);
      bool taken = false;
      pvlm_status pst = PVLM_OK;
      if (pre) {
        const auto tj = std::chrono::steady_clock::now();
        if (pre->worker.joinable()) pre->worker.join();
        if (getenv("PVLM_TRACE")) {
          char msg[128];
          snprintf(msg, sizeof msg, "spd plan prefetch: thread %.2f ms, joined after %.2f ms", pre->worker_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj).count());
          pvlm_i_trace(msg);
        }
        if (!pre->failed && pre->n == n && pre->n_blocks == n_blocks && pre->rows == plan->key_rows && pre->cols == plan->key_cols && pre->mirror == plan->key_mirror) {
          pst = spd_plan_adopt(ctx, n, pre->host, plan);
          taken = true;
        }
        spd_prefetch_drop(ctx);
      }
      ctx->spd_prefetch_hits += taken ? 1 : 0;
      if (!taken) {
        SpdHostPlan host;
        spd_plan_host(n, n_blocks, row_idx, col_idx, &host);
        pst = spd_plan_adopt(ctx, n, host, plan);
      }
      if (pst) { pvlm_i_spd_plan_release(ctx); return pst; }
      if (getenv("PVLM_TRACE")) { char msg[96]; snprintf(msg, sizeof msg, "spd plan built: %s, update fraction %.3f, n %d, blocks %d", plan->sparse ? "tile-sparse" : "dense", plan->update_fraction, n, n_blocks); pvlm_i_trace(msg); }
    }
    if (plan->sparse) {
      const std::vector<int>& nw = plan->new_of_old;
      if (!plan->idx_on_device) {
        prow.resize((size_t)n_blocks * 6); pcol.resize((size_t)n_blocks * 6);
        for (size_t k = 0; k < prow.size(); ++k) { prow[k] = row_idx[k] >= 0 && row_idx[k] < n ? nw[(size_t)row_idx[k]] : -1; pcol[k] = col_idx[k] >= 0 && col_idx[k] < n ? nw[(size_t)col_idx[k]] : -1; }
      }
      // the level plan pads the system: dummy rows are identity rows (scale 1, diagonal 1, right-hand side 0 -> solution 0)
      const size_t np = plan->levels ? (size_t)plan->n_pad : (size_t)n;
      pscale.assign(np, 1.0); pdiag.assign(np, 1.0); prhs.assign(np, 0.0);
      for (int i = 0; i < n; ++i) { const size_t q = (size_t)nw[(size_t)i]; pscale[q] = scale[i]; pdiag[q] = diag_add[i]; prhs[q] = rhs[i]; }
      if (!plan->idx_on_device) { row_idx = prow.data(); col_idx = pcol.data(); }
      scale = pscale.data(); diag_add = pdiag.data();
    }
  } catch (const std::bad_alloc&) {
    PVLM_SET_ERR(ctx, "pvlm_spd_solve_blocks: out of host memory");
    return PVLM_ERR_NOMEM;
  }
  double* rhs_io = plan->sparse ? prhs.data() : rhs;
  const int n_user = n;
  if (plan->levels) n = plan->n_pad;                     // from here on the device works on the padded system
  SpdTileLists lists{plan->d_row_tiles, plan->d_pairs, plan->row_off.data(), plan->pair_off.data()};
  const size_t linv_count = (size_t)((n + PVLM_CHOL_NB - 1) / PVLM_CHOL_NB) * PVLM_CHOL_NB * PVLM_CHOL_NB;
  const size_t need = pad256((size_t)n * n * 8) + pad256((size_t)n_blocks * 36 * 8) + 3 * pad256((size_t)n_blocks * 6 * 4) + 4 * pad256((size_t)n * 8) +
                      pad256(linv_count * 8) + 256;
  pvlm_status st = ws_reserve(ctx, need);
  if (st) return st;
  WsCarver ws{static_cast<char*>(ctx->d_ws)};
  double* d_M = ws.take<double>((size_t)n * n); double* d_blocks = ws.take<double>((size_t)n_blocks * 36);
  int* d_row = ws.take<int>((size_t)n_blocks * 6); int* d_col = ws.take<int>((size_t)n_blocks * 6); int* d_mir = ws.take<int>((size_t)n_blocks * 6);
  double* d_scale = ws.take<double>(n); double* d_diag = ws.take<double>(n); double* d_rhs = ws.take<double>(n);
  double* d_y = ws.take<double>(n); double* d_Linv = ws.take<double>(linv_count);
  int* d_info = ws.take<int>(1);
  if (!st) {
    hipStream_t s = ctx->stream;
    hipError_t e = hipSuccess;
    if (plan->levels && plan->flow_T > 0) hipLaunchKernelGGL(k_zero_flow_tiles, dim3((unsigned)plan->flow_tasks), dim3(256), 0, s, d_M, n, (const NdFlowTask*)plan->d_flow_tasks);
    else e = hipMemsetAsync(d_M, 0, (size_t)n * n * sizeof(double), s);
    // through the pinned arena (1.3 MB of blocks at Room scale, once per LM step: a pageable copy locks the caller's pages)
    if (e == hipSuccess && n_blocks) {
      st = pvlm_i_h2d_q(ctx, d_blocks, blocks, (size_t)n_blocks * 36 * sizeof(double));
      if (plan->sparse && !plan->idx_on_device && !st) {
        // the (permuted) index lists of this structure stay on the device with the plan: later solves send the values only
        st = pvlm_i_alloc(ctx, &plan->d_idx_rows, (size_t)n_blocks * 6);
        if (!st) st = pvlm_i_alloc(ctx, &plan->d_idx_cols, (size_t)n_blocks * 6);
        if (!st) st = pvlm_i_alloc(ctx, &plan->d_idx_mirror, (size_t)n_blocks);
        if (!st) st = pvlm_i_h2d_q(ctx, plan->d_idx_rows, row_idx, (size_t)n_blocks * 6 * sizeof(int));
        if (!st) st = pvlm_i_h2d_q(ctx, plan->d_idx_cols, col_idx, (size_t)n_blocks * 6 * sizeof(int));
        if (!st) st = pvlm_i_h2d_q(ctx, plan->d_idx_mirror, mirror, (size_t)n_blocks * sizeof(int));
        if (!st) plan->idx_on_device = true;
      }
      if (plan->idx_on_device) { d_row = plan->d_idx_rows; d_col = plan->d_idx_cols; d_mir = plan->d_idx_mirror; }
      else {
        if (!st) st = pvlm_i_h2d_q(ctx, d_row, row_idx, (size_t)n_blocks * 6 * sizeof(int));
        if (!st) st = pvlm_i_h2d_q(ctx, d_col, col_idx, (size_t)n_blocks * 6 * sizeof(int));
        if (!st) st = pvlm_i_h2d_q(ctx, d_mir, mirror, (size_t)n_blocks * sizeof(int));
      }
    }
    if (e == hipSuccess && !st) st = pvlm_i_h2d_q(ctx, d_scale, scale, (size_t)n * sizeof(double));
    if (e == hipSuccess && !st) st = pvlm_i_h2d_q(ctx, d_diag, diag_add, (size_t)n * sizeof(double));
    if (e == hipSuccess && !st) st = pvlm_i_h2d_q(ctx, d_rhs, rhs_io, (size_t)n * sizeof(double));
    int info = 0;
    if (e == hipSuccess && !st) {
      if (n_blocks) hipLaunchKernelGGL(k_scatter_blocks, dim3((unsigned)(((long long)n_blocks * 36 + 255) / 256)), dim3(256), 0, s, n, n_blocks, d_row, d_col, d_mir, d_blocks, d_scale, d_M);
      hipLaunchKernelGGL(k_add_diag, dim3((n + 255) / 256), dim3(256), 0, s, n, d_diag, d_M);
      e = hipGetLastError();
    }
    if (e == hipSuccess && !st) e = hipMemsetAsync(d_info, 0, sizeof(int), s);
    if (e == hipSuccess && !st) {
      if (plan->levels) chol_factor_solve_levels(ctx, n, d_M, d_rhs, d_Linv, d_y, d_info, plan);
      else chol_factor_solve(ctx, n, d_M, d_rhs, 1, d_Linv, d_y, d_info, plan->sparse ? &lists : nullptr);
      e = hipGetLastError();
    }
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_spd_solve_blocks: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
    if (!st) st = pvlm_i_d2h_q(ctx, &info, d_info, sizeof(int));
    if (!st) st = pvlm_i_d2h_q(ctx, rhs_io, d_rhs, (size_t)n * sizeof(double));
    { const pvlm_status s2 = pvlm_i_sync(ctx); if (!st) st = s2; }
    if (!st && info == -1) {                       // a wait inside the one-launch factorisation ran into its limit: say where (the step is reported as failed)
      unsigned w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const unsigned zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const bool got = hipMemcpyFromSymbol(w, HIP_SYMBOL(g_tail_timeout), sizeof w) == hipSuccess;
      if (got) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tail_timeout), zero8, sizeof zero8);        // the next event records its own site
      if (got)
      {
        fprintf(stderr, "pvlm_spd_solve_blocks: a wait of the one-launch factorisation ran into its limit (site %u, task / ticket %u, flag word %u of %d tasks + 3 x %d, tickets %u / %u, workgroup %u)\n",
                w[1], w[2], w[3], plan->flow_tasks, plan->flow_T, w[4], w[5], w[6]);
        std::vector<unsigned> wg(2048, 0);
        if (getenv("PVLM_SPD_FLOW_DEBUG") && hipMemcpyFromSymbol(wg.data(), HIP_SYMBOL(g_wg_state), wg.size() * 4) == hipSuccess) {
          std::vector<NdFlowTask> tk((size_t)plan->flow_tasks);
          (void)hipMemcpy(tk.data(), plan->d_flow_tasks, tk.size() * sizeof(NdFlowTask), hipMemcpyDeviceToHost);
          std::vector<unsigned> fl((size_t)2 + tk.size() + 3 * (size_t)plan->flow_T);
          (void)hipMemcpy(fl.data(), plan->d_tail_flags, fl.size() * 4, hipMemcpyDeviceToHost);
          for (int g = 0; g < 1024; ++g) {
            const unsigned task = wg[2 * g], ph = wg[2 * g + 1];
            if ((ph & 255u) >= 9u || task >= tk.size()) continue;
            const NdFlowTask& q = tk[task];
            fprintf(stderr, "  wg %4d task %6u (I %3d J %3d n_src %3d prev %6d final %d) phase %u source %u | own flag %u prev flag %d inv flag %u\n", g, task, q.I, q.J, q.n_src, q.prev, q.final_,
                    ph & 255u, ph >> 8, fl[2 + task], q.prev >= 0 ? (int)fl[2 + (size_t)q.prev] : -1, fl[2 + tk.size() + (size_t)q.J]);
          }
        }
      }
      else (void)hipGetLastError();
    }
    if (!st && info == -1 && plan->levels && plan->flow_T > 0) {
      // Not a property of the matrix: the launch did not get through (see tail_wait).  This context factorises by level launches from here on and this solve is
      // redone with them — same inputs (the caller's right-hand side has not been touched yet), a deterministic result again.
      ctx->spd_one_launch = 0; ctx->spd_fallbacks += 1;
      fprintf(stderr, "pvlm_spd_solve_blocks: redone with the level launches; this context keeps them (pvlm_spd_one_launch)\n");
      return pvlm_spd_solve_blocks(ctx, n_user, n_blocks, user_rows, user_cols, mirror, blocks, user_scale, user_diag, rhs, info_out);
    }
    pvlm_i_trace("spd solve: factorised and solved");
    if (!st && plan->sparse) for (int i = 0; i < n_user; ++i) rhs[i] = prhs[(size_t)plan->new_of_old[(size_t)i]];
    *info_out = info;
  } else {
    hipStreamSynchronize(ctx->stream);
  }
  return st;
}

// Starts the host half of the plan for this structure on a thread of its own and returns at once; the next pvlm_spd_solve_blocks with EXACTLY these lists takes it
// (any other structure: the prefetch is dropped and the plan made as before).  The lists are copied.  No device work here.
pvlm_status pvlm_spd_plan_prefetch(pvlm_ctx* ctx, int n, int n_blocks, const int* row_idx, const int* col_idx, const int* mirror) {
  if (!ctx || n < 0 || n_blocks < 0 || (n_blocks > 0 && (!row_idx || !col_idx || !mirror))) return PVLM_ERR_ARG;
  spd_prefetch_drop(ctx);
  // a structure the context already holds a plan for needs none
  const SpdPlan* have = static_cast<const SpdPlan*>(ctx->spd_plan);
  try {
    if (have && have->n == n && have->key_mirror.size() == (size_t)n_blocks && have->key_rows.size() == (size_t)n_blocks * 6 &&
        (n_blocks == 0 || (std::memcmp(have->key_rows.data(), row_idx, (size_t)n_blocks * 6 * sizeof(int)) == 0 && std::memcmp(have->key_cols.data(), col_idx, (size_t)n_blocks * 6 * sizeof(int)) == 0 &&
                           std::memcmp(have->key_mirror.data(), mirror, (size_t)n_blocks * sizeof(int)) == 0)))
      return PVLM_OK;
    SpdPrefetch* f = new SpdPrefetch();
    ctx->spd_prefetch = f;
    f->n = n; f->n_blocks = n_blocks;
    if (n_blocks > 0) { f->rows.assign(row_idx, row_idx + (size_t)n_blocks * 6); f->cols.assign(col_idx, col_idx + (size_t)n_blocks * 6); f->mirror.assign(mirror, mirror + n_blocks); }
    f->worker = std::thread([f]() {
      const auto t0 = std::chrono::steady_clock::now();
      try { spd_plan_host(f->n, f->n_blocks, f->rows.data(), f->cols.data(), &f->host); }
      catch (...) { f->failed = true; }
      f->worker_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();                       // out of memory on the thread: the solve makes its plan itself (and reports what it meets)
    });
  } catch (const std::bad_alloc&) {
    spd_prefetch_drop(ctx);
    PVLM_SET_ERR(ctx, "pvlm_spd_plan_prefetch: out of host memory");
    return PVLM_ERR_NOMEM;
  } catch (const std::system_error&) {                         // no thread to be had: not an error of the caller's, the solve plans for itself
    spd_prefetch_drop(ctx);
  }
  return PVLM_OK;
}
pvlm_status pvlm_spd_plan_prefetch_hits(const pvlm_ctx* ctx, long long* hits) {
  if (!ctx || !hits) return PVLM_ERR_ARG;
  *hits = ctx->spd_prefetch_hits;
  return PVLM_OK;
}

pvlm_status pvlm_spd_plan_schedule(const pvlm_ctx* ctx, int* levels, int* block_columns, int* padded_rows) {
  if (!ctx) return PVLM_ERR_ARG;
  const SpdPlan* p = static_cast<const SpdPlan*>(ctx->spd_plan);
  const bool lv = p && p->levels;
  if (levels) *levels = lv ? p->n_levels : 0;
  if (block_columns) *block_columns = lv ? p->n_pad / PVLM_CHOL_NB : (p ? (p->n + PVLM_CHOL_NB - 1) / PVLM_CHOL_NB : 0);
  if (padded_rows) *padded_rows = lv ? p->n_pad : (p ? p->n : 0);
  return PVLM_OK;
}

pvlm_status pvlm_spd_one_launch(pvlm_ctx* ctx, int enable, long long* fallbacks) {
  if (!ctx) return PVLM_ERR_ARG;
  if (enable >= 0) ctx->spd_one_launch = enable ? 1 : 0;
  ctx->spd_withhold_task = enable >= 2 ? enable - 2 : -1;          // test hook: enable = 2 + task makes that task of the next one-launch solves withhold its tile
  if (fallbacks) *fallbacks = ctx->spd_fallbacks;
  return PVLM_OK;
}

pvlm_status pvlm_spd_plan_tail(const pvlm_ctx* ctx, int* tail_block_columns, int* launched_levels) {
  if (!ctx) return PVLM_ERR_ARG;
  const SpdPlan* p = static_cast<const SpdPlan*>(ctx->spd_plan);
  const bool lv = p && p->levels, tl = lv && p->tail_T > 0, fl = lv && p->flow_T > 0;
  if (tail_block_columns) *tail_block_columns = fl ? p->flow_T * (64 / PVLM_CHOL_NB) : tl ? p->tail_T * (64 / PVLM_CHOL_NB) : 0;
  if (launched_levels) *launched_levels = fl ? 0 : tl ? p->n_main : (lv ? p->n_levels : 0);
  return PVLM_OK;
}

pvlm_status pvlm_spd_plan_info(const pvlm_ctx* ctx, int* tile_sparse, double* update_fraction) {
  if (!ctx) return PVLM_ERR_ARG;
  const SpdPlan* p = static_cast<const SpdPlan*>(ctx->spd_plan);
  if (tile_sparse) *tile_sparse = p && p->sparse ? 1 : 0;
  if (update_fraction) *update_fraction = p ? p->update_fraction : 1.0;
  return PVLM_OK;
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
  double *d_A = nullptr, *d_B = nullptr, *d_Linv = nullptr, *d_y = nullptr; int* d_info = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_A, (size_t)n * n);
  if (!st) st = pvlm_i_alloc(ctx, &d_B, (size_t)n * nrhs);
  if (!st) st = pvlm_i_alloc(ctx, &d_Linv, (size_t)((n + PVLM_CHOL_NB - 1) / PVLM_CHOL_NB) * PVLM_CHOL_NB * PVLM_CHOL_NB);
  if (!st) st = pvlm_i_alloc(ctx, &d_y, (size_t)n);
  if (!st) st = pvlm_i_alloc(ctx, &d_info, (size_t)1);
  if (!st) {
    hipError_t e = hipMemcpyAsync(d_A, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_B, B, (size_t)n * nrhs * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    int info = 0;
    if (e == hipSuccess) e = hipMemsetAsync(d_info, 0, sizeof(int), ctx->stream);
    if (e == hipSuccess) {
      chol_factor_solve(ctx, n, d_A, d_B, nrhs, d_Linv, d_y, d_info);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(B, d_B, (size_t)n * nrhs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    *info_out = info;
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_spd_solve: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  hipStreamSynchronize(ctx->stream);
  pvlm_i_free(ctx, d_A); pvlm_i_free(ctx, d_B); pvlm_i_free(ctx, d_Linv); pvlm_i_free(ctx, d_y); pvlm_i_free(ctx, d_info);
  return st;
}

}  // extern "C"

// pvlm_preload: HIP loads the code object of a translation unit at the first launch of one of its kernels (15 ms for the larger ones) — an empty launch from here
// moves that out of the first call that needs this file's kernels
__global__ void k_preload_linalg() {}
void pvlm_i_preload_linalg(hipStream_t s) { hipLaunchKernelGGL(k_preload_linalg, dim3(1), dim3(1), 0, s); }
