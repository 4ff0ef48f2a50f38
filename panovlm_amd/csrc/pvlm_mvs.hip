// Scoring pass of the reference's panoramic PatchMatch MVS: MVS::InitPatchMap + MVS::InitConfMap (mvs/MVS.cpp:586-636), photometric
// and (use_geometry) geometric-consistency terms = FillPixelPatch (:637-680) + the photometric term of ScorePixel (:774-923) for the current
// depth / normal hypothesis of every pixel.  First kernel of SURVEY.md §8 N4's second part.
//
// One wave per reference pixel, one lane per texel of the NCC window (7 x 7 = 49 at the Room settings; larger windows
// loop): the bilateral weights, the plane-induced homography H = R_nr + t_nr n^T / d, the equirectangular re-projection
// (the same FastAtan2 arithmetic as K7) and the bilinear samples are per-lane work, the weighted means / variances are
// summed in the reference's own sequential order (strip_seq_sum* below).
// Threshold decisions (d > 0, projection inside the image, sq0 <= 1e-6) use the reference's float arithmetic —
// compiled with -ffp-contract=off.  Images stay in HBM as uint8; the PreComputeI2C table is built on the device.
#include <algorithm>
#include <vector>

#include "pvlm_internal.h"

#define PVLM_HD __host__ __device__
#ifndef PVLM_MVS_FLOW_CLOCK
#define PVLM_MVS_FLOW_CLOCK 0      // measured variant: see FLOW_CLOCK_* below
#endif
#if PVLM_MVS_FLOW_CLOCK
__device__ unsigned long long g_flow_clock[24];
#define PVLM_MVS_SPEC_STAT(i) do { if (threadIdx.x == 0) atomicAdd(&g_flow_clock[16 + (i)], 1ull); } while (0)
#endif
#include "pvlm_mvs_core.h"

#define PVLM_MVS_MAXM 4   // texels per lane: windows up to 256 texels.  Every wave-level piece below is a template on the
                          // texels per lane M: M = 1 (windows up to 64 texels: the reference's 7 x 7 and 5 x 5) keeps one texel,
                          // one weight and four neighbour samples per lane in registers and 3 KB of LDS per wave, M = 4 the rest

// Sums of per-texel values IN INDEX ORDER: s = 0; s += v_0; s += v_1; ... — exactly the reference's sequential float loops
// (mvs/MVS.cpp:659-673, :826-833).  A wave tree would be six steps instead of n, but float addition does not associate: on
// texture-less windows sq0, sq1 and nrm = sq0 * sq1 are sums of rounding residues, and the reference's own tests
// `sq0 > 0` (:602), `nrm <= 0` (:835) then depend on the order of the additions — measured at 5.7K, a tree order took the
// other branch on 0.4 % of the pixels; near-ties between PatchMatch hypotheses fall the other way for the same reason.
// Every lane parks its values in the wave's LDS strip (one float4 per texel = the same quantity for four neighbour images);
// then lane c walks COLUMN c of the strip — lanes 0-3 the four neighbours of one quantity, lanes 0-7 two quantities at
// once — with one 4-byte LDS read and one add per texel, and the totals are handed to every lane with v_readlane.  History:
// v_readlane chains over the registers cost 2.5x the rest of the kernel; all 64 lanes walking the strip with broadcast
// 16-byte reads and adding all four columns (twelve dependent add chains per neighbour group, every lane redundantly)
// took K11 from 2.75 to 5.2 ms; one chain per lane does the same additions in the same order in a sixth of the issue slots.
#define PVLM_MVS_STRIP(M) (64 * (M))                       // texels per strip
#define PVLM_MVS_LDS_PER_WAVE(M) (3 * PVLM_MVS_STRIP(M))   // float4 elements: products A | products B1 | products B2
__device__ __forceinline__ float lane_bcast(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }
// column 0 of the strip (one quantity, e.g. of the reference patch): every lane walks it, every lane ends with the sum
__device__ inline float strip_seq_sum1(const float4* __restrict__ strip, int n) {
  const float* f = reinterpret_cast<const float*>(strip);
  __asm__ volatile("" ::: "memory");             // the strip was just written through another pointer type
  float s = 0.f;
#pragma unroll 8
  for (int i = 0; i < n; ++i) s += f[4 * i];
  return s;
}
// the four columns of one strip: lane c (mod 4) walks column c
__device__ inline void strip_seq_sum4(const float4* __restrict__ strip, int n, int lane, float out[4]) {
  const float* f = reinterpret_cast<const float*>(strip) + (lane & 3);
  __asm__ volatile("" ::: "memory");
  float s = 0.f;
#pragma unroll 8
  for (int i = 0; i < n; ++i) s += f[4 * i];
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = lane_bcast(s, j);
}
// the four columns of two strips (b follows a at a fixed distance): lane c (mod 8) walks column c & 3 of strip c >> 2
__device__ inline void strip_seq_sum8(const float4* __restrict__ a, const float4* __restrict__ b, int n, int lane, float sa[4], float sb[4]) {
  const float* f = reinterpret_cast<const float*>((lane & 4) ? b : a) + (lane & 3);
  __asm__ volatile("" ::: "memory");
  float s = 0.f;
#pragma unroll 8
  for (int i = 0; i < n; ++i) s += f[4 * i];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sa[j] = lane_bcast(s, j); sb[j] = lane_bcast(s, 4 + j); }
}

__global__ void k_mvs_unit_table(int rows, int cols, float* __restrict__ unit) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)rows * cols) return;
  pvlm_mvs::unit_ray(rows, cols, (int)(e % cols), (int)(e / cols), unit + 3 * e);
}

// host <-> device copies of whole maps through the context's pinned staging arena (a pageable copy of a 5.7K map locks and
// unlocks 66 MB of the caller's pages); downloads reach the caller's buffer at mvs_sync
static inline hipError_t mvs_up(pvlm_ctx* ctx, void* d, const void* h, size_t bytes) { return pvlm_i_h2d_q(ctx, d, h, bytes) == PVLM_OK ? hipSuccess : hipErrorUnknown; }
static inline hipError_t mvs_down(pvlm_ctx* ctx, void* h, const void* d, size_t bytes) { return pvlm_i_d2h_q(ctx, h, d, bytes) == PVLM_OK ? hipSuccess : hipErrorUnknown; }
static inline hipError_t mvs_sync(pvlm_ctx* ctx) { return pvlm_i_sync(ctx) == PVLM_OK ? hipSuccess : hipErrorUnknown; }

// neighbour images of one reference view: quad[b] = the 2x2-quad copy of grey image b (pvlm_mvs_core.h, QuadImage) — what every tap reads
struct pvlm_mvs_neighbours {
  const unsigned* quad[16]; const float* depth[16]; float R[16][9]; float t[16][3]; int n; int geometric;
  __host__ __device__ pvlm_mvs::QuadImage image(int b) const { return pvlm_mvs::QuadImage{quad[b]}; }
};
// quad copy of a grey image: word (x, y) = I(x,y) | I(x+1,y) << 8 | I(x,y+1) << 16 | I(x+1,y+1) << 24 (the last column / row repeat themselves: a
// tap that is used never starts there — frame.IsInside(x1, 1, 1))
__global__ __launch_bounds__(256) void k_mvs_quad(int rows, int cols, const unsigned char* __restrict__ gray, unsigned* __restrict__ quad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int y = (int)(i / cols), x = (int)(i - (size_t)y * cols);
  const int x1 = x + 1 < cols ? x + 1 : x, y1 = y + 1 < rows ? y + 1 : y;
  quad[i] = (unsigned)gray[(size_t)y * cols + x] | ((unsigned)gray[(size_t)y * cols + x1] << 8) | ((unsigned)gray[(size_t)y1 * cols + x] << 16) |
            ((unsigned)gray[(size_t)y1 * cols + x1] << 24);
}

// MEASURED VARIANT (-DPVLM_MVS_FLOW_CLOCK=1): where a pixel of K13q spends its time, summed over all workgroups by thread 0 with the
// 100 MHz wall clock: [0] ticket + walk + patch preparation, [1] waiting for the two stamps, [2] the dependent chain, [3] store + stamp,
// [4] pixels, [5] batch rounds; inside the chain: [8] close neighbours (fence -> first batch), [9] hypothesis build, [10] score_hypothesis,
// [11] barrier + exchange, and inside wave_score [12] homographies + neighbour texels, [13] strip sums, [14] NCC + geometric adjustment; printed by mvs_flow_end.
#if PVLM_MVS_FLOW_CLOCK
#define FLOW_CLOCK_NOW() wall_clock64()
#define FLOW_CLOCK_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_flow_clock[i], (unsigned long long)(v)); } while (0)
#else
#define FLOW_CLOCK_NOW() 0ull
#define FLOW_CLOCK_ADD(i, v) do { (void)(v); } while (0)
#endif
// ---- wave-level pieces shared by the scoring pass and the PatchMatch sweep (one wave per pixel, lane = texel) ----
template <int M> struct PatchRegs { float w[M], t0[M]; float sq0; bool inside; };

// FillPixelPatch (mvs/MVS.cpp:637-680).  lds: the wave's strip (PVLM_MVS_LDS_PER_WAVE(M) float4).
template <int M>
__device__ inline void wave_fill_patch(const unsigned char* __restrict__ ref_gray, int rows, int cols, int px, int py, int half_window, int step, int n, int lane,
                                       float4* __restrict__ lds, PatchRegs<M>& P) {
  P.inside = px >= half_window && py >= half_window && px < cols - half_window && py < rows - half_window;
  P.sq0 = 0.f;
#pragma unroll
  for (int m = 0; m < M; ++m) { P.w[m] = 0.f; P.t0[m] = 0.f; }
  if (!P.inside) return;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int k = lane + 64 * m;
    if (k < n) { pvlm_mvs::patch_texel(ref_gray, cols, px, py, half_window, step, k, &P.w[m], &P.t0[m]); lds[k] = make_float4(P.w[m], 0.f, 0.f, 0.f); }
  }
  const float wsum = strip_seq_sum1(lds, n);                  // accumulate(weight.begin(), weight.end(), 0.f)   :659
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int k = lane + 64 * m;
    P.w[m] /= wsum;
    if (k < n) lds[k] = make_float4(P.w[m] * P.t0[m], 0.f, 0.f, 0.f);
  }
  const float mean = strip_seq_sum1(lds, n);                  // sum += weight[i] * texels0[i]                  :662-664
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int k = lane + 64 * m;
    if (k < n) { P.t0[m] -= mean; const float tmp = P.t0[m] * P.w[m]; lds[k] = make_float4(P.t0[m] * tmp, 0.f, 0.f, 0.f); P.t0[m] = tmp; } else P.t0[m] = 0.f;
  }
  P.sq0 = strip_seq_sum1(lds, n);                             // sq0 += texels0[i] * tmp                         :668-672
}

// ScorePixel (mvs/MVS.cpp:774-923) for one hypothesis (nrm3, dep) of pixel (px, py): photometric NCC per neighbour image,
// optional smoothness factors (n_close > 0) and geometric-consistency adjustment (nb.geometric), best-two average.
// Every lane returns the same value.
template <int M>
__device__ inline float wave_score(int rows, int cols, int half_window, int step, int n, int lane, const float* __restrict__ unit,
                                   const pvlm_mvs_neighbours& nb, int px, int py, const PatchRegs<M>& P, const float* nrm3, float dep, const float* factors,
                                   int n_close, float4* __restrict__ lds) {
  const float* u0 = unit + 3 * ((size_t)py * cols + px);
  const float X0[3] = {u0[0] * dep, u0[1] * dep, u0[2] * dep};
  const float d = X0[0] * nrm3[0] + X0[1] * nrm3[1] + X0[2] * nrm3[2];
  if (d > 0) return -1.f;
  float best1 = 0.f, best2 = 0.f; int count = 0;
  float4* sA = lds; float4* sB1 = lds + PVLM_MVS_STRIP(M); float4* sB2 = lds + 2 * PVLM_MVS_STRIP(M);
  // the unit rays of this lane's texels: one read per scoring (inside the loop over the neighbour images the compiler re-read them per image,
  // each time a load the projection had to wait for).  Lanes past the window's last texel repeat it (M = 1: 49 of 64 lanes carry a texel):
  // same verdict, value unused — an `if (k < n)` per piece of the chain is an exec-mask save / restore each time
  float uv[M][3];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float* up = pvlm_mvs::texel_ray(unit, cols, px, py, half_window, step, min(lane + 64 * m, n - 1));
    uv[m][0] = up[0]; uv[m][1] = up[1]; uv[m][2] = up[2];
  }
  for (int b0 = 0; b0 < nb.n; b0 += 4) {                                 // four neighbour images per pass: one float4 per texel in LDS
    float t1[4][M];
    bool okj[4];
    const unsigned long long wk0 = FLOW_CLOCK_NOW();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool ok = b0 + j < nb.n;
      if (ok) {
        float H[9];
        pvlm_mvs::homography(nb.R[b0 + j], nb.t[b0 + j], nrm3, d, H);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          t1[j][m] = 0.f;
          ok = pvlm_mvs::neighbour_texel_ray(uv[m], nb.image(b0 + j), rows, cols, H, &t1[j][m]) && ok;
        }
      } else {
#pragma unroll
        for (int m = 0; m < M; ++m) t1[j][m] = 0.f;
      }
      okj[j] = !__any(!ok);                                              // goto next_image
    }
    if (!(okj[0] || okj[1] || okj[2] || okj[3])) continue;
    const unsigned long long wk1 = FLOW_CLOCK_NOW();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;                              // strips hold 64 M entries: the surplus lanes write slots the sums never read
      sA[k] = make_float4(t1[0][m] * P.w[m], t1[1][m] * P.w[m], t1[2][m] * P.w[m], t1[3][m] * P.w[m]);
    }
    float sj[4];
    strip_seq_sum4(sA, n, lane, sj);                                     // sum += texels1[i] * weight[i]            :826-827
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;
      {
        float p1[4], p01[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { t1[j][m] -= sj[j]; p1[j] = t1[j][m] * t1[j][m] * P.w[m]; p01[j] = P.t0[m] * t1[j][m]; }
        sB1[k] = make_float4(p1[0], p1[1], p1[2], p1[3]);
        sB2[k] = make_float4(p01[0], p01[1], p01[2], p01[3]);
      }
    }
    float sq1j[4], sq01j[4];
    strip_seq_sum8(sB1, sB2, n, lane, sq1j, sq01j);                      // :830-831, :834-835
    const unsigned long long wk2 = FLOW_CLOCK_NOW();
    // NCC -> clamp -> smoothness -> (use_geometry) geometric-consistency adjustment of neighbour j in lane j (mod 4): the
    // adjustment is ~600 wave-uniform instructions per neighbour (a projection, a depth sample, a back-projection with three
    // double-evaluated sin / cos and an acos) — four lanes do the four neighbours of the pass in one go
    const int jl = lane & 3;
    float sq1 = sq1j[0], sq01 = sq01j[0]; bool okl = okj[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (jl == j) { sq1 = sq1j[j]; sq01 = sq01j[j]; okl = okj[j]; }
    const float nrm = P.sq0 * sq1;
    const bool valid = okl && !(nrm <= 0.f);                              // goto next_image / `if (nrm <= 0) continue`
    float score = 0.f;
    if (valid) {
      score = sq01 / sqrtf(nrm);
      score = fminf(fmaxf(score, -1.f), 1.f);
      score = pvlm_mvs::smooth_score_static(score, factors, n_close);
      if (nb.geometric) {
        float Rl[9], tl[3]; const float* dl = nb.depth[b0];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rl[k] = nb.R[b0][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tl[k] = nb.t[b0][k];
#pragma unroll
        for (int j = 1; j < 4; ++j)
          if (jl == j) {                                                 // b0 + j <= 15: inside the tables; a slot past nb.n is never `valid`
#pragma unroll
            for (int k = 0; k < 9; ++k) Rl[k] = nb.R[b0 + j][k];
#pragma unroll
            for (int k = 0; k < 3; ++k) tl[k] = nb.t[b0 + j][k];
            dl = nb.depth[b0 + j];
          }
        score = pvlm_mvs::geometric_adjust(score, rows, cols, X0, Rl, tl, dl);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                        // the neighbours in their own order, as the reference's loop visits them
      if (!__builtin_amdgcn_readlane((int)valid, j)) continue;
      const float sc = lane_bcast(score, j);
      if (count == 0 || sc > best1) { best2 = best1; best1 = sc; } else if (count == 1 || sc > best2) best2 = sc;
      ++count;
    }
    FLOW_CLOCK_ADD(12, wk1 - wk0); FLOW_CLOCK_ADD(13, wk2 - wk1); FLOW_CLOCK_ADD(14, FLOW_CLOCK_NOW() - wk2);
  }
  if (count == 1) return best1;
  if (count >= 2) { float avg = 0.f; avg += best1; avg += best2; return avg / 2; }
  return -1.f;
}

template <int M>
__global__ __launch_bounds__(256) void k_mvs_conf(int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray,
                                                  const float* __restrict__ unit, pvlm_mvs_neighbours nb, float* __restrict__ depth,
                                                  float* __restrict__ normal, float* __restrict__ conf) {
  const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per pixel
  const int lane = threadIdx.x & 63;
  if (e >= (long long)rows * cols) return;
  const float dep = depth[e];
  if (dep <= 0) return;                                                   // InitConfMap :594-595
  const int py = (int)(e / cols), px = (int)(e % cols);
  const int n = pvlm_mvs::num_texels(half_window, step);
  float c = -1.f;
  __shared__ float4 strips[4][PVLM_MVS_LDS_PER_WAVE(M)];
  float4* lds = strips[threadIdx.x >> 6];
  PatchRegs<M> P;
  wave_fill_patch<M>(ref_gray, rows, cols, px, py, half_window, step, n, lane, lds, P);
  if (P.inside && P.sq0 > 0) {   // InitConfMap :602 tests sq0 > 0 only (InitPatchMap ignores FillPixelPatch's 1e-6 verdict; that gate is PropagateCheckerBoard's, :1116)
    const float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
    c = wave_score<M>(rows, cols, half_window, step, n, lane, unit, nb, px, py, P, nrm3, dep, nullptr, 0, lds);
  }
  if (lane == 0) {
    conf[e] = c;
    if (c <= -1) { depth[e] = 0; normal[3 * e] = 0; normal[3 * e + 1] = 0; normal[3 * e + 2] = 0; }
  }
}

// PatchMatch sweep, one colour of the checkerboard (PropagateCheckerBoard :1098-1129): pixels of colour `offset` read
// the depth / normal of the other colour and update their own, so one launch is race-free.
template <int M>
struct WaveScorer {
  int rows, cols, half_window, step, n, lane, px, py;
  const float* unit; const pvlm_mvs_neighbours* nb; const PatchRegs<M>* P; float4* lds;
  // close neighbour c's factor in lane c (mod 4): one evaluation of the two exp + one acos for the wave instead of four in a row
  __device__ void smooth_factors(const float* plane, const pvlm_mvs::ClosePixel* close, int n_close, const float* normal, float depth, float* factors) const {
    if (n_close <= 0) return;
    const int c = lane & 3;
    pvlm_mvs::ClosePixel mine = close[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (c == q) mine = close[q];                       // entries past n_close are zero-initialised and their factor is never read
    const float f = pvlm_mvs::smooth_factor(plane, mine, normal, depth);
#pragma unroll
    for (int q = 0; q < 4; ++q) factors[q] = lane_bcast(f, q);
  }
  // angle k in lane k (mod 4)
  __device__ void sincos3(const float* a, float* sn, float* cs) const {
    const int c = lane & 3;
    float mine = a[0];
    if (c == 1) mine = a[1];
    if (c == 2) mine = a[2];
    const float s = pvlm_mvs::f_sin(mine), co = pvlm_mvs::f_cos(mine);
#pragma unroll
    for (int k = 0; k < 3; ++k) { sn[k] = lane_bcast(s, k); cs[k] = lane_bcast(co, k); }
  }
  __device__ float operator()(const float* nrm3, float dep, const float* factors, int n_close) const {
    return wave_score<M>(rows, cols, half_window, step, n, lane, unit, *nb, px, py, *P, nrm3, dep, factors, n_close, lds);
  }
};
// M = 1: 169 VGPRs fall two registers short of three waves per SIMD; asking for three costs 2 spilled registers
#ifndef PVLM_K13_WAVES
#define PVLM_K13_WAVES 3
#endif
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M == 1 ? PVLM_K13_WAVES : 2))) void k_mvs_propagate(int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray,
                                                       const float* __restrict__ unit, pvlm_mvs_neighbours nb, float* depth, float* normal, float* conf,
                                                       const unsigned char* __restrict__ depth_constant, float min_depth, float max_depth,
                                                       unsigned long long pass_seed, int offset) {
  const int half = (cols + 1) / 2;
  const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per pixel of this colour
  const int lane = threadIdx.x & 63;
  if (wv >= (long long)rows * half) return;
  const int py = (int)(wv / half);
  const int px = ((py % 2 + offset) % 2) + 2 * (int)(wv % half);
  if (px >= cols) return;
  const long long e = (long long)py * cols + px;
  float dep = depth[e];
  if (dep <= 0) return;
  const int n = pvlm_mvs::num_texels(half_window, step);
  __shared__ float4 strips[4][PVLM_MVS_LDS_PER_WAVE(M)];
  float4* lds = strips[threadIdx.x >> 6];
  PatchRegs<M> P;
  wave_fill_patch<M>(ref_gray, rows, cols, px, py, half_window, step, n, lane, lds, P);
  if (!P.inside || P.sq0 <= 1e-6) return;                                 // patch.sq0 <= 1e-6 (:1116-1117; patches outside the margin have sq0 = 0)
  float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
  float c = conf[e];
  pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, depth_constant, min_depth, max_depth};
  pvlm_mvs::Rng rng{pass_seed, (unsigned long long)e, 0u};
  WaveScorer<M> scorer{rows, cols, half_window, step, n, lane, px, py, unit, &nb, &P, lds};
  const int pdx[4] = {-1, 0, 1, 0}, pdy[4] = {0, -1, 0, 1};
  pvlm_mvs::process_pixel(A, rng, px, py, scorer, dep, nrm3, c, 4, pdx, pdy);
  if (lane == 0) { depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c; }
}

// The same colour pass with one pixel per THREAD (pvlm_mvs::ColumnScorer): the reference's own per-pixel program — window texel after
// texel, sums in index order — run by 64 pixels side by side.  Against the wave-per-pixel kernel above: all 64 lanes carry texels
// (there 49 of 64 on a 7 x 7 window), nothing wave-uniform is evaluated 64 times (homographies, perturbation trigonometry, the
// smoothness factors, the decision logic of process_pixel), and the index-ordered sums are a register add per texel instead of an
// LDS strip walked by one lane.  The neighbour texels of the image being scored are a column of the workgroup's LDS table ([k][lane]:
// bank-conflict free, 12.5 KB per wave at 7 x 7, i.e. three waves per SIMD); the patch weights a column of `wtab` ([k][pixel of the
// band of rows this launch serves], n x band_rows x ceil(cols / 2) floats of scratch, mvs_lane_bands).  Windows of at most 64 texels.
#ifndef PVLM_K13L_WAVES
#define PVLM_K13L_WAVES 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PVLM_K13L_WAVES, 4))) void k_mvs_propagate_lane(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray, const float* __restrict__ unit, pvlm_mvs_neighbours nb, float* depth,
    float* normal, float* conf, const unsigned char* __restrict__ depth_constant, float min_depth, float max_depth, unsigned long long pass_seed, int offset,
    float* __restrict__ wtab, int row0, int band_rows) {
  extern __shared__ float lane_tab[];                                      // [n][64]
  const int half = (cols + 1) / 2;
  // a workgroup is an 8 x 8 TILE of pixels of this colour (16 image columns x 8 rows) inside the band of rows [row0, row0 + band_rows):
  // the 64 windows overlap — a third of the cache lines a 64-pixel run along one row touches
  const int tiles_x = (half + 7) / 8;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int ry = ty * 8 + (int)(threadIdx.x >> 3), hx = tx * 8 + (int)(threadIdx.x & 7);
  if (ry >= band_rows || hx >= half) return;
  const long long wv = (long long)blockIdx.x * 64 + threadIdx.x;           // this thread's column of the weight table
  const int py = row0 + ry;
  const int px = ((py % 2 + offset) % 2) + 2 * hx;
  if (px >= cols) return;
  const long long e = (long long)py * cols + px;
  float dep = depth[e];
  if (dep <= 0) return;
  const int n = pvlm_mvs::num_texels(half_window, step);
  pvlm_mvs::ColumnPatch P{wtab + wv, (size_t)gridDim.x * 64, lane_tab + threadIdx.x, 64, 0.f, 0.f, false};
  pvlm_mvs::fill_patch_column(ref_gray, rows, cols, px, py, half_window, step, n, P);
  if (!P.inside || P.sq0 <= 1e-6) return;                                 // patch.sq0 <= 1e-6 (:1116-1117)
  float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
  float c = conf[e];
  pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, depth_constant, min_depth, max_depth};
  pvlm_mvs::Rng rng{pass_seed, (unsigned long long)e, 0u};
  pvlm_mvs::ColumnScorer<pvlm_mvs_neighbours> scorer{{}, {}, rows, cols, half_window, step, n, px, py, unit, ref_gray, &nb, P};
  const int pdx[4] = {-1, 0, 1, 0}, pdy[4] = {0, -1, 0, 1};
  pvlm_mvs::process_pixel(A, rng, px, py, scorer, dep, nrm3, c, 4, pdx, pdy);
  depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c;
}
// ... and the scoring pass (InitConfMap, k_mvs_conf) in the same form: every pixel of the band, one ScorePixel each
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PVLM_K13L_WAVES, 4))) void k_mvs_conf_lane(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray, const float* __restrict__ unit, pvlm_mvs_neighbours nb,
    float* __restrict__ depth, float* __restrict__ normal, float* __restrict__ conf, float* __restrict__ wtab, int row0, int band_rows) {
  extern __shared__ float lane_tab[];
  const int tiles_x = (cols + 7) / 8;                                      // 8 x 8 pixel tiles, as in k_mvs_propagate_lane
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int ry = ty * 8 + (int)(threadIdx.x >> 3), px = tx * 8 + (int)(threadIdx.x & 7);
  if (ry >= band_rows || px >= cols) return;
  const long long wv = (long long)blockIdx.x * 64 + threadIdx.x;
  const int py = row0 + ry;
  const long long e = (long long)py * cols + px;
  const float dep = depth[e];
  if (dep <= 0) return;                                                   // InitConfMap :594-595
  const int n = pvlm_mvs::num_texels(half_window, step);
  pvlm_mvs::ColumnPatch P{wtab + wv, (size_t)gridDim.x * 64, lane_tab + threadIdx.x, 64, 0.f, 0.f, false};
  pvlm_mvs::fill_patch_column(ref_gray, rows, cols, px, py, half_window, step, n, P);
  float c = -1.f;
  if (P.inside && P.sq0 > 0) {                                            // :602 tests sq0 > 0 only
    const float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
    pvlm_mvs::ColumnScorer<pvlm_mvs_neighbours> scorer{{}, {}, rows, cols, half_window, step, n, px, py, unit, ref_gray, &nb, P};
    c = scorer(nrm3, dep, nullptr, 0);
  }
  conf[e] = c;
  if (c <= -1) { depth[e] = 0; normal[3 * e] = 0; normal[3 * e + 1] = 0; normal[3 * e + 2] = 0; }
}

// Sequential sweep (PropagateSequential :1057-1097, the strategy config/Room.txt and config/Floor.txt select: propagate_strategy = 2):
// upstream walks the image in raster order (even iterations; odd ones backwards) and a pixel takes the hypotheses of its left and
// upper neighbour, which that very walk has just updated.  A pixel only ever reads its four direct neighbours, so all pixels of an
// anti-diagonal col + row = d are independent of each other and see exactly what the raster walk shows them: the neighbours on
// d - 1 already updated, those on d + 1 not yet.  One launch per anti-diagonal (ascending d, or descending for the backward
// walk), one wave per pixel: the same result as the sequential loop, bit for bit.
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M == 1 ? PVLM_K13_WAVES : 2))) void k_mvs_propagate_diag(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray, const float* __restrict__ unit, pvlm_mvs_neighbours nb, float* depth,
    float* normal, float* conf, const unsigned char* __restrict__ depth_constant, float min_depth, float max_depth, unsigned long long pass_seed, int diag, int backward) {
  const int r0 = max(0, diag - (cols - 1)), r1 = min(rows - 1, diag);
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int py = r0 + w;
  if (py > r1) return;
  const int px = diag - py;
  const long long e = (long long)py * cols + px;
  float dep = depth[e];
  if (dep <= 0) return;
  const int n = pvlm_mvs::num_texels(half_window, step);
  __shared__ float4 strips[4][PVLM_MVS_LDS_PER_WAVE(M)];
  float4* lds = strips[threadIdx.x >> 6];
  PatchRegs<M> P;
  wave_fill_patch<M>(ref_gray, rows, cols, px, py, half_window, step, n, lane, lds, P);
  if (!P.inside || P.sq0 <= 0) return;                                    // patch.sq0 <= 0 (:1069, :1087)
  float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
  float c = conf[e];
  pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, depth_constant, min_depth, max_depth};
  pvlm_mvs::Rng rng{pass_seed, (unsigned long long)e, 0u};
  WaveScorer<M> scorer{rows, cols, half_window, step, n, lane, px, py, unit, &nb, &P, lds};
  const int sgn = backward ? 1 : -1;
  const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};                          // (col - 1, row), (col, row - 1)  /  (col + 1, row), (col, row + 1)
  pvlm_mvs::process_pixel(A, rng, px, py, scorer, dep, nrm3, c, 2, pdx, pdy);
  if (lane == 0) { depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c; }
}
// K13p — the same sweep as ONE persistent launch per iteration, ordered by data flow instead of by diagonal.
// Round 3 launched one kernel per anti-diagonal (2159 at 1440 x 720, 8639 at 5760 x 2880): every diagonal waited for its slowest pixel
// and for a launch, and every pixel paid its state-independent preparation (patch statistics, texel products) inside that window.
// A pixel only needs its two predecessors (left / up of the walk) to be FINAL; everything of the preparation depends on the reference
// image alone.  Here a wave draws pixels from a ticket counter in walking order (diagonal after diagonal — so the wave that owns a
// predecessor drew it earlier and is running: no deadlock whatever the number of resident waves), prepares the pixel, THEN waits for the
// state of its two predecessors, runs the dependent chain and publishes its own.  The critical path per diagonal is the chain of dependent
// scorings only; preparation and launch latency are off it.  Neighbours on the NEXT diagonal are read before they change: they wait for
// this pixel.  Results: those of the diagonal launches, bit for bit.  The hand-off itself is described at k_mvs_propagate_flow_spec below:
// four 8-byte {value, epoch} cells per pixel, agent-scope atomics on both sides, no fence.
__device__ __forceinline__ void mvs_walk_pixel(long long t, int rows, int cols, int* wx, int* wy) {     // ticket -> pixel in the walk's own frame
  const long long m = rows < cols ? rows : cols, mx = rows < cols ? cols : rows, npix = (long long)rows * cols;
  const long long ta = m * (m - 1) / 2, tb = ta + (mx - m + 1) * m;
  auto tri = [](long long u, long long* d, long long* k) {                  // u-th element of 1 + 2 + 3 + ...: row d, position k
    long long q = (long long)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
    while ((q + 1) * (q + 2) / 2 <= u) ++q;
    while (q * (q + 1) / 2 > u) --q;
    *d = q; *k = u - q * (q + 1) / 2;
  };
  long long d, k;
  if (t < ta) tri(t, &d, &k);
  else if (t < tb) { const long long u = t - ta; d = m - 1 + u / m; k = u % m; }
  else { long long dd, kk; tri(npix - 1 - t, &dd, &kk); d = (long long)rows + cols - 2 - dd; k = dd - kk; }
  const long long r0 = d - (cols - 1) > 0 ? d - (cols - 1) : 0;
  *wy = (int)(r0 + k); *wx = (int)(d - (r0 + k));
}
#ifndef PVLM_MVS_FLOW_SLEEP
#define PVLM_MVS_FLOW_SLEEP 8     // x 64 cycles between two polls of a stamp
#endif
// the four {value, epoch} cells of a pixel: depth, normal x / y / z
__device__ __forceinline__ unsigned long long mvs_cell(int epoch, float f) { return ((unsigned long long)(unsigned)epoch << 32) | pvlm_mvs::float_bits(f); }
// State of the four direct neighbours for the pixel of this wave (all 64 lanes call it; every lane returns the whole Around): lane 4 q + k
// fetches value k of slot q — from the cells, polled until they carry this iteration's epoch, for the two predecessors of the walk; from
// the maps for the other two (and nothing for a slot outside the image).
__device__ inline void mvs_wave_around(int rows, int cols, int px, int py, int sgn, int lane, const float* depth, const float* normal, const unsigned long long* cell,
                                       int epoch, pvlm_mvs::Around& ar) {
  const int q = (lane >> 2) & 3, k = lane & 3;
  const int cx = px + (q == 0 ? -1 : (q == 3 ? 1 : 0)), cy = py + (q == 1 ? -1 : (q == 2 ? 1 : 0));
  const bool in = cx >= 0 && cy >= 0 && cx < cols && cy < rows;
  const bool pred = q == pvlm_mvs::around_slot(sgn, 0) || q == pvlm_mvs::around_slot(0, sgn);
  const long long ne = in ? (long long)cy * cols + cx : 0;
  float f = 0.f;
  if (lane < 16 && in && !pred) f = k == 0 ? depth[ne] : normal[3 * ne + k - 1];
  const bool need = lane < 16 && in && pred;
  bool have = !need;
  while (true) {
    if (!have) {
      const unsigned long long v = __hip_atomic_load(cell + 4 * ne + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      have = (int)(v >> 32) == epoch;
      f = pvlm_mvs::bits_float((unsigned)v);
    }
    if (__all(have)) break;
    __builtin_amdgcn_s_sleep(PVLM_MVS_FLOW_SLEEP);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ar.inside[s] = __builtin_amdgcn_readlane((int)in, 4 * s);
    ar.depth[s] = lane_bcast(f, 4 * s);
    ar.normal[s][0] = lane_bcast(f, 4 * s + 1); ar.normal[s][1] = lane_bcast(f, 4 * s + 2); ar.normal[s][2] = lane_bcast(f, 4 * s + 3);
  }
}
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M == 1 ? PVLM_K13_WAVES : 2))) void k_mvs_propagate_flow(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray, const float* __restrict__ unit, pvlm_mvs_neighbours nb, float* depth,
    float* normal, float* conf, const unsigned char* __restrict__ depth_constant, float min_depth, float max_depth, unsigned long long pass_seed, int backward,
    unsigned long long* ticket, unsigned long long* cell, int epoch) {
  const int lane = threadIdx.x & 63;
  const int n = pvlm_mvs::num_texels(half_window, step);
  __shared__ float4 strips[4][PVLM_MVS_LDS_PER_WAVE(M)];
  __shared__ pvlm_mvs::ClosePixel close_s[4][4];
  float4* lds = strips[threadIdx.x >> 6];
  const long long npix = (long long)rows * cols;
  const int sgn = backward ? 1 : -1;
  while (true) {
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1ull);
    t = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    if ((long long)t >= npix) return;
    int wx, wy;
    mvs_walk_pixel((long long)t, rows, cols, &wx, &wy);
    const int px = backward ? cols - 1 - wx : wx, py = backward ? rows - 1 - wy : wy;
    const long long e = (long long)py * cols + px;
    // ---- preparation: nothing here reads what the sweep writes (the pixel's own state is only written by this wave)
    float dep = depth[e];
    float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
    bool live = dep > 0;
    PatchRegs<M> P;
    if (live) {
      wave_fill_patch<M>(ref_gray, rows, cols, px, py, half_window, step, n, lane, lds, P);
      live = P.inside && P.sq0 > 0;                                         // patch.sq0 <= 0 (:1069, :1087)
    }
    if (live) {
      float c = conf[e];
      pvlm_mvs::Around ar;
      mvs_wave_around(rows, cols, px, py, sgn, lane, depth, normal, cell, epoch, ar);   // waits for the two predecessors of the walk
      pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, depth_constant, min_depth, max_depth};
      // the close pixels in the wave's LDS slot (see k_mvs_propagate_flow_spec)
      pvlm_mvs::ClosePixel* close_w = close_s[threadIdx.x >> 6];
      int n_close;
      {
        pvlm_mvs::ClosePixel cl[4];
        n_close = pvlm_mvs::build_close(A, px, py, ar, cl);
        if (lane < 4) {
          pvlm_mvs::ClosePixel mine = cl[0];
#pragma unroll
          for (int q = 1; q < 4; ++q) if (lane == q) mine = cl[q];
          close_w[lane] = mine;
        }
        __builtin_amdgcn_wave_barrier();
      }
      pvlm_mvs::Rng rng{pass_seed, (unsigned long long)e, 0u};
      WaveScorer<M> scorer{rows, cols, half_window, step, n, lane, px, py, unit, &nb, &P, lds};
      pvlm_mvs::SerialBatch<WaveScorer<M>> batch{&scorer, 1};                 // one hypothesis after the other: process_pixel's chain
      const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};
      pvlm_mvs::process_pixel_close(A, rng, px, py, batch, ar, close_w, n_close, dep, nrm3, c, 2, pdx, pdy);
      __builtin_amdgcn_wave_barrier();                                         // the slot is rewritten for the wave's next pixel
      if (lane < 4) __hip_atomic_store(cell + 4 * e + lane, mvs_cell(epoch, lane == 0 ? dep : nrm3[lane - 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) { depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c; }
    } else if (lane < 4) {
      // a pixel the sweep skips keeps its state: its successors may read it at once
      __hip_atomic_store(cell + 4 * e + lane, mvs_cell(epoch, lane == 0 ? dep : nrm3[lane - 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Four waves per pixel (one workgroup = one pixel): a single view's diagonal is at most
// min(rows, cols) pixels, i.e. with a wave per pixel a sixth of the chip's wave slots, each running a chain of 8-14 dependent
// scorings of ~1300 instructions.  pvlm_mvs::process_pixel_spec scores the independent hypotheses of a pixel side by side — the two
// propagated ones; batches of four consecutive refinements built as if none of them were accepted — one per wave, exchanges the four
// confidences through LDS and resolves them in order: the same result as the chain, bit for bit, in 1 + ~3 dependent scorings.
template <int M>
struct BlockBatch {
  WaveScorer<M>* scorer; int wave, lane; pvlm_mvs::Hypothesis* xchg; int phase; unsigned long long since;
  __device__ int width() const { return 4; }
  template <class Build>
  __device__ void run(int n, const float* view_ray, const pvlm_mvs::ClosePixel* close, int n_close, Build&& build, pvlm_mvs::Hypothesis* out) {
    pvlm_mvs::Hypothesis mine;
    FLOW_CLOCK_ADD(5, 1);
    const unsigned long long bk0 = FLOW_CLOCK_NOW();
    if (since) { FLOW_CLOCK_ADD(8, bk0 - since); since = 0; }
    unsigned long long bk1 = bk0, bk2 = bk0;
    mine.normal[0] = mine.normal[1] = mine.normal[2] = 0.f; mine.depth = 0.f; mine.conf = -1.f; mine.valid = 0;
    if (wave < n) {
      build(wave, mine);
      bk1 = FLOW_CLOCK_NOW();
      if (mine.valid) mine.conf = pvlm_mvs::score_hypothesis(*scorer, view_ray, close, n_close, mine);
      if (lane == 0) xchg[phase * 4 + wave] = mine;
      bk2 = FLOW_CLOCK_NOW();
    }
    __syncthreads();
    FLOW_CLOCK_ADD(9, bk1 - bk0); FLOW_CLOCK_ADD(10, bk2 - bk1); FLOW_CLOCK_ADD(11, FLOW_CLOCK_NOW() - bk2);
    // two exchange buffers: a wave that runs ahead writes batch t + 1 into the other half and then waits at ITS barrier, which the
    // slowest wave only reaches after it has read batch t
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < n) out[w] = xchg[phase * 4 + w];
    phase ^= 1;
  }
  __device__ WaveScorer<M>& single() { return *scorer; }
};

// K13q — the data-flow sweep with FOUR waves per pixel (one workgroup = one pixel).  Once the preparation is off the critical path
// (K13p) what is left per pixel is the chain of 8-14 dependent scorings; pvlm_mvs::process_pixel_around scores the independent hypotheses
// of a pixel side by side — one per wave, confidences exchanged through LDS and resolved in order: the same result as the chain, bit
// for bit, in 1 + ~2 dependent batches (tests/test_mvs_cpu.py).
//
// Hand-off between pixels.  The first form (K13p's: plain stores -> agent release -> stamp; poll -> agent acquire -> plain loads) cost
// every pixel an L2 write-back, an L1 invalidate of its CU (three workgroups share one: every texel and unit-ray line re-fetched from
// L2 all the time) and ~8 dependent L2 round trips for the neighbours' state behind the fence.  A pixel needs exactly 16 bytes from each
// of its two predecessors — depth + normal — so they travel as four 8-byte cells {value, epoch} per pixel, written with agent-scope
// atomic stores (write-through) and polled with agent-scope atomic loads by eight lanes at once: a cell is complete or absent, there is
// nothing to order and nothing to fence (MI355X_MICROARCH.md, "8-B agent atomics both sides").  The maps themselves are written with plain
// stores as before — this launch reads them only where the sweep has not been yet (the other two neighbours, read BEFORE the wait) — and
// are whole when the kernel ends.
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M == 1 ? PVLM_K13_WAVES : 2))) void k_mvs_propagate_flow_spec(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray, const float* __restrict__ unit, pvlm_mvs_neighbours nb, float* depth,
    float* normal, float* conf, const unsigned char* __restrict__ depth_constant, float min_depth, float max_depth, unsigned long long pass_seed, int backward,
    unsigned long long* ticket, unsigned long long* cell, int epoch) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = pvlm_mvs::num_texels(half_window, step);
  __shared__ float4 strips[4][PVLM_MVS_LDS_PER_WAVE(M)];
  __shared__ pvlm_mvs::Hypothesis xchg[8];
  __shared__ pvlm_mvs::Around around;
  __shared__ pvlm_mvs::ClosePixel close_s[4];
  __shared__ int n_close_s;
  __shared__ unsigned long long drawn;
  float4* lds = strips[wave];
  const long long npix = (long long)rows * cols;
  const int sgn = backward ? 1 : -1;
  const int slot_h = pvlm_mvs::around_slot(sgn, 0), slot_v = pvlm_mvs::around_slot(0, sgn);   // the walk's two predecessors
  while (true) {
    const unsigned long long ck0 = FLOW_CLOCK_NOW();
    if (threadIdx.x == 0) drawn = atomicAdd(ticket, 1ull);
    __syncthreads();
    const unsigned long long t = drawn;
    if ((long long)t >= npix) return;                                       // the whole workgroup alike
    int wx, wy;
    mvs_walk_pixel((long long)t, rows, cols, &wx, &wy);
    const int px = backward ? cols - 1 - wx : wx, py = backward ? rows - 1 - wy : wy;
    const long long e = (long long)py * cols + px;
    float dep = depth[e];
    float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
    bool live = dep > 0;
    PatchRegs<M> P;
    if (live) {
      wave_fill_patch<M>(ref_gray, rows, cols, px, py, half_window, step, n, lane, lds, P);
      live = P.inside && P.sq0 > 0;
    }
    if (live) {
      float c = conf[e];
      // the two neighbours the walk has not reached yet: their state cannot change before this pixel is published
      if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        const int cx = px + (q == 0 ? -1 : (q == 3 ? 1 : 0)), cy = py + (q == 1 ? -1 : (q == 2 ? 1 : 0));
        const bool in = cx >= 0 && cy >= 0 && cx < cols && cy < rows;
        around.inside[q] = in;
        if (q != slot_h && q != slot_v) {
          const long long ne = in ? (long long)cy * cols + cx : e;
          around.depth[q] = in ? depth[ne] : 0.f;
          around.normal[q][0] = normal[3 * ne]; around.normal[q][1] = normal[3 * ne + 1]; around.normal[q][2] = normal[3 * ne + 2];
        }
      }
      const unsigned long long ck1 = FLOW_CLOCK_NOW();
      // lanes 0-3: the four cells of the horizontal predecessor, lanes 4-7: of the vertical one
      if (wave == 0) {
        const int p = lane >> 2, k = lane & 3;
        const int qx = px + (p ? 0 : sgn), qy = py + (p ? sgn : 0);
        const bool need = lane < 8 && qx >= 0 && qx < cols && qy >= 0 && qy < rows;
        const unsigned long long* src = cell + 4 * ((long long)qy * cols + qx) + k;
        unsigned long long v = 0;
        bool have = !need;
        while (true) {
          if (!have) { v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); have = (int)(v >> 32) == epoch; }
          if (__all(have)) break;
          __builtin_amdgcn_s_sleep(PVLM_MVS_FLOW_SLEEP);
        }
        if (need) {
          const float f = pvlm_mvs::bits_float((unsigned)v);
          const int slot = p ? slot_v : slot_h;
          if (k == 0) around.depth[slot] = f; else around.normal[slot][k - 1] = f;
        }
        // the close pixels of this pixel, once, in LDS: every scoring of every wave reads all of them (a per-thread array would sit in
        // scratch memory: 18 stores + 6 loads per scoring behind the wait)
        __builtin_amdgcn_wave_barrier();                                       // LDS operations of one wave execute in order
        pvlm_mvs::SweepArgs A0{rows, cols, unit, depth, normal, depth_constant, min_depth, max_depth};
        pvlm_mvs::ClosePixel cl[4];
        const int nc = pvlm_mvs::build_close(A0, px, py, around, cl);
        if (lane < 4) {
          pvlm_mvs::ClosePixel mine = cl[0];
#pragma unroll
          for (int q = 1; q < 4; ++q) if (lane == q) mine = cl[q];
          close_s[lane] = mine;
        }
        if (lane == 0) n_close_s = nc;
      }
      __syncthreads();
      const unsigned long long ck2 = FLOW_CLOCK_NOW();
      pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, depth_constant, min_depth, max_depth};
      pvlm_mvs::Rng rng{pass_seed, (unsigned long long)e, 0u};
      WaveScorer<M> scorer{rows, cols, half_window, step, n, lane, px, py, unit, &nb, &P, lds};
      BlockBatch<M> batch{&scorer, wave, lane, xchg, 0, ck2};
      const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};
      pvlm_mvs::process_pixel_close(A, rng, px, py, batch, around, close_s, n_close_s, dep, nrm3, c, 2, pdx, pdy);
      const unsigned long long ck3 = FLOW_CLOCK_NOW();
      if (threadIdx.x < 4) {
        const float f = threadIdx.x == 0 ? dep : nrm3[threadIdx.x - 1];
        __hip_atomic_store(cell + 4 * e + threadIdx.x, mvs_cell(epoch, f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (threadIdx.x == 0) { depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c; }
      FLOW_CLOCK_ADD(0, ck1 - ck0); FLOW_CLOCK_ADD(1, ck2 - ck1); FLOW_CLOCK_ADD(2, ck3 - ck2); FLOW_CLOCK_ADD(3, FLOW_CLOCK_NOW() - ck3); FLOW_CLOCK_ADD(4, 1);
    } else if (threadIdx.x < 4) {
      // a pixel the sweep skips keeps its state: its successors may read it at once
      const float f = threadIdx.x == 0 ? dep : nrm3[threadIdx.x - 1];
      __hip_atomic_store(cell + 4 * e + threadIdx.x, mvs_cell(epoch, f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                                          // `drawn`, `around` and the exchange buffers are reused by the next pixel
  }
}

// MEASURED VARIANT, not in the default library (built with -DPVLM_MEASURED_VARIANTS=1, selected with PVLM_MVS_SPEC=1).  Round 3 built
// it to shorten the per-pixel chain of a single view's sequential sweep and measured NO gain
// (profiles/r3_mvs_spec_ab.txt, 1440 x 720, 4 neighbours): 156.9 ms per iteration against 152.9 ms for the wave-per-pixel kernel;
// per launch 72 us against 70 us on full 720-pixel diagonals, 38 us against 50 us on diagonals shorter than 50 pixels.  Why: a
// pixel costs a fixed ~20 us (patch statistics, close neighbours, dependent global loads) plus ~4-5 us per CHAINED scoring; the
// speculation cuts the chain from 8 scorings to 3, but every one of the four waves repeats the fixed part, so a full diagonal is
// 2880 waves of (fixed + 3 scorings) on 1024 SIMDs — throughput-bound at about the time one wave per SIMD needs for its whole chain.
// The exactness argument (process_pixel_spec == process_pixel, any batch width) stays tested on the CPU (tests/test_mvs_cpu.py).
#ifndef PVLM_MEASURED_VARIANTS
#define PVLM_MEASURED_VARIANTS 0
#endif
#if PVLM_MEASURED_VARIANTS
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M == 1 ? PVLM_K13_WAVES : 2))) void k_mvs_propagate_diag_spec(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ ref_gray, const float* __restrict__ unit, pvlm_mvs_neighbours nb, float* depth,
    float* normal, float* conf, const unsigned char* __restrict__ depth_constant, float min_depth, float max_depth, unsigned long long pass_seed, int diag, int backward) {
  const int r0 = max(0, diag - (cols - 1));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int py = r0 + (int)blockIdx.x;                                    // one workgroup per pixel of the diagonal
  const int px = diag - py;
  const long long e = (long long)py * cols + px;
  float dep = depth[e];
  if (dep <= 0) return;                                                   // every exit before the first batch is taken by the four waves alike
  const int n = pvlm_mvs::num_texels(half_window, step);
  __shared__ float4 strips[4][PVLM_MVS_LDS_PER_WAVE(M)];
  __shared__ pvlm_mvs::Hypothesis xchg[8];
  float4* lds = strips[wave];
  PatchRegs<M> P;
  wave_fill_patch<M>(ref_gray, rows, cols, px, py, half_window, step, n, lane, lds, P);
  if (!P.inside || P.sq0 <= 0) return;                                    // patch.sq0 <= 0 (:1069, :1087)
  float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
  float c = conf[e];
  pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, depth_constant, min_depth, max_depth};
  pvlm_mvs::Rng rng{pass_seed, (unsigned long long)e, 0u};
  WaveScorer<M> scorer{rows, cols, half_window, step, n, lane, px, py, unit, &nb, &P, lds};
  BlockBatch<M> batch{&scorer, wave, lane, xchg, 0, 0ull};
  const int sgn = backward ? 1 : -1;
  const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};
  pvlm_mvs::process_pixel_spec(A, rng, px, py, batch, dep, nrm3, c, 2, pdx, pdy);
  // the pixel's neighbours on this diagonal are other workgroups' pixels and nobody reads (px, py) before the next launch
  if (threadIdx.x == 0) { depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c; }
}
#endif  // PVLM_MEASURED_VARIANTS

// Several views per launch (grid.y = job): a single diagonal is at most min(rows, cols) waves — a third of the chip's SIMDs at
// 1440 x 720, each running ONE wave with nothing to overlap its latencies — so the sequential sweep of one view is launch- and
// latency-bound (134 ms per iteration against 29 ms for the checkerboard).  Upstream parallelises this strategy over IMAGES
// (one view per OpenMP thread, mvs/MVS.cpp:87-93); so does this kernel: the views of a batch are independent (a view writes its
// own depth / normal / conf and reads the others' grey image and, with use_geometry, their depth_filter snapshot).
struct pvlm_mvs_job { pvlm_mvs_neighbours nb; size_t off; unsigned long long seed; const unsigned char* dconst; };
template <int M>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M == 1 ? PVLM_K13_WAVES : 2))) void k_mvs_propagate_diag_batch(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ gray_base, const float* __restrict__ unit, const pvlm_mvs_job* __restrict__ jobs,
    float* depth_base, float* normal_base, float* conf_base, float min_depth, float max_depth, int iter, int diag) {
  const pvlm_mvs_job& J = jobs[blockIdx.y];
  const int backward = iter & 1;
  const int r0 = max(0, diag - (cols - 1)), r1 = min(rows - 1, diag);
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int py = r0 + w;
  if (py > r1) return;
  const int px = diag - py;
  const long long e = (long long)py * cols + px;
  float* depth = depth_base + J.off; float* normal = normal_base + 3 * J.off; float* conf = conf_base + J.off;
  float dep = depth[e];
  if (dep <= 0) return;
  const int n = pvlm_mvs::num_texels(half_window, step);
  __shared__ float4 strips[4][PVLM_MVS_LDS_PER_WAVE(M)];
  float4* lds = strips[threadIdx.x >> 6];
  PatchRegs<M> P;
  wave_fill_patch<M>(gray_base + J.off, rows, cols, px, py, half_window, step, n, lane, lds, P);
  if (!P.inside || P.sq0 <= 0) return;
  float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
  float c = conf[e];
  pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, J.dconst, min_depth, max_depth};
  pvlm_mvs::Rng rng{pvlm_mvs::pass_seed(J.seed, iter), (unsigned long long)e, 0u};
  WaveScorer<M> scorer{rows, cols, half_window, step, n, lane, px, py, unit, &J.nb, &P, lds};
  const int sgn = backward ? 1 : -1;
  const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};
  pvlm_mvs::process_pixel(A, rng, px, py, scorer, dep, nrm3, c, 2, pdx, pdy);
  if (lane == 0) { depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c; }
}

// MEASURED VARIANT, not in the default library (-DPVLM_MEASURED_VARIANTS=1, used from PVLM_MVS_LANE_BATCH_MIN pixels per diagonal):
// the anti-diagonal of many views with one pixel per THREAD (see k_mvs_propagate_lane), workgroup = 64 pixels of one job's diagonal.
// Bit-identical (tests/test_mvs_gpu.py with PVLM_MVS_LANE_BATCH_MIN=1) and NOT faster at any batch size measured, 1440 x 720 x 4
// neighbours, ms per view and iteration (tools/mvs_batch_bench.py, profiles/r3_mvs_lane_ab.txt): one wave per pixel 37.1 / 27.8 / 26.1 /
// 25.4 / 24.6 for 8 / 32 / 64 / 128 / 320 views; thread per pixel on the long diagonals 43.2 (64 views), 30.6 (128), 26.1 (320).  Why the
// colour pass gains 1.65x and this does not: a thread runs the whole ~400 k-instruction program of its pixel (1.6 ms alone on a SIMD),
// and a diagonal of V views is only V x 720 / 64 such waves per launch — 3 840 for 320 views against 3 072 resident slots: one and a
// quarter rounds, the second almost empty; the colour pass launches 8 100.  It would take > 500 resident views to fill two rounds.
#if PVLM_MEASURED_VARIANTS
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PVLM_K13L_WAVES, 4))) void k_mvs_propagate_diag_batch_lane(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ gray_base, const float* __restrict__ unit, const pvlm_mvs_job* __restrict__ jobs,
    float* depth_base, float* normal_base, float* conf_base, float min_depth, float max_depth, int iter, int diag, int groups_per_job, float* __restrict__ wtab) {
  extern __shared__ float lane_tab[];
  const int job = blockIdx.x / groups_per_job, w = (blockIdx.x % groups_per_job) * 64 + threadIdx.x;
  const pvlm_mvs_job& J = jobs[job];
  const int backward = iter & 1;
  const int r0 = max(0, diag - (cols - 1)), r1 = min(rows - 1, diag);
  const int py = r0 + w;
  if (py > r1) return;
  const int px = diag - py;
  const long long e = (long long)py * cols + px;
  float* depth = depth_base + J.off; float* normal = normal_base + 3 * J.off; float* conf = conf_base + J.off;
  float dep = depth[e];
  if (dep <= 0) return;
  const int n = pvlm_mvs::num_texels(half_window, step);
  const unsigned char* ref_gray = gray_base + J.off;
  pvlm_mvs::ColumnPatch P{wtab + (size_t)blockIdx.x * 64 + threadIdx.x, (size_t)gridDim.x * 64, lane_tab + threadIdx.x, 64, 0.f, 0.f, false};
  pvlm_mvs::fill_patch_column(ref_gray, rows, cols, px, py, half_window, step, n, P);
  if (!P.inside || P.sq0 <= 0) return;                                    // patch.sq0 <= 0 (:1069, :1087)
  float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
  float c = conf[e];
  pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, J.dconst, min_depth, max_depth};
  pvlm_mvs::Rng rng{pvlm_mvs::pass_seed(J.seed, iter), (unsigned long long)e, 0u};
  pvlm_mvs::ColumnScorer<pvlm_mvs_neighbours> scorer{{}, {}, rows, cols, half_window, step, n, px, py, unit, ref_gray, &J.nb, P};
  const int sgn = backward ? 1 : -1;
  const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};
  pvlm_mvs::process_pixel(A, rng, px, py, scorer, dep, nrm3, c, 2, pdx, pdy);
  depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c;
}
#endif

// The anti-diagonal of many views with FOUR threads per pixel, one neighbour image each (pvlm_mvs::QuadScorer): a workgroup is 64 pixels of
// one job's diagonal.  Between the wave per pixel (every uniform value computed 64 times, 49 of 64 lanes busy, LDS strip walks for the
// ordered sums) and the thread per pixel (a 400 k-instruction program per thread: too few, too long waves on a diagonal — the measured
// variant above) — a quarter of that program per thread, four times the waves, the per-image NCC sums still a register add per texel.
// Chosen per diagonal by launch threshold (pvlm_mvs_views_estimate_sequential_batch); same maps as the other forms, bit for bit.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PVLM_K13L_WAVES, 4))) void k_mvs_propagate_diag_batch_quad(
    int rows, int cols, int half_window, int step, const unsigned char* __restrict__ gray_base, const float* __restrict__ unit, const pvlm_mvs_job* __restrict__ jobs,
    float* depth_base, float* normal_base, float* conf_base, float min_depth, float max_depth, int iter, int diag, int groups_per_job, float* __restrict__ wtab) {
  extern __shared__ float lane_tab[];                                      // [n][256]: one texel column per thread
  const int job = blockIdx.x / groups_per_job, w = (blockIdx.x % groups_per_job) * 64 + (threadIdx.x >> 2);
  const pvlm_mvs_job& J = jobs[job];
  const int backward = iter & 1;
  const int r0 = max(0, diag - (cols - 1)), r1 = min(rows - 1, diag);
  const int py = r0 + w;
  if (py > r1) return;                                                    // whole quads leave together: a pixel's four threads take every branch alike
  const int px = diag - py;
  const long long e = (long long)py * cols + px;
  float* depth = depth_base + J.off; float* normal = normal_base + 3 * J.off; float* conf = conf_base + J.off;
  float dep = depth[e];
  if (dep <= 0) return;
  const int n = pvlm_mvs::num_texels(half_window, step);
  const unsigned char* ref_gray = gray_base + J.off;
  // the pixel's weight column: its four threads compute and store the same values
  pvlm_mvs::ColumnPatch P{wtab + (size_t)blockIdx.x * 64 + (threadIdx.x >> 2), (size_t)gridDim.x * 64, lane_tab + threadIdx.x, 256, 0.f, 0.f, false};
  pvlm_mvs::fill_patch_column<true>(ref_gray, rows, cols, px, py, half_window, step, n, P);     // shared column: only final values are stored
  if (!P.inside || P.sq0 <= 0) return;                                    // patch.sq0 <= 0 (:1069, :1087)
  float nrm3[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
  float c = conf[e];
  pvlm_mvs::SweepArgs A{rows, cols, unit, depth, normal, J.dconst, min_depth, max_depth};
  pvlm_mvs::Rng rng{pvlm_mvs::pass_seed(J.seed, iter), (unsigned long long)e, 0u};
  pvlm_mvs::QuadScorer<pvlm_mvs_neighbours> scorer{{{}, {}, rows, cols, half_window, step, n, px, py, unit, ref_gray, &J.nb, P}};
  const int sgn = backward ? 1 : -1;
  const int pdx[2] = {sgn, 0}, pdy[2] = {0, sgn};
  pvlm_mvs::process_pixel(A, rng, px, py, scorer, dep, nrm3, c, 2, pdx, pdy);
  if ((threadIdx.x & 3) == 0) { depth[e] = dep; normal[3 * e] = nrm3[0]; normal[3 * e + 1] = nrm3[1]; normal[3 * e + 2] = nrm3[2]; conf[e] = c; }
}

// launch helpers: M = 1 for windows of at most 64 texels (one texel per lane), M = PVLM_MVS_MAXM otherwise
static bool mvs_lane_form(int n_tex);
static void launch_mvs_conf_lane(hipStream_t s, int rows, int cols, int half_window, int step, const unsigned char* img, const float* unit,
                                 const pvlm_mvs_neighbours& nb, float* depth, float* normal, float* conf, float* wtab);
static void launch_mvs_conf(hipStream_t s, int rows, int cols, int half_window, int step, const unsigned char* img, const float* unit,
                            const pvlm_mvs_neighbours& nb, float* depth, float* normal, float* conf, float* wtab) {
  const size_t npix = (size_t)rows * cols;
  if (wtab && pvlm_mvs::num_texels(half_window, step) <= 64) { launch_mvs_conf_lane(s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, wtab); return; }
  const dim3 grid((unsigned)((npix + 3) / 4)), block(256);
  if (pvlm_mvs::num_texels(half_window, step) <= 64) hipLaunchKernelGGL(k_mvs_conf<1>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf);
  else hipLaunchKernelGGL(k_mvs_conf<PVLM_MVS_MAXM>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf);
}
// The colour pass takes the thread-per-pixel form when the caller hands it the weight table (mvs_lane_table: windows of at most 64
// texels, PVLM_MVS_LANE=0 switches back for A/B runs): 8.9 against 14.6 ms per pass at 1440 x 720 x 4 neighbours, 9.8 against 19.5 ms with the
// geometric term (profiles/r3_mvs_lane_ab.txt).
static bool mvs_lane_form(int n_tex) {
  static const bool off = getenv("PVLM_MVS_LANE") && atoi(getenv("PVLM_MVS_LANE")) == 0;
  return !off && n_tex <= 64;
}
// The weight table serves a BAND of image rows at a time (one launch per band, in stream order), so that its size does not grow with
// the image: at most PVLM_MVS_LANE_TABLE_MB (128) — the whole 1440 x 720 pass in one band, 13 bands per pass at 5760 x 2880.
struct LaneBands { int band_rows; size_t floats; };
static LaneBands mvs_lane_bands(int rows, int per_row_pixels, int half_window, int step) {
  const int n_tex = pvlm_mvs::num_texels(half_window, step);
  if (!mvs_lane_form(n_tex)) return {0, 0};
  static const size_t cap_mb = getenv("PVLM_MVS_LANE_TABLE_MB") ? (size_t)std::max(1, atoi(getenv("PVLM_MVS_LANE_TABLE_MB"))) : 128;
  const size_t per_row = (size_t)n_tex * (((size_t)per_row_pixels + 7) / 8 * 8);   // floats per image row, in whole 8 x 8 tiles
  int band = (int)std::max<size_t>(8, std::min<size_t>(((size_t)rows + 7) / 8 * 8, (cap_mb << 18) / std::max<size_t>(per_row, 1) / 8 * 8));
  return {band, per_row * band};
}
static void launch_mvs_conf_lane(hipStream_t s, int rows, int cols, int half_window, int step, const unsigned char* img, const float* unit,
                                 const pvlm_mvs_neighbours& nb, float* depth, float* normal, float* conf, float* wtab) {
  const int n_tex = pvlm_mvs::num_texels(half_window, step);
  const int band = mvs_lane_bands(rows, cols, half_window, step).band_rows;
  for (int row0 = 0; row0 < rows; row0 += band) {
    const int br = std::min(band, rows - row0);
    hipLaunchKernelGGL(k_mvs_conf_lane, dim3((unsigned)(((cols + 7) / 8) * ((br + 7) / 8))), dim3(64), (size_t)n_tex * 64 * sizeof(float), s, rows, cols, half_window, step, img, unit,
                       nb, depth, normal, conf, wtab, row0, br);
  }
}
static void launch_mvs_propagate(hipStream_t s, int rows, int cols, int half_window, int step, const unsigned char* img, const float* unit,
                                 const pvlm_mvs_neighbours& nb, float* depth, float* normal, float* conf, const unsigned char* depth_constant, float min_depth,
                                 float max_depth, unsigned long long pass_seed, int offset, float* wtab) {
  const int half = (cols + 1) / 2;
  const size_t waves = (size_t)rows * (size_t)half;
  const int n_tex = pvlm_mvs::num_texels(half_window, step);
  if (wtab && n_tex <= 64) {
    const int band = mvs_lane_bands(rows, half, half_window, step).band_rows;
    for (int row0 = 0; row0 < rows; row0 += band) {                       // a colour's pixels do not read each other: the bands are independent
      const int br = std::min(band, rows - row0);
      hipLaunchKernelGGL(k_mvs_propagate_lane, dim3((unsigned)(((half + 7) / 8) * ((br + 7) / 8))), dim3(64), (size_t)n_tex * 64 * sizeof(float), s, rows, cols, half_window, step,
                         img, unit, nb, depth, normal, conf, depth_constant, min_depth, max_depth, pass_seed, offset, wtab, row0, br);
    }
    return;
  }
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (n_tex <= 64)
    hipLaunchKernelGGL(k_mvs_propagate<1>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant, min_depth, max_depth, pass_seed, offset);
  else
    hipLaunchKernelGGL(k_mvs_propagate<PVLM_MVS_MAXM>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant, min_depth, max_depth, pass_seed, offset);
}

// one iteration of the sequential sweep (iteration parity = direction, :1061, :1079): one persistent launch (k_mvs_propagate_flow) when
// the caller provides the ticket counter + per-pixel stamps (flow != nullptr; stamps of iteration `iter` = iter + 1, the caller zeroes
// them once), otherwise — PVLM_MVS_FLOW=0, or no memory for the stamps — every anti-diagonal in walking order
struct MvsFlow { unsigned long long* ticket; unsigned long long* cell; int blocks; bool spec; };
static void launch_mvs_propagate_sequential(hipStream_t s, int rows, int cols, int half_window, int step, const unsigned char* img, const float* unit,
                                            const pvlm_mvs_neighbours& nb, float* depth, float* normal, float* conf, const unsigned char* depth_constant,
                                            float min_depth, float max_depth, unsigned long long pass_seed, int iter, const MvsFlow* flow = nullptr) {
  const int backward = iter % 2, n_diag = rows + cols - 1;
  const bool small = pvlm_mvs::num_texels(half_window, step) <= 64;
  if (flow) {
    (void)hipMemsetAsync(flow->ticket, 0, sizeof(unsigned long long), s);
    const dim3 grid((unsigned)flow->blocks), block(256);
    if (flow->spec) {
      if (small)
        hipLaunchKernelGGL(k_mvs_propagate_flow_spec<1>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant, min_depth, max_depth,
                           pass_seed, backward, flow->ticket, flow->cell, iter + 1);
      else
        hipLaunchKernelGGL(k_mvs_propagate_flow_spec<PVLM_MVS_MAXM>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant,
                           min_depth, max_depth, pass_seed, backward, flow->ticket, flow->cell, iter + 1);
      return;
    }
    if (small)
      hipLaunchKernelGGL(k_mvs_propagate_flow<1>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant, min_depth, max_depth,
                         pass_seed, backward, flow->ticket, flow->cell, iter + 1);
    else
      hipLaunchKernelGGL(k_mvs_propagate_flow<PVLM_MVS_MAXM>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant, min_depth,
                         max_depth, pass_seed, backward, flow->ticket, flow->cell, iter + 1);
    return;
  }
#if PVLM_MEASURED_VARIANTS
  static const bool spec = getenv("PVLM_MVS_SPEC") && atoi(getenv("PVLM_MVS_SPEC")) != 0;   // four waves per pixel (measured: no gain, see above)
#else
  const bool spec = false;
#endif
  for (int q = 0; q < n_diag; ++q) {
    const int d = backward ? n_diag - 1 - q : q;
    const int len = std::min(rows - 1, d) - std::max(0, d - (cols - 1)) + 1;
    const dim3 grid((unsigned)(spec ? len : (len + 3) / 4)), block(256);
#if PVLM_MEASURED_VARIANTS
    if (spec) {
      if (small)
        hipLaunchKernelGGL(k_mvs_propagate_diag_spec<1>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant, min_depth,
                           max_depth, pass_seed, d, backward);
      else
        hipLaunchKernelGGL(k_mvs_propagate_diag_spec<PVLM_MVS_MAXM>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant,
                           min_depth, max_depth, pass_seed, d, backward);
      continue;
    }
#endif
    if (small)
      hipLaunchKernelGGL(k_mvs_propagate_diag<1>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant, min_depth, max_depth,
                         pass_seed, d, backward);
    else
      hipLaunchKernelGGL(k_mvs_propagate_diag<PVLM_MVS_MAXM>, grid, block, 0, s, rows, cols, half_window, step, img, unit, nb, depth, normal, conf, depth_constant,
                         min_depth, max_depth, pass_seed, d, backward);
  }
}

#ifndef PVLM_MVS_FLOW_MAX_DIAG
#define PVLM_MVS_FLOW_MAX_DIAG 1024   // longest anti-diagonal (pixels) up to which a pixel gets a workgroup of four waves (K13q); beyond: one wave (K13p)
#endif
// scratch of the persistent sweep: ticket counter + four hand-off cells per pixel (zeroed here), grid = what the chip keeps resident
static bool mvs_flow_begin(pvlm_ctx* ctx, int rows, int cols, int half_window, int step, MvsFlow* f) {
  // PVLM_MVS_FLOW: 0 = one launch per anti-diagonal, 1 = persistent data-flow launch with a wave per pixel (K13p), 2 = with four waves per
  // pixel (K13q).  Unset: by size — four waves per pixel while an anti-diagonal of workgroups fits the chip (768 resident workgroups), one
  // wave per pixel beyond.  Measured per iteration, 4 neighbours (profiles/r4_mvs_seq_forms.txt): 1440 x 720: 127.6 / 119.4 / 82.4 ms for
  // 0 / 1 / 2; 5760 x 2880: 697 / 607 / 879 ms.
  static const int forced = getenv("PVLM_MVS_FLOW") ? atoi(getenv("PVLM_MVS_FLOW")) : -1;
  const int mode = forced >= 0 ? forced : (std::min(rows, cols) <= PVLM_MVS_FLOW_MAX_DIAG ? 2 : 1);
  f->ticket = nullptr; f->cell = nullptr; f->blocks = 0; f->spec = mode == 2;
  if (mode == 0) return false;
  const size_t npix = (size_t)rows * cols;
  if (pvlm_i_alloc(ctx, &f->ticket, (size_t)1)) { f->ticket = nullptr; return false; }
  if (pvlm_i_alloc(ctx, &f->cell, 4 * npix)) { pvlm_i_free(ctx, f->ticket); f->ticket = nullptr; f->cell = nullptr; return false; }   // four {value, epoch} cells per pixel
  (void)hipMemsetAsync(f->cell, 0, 4 * npix * sizeof(unsigned long long), ctx->stream);
  int per_cu = 0;
  const bool small = pvlm_mvs::num_texels(half_window, step) <= 64;
  hipError_t e;
  if (f->spec) e = small ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mvs_propagate_flow_spec<1>, 256, 0)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mvs_propagate_flow_spec<PVLM_MVS_MAXM>, 256, 0);
  else e = small ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mvs_propagate_flow<1>, 256, 0)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mvs_propagate_flow<PVLM_MVS_MAXM>, 256, 0);
  if (e != hipSuccess || per_cu < 1) per_cu = 1;
  f->blocks = std::max(1, ctx->cu_count > 0 ? ctx->cu_count : 256) * per_cu;
  // more waves than ~2.5 anti-diagonals hold pixels only wait: one diagonal in its dependent chain, the next ones preparing
  static const double depth_diags = getenv("PVLM_MVS_FLOW_DIAGS") ? atof(getenv("PVLM_MVS_FLOW_DIAGS")) : 2.5;
  const int useful = (int)(depth_diags * std::min(rows, cols) / (f->spec ? 1.0 : 4.0)) + 1;
  f->blocks = std::max(1, std::min(f->blocks, useful));
  return true;
}
static void mvs_flow_end(pvlm_ctx* ctx, MvsFlow* f) {
#if PVLM_MVS_FLOW_CLOCK
  if (f->ticket) {
    unsigned long long h[24] = {}, z[24] = {};
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_flow_clock), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_flow_clock), z, sizeof z);
    const double px = (double)std::max<unsigned long long>(h[4], 1);
    fprintf(stderr, "[flow clock] pixels %llu | us per pixel: prepare %.2f  wait %.2f  chain %.2f  publish %.2f | batch rounds per pixel %.2f | blocks %d\n", h[4],
            h[0] * 0.01 / px, h[1] * 0.01 / px, h[2] * 0.01 / px, h[3] * 0.01 / px, (double)h[5] / px, f->blocks);
    fprintf(stderr, "[flow clock]   chain, us per pixel (wave 0): close neighbours %.2f  build %.2f  score %.2f  barrier + exchange %.2f | inside score: texels %.2f  strip sums %.2f  ncc + geometry %.2f\n",
            h[8] * 0.01 / px, h[9] * 0.01 / px, h[10] * 0.01 / px, h[11] * 0.01 / px, h[12] * 0.01 / px, h[13] * 0.01 / px, h[14] * 0.01 / px);
    fprintf(stderr, "[flow clock]   per pixel: propagated hypotheses accepted %.3f  random phase entered %.3f  refinements accepted %.3f\n", h[16] / px, h[17] / px, h[19] / px);
  }
#endif
  pvlm_i_free(ctx, f->ticket); pvlm_i_free(ctx, f->cell); f->ticket = nullptr; f->cell = nullptr; }

// MVS::InitDepthNormal (mvs/MVS.cpp:496-584, the `#elif 1` branch :511-514): LiDAR depth image (uint16, depth * 256) where it has a
// value, a uniform random depth elsewhere, optional mask, a random normal facing the camera for every pixel the mask keeps.
// Draw 0 of pixel e = its random depth, the following draws = GenerateRandomNormal (counter-based stream, as in the sweep).
__global__ void k_mvs_init_depth_normal(int rows, int cols, const unsigned short* __restrict__ lidar16, const float* __restrict__ mask, float min_depth,
                                        float max_depth, int keep_const, unsigned long long ps, float* __restrict__ depth, float* __restrict__ normal,
                                        unsigned char* __restrict__ depth_constant) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)rows * cols) return;
  pvlm_mvs::Rng rng{ps, (unsigned long long)e, 0u};
  float d = lidar16 ? (float)lidar16[e] : 0.f;
  d /= 256.f;
  const float depth_random = rng.next01() * (min_depth - max_depth) + max_depth;      // rng.fill(UNIFORM, max_depth, min_depth)  :546
  const float lidar_mask = d > 0 ? 0.f : 1.f;                                          // THRESH_BINARY_INV                         :552
  d = d + depth_random * lidar_mask;
  if (lidar16 && keep_const && depth_constant) depth_constant[e] = (unsigned char)(1.f - lidar_mask);
  const float m = mask ? mask[e] : 1.f;
  depth[e] = d * m;
  float nrm[3] = {0.f, 0.f, 0.f};
  if (!(m < 1)) {
    float ray[3];
    pvlm_mvs::unit_ray(rows, cols, (int)(e % cols), (int)(e / cols), ray);
    pvlm_mvs::generate_random_normal(rng, ray, nrm);
  }
  normal[3 * e] = nrm[0]; normal[3 * e + 1] = nrm[1]; normal[3 * e + 2] = nrm[2];
}

// EstimateDepthMapSingle :698-714: hypotheses under the confidence threshold are dropped
__global__ void k_mvs_threshold(long long npix, const unsigned char* __restrict__ depth_constant, float thr, float* __restrict__ depth, float* __restrict__ normal,
                                float* __restrict__ conf) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npix) return;
  if (depth_constant && depth_constant[e]) return;
  if (conf[e] < thr) { depth[e] = 0.f; conf[e] = -1.f; normal[3 * e] = 0.f; normal[3 * e + 1] = 0.f; normal[3 * e + 2] = 0.f; }
}

// depth-map fusion filter (FilterDepthImage + ProjectDepthConfToRef)
struct pvlm_mvs_pose { float R_rn[9]; float t_rn[3]; };
__global__ void k_mvs_fill_u32(long long n, unsigned v, unsigned* __restrict__ p) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] = v;
}
__global__ void k_mvs_project(int rows, int cols, const float* __restrict__ unit, const float* __restrict__ nei_depth, pvlm_mvs_pose pose,
                              unsigned* __restrict__ proj_bits) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (long long)rows * cols) pvlm_mvs::project_splat(rows, cols, unit, nei_depth, pose.R_rn, pose.t_rn, e, proj_bits);
}
__global__ void k_mvs_filter(int rows, int cols, int n_neighbors, const unsigned* __restrict__ proj_bits, const float* __restrict__ depth,
                             const float* __restrict__ conf, const unsigned char* __restrict__ depth_constant, float thr,
                             float* __restrict__ depth_filter, float* __restrict__ conf_filter) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (long long)rows * cols) pvlm_mvs::filter_pixel(rows, cols, n_neighbors, proj_bits, depth, conf, depth_constant, thr, e, depth_filter, conf_filter);
}

// FilterDepthImageRefine: keyed splat (range, last raster-order source) + per-pixel confidence fusion
__global__ void k_mvs_fill_u64(long long n, unsigned long long v, unsigned long long* __restrict__ p) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] = v;
}
__global__ void k_mvs_project_conf(int rows, int cols, const float* __restrict__ unit, const float* __restrict__ nei_depth, pvlm_mvs_pose pose,
                                   unsigned long long* __restrict__ proj_key) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (long long)rows * cols) pvlm_mvs::project_splat_conf(rows, cols, unit, nei_depth, pose.R_rn, pose.t_rn, e, proj_key);
}
__global__ void k_mvs_refine(int rows, int cols, pvlm_mvs::RefineViews nv, const unsigned long long* __restrict__ proj_key, const float* __restrict__ unit,
                             const float* __restrict__ depth, float* __restrict__ conf, const unsigned char* __restrict__ depth_constant, float thr,
                             float min_depth, float max_depth, float* __restrict__ depth_filter, float* __restrict__ conf_filter) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (long long)rows * cols)
    pvlm_mvs::refine_pixel(rows, cols, nv, proj_key, unit, depth, conf, depth_constant, thr, min_depth, max_depth, e, depth_filter, conf_filter);
}

extern "C" {

pvlm_status pvlm_mvs_filter_depth(pvlm_ctx* ctx, int rows, int cols, int n_neighbors, const float* const* nei_depth, const float* R_nr, const float* t_nr,
                                  const float* depth, const float* conf, const unsigned char* depth_constant, float depth_diff_threshold,
                                  float* depth_filter, float* conf_filter) {
  if (!ctx || rows <= 0 || cols <= 0 || n_neighbors < 0 || n_neighbors > 16 || !depth || !depth_filter || (n_neighbors > 0 && (!nei_depth || !R_nr || !t_nr)) ||
      (conf && !conf_filter))
    return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t npix = (size_t)rows * cols;
  float *d_unit = nullptr, *d_nd = nullptr, *d_depth = nullptr, *d_conf = nullptr, *d_out = nullptr, *d_cout = nullptr;
  unsigned* d_proj = nullptr; unsigned char* d_const = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_unit, npix * 3);
  if (!st) st = pvlm_i_alloc(ctx, &d_nd, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_proj, npix * (size_t)std::max(n_neighbors, 1));
  if (!st) st = pvlm_i_alloc(ctx, &d_depth, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_out, npix);
  if (!st && conf) st = pvlm_i_alloc(ctx, &d_conf, npix);
  if (!st && conf) st = pvlm_i_alloc(ctx, &d_cout, npix);
  if (!st && depth_constant) st = pvlm_i_alloc(ctx, &d_const, npix);
  if (!st) {
    hipStream_t s = ctx->stream;
    const unsigned grid = (unsigned)((npix + 255) / 256);
    hipLaunchKernelGGL(k_mvs_unit_table, dim3(grid), dim3(256), 0, s, rows, cols, d_unit);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && n_neighbors > 0) {
      hipLaunchKernelGGL(k_mvs_fill_u32, dim3((unsigned)((npix * n_neighbors + 255) / 256)), dim3(256), 0, s, (long long)(npix * n_neighbors), 0x7f800000u, d_proj);
      e = hipGetLastError();
    }
    for (int b = 0; b < n_neighbors && e == hipSuccess; ++b) {
      if (!nei_depth[b]) { e = hipErrorInvalidValue; break; }
      e = mvs_up(ctx, d_nd, nei_depth[b], npix * sizeof(float));
      pvlm_mvs_pose pose;
      pvlm_mvs::inverse_pose(R_nr + 9 * b, t_nr + 3 * b, pose.R_rn, pose.t_rn);
      if (e == hipSuccess) { hipLaunchKernelGGL(k_mvs_project, dim3(grid), dim3(256), 0, s, rows, cols, d_unit, d_nd, pose, d_proj + npix * (size_t)b); e = hipGetLastError(); }
      if (e == hipSuccess) e = mvs_sync(ctx);   // d_nd is reused by the next neighbour; nei_depth[b] is pageable host memory
    }
    if (e == hipSuccess) e = mvs_up(ctx, d_depth, depth, npix * sizeof(float));
    if (e == hipSuccess && conf) e = mvs_up(ctx, d_conf, conf, npix * sizeof(float));
    if (e == hipSuccess && depth_constant) e = mvs_up(ctx, d_const, depth_constant, npix);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_mvs_filter, dim3(grid), dim3(256), 0, s, rows, cols, n_neighbors, d_proj, d_depth, d_conf, d_const, depth_diff_threshold, d_out, d_cout);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = mvs_down(ctx, depth_filter, d_out, npix * sizeof(float));
    if (e == hipSuccess && conf) e = mvs_down(ctx, conf_filter, d_cout, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_sync(ctx);
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_filter_depth: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  mvs_sync(ctx);
  pvlm_i_free(ctx, d_unit); pvlm_i_free(ctx, d_nd); pvlm_i_free(ctx, d_proj); pvlm_i_free(ctx, d_depth); pvlm_i_free(ctx, d_conf); pvlm_i_free(ctx, d_out); pvlm_i_free(ctx, d_cout); pvlm_i_free(ctx, d_const);
  return st;
}

pvlm_status pvlm_mvs_filter_depth_refine(pvlm_ctx* ctx, int rows, int cols, int n_neighbors, const float* const* nei_depth, const float* const* nei_conf,
                                         const float* R_nr, const float* t_nr, const float* depth, float* conf, const unsigned char* depth_constant,
                                         float depth_diff_threshold, float min_depth, float max_depth, float* depth_filter, float* conf_filter) {
  if (!ctx || rows <= 0 || cols <= 0 || n_neighbors < 0 || n_neighbors > 16 || !depth || !conf || !depth_filter || !conf_filter ||
      (n_neighbors > 0 && (!nei_depth || !nei_conf || !R_nr || !t_nr)))
    return PVLM_ERR_ARG;
  const size_t npix = (size_t)rows * cols;
  if (npix > 0xffffffffull) { PVLM_SET_ERR(ctx, "pvlm_mvs_filter_depth_refine: %zu pixels exceed the 32-bit source index of the splat key", npix); return PVLM_ERR_ARG; }
  for (int b = 0; b < n_neighbors; ++b) if (!nei_depth[b] || !nei_conf[b]) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t nn = (size_t)std::max(n_neighbors, 1);
  float *d_unit = nullptr, *d_nd = nullptr, *d_nc = nullptr, *d_depth = nullptr, *d_conf = nullptr, *d_out = nullptr, *d_cout = nullptr;
  unsigned long long* d_key = nullptr; unsigned char* d_const = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_unit, npix * 3);
  if (!st) st = pvlm_i_alloc(ctx, &d_nd, npix * nn);
  if (!st) st = pvlm_i_alloc(ctx, &d_nc, npix * nn);
  if (!st) st = pvlm_i_alloc(ctx, &d_key, npix * nn);
  if (!st) st = pvlm_i_alloc(ctx, &d_depth, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_conf, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_out, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_cout, npix);
  if (!st && depth_constant) st = pvlm_i_alloc(ctx, &d_const, npix);
  if (!st) {
    hipStream_t s = ctx->stream;
    const unsigned grid = (unsigned)((npix + 255) / 256);
    hipLaunchKernelGGL(k_mvs_unit_table, dim3(grid), dim3(256), 0, s, rows, cols, d_unit);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && n_neighbors > 0) {
      hipLaunchKernelGGL(k_mvs_fill_u64, dim3((unsigned)((npix * n_neighbors + 255) / 256)), dim3(256), 0, s, (long long)(npix * n_neighbors), ~0ull, d_key);
      e = hipGetLastError();
    }
    pvlm_mvs::RefineViews nv;
    nv.n = n_neighbors;
    for (int b = 0; b < n_neighbors && e == hipSuccess; ++b) {
      e = mvs_up(ctx, d_nd + npix * (size_t)b, nei_depth[b], npix * sizeof(float));
      if (e == hipSuccess) e = mvs_up(ctx, d_nc + npix * (size_t)b, nei_conf[b], npix * sizeof(float));
      pvlm_mvs_pose pose;
      pvlm_mvs::inverse_pose(R_nr + 9 * b, t_nr + 3 * b, pose.R_rn, pose.t_rn);
      nv.conf[b] = d_nc + npix * (size_t)b;
      for (int k = 0; k < 9; ++k) nv.R[b][k] = R_nr[9 * b + k];
      for (int k = 0; k < 3; ++k) nv.t[b][k] = t_nr[3 * b + k];
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_mvs_project_conf, dim3(grid), dim3(256), 0, s, rows, cols, d_unit, d_nd + npix * (size_t)b, pose, d_key + npix * (size_t)b);
        e = hipGetLastError();
      }
    }
    if (e == hipSuccess) e = mvs_up(ctx, d_depth, depth, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_up(ctx, d_conf, conf, npix * sizeof(float));
    if (e == hipSuccess && depth_constant) e = mvs_up(ctx, d_const, depth_constant, npix);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_mvs_refine, dim3(grid), dim3(256), 0, s, rows, cols, nv, d_key, d_unit, d_depth, d_conf, d_const, depth_diff_threshold, min_depth,
                         max_depth, d_out, d_cout);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = mvs_down(ctx, depth_filter, d_out, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_down(ctx, conf_filter, d_cout, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_down(ctx, conf, d_conf, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_sync(ctx);
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_filter_depth_refine: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  mvs_sync(ctx);
  pvlm_i_free(ctx, d_unit); pvlm_i_free(ctx, d_nd); pvlm_i_free(ctx, d_nc); pvlm_i_free(ctx, d_key); pvlm_i_free(ctx, d_depth); pvlm_i_free(ctx, d_conf); pvlm_i_free(ctx, d_out); pvlm_i_free(ctx, d_cout); pvlm_i_free(ctx, d_const);
  return st;
}

// shared by the scoring pass (max_iter < 0) and the PatchMatch sweep: upload, launch, download
static pvlm_status mvs_run(pvlm_ctx* ctx, const char* what, int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                           const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                           const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth, unsigned long long seed,
                           int max_iter, float conf_threshold, int strategy = 1) {
  if (!ctx || rows <= 0 || cols <= 0 || half_window < 1 || step < 1 || !ref_gray || n_neighbors < 0 || n_neighbors > 16 || !depth || !normal || !conf ||
      (n_neighbors > 0 && (!nei_gray || !R_nr || !t_nr)))
    return PVLM_ERR_ARG;
  if (pvlm_mvs::num_texels(half_window, step) > 64 * PVLM_MVS_MAXM) { PVLM_SET_ERR(ctx, "NCC window of %d texels exceeds %d", pvlm_mvs::num_texels(half_window, step), 64 * PVLM_MVS_MAXM); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t npix = (size_t)rows * cols;
  unsigned char *d_img = nullptr, *d_const = nullptr; float *d_unit = nullptr, *d_depth = nullptr, *d_normal = nullptr, *d_conf = nullptr, *d_ndepth = nullptr;
  unsigned* d_quad = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_img, npix * (size_t)(n_neighbors + 1));
  if (!st) st = pvlm_i_alloc(ctx, &d_quad, npix * (size_t)std::max(n_neighbors, 1));
  if (!st && nei_depth) st = pvlm_i_alloc(ctx, &d_ndepth, npix * (size_t)std::max(n_neighbors, 1));
  if (!st && depth_constant) st = pvlm_i_alloc(ctx, &d_const, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_unit, npix * 3);
  if (!st) st = pvlm_i_alloc(ctx, &d_depth, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_normal, npix * 3);
  if (!st) st = pvlm_i_alloc(ctx, &d_conf, npix);
  if (!st) {
    hipStream_t s = ctx->stream;
    pvlm_mvs_neighbours nb;
    nb.n = n_neighbors; nb.geometric = nei_depth ? 1 : 0;
    hipError_t e = mvs_up(ctx, d_img, ref_gray, npix);
    for (int b = 0; b < n_neighbors && e == hipSuccess; ++b) {
      if (!nei_gray[b]) { e = hipErrorInvalidValue; break; }
      e = mvs_up(ctx, d_img + npix * (size_t)(b + 1), nei_gray[b], npix);
      if (e == hipSuccess) hipLaunchKernelGGL(k_mvs_quad, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, rows, cols, d_img + npix * (size_t)(b + 1), d_quad + npix * (size_t)b);
      nb.quad[b] = d_quad + npix * (size_t)b;
      nb.depth[b] = nullptr;
      if (nei_depth && e == hipSuccess) {
        if (!nei_depth[b]) { e = hipErrorInvalidValue; break; }
        e = mvs_up(ctx, d_ndepth + npix * (size_t)b, nei_depth[b], npix * sizeof(float));
        nb.depth[b] = d_ndepth + npix * (size_t)b;
      }
      for (int k = 0; k < 9; ++k) nb.R[b][k] = R_nr[9 * b + k];
      for (int k = 0; k < 3; ++k) nb.t[b][k] = t_nr[3 * b + k];
    }
    if (e == hipSuccess) e = mvs_up(ctx, d_depth, depth, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_up(ctx, d_normal, normal, npix * 3 * sizeof(float));
    if (e == hipSuccess) e = mvs_up(ctx, d_conf, conf, npix * sizeof(float));
    if (e == hipSuccess && depth_constant) e = mvs_up(ctx, d_const, depth_constant, npix);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_mvs_unit_table, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, rows, cols, d_unit);
      if (max_iter < 0) {
        float* d_wtab = nullptr;
        const size_t wtab_floats = mvs_lane_bands(rows, cols, half_window, step).floats;
        if (wtab_floats && pvlm_i_alloc(ctx, &d_wtab, wtab_floats)) d_wtab = nullptr;
        {
          pvlm_prof_scope prof(ctx, 1);   // timed with the "materialise" slot of pvlm_profile_* (bench / tools)
          launch_mvs_conf(s, rows, cols, half_window, step, d_img, d_unit, nb, d_depth, d_normal, d_conf, d_wtab);
        }
        pvlm_i_free(ctx, d_wtab);
      } else if (strategy == 2) {
        MvsFlow flow;
        const bool have_flow = mvs_flow_begin(ctx, rows, cols, half_window, step, &flow);
        for (int iter = 0; iter < max_iter; ++iter) {
          pvlm_prof_scope prof(ctx, 1);                  // one profile interval per iteration (one persistent launch, or rows + cols - 1 launches)
          launch_mvs_propagate_sequential(s, rows, cols, half_window, step, d_img, d_unit, nb, d_depth, d_normal, d_conf, d_const, min_depth, max_depth,
                                          pvlm_mvs::pass_seed(seed, iter), iter, have_flow ? &flow : nullptr);
        }
        mvs_flow_end(ctx, &flow);
        hipLaunchKernelGGL(k_mvs_threshold, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, (long long)npix, d_const, conf_threshold, d_depth, d_normal, d_conf);
      } else {
        float* d_wtab = nullptr;                       // patch weights of the thread-per-pixel form, [texel][pixel of the colour in a band of rows]
        const size_t wtab_floats = mvs_lane_bands(rows, (cols + 1) / 2, half_window, step).floats;
        if (wtab_floats && pvlm_i_alloc(ctx, &d_wtab, wtab_floats)) d_wtab = nullptr;    // no room: the wave-per-pixel form needs none
        for (int iter = 0; iter < max_iter; ++iter)
          for (int offset = 0; offset <= 1; ++offset) {
            pvlm_prof_scope prof(ctx, 1);
            launch_mvs_propagate(s, rows, cols, half_window, step, d_img, d_unit, nb, d_depth, d_normal, d_conf, d_const, min_depth, max_depth,
                                 pvlm_mvs::pass_seed(seed, 2 * iter + offset), offset, d_wtab);
          }
        pvlm_i_free(ctx, d_wtab);                        // stream-ordered: behind the launches above
        hipLaunchKernelGGL(k_mvs_threshold, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, (long long)npix, d_const, conf_threshold, d_depth, d_normal, d_conf);
      }
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = mvs_down(ctx, depth, d_depth, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_down(ctx, normal, d_normal, npix * 3 * sizeof(float));
    if (e == hipSuccess) e = mvs_down(ctx, conf, d_conf, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_sync(ctx);
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "%s: %s", what, hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  mvs_sync(ctx);
  pvlm_i_free(ctx, d_quad);
  pvlm_i_free(ctx, d_img); pvlm_i_free(ctx, d_unit); pvlm_i_free(ctx, d_depth); pvlm_i_free(ctx, d_normal); pvlm_i_free(ctx, d_conf); pvlm_i_free(ctx, d_ndepth); pvlm_i_free(ctx, d_const);
  return st;
}

pvlm_status pvlm_mvs_init_depth_normal(pvlm_ctx* ctx, int rows, int cols, const uint16_t* lidar_depth, const float* mask, float min_depth, float max_depth,
                                       int keep_lidar_constant, unsigned long long seed, float* depth, float* normal, unsigned char* depth_constant) {
  if (!ctx || rows <= 0 || cols <= 0 || !depth || !normal) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t npix = (size_t)rows * cols;
  unsigned short* d_l = nullptr; float *d_m = nullptr, *d_d = nullptr, *d_n = nullptr; unsigned char* d_c = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_d, npix);
  if (!st) st = pvlm_i_alloc(ctx, &d_n, npix * 3);
  if (!st && lidar_depth) st = pvlm_i_alloc(ctx, &d_l, npix);
  if (!st && mask) st = pvlm_i_alloc(ctx, &d_m, npix);
  const bool want_const = lidar_depth && keep_lidar_constant && depth_constant;
  if (!st && want_const) st = pvlm_i_alloc(ctx, &d_c, npix);
  if (!st) {
    hipStream_t s = ctx->stream;
    hipError_t e = hipSuccess;
    if (lidar_depth) e = mvs_up(ctx, d_l, lidar_depth, npix * sizeof(unsigned short));
    if (e == hipSuccess && mask) e = mvs_up(ctx, d_m, mask, npix * sizeof(float));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_mvs_init_depth_normal, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, rows, cols, d_l, d_m, min_depth, max_depth,
                         keep_lidar_constant, pvlm_mvs::pass_seed(seed, -2), d_d, d_n, d_c);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = mvs_down(ctx, depth, d_d, npix * sizeof(float));
    if (e == hipSuccess) e = mvs_down(ctx, normal, d_n, npix * 3 * sizeof(float));
    if (e == hipSuccess && want_const) e = mvs_down(ctx, depth_constant, d_c, npix);
    if (e == hipSuccess) e = mvs_sync(ctx);
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_init_depth_normal: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  mvs_sync(ctx);
  pvlm_i_free(ctx, d_l); pvlm_i_free(ctx, d_m); pvlm_i_free(ctx, d_d); pvlm_i_free(ctx, d_n); pvlm_i_free(ctx, d_c);
  return st;
}

// MVS::RemoveSmallSegments (mvs/MVS.cpp:1504-1577).  HOST code on purpose: the region a pixel ends up in depends on the order in
// which seeds are visited (column-major) because the similarity test divides by the depth of the pixel a neighbour is reached
// FROM — a data-parallel labelling would merge regions the reference keeps apart.  One pass over the image with an explicit
// frontier; 1440 x 720 takes a few milliseconds, once per view at the end of its estimation (:102, :138).
pvlm_status pvlm_mvs_remove_small_segments(pvlm_ctx* ctx, int rows, int cols, float depth_diff_threshold, int min_segment, float* depth, float* normal,
                                           float* conf, int64_t* removed) {
  if (!ctx || rows <= 0 || cols <= 0 || !depth || !normal || !conf) return PVLM_ERR_ARG;
  const size_t npix = (size_t)rows * cols;
  std::vector<unsigned char> claimed(npix, 0);
  std::vector<int> frontier(npix);
  int64_t gone = 0;
  const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
  for (int u = 0; u < cols; ++u)
    for (int v = 0; v < rows; ++v) {
      const size_t seed = (size_t)v * cols + u;
      if (claimed[seed]) continue;
      size_t head = 0, tail = 0;
      frontier[tail++] = (int)seed;
      while (head < tail) {
        const int cur = frontier[head++];
        const int cx = cur % cols, cy = cur / cols;
        const float here = depth[cur];
        for (int k = 0; k < 4; ++k) {
          const int x = cx + dx[k], y = cy + dy[k];
          if (x < 0 || y < 0 || x >= cols || y >= rows) continue;
          const size_t nb = (size_t)y * cols + x;
          if (claimed[nb]) continue;
          const float there = depth[nb];
          if (there > 0 && std::fabs((here - there) / here) < depth_diff_threshold) { frontier[tail++] = (int)nb; claimed[nb] = 1; }
        }
        claimed[(size_t)cur] = 1;
      }
      if (tail < (size_t)min_segment) {
        for (size_t i = 0; i < tail; ++i) {
          const size_t e = (size_t)frontier[i];
          depth[e] = 0.f; normal[3 * e] = 0.f; normal[3 * e + 1] = 0.f; normal[3 * e + 2] = 0.f; conf[e] = -1.f;
        }
        gone += (int64_t)tail;
      }
    }
  if (removed) *removed = gone;
  return PVLM_OK;
}

pvlm_status pvlm_mvs_init_conf_map(pvlm_ctx* ctx, int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                   const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                                   const float* const* nei_depth) {
  return mvs_run(ctx, "pvlm_mvs_init_conf_map", rows, cols, half_window, step, ref_gray, n_neighbors, nei_gray, R_nr, t_nr, depth, normal, conf, nei_depth, nullptr,
                 0.f, 0.f, 0ull, -1, 0.f);
}

pvlm_status pvlm_mvs_propagate(pvlm_ctx* ctx, int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                               const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                               const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth,
                               unsigned long long seed, int max_iter, float conf_threshold) {
  if (max_iter < 0) return PVLM_ERR_ARG;
  return mvs_run(ctx, "pvlm_mvs_propagate", rows, cols, half_window, step, ref_gray, n_neighbors, nei_gray, R_nr, t_nr, depth, normal, conf, nei_depth,
                 depth_constant, min_depth, max_depth, seed, max_iter, conf_threshold);
}

pvlm_status pvlm_mvs_propagate_sequential(pvlm_ctx* ctx, int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                          const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                                          const float* const* nei_depth, const unsigned char* depth_constant, float min_depth, float max_depth,
                                          unsigned long long seed, int max_iter, float conf_threshold) {
  if (max_iter < 0) return PVLM_ERR_ARG;
  return mvs_run(ctx, "pvlm_mvs_propagate_sequential", rows, cols, half_window, step, ref_gray, n_neighbors, nei_gray, R_nr, t_nr, depth, normal, conf, nei_depth,
                 depth_constant, min_depth, max_depth, seed, max_iter, conf_threshold, 2);
}


// ---- resident view set: the same kernels with the maps kept in HBM between the scoring pass, the sweeps and the
// fusion filter (the per-call entry points above move 30-50 MB over PCIe per call, which is most of their wall time) ----
// ---- MVS::DepthImageToCloud / DepthNormalToCloud (mvs/MVS.cpp:2073-2142): ordered stream compaction of a depth map into world points.
// Three launches: kept pixels per 256-pixel block, an exclusive scan of the block counts by one workgroup, and the emit pass
// (block base + rank of the lane among the kept lanes before it), so that the points come out in the reference's raster order.
struct pvlm_cloud_pose { double T[12]; };

__device__ __forceinline__ bool cloud_lane_keeps(long long e, long long npix, const float* depth, const unsigned char* bgr, float max_depth, int filter_sky) {
  if (e >= npix) return false;
  return pvlm_mvs::cloud_keeps(depth[e], max_depth, bgr + 3 * e, filter_sky != 0);
}

__global__ void __launch_bounds__(256) k_cloud_count(long long npix, const float* __restrict__ depth, const unsigned char* __restrict__ bgr, float max_depth,
                                                     int filter_sky, unsigned* __restrict__ block_count) {
  __shared__ unsigned wave_n[4];
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long m = __ballot(cloud_lane_keeps(e, npix, depth, bgr, max_depth, filter_sky));
  if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
}

// exclusive scan in place; block_count[n_blocks] = total.  One workgroup of 1024 lanes, 1024 counts per round.
__global__ void __launch_bounds__(1024) k_cloud_scan(unsigned n_blocks, unsigned* __restrict__ block_count) {
  __shared__ unsigned long long wave_sum[16];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (unsigned base = 0; base < n_blocks; base += 1024) {
    const unsigned i = base + threadIdx.x;
    const unsigned long long v = i < n_blocks ? block_count[i] : 0;
    unsigned long long inc = v;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wave_sum[w] = inc;
    __syncthreads();
    unsigned long long before = carry;
    for (int k = 0; k < w; ++k) before += wave_sum[k];
    if (i < n_blocks) block_count[i] = (unsigned)(before + inc - v);
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) block_count[n_blocks] = (unsigned)carry;
}

__global__ void __launch_bounds__(256) k_cloud_emit(int rows, int cols, const float* __restrict__ depth, const unsigned char* __restrict__ bgr,
                                                    const float* __restrict__ normal, const float* __restrict__ unit, pvlm_cloud_pose pose, float max_depth,
                                                    int filter_sky, const unsigned* __restrict__ block_base, float* __restrict__ xyz,
                                                    unsigned char* __restrict__ rgb, float* __restrict__ normal_out) {
  __shared__ unsigned wave_n[4];
  const long long npix = (long long)rows * cols, e = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool keep = cloud_lane_keeps(e, npix, depth, bgr, max_depth, filter_sky);
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wave_n[w] = (unsigned)__popcll(m);
  __syncthreads();
  if (!keep) return;
  unsigned at = block_base[blockIdx.x] + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
  for (int k = 0; k < w; ++k) at += wave_n[k];
  float ray[3];
  if (unit) { ray[0] = unit[3 * e]; ray[1] = unit[3 * e + 1]; ray[2] = unit[3 * e + 2]; }
  else pvlm_mvs::unit_ray(rows, cols, (int)(e % cols), (int)(e / cols), ray);
  float p[3];
  pvlm_mvs::cloud_point(ray, depth[e], pose.T, p);
  xyz[3 * (size_t)at] = p[0]; xyz[3 * (size_t)at + 1] = p[1]; xyz[3 * (size_t)at + 2] = p[2];
  rgb[3 * (size_t)at] = bgr[3 * e + 2]; rgb[3 * (size_t)at + 1] = bgr[3 * e + 1]; rgb[3 * (size_t)at + 2] = bgr[3 * e];
  if (normal_out) {
    const float n[3] = {normal[3 * e], normal[3 * e + 1], normal[3 * e + 2]};
    pvlm_mvs::cloud_normal(n, pose.T, p);
    normal_out[3 * (size_t)at] = p[0]; normal_out[3 * (size_t)at + 1] = p[1]; normal_out[3 * (size_t)at + 2] = p[2];
  }
}

// depth / normal / unit: device maps (normal, unit may be null); bgr, outputs: host.  Synchronises.
static pvlm_status depth_to_cloud(pvlm_ctx* ctx, const char* who, int rows, int cols, const float* d_depth, const float* d_normal, const float* d_unit,
                                  const unsigned char* bgr, const double* T_wc, float max_depth, int filter_sky, float* xyz, unsigned char* rgb,
                                  float* normal_out, long long* n_points) {
  const size_t npix = (size_t)rows * cols;
  const unsigned n_blocks = (unsigned)((npix + 255) / 256);
  unsigned char *d_bgr = nullptr, *d_rgb = nullptr; unsigned* d_cnt = nullptr; float *d_xyz = nullptr, *d_nout = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_bgr, npix * 3);
  if (!st) st = pvlm_i_alloc(ctx, &d_cnt, (size_t)n_blocks + 1);
  if (!st) st = pvlm_i_alloc(ctx, &d_xyz, npix * 3);
  if (!st) st = pvlm_i_alloc(ctx, &d_rgb, npix * 3);
  if (!st && normal_out) st = pvlm_i_alloc(ctx, &d_nout, npix * 3);
  if (!st) {
    hipStream_t s = ctx->stream;
    pvlm_cloud_pose pose;
    for (int k = 0; k < 12; ++k) pose.T[k] = T_wc[k];
    hipError_t e = mvs_up(ctx, d_bgr, bgr, npix * 3);
    if (e == hipSuccess) {
      pvlm_prof_scope prof(ctx, 1);   // the three launches are one interval of the "materialise" slot (tools/mvs_bench.py)
      hipLaunchKernelGGL(k_cloud_count, dim3(n_blocks), dim3(256), 0, s, (long long)npix, d_depth, d_bgr, max_depth, filter_sky, d_cnt);
      hipLaunchKernelGGL(k_cloud_scan, dim3(1), dim3(1024), 0, s, n_blocks, d_cnt);
      hipLaunchKernelGGL(k_cloud_emit, dim3(n_blocks), dim3(256), 0, s, rows, cols, d_depth, d_bgr, d_normal, d_unit, pose, max_depth, filter_sky, d_cnt, d_xyz, d_rgb, d_nout);
      e = hipGetLastError();
    }
    unsigned total = 0;
    if (e == hipSuccess) e = mvs_down(ctx, &total, d_cnt + n_blocks, sizeof(unsigned));
    if (e == hipSuccess) e = mvs_sync(ctx);
    if (e == hipSuccess && total > 0) {
      e = mvs_down(ctx, xyz, d_xyz, (size_t)total * 3 * sizeof(float));
      if (e == hipSuccess) e = mvs_down(ctx, rgb, d_rgb, (size_t)total * 3);
      if (e == hipSuccess && normal_out) e = mvs_down(ctx, normal_out, d_nout, (size_t)total * 3 * sizeof(float));
      if (e == hipSuccess) e = mvs_sync(ctx);
    }
    if (e == hipSuccess) *n_points = (long long)total;
    else { PVLM_SET_ERR(ctx, "%s: %s", who, hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  mvs_sync(ctx);
  pvlm_i_free(ctx, d_bgr); pvlm_i_free(ctx, d_cnt); pvlm_i_free(ctx, d_xyz); pvlm_i_free(ctx, d_rgb); pvlm_i_free(ctx, d_nout);
  return st;
}

pvlm_status pvlm_mvs_depth_to_cloud(pvlm_ctx* ctx, int rows, int cols, const float* depth, const unsigned char* bgr, const float* normal, const double* T_wc,
                                    float max_depth, int filter_sky, float* xyz, unsigned char* rgb, float* normal_out, long long* n_points) {
  if (!ctx || rows <= 0 || cols <= 0 || !depth || !bgr || !T_wc || !xyz || !rgb || !n_points || (normal_out && !normal)) return PVLM_ERR_ARG;
  if ((size_t)rows * cols > 0xfffffff0ull) { PVLM_SET_ERR(ctx, "pvlm_mvs_depth_to_cloud: %d x %d pixels exceed the 32-bit point index", rows, cols); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t npix = (size_t)rows * cols;
  float *d_depth = nullptr, *d_normal = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_depth, npix);
  if (!st && normal_out) st = pvlm_i_alloc(ctx, &d_normal, npix * 3);
  if (!st && mvs_up(ctx, d_depth, depth, npix * sizeof(float)) != hipSuccess) st = PVLM_ERR_HIP;
  if (!st && normal_out && mvs_up(ctx, d_normal, normal, npix * 3 * sizeof(float)) != hipSuccess) st = PVLM_ERR_HIP;
  if (!st) st = depth_to_cloud(ctx, "pvlm_mvs_depth_to_cloud", rows, cols, d_depth, d_normal, nullptr, bgr, T_wc, max_depth, filter_sky, xyz, rgb, normal_out, n_points);
  mvs_sync(ctx);
  pvlm_i_free(ctx, d_depth); pvlm_i_free(ctx, d_normal);
  return st;
}

struct pvlm_mvs_views {
  int rows = 0, cols = 0, n = 0;
  size_t npix = 0;
  unsigned char* d_gray = nullptr;                      // n x npix
  unsigned* d_quad = nullptr;                           // n x npix: the 2x2-quad copies the taps read (k_mvs_quad, rebuilt whenever a grey image is uploaded)
  float *d_depth = nullptr, *d_normal = nullptr, *d_conf = nullptr, *d_depth_filter = nullptr, *d_conf_filter = nullptr;   // n x npix (normal: x 3)
  float* d_unit = nullptr;                              // npix x 3 (PreComputeI2C)
  unsigned long long* d_key = nullptr;                  // 16 x npix splat keys (fusion filter scratch)
  unsigned char* d_const = nullptr;                     // npix (depth_constant of the view being processed)
};

static bool views_ids_ok(const pvlm_mvs_views* v, int ref, int n_neighbors, const int* nei) {
  if (!v || ref < 0 || ref >= v->n || n_neighbors < 0 || n_neighbors > 16 || (n_neighbors > 0 && !nei)) return false;
  for (int b = 0; b < n_neighbors; ++b) if (nei[b] < 0 || nei[b] >= v->n || nei[b] == ref) return false;
  return true;
}

pvlm_status pvlm_mvs_views_depth_to_cloud(pvlm_ctx* ctx, pvlm_mvs_views* v, int view, int use_filtered_depth, const unsigned char* bgr, const double* T_wc,
                                          float max_depth, int filter_sky, float* xyz, unsigned char* rgb, float* normal_out, long long* n_points) {
  if (!ctx || !v || view < 0 || view >= v->n || !bgr || !T_wc || !xyz || !rgb || !n_points) return PVLM_ERR_ARG;
  if (v->npix > 0xfffffff0ull) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const float* d_depth = (use_filtered_depth ? v->d_depth_filter : v->d_depth) + (size_t)view * v->npix;
  return depth_to_cloud(ctx, "pvlm_mvs_views_depth_to_cloud", v->rows, v->cols, d_depth, v->d_normal + (size_t)view * v->npix * 3, v->d_unit, bgr, T_wc, max_depth,
                        filter_sky, xyz, rgb, normal_out, n_points);
}

pvlm_status pvlm_mvs_views_destroy(pvlm_ctx* ctx, pvlm_mvs_views* v) {
  if (!ctx) return PVLM_ERR_ARG;
  if (!v) return PVLM_OK;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  mvs_sync(ctx);
  pvlm_i_free(ctx, v->d_quad);
  pvlm_i_free(ctx, v->d_gray); pvlm_i_free(ctx, v->d_depth); pvlm_i_free(ctx, v->d_normal); pvlm_i_free(ctx, v->d_conf); pvlm_i_free(ctx, v->d_depth_filter); pvlm_i_free(ctx, v->d_conf_filter);
  pvlm_i_free(ctx, v->d_unit); pvlm_i_free(ctx, v->d_key); pvlm_i_free(ctx, v->d_const);
  delete v;
  return PVLM_OK;
}

pvlm_status pvlm_mvs_views_create(pvlm_ctx* ctx, int rows, int cols, int n_views, pvlm_mvs_views** out) {
  if (!ctx || !out || rows <= 0 || cols <= 0 || n_views <= 0 || (size_t)rows * cols > 0xffffffffull) return PVLM_ERR_ARG;
  *out = nullptr;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  pvlm_mvs_views* v = new pvlm_mvs_views();
  v->rows = rows; v->cols = cols; v->n = n_views; v->npix = (size_t)rows * cols;
  const size_t all = v->npix * (size_t)n_views;
  pvlm_status st = pvlm_i_alloc(ctx, &v->d_gray, all);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_quad, all);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_depth, all);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_normal, all * 3);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_conf, all);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_depth_filter, all);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_conf_filter, all);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_unit, v->npix * 3);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_key, v->npix * 16);
  if (!st) st = pvlm_i_alloc(ctx, &v->d_const, v->npix);
  if (!st) {
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemsetAsync(v->d_gray, 0, all, s);
    if (e == hipSuccess) e = hipMemsetAsync(v->d_quad, 0, all * sizeof(unsigned), s);
    float* zero[5] = {v->d_depth, v->d_conf, v->d_depth_filter, v->d_conf_filter, v->d_normal};
    for (int k = 0; k < 5 && e == hipSuccess; ++k) e = hipMemsetAsync(zero[k], 0, all * sizeof(float) * (k == 4 ? 3 : 1), s);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_mvs_unit_table, dim3((unsigned)((v->npix + 255) / 256)), dim3(256), 0, s, rows, cols, v->d_unit); e = hipGetLastError(); }
    if (e == hipSuccess) e = mvs_sync(ctx);
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_views_create: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  if (st) { pvlm_mvs_views_destroy(ctx, v); return st; }
  *out = v;
  return PVLM_OK;
}

pvlm_status pvlm_mvs_views_upload(pvlm_ctx* ctx, pvlm_mvs_views* v, int view, const unsigned char* gray, const float* depth, const float* normal,
                                  const float* conf) {
  if (!ctx || !v || view < 0 || view >= v->n) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t o = v->npix * (size_t)view;
  hipError_t e = hipSuccess;
  if (gray) {
    e = mvs_up(ctx, v->d_gray + o, gray, v->npix);
    if (e == hipSuccess) hipLaunchKernelGGL(k_mvs_quad, dim3((unsigned)((v->npix + 255) / 256)), dim3(256), 0, ctx->stream, v->rows, v->cols, v->d_gray + o, v->d_quad + o);
  }
  if (e == hipSuccess && depth) e = mvs_up(ctx, v->d_depth + o, depth, v->npix * sizeof(float));
  if (e == hipSuccess && normal) e = mvs_up(ctx, v->d_normal + 3 * o, normal, v->npix * 3 * sizeof(float));
  if (e == hipSuccess && conf) e = mvs_up(ctx, v->d_conf + o, conf, v->npix * sizeof(float));
  if (e == hipSuccess) e = mvs_sync(ctx);
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_views_upload: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  return PVLM_OK;
}

pvlm_status pvlm_mvs_views_download(pvlm_ctx* ctx, pvlm_mvs_views* v, int view, float* depth, float* normal, float* conf, float* depth_filter,
                                    float* conf_filter) {
  if (!ctx || !v || view < 0 || view >= v->n) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t o = v->npix * (size_t)view, bytes = v->npix * sizeof(float);
  hipError_t e = hipSuccess;
  if (depth) e = mvs_down(ctx, depth, v->d_depth + o, bytes);
  if (e == hipSuccess && normal) e = mvs_down(ctx, normal, v->d_normal + 3 * o, 3 * bytes);
  if (e == hipSuccess && conf) e = mvs_down(ctx, conf, v->d_conf + o, bytes);
  if (e == hipSuccess && depth_filter) e = mvs_down(ctx, depth_filter, v->d_depth_filter + o, bytes);
  if (e == hipSuccess && conf_filter) e = mvs_down(ctx, conf_filter, v->d_conf_filter + o, bytes);
  { const hipError_t e2 = mvs_sync(ctx); if (e == hipSuccess) e = e2; }
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_views_download: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  return PVLM_OK;
}

pvlm_status pvlm_mvs_views_snapshot_depth(pvlm_ctx* ctx, pvlm_mvs_views* v, int view) {
  if (!ctx || !v || view < 0 || view >= v->n) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  const size_t o = v->npix * (size_t)view;
  hipError_t e = hipMemcpyAsync(v->d_depth_filter + o, v->d_depth + o, v->npix * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_views_snapshot_depth: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  return PVLM_OK;
}

static void views_neighbours(const pvlm_mvs_views* v, int n_neighbors, const int* nei, const float* R_nr, const float* t_nr, bool geometry, pvlm_mvs_neighbours& nb) {
  nb.n = n_neighbors; nb.geometric = geometry ? 1 : 0;
  for (int b = 0; b < n_neighbors; ++b) {
    nb.quad[b] = v->d_quad + v->npix * (size_t)nei[b];
    nb.depth[b] = geometry ? v->d_depth_filter + v->npix * (size_t)nei[b] : nullptr;
    for (int k = 0; k < 9; ++k) nb.R[b][k] = R_nr[9 * b + k];
    for (int k = 0; k < 3; ++k) nb.t[b][k] = t_nr[3 * b + k];
  }
}

// the scoring pass (max_iter < 0) or the PatchMatch sweep of view `ref` against resident neighbours; asynchronous on ctx->stream
static pvlm_status views_estimate(pvlm_ctx* ctx, pvlm_mvs_views* v, int ref, int n_neighbors, const int* nei, const float* R_nr, const float* t_nr,
                                  int half_window, int step, int use_geometry, const unsigned char* depth_constant, float min_depth, float max_depth,
                                  unsigned long long seed, int max_iter, float conf_threshold, int strategy) {
  if (!ctx || !views_ids_ok(v, ref, n_neighbors, nei) || half_window < 1 || step < 1 || (n_neighbors > 0 && (!R_nr || !t_nr))) return PVLM_ERR_ARG;
  if (pvlm_mvs::num_texels(half_window, step) > 64 * PVLM_MVS_MAXM) { PVLM_SET_ERR(ctx, "NCC window of %d texels exceeds %d", pvlm_mvs::num_texels(half_window, step), 64 * PVLM_MVS_MAXM); return PVLM_ERR_ARG; }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  hipStream_t s = ctx->stream;
  pvlm_mvs_neighbours nb;
  views_neighbours(v, n_neighbors, nei, R_nr, t_nr, use_geometry != 0, nb);
  const size_t o = v->npix * (size_t)ref;
  hipError_t e = hipSuccess;
  if (depth_constant) { e = mvs_up(ctx, v->d_const, depth_constant, v->npix); if (e == hipSuccess) e = mvs_sync(ctx); }
  unsigned char* d_const = depth_constant ? v->d_const : nullptr;
  if (e == hipSuccess) {
    if (max_iter < 0) {
      float* d_wtab = nullptr;
      const size_t wtab_floats = mvs_lane_bands(v->rows, v->cols, half_window, step).floats;
      if (wtab_floats && pvlm_i_alloc(ctx, &d_wtab, wtab_floats)) d_wtab = nullptr;
      {
        pvlm_prof_scope prof(ctx, 1);
        launch_mvs_conf(s, v->rows, v->cols, half_window, step, v->d_gray + o, v->d_unit, nb, v->d_depth + o, v->d_normal + 3 * o, v->d_conf + o, d_wtab);
      }
      pvlm_i_free(ctx, d_wtab);
    } else {
      if (strategy == 2) {
        MvsFlow flow;
        const bool have_flow = mvs_flow_begin(ctx, v->rows, v->cols, half_window, step, &flow);
        for (int iter = 0; iter < max_iter; ++iter) {
          pvlm_prof_scope prof(ctx, 1);
          launch_mvs_propagate_sequential(s, v->rows, v->cols, half_window, step, v->d_gray + o, v->d_unit, nb, v->d_depth + o, v->d_normal + 3 * o, v->d_conf + o,
                                          d_const, min_depth, max_depth, pvlm_mvs::pass_seed(seed, iter), iter, have_flow ? &flow : nullptr);
        }
        mvs_flow_end(ctx, &flow);
      } else {
        float* d_wtab = nullptr;
        const size_t wtab_floats = mvs_lane_bands(v->rows, (v->cols + 1) / 2, half_window, step).floats;
        if (wtab_floats && pvlm_i_alloc(ctx, &d_wtab, wtab_floats)) d_wtab = nullptr;
        for (int iter = 0; iter < max_iter; ++iter)
          for (int offset = 0; offset <= 1; ++offset) {
            pvlm_prof_scope prof(ctx, 1);
            launch_mvs_propagate(s, v->rows, v->cols, half_window, step, v->d_gray + o, v->d_unit, nb, v->d_depth + o, v->d_normal + 3 * o, v->d_conf + o, d_const,
                                 min_depth, max_depth, pvlm_mvs::pass_seed(seed, 2 * iter + offset), offset, d_wtab);
          }
        pvlm_i_free(ctx, d_wtab);
      }
      hipLaunchKernelGGL(k_mvs_threshold, dim3((unsigned)((v->npix + 255) / 256)), dim3(256), 0, s, (long long)v->npix, d_const, conf_threshold, v->d_depth + o,
                         v->d_normal + 3 * o, v->d_conf + o);
    }
    e = hipGetLastError();
  }
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_views_estimate: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  return PVLM_OK;
}

pvlm_status pvlm_mvs_views_estimate(pvlm_ctx* ctx, pvlm_mvs_views* v, int ref, int n_neighbors, const int* nei, const float* R_nr, const float* t_nr,
                                    int half_window, int step, int use_geometry, const unsigned char* depth_constant, float min_depth, float max_depth,
                                    unsigned long long seed, int max_iter, float conf_threshold) {
  return views_estimate(ctx, v, ref, n_neighbors, nei, R_nr, t_nr, half_window, step, use_geometry, depth_constant, min_depth, max_depth, seed, max_iter, conf_threshold, 1);
}
pvlm_status pvlm_mvs_views_estimate_sequential(pvlm_ctx* ctx, pvlm_mvs_views* v, int ref, int n_neighbors, const int* nei, const float* R_nr, const float* t_nr,
                                               int half_window, int step, int use_geometry, const unsigned char* depth_constant, float min_depth, float max_depth,
                                               unsigned long long seed, int max_iter, float conf_threshold) {
  if (max_iter < 0) return PVLM_ERR_ARG;
  return views_estimate(ctx, v, ref, n_neighbors, nei, R_nr, t_nr, half_window, step, use_geometry, depth_constant, min_depth, max_depth, seed, max_iter, conf_threshold, 2);
}

// EstimateDepthMapSingle(SEQUENTIAL) of several views at once — the views of one call must be distinct; job j uses the nei_counts[j]
// neighbours that follow those of job j - 1 in nei / R_nr / t_nr.  depth_constant: NULL, or n_jobs pointers (NULL entries allowed).
pvlm_status pvlm_mvs_views_estimate_sequential_batch(pvlm_ctx* ctx, pvlm_mvs_views* v, int n_jobs, const int* refs, const int* nei_counts, const int* nei,
                                                     const float* R_nr, const float* t_nr, int half_window, int step, int use_geometry,
                                                     const unsigned char* const* depth_constant, float min_depth, float max_depth,
                                                     const unsigned long long* seeds, int max_iter, float conf_threshold) {
  if (!ctx || !v || n_jobs < 0 || max_iter < 0 || half_window < 1 || step < 1 || (n_jobs > 0 && (!refs || !nei_counts || !seeds))) return PVLM_ERR_ARG;
  if (n_jobs == 0) return PVLM_OK;
  if (pvlm_mvs::num_texels(half_window, step) > 64 * PVLM_MVS_MAXM) { PVLM_SET_ERR(ctx, "NCC window of %d texels exceeds %d", pvlm_mvs::num_texels(half_window, step), 64 * PVLM_MVS_MAXM); return PVLM_ERR_ARG; }
  std::vector<pvlm_mvs_job> jobs((size_t)n_jobs);
  std::vector<char> seen((size_t)v->n, 0);
  size_t at = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (!views_ids_ok(v, refs[j], nei_counts[j], nei ? nei + at : nullptr) || (nei_counts[j] > 0 && (!R_nr || !t_nr))) { PVLM_SET_ERR(ctx, "job %d: bad view ids", j); return PVLM_ERR_ARG; }
    if (seen[(size_t)refs[j]]) { PVLM_SET_ERR(ctx, "view %d is the reference of two jobs of one batch", refs[j]); return PVLM_ERR_ARG; }
    seen[(size_t)refs[j]] = 1;
    views_neighbours(v, nei_counts[j], nei + at, R_nr + 9 * at, t_nr + 3 * at, use_geometry != 0, jobs[(size_t)j].nb);
    jobs[(size_t)j].off = v->npix * (size_t)refs[j]; jobs[(size_t)j].seed = seeds[j]; jobs[(size_t)j].dconst = nullptr;
    at += (size_t)nei_counts[j];
  }
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  hipStream_t s = ctx->stream;
  pvlm_mvs_job* d_jobs = nullptr; unsigned char* d_cb = nullptr;
  pvlm_status st = pvlm_i_alloc(ctx, &d_jobs, (size_t)n_jobs);
  hipError_t e = hipSuccess;
  if (!st && depth_constant) {
    st = pvlm_i_alloc(ctx, &d_cb, (size_t)n_jobs * v->npix);
    for (int j = 0; j < n_jobs && !st; ++j)
      if (depth_constant[j]) {
        if (mvs_up(ctx, d_cb + (size_t)j * v->npix, depth_constant[j], v->npix) != hipSuccess) st = PVLM_ERR_HIP;
        jobs[(size_t)j].dconst = d_cb + (size_t)j * v->npix;
      }
  }
  if (!st && mvs_up(ctx, d_jobs, jobs.data(), jobs.size() * sizeof(pvlm_mvs_job)) != hipSuccess) st = PVLM_ERR_HIP;
  // Per anti-diagonal: one wave per pixel (k_mvs_propagate_diag_batch); the thread-per-pixel form of a diagonal is a measured variant (above).
  const int n_tex = pvlm_mvs::num_texels(half_window, step);
  float* d_wtab = nullptr;
  // four threads per pixel from PVLM_MVS_QUAD_MIN pixels per diagonal over all jobs (0 = never); below that, one wave per pixel
  static const long long quad_min = getenv("PVLM_MVS_QUAD_MIN") ? atoll(getenv("PVLM_MVS_QUAD_MIN")) : 32768;
  const int longest_diag = std::min(v->rows, v->cols);
  float* d_qtab = nullptr;
  if (!st && quad_min > 0 && mvs_lane_form(n_tex) && (long long)n_jobs * longest_diag >= quad_min)
    if (pvlm_i_alloc(ctx, &d_qtab, (size_t)n_tex * (size_t)n_jobs * (size_t)((longest_diag + 63) / 64) * 64)) d_qtab = nullptr;
#if PVLM_MEASURED_VARIANTS
  static const long long lane_min = getenv("PVLM_MVS_LANE_BATCH_MIN") ? atoll(getenv("PVLM_MVS_LANE_BATCH_MIN")) : (1ll << 40);
  const int longest = std::min(v->rows, v->cols);
  if (!st && mvs_lane_form(n_tex) && (long long)n_jobs * longest >= lane_min)
    if (pvlm_i_alloc(ctx, &d_wtab, (size_t)n_tex * (size_t)n_jobs * (size_t)((longest + 63) / 64) * 64)) d_wtab = nullptr;
#endif
  if (!st) {
    const int n_diag = v->rows + v->cols - 1;
    const bool small = n_tex <= 64;
    for (int iter = 0; iter < max_iter; ++iter) {
      pvlm_prof_scope prof(ctx, 1);
      for (int q = 0; q < n_diag; ++q) {
        const int d = (iter & 1) ? n_diag - 1 - q : q;
        const int len = std::min(v->rows - 1, d) - std::max(0, d - (v->cols - 1)) + 1;
        if (d_qtab && (long long)n_jobs * len >= quad_min) {
          const int groups = (len + 63) / 64;
          hipLaunchKernelGGL(k_mvs_propagate_diag_batch_quad, dim3((unsigned)(groups * n_jobs)), dim3(256), (size_t)n_tex * 256 * sizeof(float), s, v->rows, v->cols, half_window,
                             step, v->d_gray, v->d_unit, d_jobs, v->d_depth, v->d_normal, v->d_conf, min_depth, max_depth, iter, d, groups, d_qtab);
          continue;
        }
#if PVLM_MEASURED_VARIANTS
        if (d_wtab && (long long)n_jobs * len >= lane_min) {
          const int groups = (len + 63) / 64;
          hipLaunchKernelGGL(k_mvs_propagate_diag_batch_lane, dim3((unsigned)(groups * n_jobs)), dim3(64), (size_t)n_tex * 64 * sizeof(float), s, v->rows, v->cols, half_window,
                             step, v->d_gray, v->d_unit, d_jobs, v->d_depth, v->d_normal, v->d_conf, min_depth, max_depth, iter, d, groups, d_wtab);
          continue;
        }
#endif
        const dim3 grid((unsigned)((len + 3) / 4), (unsigned)n_jobs), block(256);
        if (small)
          hipLaunchKernelGGL(k_mvs_propagate_diag_batch<1>, grid, block, 0, s, v->rows, v->cols, half_window, step, v->d_gray, v->d_unit, d_jobs, v->d_depth, v->d_normal,
                             v->d_conf, min_depth, max_depth, iter, d);
        else
          hipLaunchKernelGGL(k_mvs_propagate_diag_batch<PVLM_MVS_MAXM>, grid, block, 0, s, v->rows, v->cols, half_window, step, v->d_gray, v->d_unit, d_jobs, v->d_depth,
                             v->d_normal, v->d_conf, min_depth, max_depth, iter, d);
      }
    }
    for (int j = 0; j < n_jobs; ++j) {
      const size_t o = jobs[(size_t)j].off;
      hipLaunchKernelGGL(k_mvs_threshold, dim3((unsigned)((v->npix + 255) / 256)), dim3(256), 0, s, (long long)v->npix, jobs[(size_t)j].dconst, conf_threshold, v->d_depth + o,
                         v->d_normal + 3 * o, v->d_conf + o);
    }
    e = hipGetLastError();
    if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_views_estimate_sequential_batch: %s", hipGetErrorString(e)); st = PVLM_ERR_HIP; }
  }
  // the job table and the depth_constant copies go back to the pool in stream order; the staged uploads must have left the arena
  if (mvs_sync(ctx) != hipSuccess && !st) st = PVLM_ERR_HIP;
  pvlm_i_free(ctx, d_jobs); pvlm_i_free(ctx, d_cb); pvlm_i_free(ctx, d_wtab); pvlm_i_free(ctx, d_qtab);
  return st;
}

// FilterDepthImageRefine of view `ref`: reads depth / conf of the neighbours, writes depth_filter / conf_filter of ref and zeroes
// conf of ref where depth <= 0 (in place, as upstream); asynchronous on ctx->stream
pvlm_status pvlm_mvs_views_filter_refine(pvlm_ctx* ctx, pvlm_mvs_views* v, int ref, int n_neighbors, const int* nei, const float* R_nr, const float* t_nr,
                                         const unsigned char* depth_constant, float depth_diff_threshold, float min_depth, float max_depth) {
  if (!ctx || !views_ids_ok(v, ref, n_neighbors, nei) || (n_neighbors > 0 && (!R_nr || !t_nr))) return PVLM_ERR_ARG;
  if (pvlm_i_bind(ctx)) return PVLM_ERR_HIP;
  hipStream_t s = ctx->stream;
  const size_t npix = v->npix, o = npix * (size_t)ref;
  const unsigned grid = (unsigned)((npix + 255) / 256);
  hipError_t e = hipSuccess;
  if (depth_constant) { e = mvs_up(ctx, v->d_const, depth_constant, npix); if (e == hipSuccess) e = mvs_sync(ctx); }
  if (e == hipSuccess && n_neighbors > 0) {
    hipLaunchKernelGGL(k_mvs_fill_u64, dim3((unsigned)((npix * n_neighbors + 255) / 256)), dim3(256), 0, s, (long long)(npix * n_neighbors), ~0ull, v->d_key);
    e = hipGetLastError();
  }
  pvlm_mvs::RefineViews nv;
  nv.n = n_neighbors;
  for (int b = 0; b < n_neighbors && e == hipSuccess; ++b) {
    pvlm_mvs_pose pose;
    pvlm_mvs::inverse_pose(R_nr + 9 * b, t_nr + 3 * b, pose.R_rn, pose.t_rn);
    nv.conf[b] = v->d_conf + npix * (size_t)nei[b];
    for (int k = 0; k < 9; ++k) nv.R[b][k] = R_nr[9 * b + k];
    for (int k = 0; k < 3; ++k) nv.t[b][k] = t_nr[3 * b + k];
    hipLaunchKernelGGL(k_mvs_project_conf, dim3(grid), dim3(256), 0, s, v->rows, v->cols, v->d_unit, v->d_depth + npix * (size_t)nei[b], pose, v->d_key + npix * (size_t)b);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_mvs_refine, dim3(grid), dim3(256), 0, s, v->rows, v->cols, nv, v->d_key, v->d_unit, v->d_depth + o, v->d_conf + o,
                       depth_constant ? v->d_const : nullptr, depth_diff_threshold, min_depth, max_depth, v->d_depth_filter + o, v->d_conf_filter + o);
    e = hipGetLastError();
  }
  if (e != hipSuccess) { PVLM_SET_ERR(ctx, "pvlm_mvs_views_filter_refine: %s", hipGetErrorString(e)); return PVLM_ERR_HIP; }
  return PVLM_OK;
}

}  // extern "C"

// pvlm_preload: HIP loads the code object of a translation unit at the first launch of one of its kernels (15 ms for the larger ones) — an empty launch from here
// moves that out of the first call that needs this file's kernels
__global__ void k_preload_mvs() {}
void pvlm_i_preload_mvs(hipStream_t s) { hipLaunchKernelGGL(k_preload_mvs, dim3(1), dim3(1), 0, s); }
