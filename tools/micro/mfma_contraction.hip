// micro-benchmark: the 7 x N . N x 7 contraction (S = sum a a^T, a = sqrt(rho')[v; r]) of the fused
// kernel on the fp64 VALU (28 FMAs per evaluation per lane, wave tree at the end) versus on the matrix
// core (v_mfma_f64_16x16x4_f64: per-lane data transposed through LDS, 16 MFMAs per 64 evaluations,
// 49 of 256 tile entries useful).  Prints ns per 64 evaluations per wave and the implied chip rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void make_a(int it, int lane, double* a) {
#pragma unroll
  for (int k = 0; k < 7; ++k) a[k] = 1e-3 * (double)((it * 7 + k) % 13) + 1e-4 * (double)lane;
}

__global__ __launch_bounds__(256) void k_valu(int iters, double* out) {
  const int lane = threadIdx.x & 63;
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  for (int it = 0; it < iters; ++it) {
    double a[7];
    make_a(it, lane, a);
    int q = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = i; j < 7; ++j) acc[q++] += a[i] * a[j];
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 28; ++k) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mfma(int iters, double* out) {
  __shared__ double stage[4][64][8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double4_t c = {0.0, 0.0, 0.0, 0.0};
  const int i = lane & 15, kq = lane >> 4;
  for (int it = 0; it < iters; ++it) {
    double a[7];
    make_a(it, lane, a);
#pragma unroll
    for (int k = 0; k < 7; ++k) stage[wv][lane][k] = a[k];
    stage[wv][lane][7] = 0.0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const double v = i < 8 ? stage[wv][4 * m + kq][i] : 0.0;
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, c, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
}

int main() {
  const int blocks = 256 * 8, iters = 4096;
  double* d;
  (void)hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0, 0);
      if (which == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, iters, d);
      else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, iters, d);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double evals = (double)blocks * 256 * iters;
    printf("%s: %.3f ms for %.3g evaluations -> %.1f G evals/s chip-wide (contraction only)\n", which == 0 ? "VALU 28 FMA/eval" : "MFMA f64 16x16x4 via LDS transpose",
           best, evals, evals / (best * 1e-3) / 1e9);
  }
  return 0;
}
