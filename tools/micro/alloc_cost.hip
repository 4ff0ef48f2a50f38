// micro-benchmark: what do hipMalloc / hipFree / hipMallocAsync / hipMemset cost on this box as a function of size?
// (decides how pvlm_assoc_point2plane gets its scratch and its output: measured, not guessed)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  hipFree(nullptr);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const size_t GB = 1ull << 30;
  const size_t sizes[] = {GB / 64, GB / 4, GB, 4 * GB, 16 * GB, 48 * GB};
  for (int rep = 0; rep < 2; ++rep)
    for (size_t sz : sizes) {
      void* p = nullptr;
      double t0 = now();
      hipError_t e = hipMalloc(&p, sz);
      double t1 = now();
      if (e != hipSuccess) { printf("hipMalloc %zu failed\n", sz); continue; }
      hipMemsetAsync(p, 0, sz, s); hipStreamSynchronize(s);
      double t2 = now();
      hipMemsetAsync(p, 0, sz, s); hipStreamSynchronize(s);
      double t3 = now();
      hipFree(p);
      double t4 = now();
      printf("rep %d size %8.3f GB: hipMalloc %9.3f ms  first memset %9.3f ms  second memset %9.3f ms  hipFree %9.3f ms\n", rep, sz / (double)GB,
             (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    }
  // stream-ordered pool with an unbounded release threshold: second round should be free
  hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
  unsigned long long thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
  for (int rep = 0; rep < 3; ++rep)
    for (size_t sz : {GB, 16 * GB}) {
      void* p = nullptr;
      double t0 = now();
      hipError_t e = hipMallocAsync(&p, sz, s); hipStreamSynchronize(s);
      double t1 = now();
      if (e != hipSuccess) { printf("hipMallocAsync %zu failed\n", sz); continue; }
      hipFreeAsync(p, s); hipStreamSynchronize(s);
      double t2 = now();
      printf("pool rep %d size %8.3f GB: hipMallocAsync %9.3f ms  hipFreeAsync %9.3f ms\n", rep, sz / (double)GB, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    }
  // many medium blocks
  {
    std::vector<void*> v(64);
    double t0 = now();
    for (auto& p : v) hipMalloc(&p, 768ull << 20);
    double t1 = now();
    for (auto& p : v) hipFree(p);
    double t2 = now();
    printf("64 x 0.75 GB: hipMalloc %9.3f ms total, hipFree %9.3f ms total\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
  }
  // pinned host memory + D2H bandwidth (Ceres-feeding boundary)
  for (size_t sz : {GB / 4, 2 * GB}) {
    void *h = nullptr, *d = nullptr;
    double t0 = now();
    hipHostMalloc(&h, sz, hipHostMallocDefault);
    double t1 = now();
    hipMalloc(&d, sz);
    hipMemsetAsync(d, 1, sz, s); hipStreamSynchronize(s);
    double t2 = now();
    hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
    double t3 = now();
    hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
    double t4 = now();
    hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
    double t5 = now();
    printf("pinned %6.3f GB: hipHostMalloc %8.1f ms, D2H first %7.2f GB/s, D2H second %7.2f GB/s, H2D %7.2f GB/s\n", sz / (double)GB, (t1 - t0) * 1e3,
           sz / (t3 - t2) / 1e9, sz / (t4 - t3) / 1e9, sz / (t5 - t4) / 1e9);
    hipFree(d); hipHostFree(h);
  }
  return 0;
}
