// Dev probe (not part of the product): does a cache-policy bit on the gather load
// change the granularity at which a random 8-byte read of a large vector is
// served (L2 line fill of 128 B vs a 32/64-B sector from the fabric / MALL)?
// Usage: gather_policy_probe [count_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int TPB = 256;
constexpr int U = 8;

template <int POL>
__device__ __forceinline__ double ld(const double *p) {
  double v;
  if (POL == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (POL == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (POL == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (POL == 4) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (POL == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  if (POL == 6) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if (POL == 7) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int POL>
__global__ __launch_bounds__(TPB) void gather_kernel(const int *__restrict__ idx, const double *__restrict__ x,
                                                     double *__restrict__ out, long count) {
  long base = ((long)blockIdx.x * TPB * U) + threadIdx.x;
  int c[U]; double v[U];
#pragma unroll
  for (int i = 0; i < U; ++i) { long k = base + (long)i * TPB; c[i] = k < count ? __builtin_nontemporal_load(idx + k) : 0; }
#pragma unroll
  for (int i = 0; i < U; ++i) v[i] = ld<POL>(x + c[i]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double s = 0;
#pragma unroll
  for (int i = 0; i < U; ++i) s += v[i];
  out[(long)blockIdx.x * TPB + threadIdx.x] = s;
}

template <typename F>
float time_it(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  long count = (argc > 1 ? atol(argv[1]) : 100) * 1000000L;
  std::vector<int> h(count);
  int *idx; CK(hipMalloc(&idx, count * 4));
  long nblocks = (count + (long)TPB * U - 1) / ((long)TPB * U);
  double *out; CK(hipMalloc(&out, nblocks * TPB * 8));
  const char *names[8] = {"plain", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  for (long N : {4096L, 16384L, 65536L, 262144L, 1000000L, 10000000L}) {   // 32 KB (L1), 128 KB, 512 KB, 2 MB (L2), 8 MB, 80 MB
    std::mt19937_64 rng(1);
    for (long i = 0; i < count; ++i) h[i] = (int)(rng() % N);
    CK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
    double *x; CK(hipMalloc(&x, N * 8)); CK(hipMemset(x, 0, N * 8));
    float t[8];
    t[0] = time_it([&] { hipLaunchKernelGGL(gather_kernel<0>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    t[1] = time_it([&] { hipLaunchKernelGGL(gather_kernel<1>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    t[2] = time_it([&] { hipLaunchKernelGGL(gather_kernel<2>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    t[3] = time_it([&] { hipLaunchKernelGGL(gather_kernel<3>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    t[4] = time_it([&] { hipLaunchKernelGGL(gather_kernel<4>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    t[5] = time_it([&] { hipLaunchKernelGGL(gather_kernel<5>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    t[6] = time_it([&] { hipLaunchKernelGGL(gather_kernel<6>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    t[7] = time_it([&] { hipLaunchKernelGGL(gather_kernel<7>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    printf("N=%9ld count=%ld random f64 gather:", N, count);
    for (int p = 0; p < 8; ++p) printf("  [%s] %.3f ms %.1f G/s", names[p], t[p], count / t[p] / 1e6);
    printf("\n");
    CK(hipFree(x));
  }
  return 0;
}
