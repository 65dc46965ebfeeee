// Dev probe: random fp64 gather rate as a function of the window size (uniform indices in
// [0, W)), with the 12 B/entry (index + value) stream of an SpMV read alongside -- what a
// column-slab pass of the stream layout would see.  Usage: window_probe [count_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int TPB = 256, U = 8;
__global__ __launch_bounds__(TPB) void k(const int *__restrict__ idx, const double *__restrict__ val,
                                         const double *__restrict__ x, double *__restrict__ out, long count) {
  long base = ((long)blockIdx.x * TPB * U) + threadIdx.x;
  int c[U]; double v[U], g[U];
#pragma unroll
  for (int i = 0; i < U; ++i) { long q = base + (long)i * TPB; bool ok = q < count; c[i] = ok ? __builtin_nontemporal_load(idx + q) : 0; v[i] = ok ? __builtin_nontemporal_load(val + q) : 0.0; }
#pragma unroll
  for (int i = 0; i < U; ++i) g[i] = x[c[i]];
  double s = 0;
#pragma unroll
  for (int i = 0; i < U; ++i) s += v[i] * g[i];
  out[(long)blockIdx.x * TPB + threadIdx.x] = s;
}
int main(int argc, char **argv) {
  long count = (argc > 1 ? atol(argv[1]) : 10) * 1000000L;
  std::vector<int> h(count);
  int *idx; double *val, *out, *x;
  CK(hipMalloc(&idx, count * 4)); CK(hipMalloc(&val, count * 8)); CK(hipMemset(val, 0, count * 8));
  long nb = (count + (long)TPB * U - 1) / ((long)TPB * U);
  CK(hipMalloc(&out, nb * TPB * 8));
  CK(hipMalloc(&x, 16000000L * 8)); CK(hipMemset(x, 0, 16000000L * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (long W : {65536L, 131072L, 262144L, 524288L, 1000000L, 2000000L, 4000000L}) {
    std::mt19937_64 rng(1);
    for (long i = 0; i < count; ++i) h[i] = (int)(rng() % W);
    CK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k, dim3(nb), dim3(TPB), 0, 0, idx, val, x, out, count); CK(hipEventRecord(b));
      CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
    }
    printf("window %8ld doubles (%5.1f MB): %ld gathers + 12 B/entry stream in %.4f ms = %.1f G gathers/s\n", W, W * 8 / 1048576.0, count, best, count / best / 1e6);
  }
  return 0;
}
