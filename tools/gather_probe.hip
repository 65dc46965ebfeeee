// Dev probe (not part of the product): what bounds a random fp64 gather on
// MI355X?  Usage: gather_probe [count_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int TPB = 256;
constexpr int U = 8;

template <int MODE>  // 0 plain, 1 nontemporal x, 2 idx plain too
__global__ __launch_bounds__(TPB) void gather_kernel(const int *__restrict__ idx, const double *__restrict__ x,
                                                     double *__restrict__ out, long count) {
  long base = ((long)blockIdx.x * TPB * U) + threadIdx.x;
  int c[U]; double v[U];
#pragma unroll
  for (int i = 0; i < U; ++i) { long k = base + (long)i * TPB; c[i] = k < count ? __builtin_nontemporal_load(idx + k) : 0; }
#pragma unroll
  for (int i = 0; i < U; ++i) v[i] = (MODE == 1) ? __builtin_nontemporal_load(x + c[i]) : x[c[i]];
  double s = 0;
#pragma unroll
  for (int i = 0; i < U; ++i) s += v[i];
  out[(long)blockIdx.x * TPB + threadIdx.x] = s;
}

// float gather (4B elements)
__global__ __launch_bounds__(TPB) void gather_f32_kernel(const int *__restrict__ idx, const float *__restrict__ x,
                                                         float *__restrict__ out, long count) {
  long base = ((long)blockIdx.x * TPB * U) + threadIdx.x;
  int c[U]; float v[U];
#pragma unroll
  for (int i = 0; i < U; ++i) { long k = base + (long)i * TPB; c[i] = k < count ? __builtin_nontemporal_load(idx + k) : 0; }
#pragma unroll
  for (int i = 0; i < U; ++i) v[i] = x[c[i]];
  float s = 0;
#pragma unroll
  for (int i = 0; i < U; ++i) s += v[i];
  out[(long)blockIdx.x * TPB + threadIdx.x] = s;
}

// stream val+col, "gather" from a tiny cached window: floor of the SpMV stream phase
__global__ __launch_bounds__(TPB) void stream_kernel(const int *__restrict__ idx, const double *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ out, long count) {
  long base = ((long)blockIdx.x * TPB * U) + threadIdx.x;
  int c[U]; double v[U];
#pragma unroll
  for (int i = 0; i < U; ++i) { long k = base + (long)i * TPB; bool ok = k < count; c[i] = ok ? __builtin_nontemporal_load(idx + k) : 0; v[i] = ok ? __builtin_nontemporal_load(val + k) : 0.0; }
  double s = 0;
#pragma unroll
  for (int i = 0; i < U; ++i) s += v[i] * x[c[i] & 1023];
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
  double *val; CK(hipMalloc(&val, count * 8)); CK(hipMemset(val, 0, count * 8));
  int *idx; CK(hipMalloc(&idx, count * 4));
  long nblocks = (count + (long)TPB * U - 1) / ((long)TPB * U);
  double *out; CK(hipMalloc(&out, nblocks * TPB * 8));
  for (long N : {1000000L, 4000000L, 10000000L, 40000000L}) {
    std::mt19937_64 rng(1);
    for (long i = 0; i < count; ++i) h[i] = (int)(rng() % N);
    CK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
    double *x; CK(hipMalloc(&x, N * 8)); CK(hipMemset(x, 0, N * 8));
    double *xu; CK(hipExtMallocWithFlags((void **)&xu, N * 8, hipDeviceMallocUncached)); CK(hipMemset(xu, 0, N * 8));
    float *xf; CK(hipMalloc(&xf, N * 4)); CK(hipMemset(xf, 0, N * 4));
    float t0 = time_it([&] { hipLaunchKernelGGL(gather_kernel<0>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    float t1 = time_it([&] { hipLaunchKernelGGL(gather_kernel<1>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    float t2 = time_it([&] { hipLaunchKernelGGL(gather_kernel<0>, dim3(nblocks), dim3(TPB), 0, 0, idx, xu, out, count); });
    float t3 = time_it([&] { hipLaunchKernelGGL(gather_f32_kernel, dim3(nblocks), dim3(TPB), 0, 0, idx, xf, (float *)out, count); });
    printf("N=%9ld count=%ld  gather f64 plain %.3f ms (%.1f G/s) | nt %.3f ms (%.1f G/s) | uncached-x %.3f ms (%.1f G/s) | f32 %.3f ms (%.1f G/s)\n",
           N, count, t0, count / t0 / 1e6, t1, count / t1 / 1e6, t2, count / t2 / 1e6, t3, count / t3 / 1e6);
    CK(hipFree(x)); CK(hipFree(xu)); CK(hipFree(xf));
  }
  // sorted-within-chunk gather: indices sorted in chunks of 2048 (what a block sees) -- any benefit?
  {
    long N = 10000000L; std::mt19937_64 rng(1);
    for (long i = 0; i < count; ++i) h[i] = (int)(rng() % N);
    double *x; CK(hipMalloc(&x, N * 8)); CK(hipMemset(x, 0, N * 8));
    CK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
    float ts = time_it([&] { hipLaunchKernelGGL(stream_kernel, dim3(nblocks), dim3(TPB), 0, 0, idx, val, x, out, count); });
    printf("stream-only (val+col, cached x): %.3f ms -> %.1f GB/s\n", ts, count * 12.0 / ts / 1e6);
    // sequential idx: the coalesced ceiling
    for (long i = 0; i < count; ++i) h[i] = (int)(i % N);
    CK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
    float tq = time_it([&] { hipLaunchKernelGGL(gather_kernel<0>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
    printf("sequential idx N=10M: %.3f ms (%.1f G/s)\n", tq, count / tq / 1e6);
    // locality: idx = row-local window (|col - row*N/count| < W)
    for (long W : {4096L, 65536L, 1000000L}) {
      for (long i = 0; i < count; ++i) { long c = (long)((double)i / count * N) + (long)(rng() % (2 * W)) - W; if (c < 0) c = 0; if (c >= N) c = N - 1; h[i] = (int)c; }
      CK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
      float tw = time_it([&] { hipLaunchKernelGGL(gather_kernel<0>, dim3(nblocks), dim3(TPB), 0, 0, idx, x, out, count); });
      printf("banded idx N=10M W=%ld: %.3f ms (%.1f G/s)\n", W, tw, count / tw / 1e6);
    }
  }
  return 0;
}
