// Dev probe (not part of the product): the tiled sweep's gather pattern (as in
// sweep_probe.hip: 8-wave workgroups, one barrier per 512 KB column tile, entries +
// values streamed, no accumulators) with CHIP-WIDE PACING BY THE WALL CLOCK: every
// workgroup finishes tile t no earlier than start + (t+1)*slot on the 100 MHz
// s_memrealtime counter, where `start` is the next multiple of the sweep period after
// the workgroup began.  Nobody runs ahead of the schedule, so the workgroups of an XCD
// gather from the same tile without exchanging a single message -- does that buy the
// L2 residency the workgroup barrier alone does not?   usage: clock_pace_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int WPB = 8, WAVE = 64;

template <int E>
__global__ __launch_bounds__(WPB * WAVE) void sweep(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                    const double *__restrict__ x, double *__restrict__ out,
                                                    int ntiles, int shift, const unsigned long long *t0p, int slot) {
  extern __shared__ double lds[];
  __shared__ unsigned long long start_s;
  constexpr int C = E / WAVE;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t w = (size_t)blockIdx.x * WPB + wid;
  const unsigned *my = pk + w * (size_t)ntiles * E;
  const double *myv = tv + w * (size_t)ntiles * E;
  const unsigned cmask = (1u << shift) - 1u;
  double s = 0.0;
  unsigned p[2][C];
  double vv[2][C];
#pragma unroll
  for (int c = 0; c < C; ++c) { p[0][c] = __builtin_nontemporal_load(my + c * WAVE + lane); vv[0][c] = __builtin_nontemporal_load(myv + c * WAVE + lane); }
  unsigned long long start = 0;
  if (slot > 0) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = *t0p, period = (unsigned long long)ntiles * slot, now = wall_clock64();
      // join the current period when it has only just begun, else wait for the next one
      const unsigned long long k = (now - t0) / period, into = (now - t0) % period;
      start_s = t0 + (into < period / 16 ? k : k + 1) * period;
    }
    __syncthreads();
    start = start_s;
    while (wall_clock64() < start) __builtin_amdgcn_s_sleep(2);
  }
  for (int t0 = 0; t0 < ntiles; t0 += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int t = t0 + b;
      if (t < ntiles) {
        const double *xt = x + ((size_t)t << shift);
        double g[C];
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] = xt[p[b][c] & cmask];
        if (t + 1 < ntiles) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            p[b ^ 1][c] = __builtin_nontemporal_load(my + (size_t)(t + 1) * E + c * WAVE + lane);
            vv[b ^ 1][c] = __builtin_nontemporal_load(myv + (size_t)(t + 1) * E + c * WAVE + lane);
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) s += g[c] * vv[b][c];
        if (slot > 0) {
          const unsigned long long due = start + (unsigned long long)(t + 1) * slot;
          while (wall_clock64() < due) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
      }
    }
  }
  out[(size_t)blockIdx.x * WPB * WAVE + threadIdx.x] = s + lds[0] * 0.0;
}

__global__ void stamp(unsigned long long *t0p) { *t0p = wall_clock64(); }

template <int E>
void run(const unsigned *pk, const double *tv, const double *x, double *out, int nwaves, int ntiles, int shift, int slot,
         unsigned long long *t0p) {
  const size_t lds = 78 << 10;
  CK(hipFuncSetAttribute((const void *)sweep<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, 0, t0p);
    hipLaunchKernelGGL((sweep<E>), dim3(nwaves / WPB), dim3(WPB * WAVE), lds, 0, pk, tv, x, out, ntiles, shift, t0p, slot);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
  }
  const double g = (double)nwaves * ntiles * E;
  printf("E=%3d waves=%5d slot=%3d ticks (%.2f us): %.3f ms  %.1f G gathers/s   (schedule floor %.3f ms)\n", E, nwaves, slot,
         slot * 0.01, best, g / best / 1e6, slot ? 0.01e-3 * slot * ntiles * ((nwaves / WPB + 511) / 512) : 0.0);
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int shift = 16, ntiles = 153, nwaves = 8192;
  const size_t N = (size_t)ntiles << shift;
  const size_t cnt = (size_t)nwaves * ntiles * 128;
  std::vector<unsigned> h(cnt);
  std::mt19937 rng(1);
  for (size_t i = 0; i < cnt; ++i) h[i] = ((rng() & 0x3FFu) << 16) | (rng() & 0xFFFFu);
  unsigned *pk; CK(hipMalloc(&pk, cnt * 4)); CK(hipMemcpy(pk, h.data(), cnt * 4, hipMemcpyHostToDevice));
  double *tv; CK(hipMalloc(&tv, cnt * 8)); CK(hipMemset(tv, 0, cnt * 8));
  double *x; CK(hipMalloc(&x, N * 8)); CK(hipMemset(x, 0, N * 8));
  double *out; CK(hipMalloc(&out, (size_t)nwaves * WAVE * 8));
  unsigned long long *t0p; CK(hipMalloc(&t0p, 8));
  for (int nw : {4096, 8192}) {
    for (int slot : {0, 130, 150, 170, 190, 210, 240}) run<64>(pk, tv, x, out, nw, ntiles, shift, slot, t0p);
    for (int slot : {0, 300, 340, 380}) run<128>(pk, tv, x, out, nw, ntiles, shift, slot, t0p);
  }
  return 0;
}
