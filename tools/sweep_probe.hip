// Dev probe (not part of the product): the gather path of the tiled sweep in
// isolation.  8-wave workgroups walk the column tiles in lock step (barrier per
// tile); every wave loads E packed 16-bit column offsets per tile and gathers
// x[tile_base + off].  No values, no accumulators: what does the gather
// instruction form (64-bit vaddr / SGPR base + 32-bit offset / buffer load) and
// the pacing cost?   usage: sweep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int WPB = 8, WAVE = 64;

template <int FORM, int E, bool BAR, bool VALS>
__global__ __launch_bounds__(WPB * WAVE) void sweep(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                    const double *__restrict__ x, double *__restrict__ out,
                                                    int ntiles, int shift) {
  extern __shared__ double lds[];
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
  for (int c = 0; c < C; ++c) { p[0][c] = __builtin_nontemporal_load(my + c * WAVE + lane); if (VALS) vv[0][c] = __builtin_nontemporal_load(myv + c * WAVE + lane); }
  for (int t0 = 0; t0 < ntiles; t0 += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int t = t0 + b;
      if (t < ntiles) {
        const double *xt = x + ((size_t)t << shift);
        double g[C];
        if (FORM == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) g[c] = xt[p[b][c] & cmask];
        } else if (FORM == 1) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const unsigned off = (p[b][c] & cmask) * 8u;
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(g[c]) : "v"(off), "s"(xt) : "memory");
          }
        } else {
          __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)xt, 0, (int)(8u << shift), 0x00020000);
#pragma unroll
          for (int c = 0; c < C; ++c) {
            auto q = __builtin_amdgcn_raw_buffer_load_b64(r, (int)((p[b][c] & cmask) * 8u), 0, 0);
            g[c] = __builtin_bit_cast(double, q);
          }
        }
        if (t + 1 < ntiles) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            p[b ^ 1][c] = __builtin_nontemporal_load(my + (size_t)(t + 1) * E + c * WAVE + lane);
            if (VALS) vv[b ^ 1][c] = __builtin_nontemporal_load(myv + (size_t)(t + 1) * E + c * WAVE + lane);
          }
        }
        if (FORM == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < C; ++c) s += VALS ? g[c] * vv[b][c] : g[c];
        if (BAR) __syncthreads();
      }
    }
  }
  out[(size_t)blockIdx.x * WPB * WAVE + threadIdx.x] = s + lds[0] * 0.0;
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


// Loader-wave variant: a 9th wave streams every (workgroup, step) block of
// packed offsets + values into an LDS ring by LDS-DMA (global_load_lds, nt); the
// 8 consumers read their entries from LDS, so their vector-memory queue holds
// gathers only.  Layout: block (wg, t) = 8 waves x E entries, contiguous.
template <int E, int S, int XPF>
__global__ __launch_bounds__((WPB + 1) * WAVE) void sweep_ring(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                               const double *__restrict__ x, double *__restrict__ out,
                                                               int ntiles, int shift) {
  extern __shared__ double lds[];
  constexpr int BLK = WPB * E;                 // entries per (wg, step)
  constexpr int NP = BLK * 4 / 1024;           // DMA instructions for the packed part
  constexpr int NV = BLK * 8 / 1024;
  double *ring_v = lds;                                            // [S][BLK] doubles
  unsigned *ring_p = reinterpret_cast<unsigned *>(lds + S * BLK);  // [S][BLK] u32
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned *gp = pk + (size_t)blockIdx.x * ntiles * BLK;
  const double *gv = tv + (size_t)blockIdx.x * ntiles * BLK;
  if (wid == WPB) {
    auto issue = [&](int j) {
      const int jj = min(j, ntiles - 1);
      const int slot = j % S;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        __builtin_amdgcn_global_load_lds(gp + (size_t)jj * BLK + i * 256 + lane * 4,
                                         (__attribute__((address_space(3))) void *)(ring_p + slot * BLK + i * 256), 16, 0, 2);
#pragma unroll
      for (int i = 0; i < NV; ++i)
        __builtin_amdgcn_global_load_lds(gv + (size_t)jj * BLK + i * 128 + lane * 2,
                                         (__attribute__((address_space(3))) void *)(ring_v + slot * BLK + i * 128), 16, 0, 2);
    };
    int sink = 0;   // destination of the never-waited touches: must stay allocated while they are in flight
    for (int j = 0; j < S - 1; ++j) issue(j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (NP + NV)) : "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ntiles; ++t) {
      issue(t + S - 1);
      if (XPF > 0 && t + XPF < ntiles) {   // touch a 1/64 slice of a later x tile (64 co-resident WGs per XCD)
        const char *xl = reinterpret_cast<const char *>(x + ((size_t)(t + XPF) << shift)) + (size_t)(((blockIdx.x >> 3) & 63) * 64 + lane) * 128;
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(xl) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (NP + NV) + 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (NP + NV)) : "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
    if (sink == 0x7fffffff) out[0] = 0.0;
    return;
  }
  constexpr int C = E / WAVE;
  const unsigned cmask = (1u << shift) - 1u;
  double s = 0.0;
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < ntiles; ++t) {
    const int slot = t % S;
    const double *xt = x + ((size_t)t << shift);
    unsigned p[C]; double vv[C], g[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = ring_p[slot * BLK + wid * E + c * WAVE + lane]; vv[c] = ring_v[slot * BLK + wid * E + c * WAVE + lane]; }
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = xt[p[c] & cmask];
#pragma unroll
    for (int c = 0; c < C; ++c) s += g[c] * vv[c];
    __syncthreads();
  }
  out[(size_t)blockIdx.x * WPB * WAVE + threadIdx.x] = s;
}

template <int E, int S, int XPF>
void run_ring(const char *name, const unsigned *pk, const double *tv, const double *x, double *out, int nwaves, int ntiles, int shift, size_t lds) {
  CK(hipFuncSetAttribute((const void *)sweep_ring<E, S, XPF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float ms = time_it([&] { hipLaunchKernelGGL((sweep_ring<E, S, XPF>), dim3(nwaves / WPB), dim3((WPB + 1) * WAVE), lds, 0, pk, tv, x, out, ntiles, shift); });
  double cnt = (double)nwaves * ntiles * E;
  printf("%-34s E=%3d lds=%3zuK: %.3f ms  %.1f G gathers/s\n", name, E, lds >> 10, ms, cnt / ms / 1e6);
}

template <int FORM, int E, bool BAR, bool VALS>
void run(const char *name, const unsigned *pk, const double *tv, const double *x, double *out, int nwaves, int ntiles, int shift, size_t lds) {
  CK(hipFuncSetAttribute((const void *)sweep<FORM, E, BAR, VALS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float ms = time_it([&] { hipLaunchKernelGGL((sweep<FORM, E, BAR, VALS>), dim3(nwaves / WPB), dim3(WPB * WAVE), lds, 0, pk, tv, x, out, ntiles, shift); });
  double cnt = (double)nwaves * ntiles * E;
  printf("%-34s E=%3d lds=%3zuK: %.3f ms  %.1f G gathers/s\n", name, E, lds >> 10, ms, cnt / ms / 1e6);
}


// Pacing with SLACK instead of a full barrier: every wave publishes the number of
// tiles it has finished in LDS and, before starting tile t, only waits until all
// waves of the workgroup have finished tile t - 1 - SLACK.  SLACK = 0 behaves like
// the barrier; SLACK = 1 lets a wave run one tile ahead of the slowest one.
template <int E, int SLACK>
__global__ __launch_bounds__(WPB * WAVE) void sweep_slack(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                          const double *__restrict__ x, double *__restrict__ out,
                                                          int ntiles, int shift) {
  extern __shared__ double lds[];
  volatile int *prog = reinterpret_cast<volatile int *>(lds);   // [WPB]
  constexpr int C = E / WAVE;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x < WPB) prog[threadIdx.x] = 0;
  __syncthreads();
  const size_t w = (size_t)blockIdx.x * WPB + wid;
  const unsigned *my = pk + w * (size_t)ntiles * E;
  const double *myv = tv + w * (size_t)ntiles * E;
  const unsigned cmask = (1u << shift) - 1u;
  double s = 0.0;
  unsigned p[2][C];
  double vv[2][C];
#pragma unroll
  for (int c = 0; c < C; ++c) { p[0][c] = __builtin_nontemporal_load(my + c * WAVE + lane); vv[0][c] = __builtin_nontemporal_load(myv + c * WAVE + lane); }
  for (int t0 = 0; t0 < ntiles; t0 += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int t = t0 + b;
      if (t < ntiles) {
        // wait until every wave has finished tile t - 1 - SLACK
        const int need = t - SLACK;
        if (need > 0) {
          for (;;) {
            const int mine = (lane < WPB) ? prog[lane] : 0x7fffffff;
            int mn = mine;
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) mn = min(mn, __shfl_down(mn, off, WAVE));
            mn = __builtin_amdgcn_readfirstlane(mn);
            if (mn >= need) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
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
        if (lane == 0) prog[wid] = t + 1;
      }
    }
  }
  out[(size_t)blockIdx.x * WPB * WAVE + threadIdx.x] = s;
}

template <int E, int SLACK>
void run_slack(const unsigned *pk, const double *tv, const double *x, double *out, int nwaves, int ntiles, int shift, size_t lds) {
  CK(hipFuncSetAttribute((const void *)sweep_slack<E, SLACK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float ms = time_it([&] { hipLaunchKernelGGL((sweep_slack<E, SLACK>), dim3(nwaves / WPB), dim3(WPB * WAVE), lds, 0, pk, tv, x, out, ntiles, shift); });
  double cnt = (double)nwaves * ntiles * E;
  printf("LDS-progress pacing, slack %d tile(s) +vals   E=%3d: %.3f ms  %.1f G gathers/s\n", SLACK, E, ms, cnt / ms / 1e6);
}

// XCD-wide pacing: a persistent grid of 512 workgroups (2 per CU, all resident),
// workgroup b on XCD b % 8; after every tile one thread bumps its XCD's counter
// and the workgroup waits (bounded spin) until all 64 workgroups of the XCD have
// finished that tile.  EVERY is how many tiles pass between two XCD syncs.
template <int E, int EVERY>
__global__ __launch_bounds__(WPB * WAVE) void sweep_xcd(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                        const double *__restrict__ x, double *__restrict__ out,
                                                        int ntiles, int shift, unsigned *__restrict__ counters,
                                                        unsigned epoch_base, int wgs_per_xcd) {
  extern __shared__ double lds[];
  constexpr int C = E / WAVE;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t w = (size_t)blockIdx.x * WPB + wid;
  const unsigned *my = pk + w * (size_t)ntiles * E;
  const double *myv = tv + w * (size_t)ntiles * E;
  const unsigned cmask = (1u << shift) - 1u;
  unsigned *ctr = counters + (blockIdx.x & 7) * 32;   // one 128-byte line per XCD
  double s = 0.0;
  unsigned p[2][C];
  double vv[2][C];
#pragma unroll
  for (int c = 0; c < C; ++c) { p[0][c] = __builtin_nontemporal_load(my + c * WAVE + lane); vv[0][c] = __builtin_nontemporal_load(myv + c * WAVE + lane); }
  int syncs = 0;
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
        __syncthreads();
        if ((t + 1) % EVERY == 0) {
          ++syncs;
          if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch_base + (unsigned)syncs * (unsigned)wgs_per_xcd;
            for (int spin = 0; spin < 200000; ++spin) {   // bounded: never hang the GPU
              if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) break;
              __builtin_amdgcn_s_sleep(1);
            }
          }
          __syncthreads();
        }
      }
    }
  }
  out[(size_t)blockIdx.x * WPB * WAVE + threadIdx.x] = s + lds[0] * 0.0;
}

template <int E, int EVERY>
void run_xcd(const unsigned *pk, const double *tv, const double *x, double *out, int ntiles, int shift, size_t lds) {
  const int grid = 512, nwaves = grid * WPB;
  unsigned *ctr; CK(hipMalloc(&ctr, 8 * 128)); CK(hipMemset(ctr, 0, 8 * 128));
  CK(hipFuncSetAttribute((const void *)sweep_xcd<E, EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  unsigned launches = 0;
  const unsigned per_launch = (unsigned)(ntiles / EVERY) * 64u;
  float ms = time_it([&] {
    hipLaunchKernelGGL((sweep_xcd<E, EVERY>), dim3(grid), dim3(WPB * WAVE), lds, 0, pk, tv, x, out, ntiles, shift, ctr,
                       launches * per_launch, 64);
    ++launches;
  });
  double cnt = (double)nwaves * ntiles * E;
  printf("XCD-wide pacing every %2d tile(s), 512 persistent workgroups +vals   E=%3d: %.3f ms  %.1f G gathers/s\n", EVERY, E, ms, cnt / ms / 1e6);
  CK(hipFree(ctr));
}

// Stream-load cache policy experiment: packed offsets and values through raw
// buffer loads with cache-policy bits AUX (gfx940+: bit0 sc0, bit1 nt, bit4 sc1);
// gathers are plain global loads.  Does any policy reduce the interference of
// the entry stream with the L2-resident gathers?
template <int E, int AUX>
__global__ __launch_bounds__(WPB * WAVE) void sweep_aux(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                        const double *__restrict__ x, double *__restrict__ out,
                                                        int ntiles, int shift, unsigned pk_bytes, unsigned tv_bytes) {
  extern __shared__ double lds[];
  constexpr int C = E / WAVE;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned w = blockIdx.x * WPB + wid;
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)pk, 0, (int)pk_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)tv, 0, (int)tv_bytes, 0x00020000);
  const unsigned base = w * (unsigned)ntiles * E;
  const unsigned cmask = (1u << shift) - 1u;
  double s = 0.0;
  unsigned p[2][C];
  double vv[2][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    p[0][c] = __builtin_amdgcn_raw_buffer_load_b32(rp, (int)((base + c * WAVE + lane) * 4u), 0, AUX);
    vv[0][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rv, (int)((base + c * WAVE + lane) * 8u), 0, AUX));
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
            const unsigned k = base + (unsigned)(t + 1) * E + c * WAVE + lane;
            p[b ^ 1][c] = __builtin_amdgcn_raw_buffer_load_b32(rp, (int)(k * 4u), 0, AUX);
            vv[b ^ 1][c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rv, (int)(k * 8u), 0, AUX));
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) s += g[c] * vv[b][c];
        __syncthreads();
      }
    }
  }
  out[(size_t)blockIdx.x * WPB * WAVE + threadIdx.x] = s + lds[0] * 0.0;
}

template <int E, int AUX>
void run_aux(const unsigned *pk, const double *tv, const double *x, double *out, int nwaves, int ntiles, int shift, size_t lds) {
  CK(hipFuncSetAttribute((const void *)sweep_aux<E, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned pkb = (unsigned)((size_t)nwaves * ntiles * E * 4), tvb = (unsigned)((size_t)nwaves * ntiles * E * 8);
  float ms = time_it([&] { hipLaunchKernelGGL((sweep_aux<E, AUX>), dim3(nwaves / WPB), dim3(WPB * WAVE), lds, 0, pk, tv, x, out, ntiles, shift, pkb, tvb); });
  double cnt = (double)nwaves * ntiles * E;
  printf("stream loads aux=%2d (sc0=%d nt=%d sc1=%d) +vals   E=%3d: %.3f ms  %.1f G gathers/s\n", AUX, AUX & 1, (AUX >> 1) & 1, (AUX >> 4) & 1, E, ms, cnt / ms / 1e6);
}

// Ceiling: every workgroup gathers from ONE window that is resident in every
// XCD's L2 (512 KB; far larger than the 32 KB L1), no barriers, no tiles, full
// occupancy -- the chip's rate for L2-hit 8-byte gathers with a 4-byte index
// stream and nothing else going on.
template <int U, bool VALS>
__global__ __launch_bounds__(256) void l2_gather_peak(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                      const double *__restrict__ x,
                                                      double *__restrict__ out, size_t count, unsigned mask) {
  size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  unsigned c[U];
  double v[U];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < U; ++i) {
    size_t k = base + (size_t)i * 256;
    c[i] = k < count ? __builtin_nontemporal_load(pk + k) : 0u;
    v[i] = (VALS && k < count) ? __builtin_nontemporal_load(tv + k) : 1.0;
  }
#pragma unroll
  for (int i = 0; i < U; ++i) s += v[i] * x[c[i] & mask];
  out[((size_t)blockIdx.x * 256 + threadIdx.x) & 0x7FFFF] = s;
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int shift = 16, ntiles = 153, nwaves = 8192;
  const size_t N = (size_t)ntiles << shift;
  const size_t maxE = 128;
  const size_t cnt = (size_t)nwaves * ntiles * maxE;
  std::vector<unsigned> h(cnt);
  std::mt19937 rng(1);
  for (size_t i = 0; i < cnt; ++i) h[i] = ((rng() & 0x3FFu) << 16) | (rng() & 0xFFFFu);
  unsigned *pk; CK(hipMalloc(&pk, cnt * 4)); CK(hipMemcpy(pk, h.data(), cnt * 4, hipMemcpyHostToDevice));
  double *tv; CK(hipMalloc(&tv, cnt * 8)); CK(hipMemset(tv, 0, cnt * 8));
  double *x; CK(hipMalloc(&x, N * 8)); CK(hipMemset(x, 0, N * 8));
  double *out; CK(hipMalloc(&out, (size_t)nwaves * WAVE * 8));
  const size_t L2WG = 78 << 10, L4WG = 38 << 10;
  run<0, 64, true, false>("vaddr64 barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<1, 64, true, false>("saddr+voff32 barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<2, 64, true, false>("buffer_load barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<0, 128, true, false>("vaddr64 barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<1, 128, true, false>("saddr+voff32 barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<2, 128, true, false>("buffer_load barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<0, 128, false, false>("vaddr64 NO barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<2, 128, false, false>("buffer_load NO barrier", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<0, 128, true, false>("vaddr64 barrier 4WG/CU", pk, tv, x, out, nwaves, ntiles, shift, L4WG);
  run<2, 128, true, false>("buffer_load barrier 4WG/CU", pk, tv, x, out, nwaves, ntiles, shift, L4WG);
  run<0, 128, true, true>("vaddr64 barrier +vals", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<2, 128, true, true>("buffer_load barrier +vals", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<2, 64, true, true>("buffer_load barrier +vals", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run<2, 128, true, true>("buffer_load barrier +vals 4WG/CU", pk, tv, x, out, nwaves, ntiles, shift, L4WG);
  for (unsigned bits : {12u, 14u, 16u, 18u}) {
    const size_t count = (size_t)nwaves * ntiles * 128;
    const size_t blocks = (count + 256 * 8 - 1) / (256 * 8);
    float ms = time_it([&] { hipLaunchKernelGGL((l2_gather_peak<8, false>), dim3((unsigned)blocks), dim3(256), 0, 0, pk, tv, x, out, count, (1u << bits) - 1u); });
    float mv = time_it([&] { hipLaunchKernelGGL((l2_gather_peak<8, true>), dim3((unsigned)blocks), dim3(256), 0, 0, pk, tv, x, out, count, (1u << bits) - 1u); });
    printf("pure gather peak, one %4u KB window shared by all workgroups: %.3f ms  %.1f G gathers/s | with the 8-byte value stream %.3f ms  %.1f G/s\n",
           (8u << bits) >> 10, ms, count / ms / 1e6, mv, count / mv / 1e6);
  }
  run<0, 64, true, true>("vaddr64 barrier +vals, 512 WGs only", pk, tv, x, out, 4096, ntiles, shift, L2WG);
  run<0, 128, true, true>("vaddr64 barrier +vals, 512 WGs only", pk, tv, x, out, 4096, ntiles, shift, L2WG);
  run_xcd<64, 1>(pk, tv, x, out, ntiles, shift, L2WG);
  run_xcd<64, 2>(pk, tv, x, out, ntiles, shift, L2WG);
  run_xcd<64, 4>(pk, tv, x, out, ntiles, shift, L2WG);
  run_xcd<128, 1>(pk, tv, x, out, ntiles, shift, L2WG);
  run_xcd<128, 2>(pk, tv, x, out, ntiles, shift, L2WG);
  run_xcd<128, 4>(pk, tv, x, out, ntiles, shift, L2WG);
  run_slack<64, 0>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_slack<64, 1>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_slack<64, 2>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_slack<64, 4>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_slack<128, 0>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_slack<128, 1>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_slack<128, 2>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 0>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 2>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 1>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 3>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 16>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 17>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 18>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<64, 19>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<128, 0>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<128, 2>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<128, 18>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_aux<128, 19>(pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_ring<64, 4, 0>("ring S=4 +vals", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_ring<64, 3, 0>("ring S=3 +vals", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_ring<64, 6, 0>("ring S=6 +vals", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_ring<128, 4, 0>("ring S=4 +vals", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_ring<64, 4, 2>("ring S=4 +vals xprefetch+2", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_ring<64, 4, 4>("ring S=4 +vals xprefetch+4", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  run_ring<128, 4, 2>("ring S=4 +vals xprefetch+2", pk, tv, x, out, nwaves, ntiles, shift, L2WG);
  return 0;
}
