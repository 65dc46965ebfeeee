// grid_barrier_probe.hip -- dev probe (not part of the library): what does a grid-wide barrier
// cost on MI355X inside one persistent kernel, and what does launching it cost?
//   flat:  every workgroup does one agent-scope atomic on ONE counter and spins on it
//   xcd:   two levels -- workgroup-scope (XCD-local L2) arrival atomics per XCD, the last arriver
//          of an XCD does the agent-scope atomic and spins on the global counter, then releases
//          its XCD through an XCD-local flag (XCC_ID read from the hardware register)
// also: plain launch vs hipLaunchCooperativeKernel round trip with a host-polled result word.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/grid_barrier_probe tools/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), bits [3:0]
  return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xF;
}

struct Sync {
  unsigned long long global;          // agent-scope counter
  unsigned long long pad0[15];
  unsigned long long xcd_arrive[8][16];   // one 128-byte line per XCD
  unsigned long long xcd_release[8][16];
  unsigned long long xcd_count[8][16];    // workgroups resident per XCD (set by a registration pass)
  unsigned long long error;
};

__device__ __forceinline__ void barrier_flat(Sync *s, unsigned long long target) {
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __threadfence();
    __hip_atomic_fetch_add(&s->global, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(&s->global, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 2000000L) { __hip_atomic_store(&s->error, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __threadfence();
  }
  __syncthreads();
}

// epoch e = 1, 2, ...: global counter reaches 8-ish * e (one per XCD that has workgroups)
__device__ __forceinline__ void barrier_xcd(Sync *s, unsigned long long epoch, unsigned nxcd_active, unsigned long long zero = 0) {
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    const unsigned x = xcc_id() & 7;
    __threadfence();                                   // release this workgroup's writes (agent scope)
    const unsigned long long cnt = s->xcd_count[x][0];
    const unsigned long long prev = __hip_atomic_fetch_add(&s->xcd_arrive[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    long spins = 0;
    if (prev + 1 == cnt * epoch) {
      __hip_atomic_fetch_add(&s->global, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&s->global, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)nxcd_active * epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000L) { __hip_atomic_store(&s->error, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      __hip_atomic_store(&s->xcd_release[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      // an RMW atomic is executed in the XCD's L2 (a workgroup-scope LOAD may be served by the CU's L1 and never see the flag)
      while (__hip_atomic_fetch_add(&s->xcd_release[x][0], zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000L) { __hip_atomic_store(&s->error, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __threadfence();                                   // acquire
  }
  __syncthreads();
}

// two levels with SCOPED FENCES: the expensive part of an agent-scope release / acquire on this chip is the L2
// write-back / invalidate (buffer_wbl2 sc1 / buffer_inv sc1), and every workgroup doing its own serialises them in
// the XCD's L2 (~37 ns per workgroup in all the variants above).  Here only the XCD's last arriver writes the L2
// back (everybody else's stores are already IN that L2: s_waitcnt vmcnt(0) before arriving) and invalidates it
// after the global phase; the other workgroups only drop their CU's L1 (buffer_inv sc0).
__device__ __forceinline__ int barrier_xcd2(Sync *s, unsigned long long epoch, unsigned nxcd_active) {
  __shared__ int was_leader;
  __syncthreads();      // includes every wave's workgroup-scope release: its stores have reached the L2
  if (threadIdx.x == 0 && __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    const unsigned x = xcc_id() & 7;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long cnt = s->xcd_count[x][0];
    const unsigned long long prev = __hip_atomic_fetch_add(&s->xcd_arrive[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    was_leader = prev + 1 == cnt * epoch;
    if (prev + 1 == cnt * epoch) {
      asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&s->global, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&s->global, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)nxcd_active * epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000L) { __hip_atomic_store(&s->error, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      asm volatile("buffer_inv sc1" ::: "memory");
      __hip_atomic_store(&s->xcd_release[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&s->xcd_release[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000L) { __hip_atomic_store(&s->error, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    // (buffer_inv sc0 -- workgroup scope -- is a no-op here: 26 % stale reads with one workgroup per CU)
    if (!was_leader) asm volatile("buffer_inv sc1" ::: "memory");
    asm volatile("s_dcache_inv" ::: "memory");
  }
  __syncthreads();
  return was_leader;
}

// timing + correctness of barrier_xcd2: every workgroup publishes a value with plain stores and, after the
// barrier, reads the value of a workgroup on ANOTHER XCD with plain loads -- only the barrier's fences make them
// visible.  Double-buffered, so that iteration it + 1's stores cannot race with iteration it's reads.
__global__ __launch_bounds__(256) void xcd2_loop_kernel(Sync *s, int iters, unsigned long long base, unsigned nxcd_active,
                                                        double *data, unsigned long long *slots, unsigned long long *bad) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  unsigned long long mism = 0;
  for (int it = 0; it < iters; ++it) {
    data[gid] = data[gid] + 1.0;
    unsigned long long *buf = slots + (size_t)(it & 1) * gridDim.x * 16;
    if (threadIdx.x < 16) buf[(size_t)blockIdx.x * 16 + threadIdx.x] = (base + it + 1) * 1000003ull + blockIdx.x;
    if (it == 0 && threadIdx.x == 0) atomicAdd(bad + 8 + (xcc_id() & 7), 1ull);     // census of this launch
    const int lead = barrier_xcd2(s, base + it + 1, nxcd_active);
    const unsigned peer = (blockIdx.x + 1 + (it % 7)) % gridDim.x;
    if (threadIdx.x < 16 && buf[(size_t)peer * 16 + threadIdx.x] != (base + it + 1) * 1000003ull + peer) { ++mism; if (lead) atomicAdd(bad + 1, 1ull); }
  }
  if (mism) atomicAdd(bad, mism);
}

__global__ __launch_bounds__(256) void register_kernel(Sync *s, unsigned *xcd_of_wg) {
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id() & 7;
    xcd_of_wg[blockIdx.x] = x;
    __hip_atomic_fetch_add(&s->xcd_count[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// two levels, every atomic at agent scope: workgroup w arrives at leaf counter (w % nleaf) (one 128-byte
// line each); the last arriver of a leaf arrives at the root; the last arriver of the root stores the epoch
// into a release word on its own line; everybody else polls that word.
// POLL 0: agent-scope loads, 1: RMW atomics (run-time zero: the compiler turns fetch_add(p, 0) into a load)
struct Tree {
  unsigned long long leaf[64][16];
  unsigned long long root[16];
  unsigned long long release[16];
};
template <int POLL>
__device__ __forceinline__ void barrier_tree(Sync *s, Tree *t, unsigned long long epoch, unsigned nleaf, unsigned long long zero) {
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    const unsigned w = blockIdx.x, l = w % nleaf;
    const unsigned long long members = (gridDim.x - l + nleaf - 1) / nleaf;     // workgroups mapped to this leaf
    __threadfence();
    bool released = false;
    if (__hip_atomic_fetch_add(&t->leaf[l][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == members * epoch) {
      const unsigned long long nl = nleaf < gridDim.x ? nleaf : gridDim.x;
      if (__hip_atomic_fetch_add(&t->root[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == nl * epoch) {
        __hip_atomic_store(&t->release[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        released = true;
      }
    }
    long spins = 0;
    while (!released) {
      const unsigned long long v = POLL ? __hip_atomic_fetch_add(&t->release[0], zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : __hip_atomic_load(&t->release[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= epoch) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 2000000L) { __hip_atomic_store(&s->error, 4ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __threadfence();
  }
  __syncthreads();
}

template <int KIND>
__global__ __launch_bounds__(256) void tree_loop_kernel(Sync *s, Tree *t, int iters, unsigned long long base, unsigned nleaf,
                                                        unsigned long long zero, double *data) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    data[gid] = data[gid] + 1.0;
    barrier_tree<KIND>(s, t, base + it + 1, nleaf, zero);
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void barrier_loop_kernel(Sync *s, int iters, unsigned long long base, unsigned nxcd_active,
                                                           double *data, volatile double *res_host) {
  // a little real traffic between the barriers so that the fences have something to flush
  const int gid = blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    data[gid] = data[gid] + 1.0;
    if (KIND == 0) barrier_flat(s, (base + it + 1) * (unsigned long long)gridDim.x);
    else barrier_xcd(s, base + it + 1, nxcd_active, (unsigned long long)(res_host != nullptr));
  }
  if (gid == 0 && res_host) { __threadfence_system(); res_host[0] = (double)(base + iters); }
}

__global__ __launch_bounds__(256) void tiny_kernel(double *data, volatile double *res_host, double v) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  data[gid] += 1.0;
  if (gid == 0) { __threadfence_system(); res_host[0] = v; }
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Sync *s;
  CK(hipMalloc((void **)&s, sizeof(Sync)));
  double *data;
  CK(hipMalloc((void **)&data, sizeof(double) * 4096 * 256));
  CK(hipMemset(data, 0, sizeof(double) * 4096 * 256));
  volatile double *res;
  CK(hipHostMalloc((void **)&res, 64, hipHostMallocCoherent | hipHostMallocMapped));
  unsigned *xcd_of;
  CK(hipMalloc((void **)&xcd_of, 4096 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200;
  for (int wgs : {256, 512, 1024, 2048}) {
    // where do the workgroups of a launch of this size land?
    CK(hipMemset(s, 0, sizeof(Sync)));
    hipLaunchKernelGGL(register_kernel, dim3(wgs), dim3(256), 0, st, s, xcd_of);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> hx(wgs);
    CK(hipMemcpy(hx.data(), xcd_of, wgs * 4, hipMemcpyDeviceToHost));
    int rr = 0;
    for (int b = 0; b < wgs; ++b) rr += (hx[b] == (unsigned)(b % 8));
    Sync hs;
    CK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost));
    unsigned active = 0;
    printf("wgs %4d: xcc_id == blockIdx %% 8 for %d of %d; per XCD:", wgs, rr, wgs);
    for (int x = 0; x < 8; ++x) { printf(" %llu", hs.xcd_count[x][0]); active += hs.xcd_count[x][0] > 0; }
    printf("\n");
    for (int kind = 0; kind < 2; ++kind) {
      float best = 1e30f;
      unsigned long long base = 0;
      // the counters keep counting across launches (monotonic), as the library would do
      CK(hipMemset(&s->global, 0, sizeof(unsigned long long)));
      CK(hipMemset(&s->error, 0, sizeof(unsigned long long)));
      CK(hipMemset(s->xcd_arrive, 0, sizeof(s->xcd_arrive)));
      CK(hipMemset(s->xcd_release, 0, sizeof(s->xcd_release)));
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        if (kind == 0) hipLaunchKernelGGL(barrier_loop_kernel<0>, dim3(wgs), dim3(256), 0, st, s, iters, base, active, data, (volatile double *)nullptr);
        else hipLaunchKernelGGL(barrier_loop_kernel<1>, dim3(wgs), dim3(256), 0, st, s, iters, base, active, data, (volatile double *)nullptr);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        base += iters;
      }
      CK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost));
      printf("  %-4s barrier: %.2f us each (%d workgroups of 256, %d barriers per launch)%s\n", kind ? "xcd" : "flat",
             1e3 * best / iters, wgs, iters, hs.error ? "  ** SPIN LIMIT HIT **" : "");
      if (hs.error) printf("     error code %llu\n", hs.error);
    }
  }
  {
    unsigned long long *slots, *bad;
    CK(hipMalloc((void **)&slots, sizeof(unsigned long long) * 2 * 2048 * 16));
    CK(hipMalloc((void **)&bad, 8 * 16));
    for (int wgs : {256, 264, 384, 512, 1024, 2048}) {
      CK(hipMemset(s, 0, sizeof(Sync)));
      CK(hipMemset(bad, 0, 8 * 16));
      CK(hipMemset(slots, 0, sizeof(unsigned long long) * 2 * 2048 * 16));
      hipLaunchKernelGGL(register_kernel, dim3(wgs), dim3(256), 0, st, s, xcd_of);
      CK(hipStreamSynchronize(st));
      float best = 1e30f;
      unsigned long long base = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(xcd2_loop_kernel, dim3(wgs), dim3(256), 0, st, s, iters, base, 8u, data, slots, bad);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        base += iters;
      }
      Sync hs;
      unsigned long long hb[16];
      CK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb, bad, 8 * 16, hipMemcpyDeviceToHost));
      printf("  xcd2 barrier (scoped fences), %4d workgroups: %.2f us each, %llu stale reads (%llu by XCD leaders) in %d x 5 barriers%s\n", wgs,
             1e3 * best / iters, hb[0], hb[1], iters, hs.error ? "  ** SPIN LIMIT HIT **" : "");
      printf("     registered per XCD:");
      for (int x = 0; x < 8; ++x) printf(" %llu", hs.xcd_count[x][0]);
      printf("; census over the 5 timed launches:");
      for (int x = 0; x < 8; ++x) printf(" %llu", hb[8 + x]);
      printf("\n");
    }
  }
  {
    Tree *t;
    CK(hipMalloc((void **)&t, sizeof(Tree)));
    for (int wgs : {256, 512, 1024, 2048})
      for (int poll = 0; poll < 2; ++poll)
        for (unsigned nleaf : {8u, 32u, 64u}) {
          CK(hipMemset(t, 0, sizeof(Tree)));
          CK(hipMemset(&s->error, 0, sizeof(unsigned long long)));
          float best = 1e30f;
          unsigned long long base = 0;
          for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, st));
            if (poll == 0) hipLaunchKernelGGL(tree_loop_kernel<0>, dim3(wgs), dim3(256), 0, st, s, t, iters, base, nleaf, 0ull, data);
            else hipLaunchKernelGGL(tree_loop_kernel<1>, dim3(wgs), dim3(256), 0, st, s, t, iters, base, nleaf, 0ull, data);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            base += iters;
          }
          Sync hs;
          CK(hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost));
          printf("  tree barrier, %4d workgroups, %2u leaves, poll by %s: %.2f us each%s\n", wgs, nleaf, poll ? "RMW " : "load",
                 1e3 * best / iters, hs.error ? "  ** SPIN LIMIT HIT **" : "");
        }
  }
  // launch round trips: launch -> result word visible on the host
  for (int mode = 0; mode < 2; ++mode) {
    double tot = 0;
    const int reps = 200;
    for (int r = 0; r < reps + 10; ++r) {
      res[0] = 0.0;
      double v = (double)(r + 1);
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) hipLaunchKernelGGL(tiny_kernel, dim3(512), dim3(256), 0, st, data, res, v);
      else {
        void *args[] = {(void *)&data, (void *)&res, (void *)&v};
        CK(hipLaunchCooperativeKernel((const void *)tiny_kernel, dim3(512), dim3(256), args, 0, st));
      }
      while (res[0] != v) {}
      const auto t1 = std::chrono::steady_clock::now();
      if (r >= 10) tot += std::chrono::duration<double>(t1 - t0).count();
    }
    printf("%s launch of a 512-workgroup kernel, until its result word is on the host: %.2f us\n",
           mode ? "hipLaunchCooperativeKernel" : "plain", 1e6 * tot / reps);
  }
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, barrier_loop_kernel<0>, 256, 0));
  printf("occupancy query: %d workgroups of 256 per CU\n", nb);
  return 0;
}
