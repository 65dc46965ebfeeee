// group_kernel.hpp -- part of the single translation unit pdhg_hip.hip (included there, after dist.hpp).
// One PDHG trial of a ROW-PARTITIONED group (dist.hpp) as ONE persistent kernel PER DEVICE: the launch's workgroups are
// dealt to the device's shards (a contiguous range each), every shard runs its own phases, and the shards -- of this
// launch and of the launches on the other devices -- exchange through each other's memory.  The low-latency form of the
// group path for LPs whose trial is latency, not bandwidth (stream layouts: L1-SVM class).  (One launch per SHARD was
// the first form: with four shards on one device two of the launches shared a hardware queue, ran one after the other
// and waited for each other at the first barrier.  One launch per device makes the co-residency a property of the
// launch.)
//
// Why.  The group's trial is ~8 launches per shard with cross-stream event barriers around two exchange kernels (peer
// back end) or two RCCL collectives: measured on the L1-SVM LP, two shards on one device, 146 us per trial (90 us of
// host issue, 54 us waiting) against 42-50 us for the plain handle (profiles/r04_l1svm_shards.txt).  Here a device's
// shards share one launch per trial; the shards meet at CROSS-SHARD barriers inside it:
//
//   phase 0   x', xbar on the OWNED column slice (K1+K2, + the deferred sum_x update); xbar is stored into EVERY
//             shard's xbar buffer (the all-gather as P stores per element: peer-mapped memory)
//   XB 1      cross-shard barrier: every shard's xbar is complete everywhere
//   phase 1   y'_p = proj(y_p + sigma (b_p - A_p xbar)), partial sum dy^2   (K3+K4 on the shard's rows)
//   barrier   (local: y'_p complete)
//   phase 2   t_p = A_p' y'_p, all n columns, into the shard's own A'y' buffer (a partial)
//   XB 2      cross-shard barrier: every partial is complete
//   phase 3   the OWNER of a column slice adds the shards' partials IN RANK ORDER (the reduce-scatter as P loads per
//             element -- the order of p2p_reduce_kernel, hence the same bits), stores A'y' and accumulates the
//             interaction sums of the slice (K6)
//   ticket    the shard's last workgroup runs the second stage and publishes the shard's five sums to pinned host
//             memory; the host adds the shards' sums in rank order and decides (as for every group path).
//
// Cross-shard barrier = the XCD-scoped grid barrier of trial_kernel.hpp with one more level: the shard's last XCD leader
// arrives on a counter all shards share (system scope) and releases its fellow leaders when every shard has arrived; the
// write-back / invalidate of each XCD's L2 around it are system-scope (`buffer_wbl2 sc0 sc1` / `buffer_inv sc0 sc1`) so
// that data another DEVICE wrote into this one's memory is read fresh.  Every spin is bounded; a barrier that cannot
// complete raises the error words and the host repeats the trial on the ordinary group path.
//
// Status: bitwise the ordinary group path with several shards on ONE device (tests/test_gpu_dist_group.py), the default
// there; on distinct devices it has never run (no multi-GPU box in this environment) and waits for PDHG_GROUP_COOP=1.
// PDHG_GROUP_COOP=0 turns it off.  No reference counterpart (single process, single thread); the arithmetic is
// src/primal_dual_hybrid_gradient.jl:442-549.
#pragma once

namespace {

struct GroupSync {                          // one per group, in memory every shard's device can reach; one line per word
  unsigned long long arrive[16];            // shards arrived at cross-shard barriers (monotonic over trials)
  unsigned long long error[16];
};

struct GroupTrialArgs {
  int rank, world;
  // phase 0 (owned slice: pointers already moved to the slice's first column)
  int cn, xbar_only;
  long long clo;
  const double *x, *c, *aty, *lb, *ub;
  double tau, theta;
  double *x_next;
  double avg_w;
  double *sum_x;
  double *xbar_peer[P2P_MAX_WORLD];         // every shard's full-length xbar (this shard's own at [rank])
  // phases 1 and 2
  TrialProduct A, T;                        // A: MODE_DUAL (xin = own xbar); T: MODE_PLAIN into the own A'y' buffer
  // phase 3
  const double *part_peer[P2P_MAX_WORLD];   // every shard's A'y' buffer (partials, full length)
  long long off;                            // first column of the owned slice inside those buffers
  double *aty_next;                         // own buffer (the owned slice is overwritten in place)
  double *pAt;                              // interaction partials, double-double: [6][pAt_stride], slot = workgroup
  int pAt_stride;
  FinalSpec sp;
  GridSync *sync;
  GroupSync *gsync;
  unsigned long long epoch, xepoch, launch, seq;
  unsigned nxcd;
  unsigned xcd_cnt[8];
  unsigned long long *seq_dev;
  volatile double *res_host;
  int relaxed;
};

// grid barrier (trial_kernel.hpp) + the cross-shard level.  epoch: this shard's barrier count; xepoch: the group's
// count of cross-shard barriers (the same on every shard).
__device__ __forceinline__ void group_barrier(GridSync *s, GroupSync *gs, unsigned long long epoch, unsigned long long xepoch,
                                              unsigned nxcd, const unsigned *xcd_cnt, int world) {
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_load(&s->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    const unsigned x = xcc_id();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long cnt = xcd_cnt[x];
    const unsigned long long prev = __hip_atomic_fetch_add(&s->xcd_arrive[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    if (prev + 1 == cnt * epoch) {
      asm volatile("buffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");      // system scope: peers read this shard's stores
      const unsigned long long g = __hip_atomic_fetch_add(&s->global[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g + 1 == (unsigned long long)nxcd * epoch) {
        // this shard's last XCD: the shard has arrived; meet the other shards, then release the fellow leaders
        __hip_atomic_fetch_add(&gs->arrive[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while (__hip_atomic_load(&gs->arrive[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)world * xepoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > GRID_SPIN_LIMIT ||
              ((spins & 0x3FF) == 0 && __hip_atomic_load(&gs->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
            __hip_atomic_store(&gs->error[0], 6ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&s->error[0], 6ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __hip_atomic_store(&s->xrelease[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(&s->xrelease[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > GRID_SPIN_LIMIT) { __hip_atomic_store(&s->error[0], 7ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          if ((spins & 0x3FF) == 0 && __hip_atomic_load(&s->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        }
      }
      asm volatile("buffer_inv sc0 sc1" ::: "memory");
      __hip_atomic_store(&s->xcd_release[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&s->xcd_release[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > GRID_SPIN_LIMIT) { __hip_atomic_store(&s->error[0], 8ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((spins & 0x3FF) == 0 && __hip_atomic_load(&s->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      }
    }
    asm volatile("s_dcache_inv" ::: "memory");
  }
  __syncthreads();
}

// the shards of one device's launch: their argument blocks (device memory) and the first workgroup of each
struct GroupDeviceArgs {
  const GroupTrialArgs *shard;
  int nshards;
  int base[P2P_MAX_WORLD + 1];
};

// census of the merged launch shape: how many workgroups of every shard land on each XCD (cf. xcd_register_kernel)
__global__ __launch_bounds__(TPB) void group_register_kernel(GroupDeviceArgs d, GridSync *const *sync) {
  if (threadIdx.x != 0) return;
  int s = 0;
  while (s + 1 < d.nshards && (int)blockIdx.x >= d.base[s + 1]) ++s;
  __hip_atomic_fetch_add(&sync[s]->xcd_count[xcc_id()][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Up to GROUP_INLINE_SHARDS shards per launch travel BY VALUE in the kernel arguments (the production case: one shard per
// device): no upload in front of the launch.  More shards on one device (the one-GPU test configuration): the argument
// blocks are copied to device memory first (GroupDeviceArgs).
constexpr int GROUP_INLINE_SHARDS = 2;

__device__ __forceinline__ void group_trial_body(const GroupTrialArgs &a, int w, int nwg, double *prod, double (*red)[TPB / WAVE],
                                                 int &done_flag);

__global__ __launch_bounds__(TPB, PDHG_TRIAL_WAVES_PER_EU) void group_trial_kernel(GroupDeviceArgs d) {
  __shared__ double prod[BLOCK_NNZ];
  __shared__ double red[6][TPB / WAVE];
  __shared__ int done_flag;
  int sh = 0;
  while (sh + 1 < d.nshards && (int)blockIdx.x >= d.base[sh + 1]) ++sh;
  group_trial_body(d.shard[sh], (int)blockIdx.x - d.base[sh], d.base[sh + 1] - d.base[sh], prod, red, done_flag);
}
// (one kernel parameter per shard: an array of blocks inside one parameter is copied to scratch as soon as a block is
//  indexed -- 2.3 KB per lane; separate parameters stay in the kernel-argument segment)
__global__ __launch_bounds__(TPB, PDHG_TRIAL_WAVES_PER_EU) void group_trial_inline_kernel(GroupTrialArgs a0, GroupTrialArgs a1, int nshards,
                                                                                          int base1, int total) {
  __shared__ double prod[BLOCK_NNZ];
  __shared__ double red[6][TPB / WAVE];
  __shared__ int done_flag;
  if (nshards > 1 && (int)blockIdx.x >= base1) group_trial_body(a1, (int)blockIdx.x - base1, total - base1, prod, red, done_flag);
  else group_trial_body(a0, (int)blockIdx.x, nshards > 1 ? base1 : total, prod, red, done_flag);
}

__device__ __forceinline__ void group_trial_body(const GroupTrialArgs &a, int w, int nwg, double *prod, double (*red)[TPB / WAVE],
                                                 int &done_flag) {
  unsigned long long epoch = a.epoch, xepoch = a.xepoch;
  Prefetched f;
  // ---- phase 0: the owned slice of x' and xbar; xbar goes to every shard (elementwise: any distribution gives the same bits)
  {
    const int stride = nwg * TPB;
    for (int j = w * TPB + threadIdx.x; j < a.cn; j += stride) {
      const double xv = a.x[j];
      double xn, xb;
      if (a.xbar_only) {                                   // Malitsky-Pock retries: xbar from the x' of pdhg_trial_primal
        xn = a.x_next[j];
        const double dlt = xn - xv;
        const double t = a.theta * dlt;
        xb = xn + t;
      } else {
        if (a.sum_x) {
          const double t = xv * a.avg_w;
          a.sum_x[j] = a.sum_x[j] + t;
        }
        primal_one<false, true>(xv, a.c[j], a.aty[j], 0.0, a.lb[j], a.ub[j], a.tau, a.theta, xn, xb);
        a.x_next[j] = xn;
      }
      for (int q = 0; q < a.world; ++q) a.xbar_peer[q][a.clo + j] = xb;
    }
  }
  product_prefetch(a.A, f, w, nwg);                        // static data: requested before the barrier
  group_barrier(a.sync, a.gsync, ++epoch, ++xepoch, a.nxcd, a.xcd_cnt, a.world);
  // ---- phase 1: y'_p and the partial sum dy^2
  product_phase<MODE_DUAL, false>(a.A, a.A.xin, a.A.e, a.A.uses, a.relaxed, f, prod, red, w, nwg);
  product_prefetch(a.T, f, w, nwg);
  grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
  // ---- phase 2: t_p = A_p' y'_p (all n columns, a partial)
  product_phase<MODE_PLAIN, false>(a.T, a.T.xin, a.T.e, a.T.uses, a.relaxed, f, prod, red, w, nwg);
  group_barrier(a.sync, a.gsync, ++epoch, ++xepoch, a.nxcd, a.xcd_cnt, a.world);
  // ---- phase 3: the owned slice of A'y' = the shards' partials added in rank order; interaction sums of the slice
  {
    Acc3 acc = acc3_zero();
    const int stride = nwg * TPB;
    for (int j = w * TPB + threadIdx.x; j < a.cn; j += stride) {
      double v = a.part_peer[0][a.off + j];
      for (int q = 1; q < a.world; ++q) {
        const double t = a.part_peer[q][a.off + j];
        v = v + t;
      }
      a.aty_next[a.off + j] = v;
      const double dx = a.x_next[j] - a.x[j];              // (own stores of phase 0 / the accepted iterate)
      const double dd = v - a.aty[j];
      dd_add(acc.hi[0], acc.lo[0], dx * dd);
      dd_add(acc.hi[1], acc.lo[1], dx * dx);
      dd_add(acc.hi[2], acc.lo[2], dd * dd);
    }
    __syncthreads();                                       // `red` is free again
    block_sum_dd<3, TPB>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        store_agent(a.pAt + q * a.pAt_stride + w, acc.hi[q]);
        store_agent(a.pAt + (3 + q) * a.pAt_stride + w, acc.lo[q]);
      }
    }
  }
  // ---- second stage on the shard's last workgroup (two-level ticket as in trial_kernel)
  __syncthreads();
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned x = xcc_id();
    const unsigned long long cnt = a.xcd_cnt[x];
    const unsigned long long t = __hip_atomic_fetch_add(&a.sync->xcd_done[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    done_flag = 0;
    if (t + 1 == (a.launch + 1) * cnt) {
      const unsigned long long u = __hip_atomic_fetch_add(&a.sync->ticket[2][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      done_flag = (u + 1 == (a.launch + 1) * (unsigned long long)a.nxcd);
    }
    if (done_flag) asm volatile("buffer_inv sc1" ::: "memory");
  }
  __syncthreads();
  if (done_flag) {
    double res[5];
    const unsigned long long errw = threadIdx.x == 0 ? (__hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                                                        __hip_atomic_load(&a.gsync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) : 0ull;
    final_reduce_body<TPB / WAVE>(a.sp, res);
    if (threadIdx.x == 0) {
      const double err = (double)errw;
      const double seq = (double)a.seq;
      unsigned long long ck = RESULT_CHECK_SALT ^ (unsigned long long)__double_as_longlong(err) ^ (unsigned long long)__double_as_longlong(seq);
#pragma unroll
      for (int k = 0; k < 5; ++k) ck ^= (unsigned long long)__double_as_longlong(res[k]);
#pragma unroll
      for (int k = 0; k < 5; ++k) a.res_host[k] = res[k];
      a.res_host[5] = __longlong_as_double((long long)ck);
      a.res_host[6] = err;
      a.res_host[7] = seq;
      *a.seq_dev = a.seq;
    }
  }
}

}  // namespace
