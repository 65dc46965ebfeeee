// trial_kernel.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// One PDHG trial step (pdhg.jl:442-549: K1+K2, K3+K4, K5+K6, second-stage reduction) as ONE
// kernel launch for stream-layout LPs: persistent workgroups, two grid-wide barriers.
//
// Why: on small / medium LPs a trial is latency, not bandwidth.  As a HIP graph (round 2) the
// L1-SVM LP's iteration was 47 us of kernels + ~15 us of launch / dependency latency between
// the 6-10 nodes + 28 us of host time inside hipGraphLaunch.  Here the chain is one launch,
// the barriers cost 4-5 us each, and -- what only a fused kernel can do -- every workgroup
// requests its first row block's (col, val) entries BEFORE the barrier that delivers the
// vector they will be multiplied with: the static half of the SpMV's dependent memory chain
// (block descriptor -> row pointers -> entries) overlaps the previous phase's tail.
// The boundary of SURVEY 8b does not move: pdhg_trial_step still returns the five sums and the
// step-size rule stays on the host.
//
// The phase bodies are the SAME device functions the separate kernels run (spmv_kernels.hpp,
// vector_kernels.hpp), every block writes the same partial slot and the second stage adds
// them in the same order, so the results are bitwise those of the separate launches
// (tests/test_gpu_native_take_step.py).
//
// Grid barrier (measured: tools/grid_barrier_probe.hip, profiles/r03_grid_barrier_probe.txt).
// An agent-scope release / acquire on this chip writes back / invalidates the XCD's L2
// (buffer_wbl2 sc1 / buffer_inv sc1), and when every workgroup issues its own they serialise
// in the L2: 9.4 us per barrier for 256 workgroups, 19 us for 512, 65 us for 2048, whatever
// the counter structure (flat, tree, per-XCD).  So the fences are scoped by hand: a
// workgroup's stores are in its XCD's L2 once `s_waitcnt vmcnt(0)` returns (the L1 is
// write-through); it then arrives on its XCD's counter; only the LAST arriver of the XCD writes
// the L2 back, arrives on the global counter, waits for all 8 XCDs, invalidates the L2 and
// releases its XCD.  3.2 / 3.8 / 5.0 / 7.4 us for 256 / 512 / 1024 / 2048 workgroups.
// The other workgroups do NOT invalidate their CU's L1 (`buffer_inv sc0` is a no-op on this
// chip, `sc1` would serialise in the L2 again): not needed for THIS kernel's data flow -- the
// L1 is clean at kernel start and no address is read before the phase that produces it has
// completed (x', xbar: written in phase 0, read from phase 1 on; y': written in phase 1, read
// in phase 2; partial sums: read at the end only), so no CU can hold a stale line.
// The XCD of a workgroup comes from the hardware register (XCC_ID); how many workgroups of a
// launch land on each XCD is counted once per handle by a registration launch of the same
// shape (the dispatcher deals workgroups round-robin; the probe saw exact, repeatable counts).
// Every spin is bounded: a barrier that cannot complete (workgroups not co-resident because
// the device is shared) raises the error word instead of hanging, and the host reports it.
#pragma once

namespace {

struct GridSync {                           // device memory, one per handle; one 128-byte line per word
  unsigned long long global[16];            // XCD leaders arrived (monotonic over launches)
  unsigned long long xcd_arrive[8][16];
  unsigned long long xcd_release[8][16];    // last completed barrier epoch of the XCD
  unsigned long long xcd_count[8][16];      // workgroups of one launch on each XCD
  unsigned long long xcd_done[8][16];       // workgroups of the XCD that have finished their last phase
  unsigned long long ticket[3][16];         // [2]: XCDs done (the last workgroup of the last XCD runs the second-stage reduction)
  unsigned long long error[16];
  unsigned long long xrelease[16];          // group_kernel.hpp: last cross-shard barrier the shard's last XCD leader has passed
};

__device__ __forceinline__ unsigned xcc_id() {
  return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7;   // hwreg(HW_REG_XCC_ID, 0, 4)
}

__global__ __launch_bounds__(TPB) void xcd_register_kernel(GridSync *s) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&s->xcd_count[xcc_id()][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef PDHG_TRIAL_WAVES_PER_EU
#define PDHG_TRIAL_WAVES_PER_EU 4
#endif
#ifndef PDHG_TRIAL_PIPE
#define PDHG_TRIAL_PIPE true
#endif
// the one-wave-per-quantity second stage of LPs (final_reduce_lp): load pairs in flight per lane; used by the single-trial kernel too
#ifndef PDHG_STEPS_LP_BATCH
#define PDHG_STEPS_LP_BATCH 8      // (L1-SVM, 856 slots per quantity: 22.8-23.2k it/s with 4, 23.1-23.4k with 6, 23.4-23.5k with 8)
#endif
#ifndef PDHG_TRIAL_LP_SECOND_STAGE
#define PDHG_TRIAL_LP_SECOND_STAGE 0   // (the single-trial kernel does not spill in the general form: no difference measured there)
#endif
constexpr unsigned long long RESULT_CHECK_SALT = 0x9E3779B97F4A7C15ull;
constexpr long GRID_SPIN_LIMIT = 4000000L;   // x s_sleep(1): ~0.1 s

// epoch = 1, 2, ... over the life of the handle; nxcd = XCDs that hold workgroups
// xcd_cnt: workgroups of this launch on each XCD (the census, passed in the kernel arguments: a load of it here
// would put one more ~1.5 us trip to memory in front of every arrival)
// err_known: the error word as thread 0 read it a little earlier (the multi-step kernel requests it at the start of the
// phase, off the critical path; ~0ull: read it here)
__device__ __forceinline__ void grid_barrier(GridSync *s, unsigned long long epoch, unsigned nxcd, const unsigned *xcd_cnt,
                                             unsigned long long err_known = ~0ull) {
  __syncthreads();       // every wave's workgroup-scope release: its stores have reached the XCD's L2
  if (threadIdx.x == 0 && (err_known != ~0ull ? err_known : __hip_atomic_load(&s->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
    const unsigned x = xcc_id();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long cnt = xcd_cnt[x];
    const unsigned long long prev = __hip_atomic_fetch_add(&s->xcd_arrive[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    if (prev + 1 == cnt * epoch) {
      asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&s->global[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&s->global[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)nxcd * epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > GRID_SPIN_LIMIT) { __hip_atomic_store(&s->error[0], 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      asm volatile("buffer_inv sc1" ::: "memory");
      __hip_atomic_store(&s->xcd_release[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&s->xcd_release[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > GRID_SPIN_LIMIT) { __hip_atomic_store(&s->error[0], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    asm volatile("s_dcache_inv" ::: "memory");
  }
  __syncthreads();
}

// The barrier of a launch whose workgroups all sit on ONE XCD (steps_kernel's XCD-local mode): they share one L2, so
// there is nothing to write back or invalidate and no second level -- an arrival counter and a release word in that L2.
// (Loads of data another compute unit rewrote still have to pass the reader's L1: the agent-scope loads the multi-step
// kernel uses anyway.)  cnt: workgroups of the launch; x: their XCD.
__device__ __forceinline__ void grid_barrier_local(GridSync *s, unsigned long long epoch, unsigned x, unsigned long long cnt,
                                                   unsigned long long err_known) {
  __syncthreads();
  if (threadIdx.x == 0 && err_known == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long prev = __hip_atomic_fetch_add(&s->xcd_arrive[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == cnt * epoch) {
      __hip_atomic_store(&s->xcd_release[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long spins = 0;
      while (__hip_atomic_load(&s->xcd_release[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > GRID_SPIN_LIMIT) { __hip_atomic_store(&s->error[0], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    asm volatile("s_dcache_inv" ::: "memory");
  }
  __syncthreads();
}

// store that is visible device-wide once `s_waitcnt vmcnt(0)` has returned (write-through,
// no L2 write-back needed): the few words a workgroup hands to a "last one finishes" ticket
__device__ __forceinline__ void store_agent(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One fused SpMV of the trial: a stream layout (row blocks + long-row chunks) with its epilogue
struct TrialProduct {
  CsrView M;
  const int2 *blks;
  int nblk, per_xcd, grid, remap;
  int nchunks, nlong, long_grid;
  const int *chunk_row, *chunk_off, *chunk_lidx;
  double *chunk_partial;
  const int *long_row, *long_chunk_ptr;
  unsigned long long *long_ticket;
  const double *xin;
  EpiArgs e;
  unsigned long long uses;    // one-launch trials that have run this product before (its long rows' tickets count up)
};

struct TrialKernelArgs {
  // phase 0: K1+K2 (or xbar alone for the Malitsky-Pock retries)
  int n, xbar_only;
  const double *x, *c, *aty, *lb, *ub;
  double tau, theta;
  double *x_next, *xbar;
  double avg_w;
  double *sum_x;
  TrialProduct A, T;
  // QP (pdhg.jl:536-541; saddle_point.jl:1093-1100): Q x before the primal step, 0.5 dx'Q dx beside the other sums
  int has_q, q_blocks;           // q_blocks: blocks of the dot-product partials (the separate dot_kernel's grid)
  TrialProduct Qx, Qtdx;         // MODE_PLAIN: qx = Q x ; tmp_n2 = Q' dx
  const double *qx;
  double *dx, *qtdx, *pq;        // dx = x' - x, Q' dx, partials of dx . (Q' dx)
  unsigned long long epoch;      // grid barriers this handle has passed so far
  FinalSpec sp;
  unsigned long long *seq_dev;
  volatile double *res_host;
  GridSync *sync;
  unsigned long long launch;     // 0, 1, 2, ...: this handle's launches of this kernel so far
  unsigned long long seq;        // sequence number to publish with the results
  unsigned nxcd;
  unsigned xcd_cnt[8];           // workgroups of one launch on each XCD (census of coop_prepare)
  int relaxed;
  unsigned long long *trace;     // PDHG_COOP_TRACE: [workgroup][8] wall-clock stamps (100 MHz) at the phase boundaries
};

__device__ __forceinline__ bool product_block_of(const TrialProduct &P, int b, int *blk) {
  const int k = P.remap ? ((b & (NUM_XCD - 1)) * P.per_xcd + (b >> 3)) : b;
  *blk = k;
  return P.remap ? ((b >> 3) < P.per_xcd && k < P.nblk) : (k < P.nblk);
}

// this workgroup's share of one product: row blocks b = w, w + nwg, ... (b mod 8 == w mod 8:
// every XCD walks the same contiguous eighth of the row blocks as in the separate launch) and
// long-row chunks dealt from the END of the grid (chunk c to workgroup nwg - 1 - c mod nwg), so
// that with a grid of blocks + chunks workgroups everybody has one item.  The workgroup that
// completes a long row's last chunk (a ticket per row) finishes the row: ordered sum of its
// chunk partials, epilogue, the row's slot of the block partials.  `pre`: block w's entries
// are already in `g`.
// the first item of a workgroup in a phase, requested before the grid barrier: a row block or
// (workgroups at the end of the grid) a long-row chunk
struct Prefetched {
  int kind;            // 0 nothing, 1 row block, 2 long-row chunk
  StreamRegs g;
};
// (w, nwg: this workgroup's index and the number of workgroups that share the product -- the launch's own by default;
//  the group kernel, one launch for several shards, passes the workgroup's place inside its shard)
__device__ __forceinline__ void product_prefetch(const TrialProduct &P, Prefetched &f, int w, int nwg) {
  int blk;
  f.kind = 0;
  const int c = nwg - 1 - w;
  if (c < P.nchunks) {                                   // chunks are processed first
    long_chunk_load(P.M, P.chunk_row[c], P.chunk_off[c], f.g);
    f.kind = 2;
  } else if (w < P.grid && product_block_of(P, w, &blk)) {
    stream_block_load(P.M, P.blks[blk], f.g);
    f.kind = 1;
  }
}
__device__ __forceinline__ void product_prefetch(const TrialProduct &P, Prefetched &f) { product_prefetch(P, f, (int)blockIdx.x, (int)gridDim.x); }

// xin, e, uses: the operands that change from trial to trial (the single-trial kernel passes P's own)
template <int MODE, bool COH = false>
__device__ __forceinline__ void product_phase(const TrialProduct &P, const double *xin, const EpiArgs &e,
                                              unsigned long long uses, int relaxed,
                                              Prefetched &f, double *prod, double (*red)[TPB / WAVE], int w, int nwg) {
  const unsigned long long launch = uses;
  __shared__ int finish_row;
  StreamRegs &g = f.g;
  const bool pre = f.kind == 1;
  for (int c = nwg - 1 - w; c < P.nchunks; c += nwg) {
    __syncthreads();
    if (!(f.kind == 2 && c == nwg - 1 - w)) long_chunk_load(P.M, P.chunk_row[c], P.chunk_off[c], f.g);
    const double part = long_chunk_finish<COH>(xin, f.g, red);
    if (threadIdx.x == 0) {
      store_agent(P.chunk_partial + c, part);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int l = P.chunk_lidx[c];
      const unsigned long long per_launch = (unsigned long long)(P.long_chunk_ptr[l + 1] - P.long_chunk_ptr[l]);
      const unsigned long long t = __hip_atomic_fetch_add(P.long_ticket + l, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      finish_row = (t + 1 == (launch + 1) * per_launch) ? l : -1;
      if (finish_row >= 0) asm volatile("buffer_inv sc1" ::: "memory");   // the row's partials are in memory: read them fresh
    }
    __syncthreads();
    if (finish_row >= 0 && threadIdx.x < WAVE)                             // one wave finishes the row
      long_final_row<MODE, true, COH>(finish_row, P.long_row, P.long_chunk_ptr, P.chunk_partial, e, P.grid);
  }
  constexpr int NQ = ModeNQ<MODE>::value;
  for (int b = w; b < P.grid; b += nwg) {
    __syncthreads();                   // `prod` and `red` are free again
    int blk;
    const bool active = product_block_of(P, b, &blk);
    Acc3 acc = acc3_zero();
    if (active) {
      if (!(pre && b == w)) stream_block_load(P.M, P.blks[blk], g);
      stream_block_finish<MODE, false, PDHG_TRIAL_PIPE, COH>(P.M, xin, g, e, relaxed, acc, prod);
    }
    if (NQ > 0) {
      block_sum_dd<NQ, TPB>(acc, red);
      if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          store_agent(e.partials + q * e.stride + b, acc.hi[q]);
          store_agent(e.partials + e.lo_offset + q * e.stride + b, acc.lo[q]);
        }
      }
    }
  }
}

template <int MODE, bool COH = false>
__device__ __forceinline__ void product_phase(const TrialProduct &P, const double *xin, const EpiArgs &e,
                                              unsigned long long uses, int relaxed,
                                              Prefetched &f, double *prod, double (*red)[TPB / WAVE]) {
  product_phase<MODE, COH>(P, xin, e, uses, relaxed, f, prod, red, (int)blockIdx.x, (int)gridDim.x);
}

// waves per SIMD: the kernel needs 96 VGPRs (5 waves per SIMD, 5 workgroups per CU).  Forcing the 8 of the separate
// stream kernel (64 VGPRs) spills 132 bytes per lane and is 1.4-2x SLOWER (L1-SVM 8.9k against 12.3k it/s, PageRank-1M
// 2.4k against 4.4k: profiles/r03_trial_kernel.txt).
template <bool COH>
__global__ __launch_bounds__(TPB, PDHG_TRIAL_WAVES_PER_EU) void trial_kernel(TrialKernelArgs a) {
  __shared__ double prod[BLOCK_NNZ];
  __shared__ double red[6][TPB / WAVE];
  __shared__ int done_flag;
  const int w = blockIdx.x, nwg = gridDim.x;
#define PDHG_STAMP(k) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)w * 8 + (k)] = wall_clock64(); } while (0)
  PDHG_STAMP(0);
  Prefetched f;
  unsigned long long epoch = a.epoch;
  if (a.has_q && !a.xbar_only) {
    // ---- QP, phase -1: Q x (the gradient's quadratic term), then a barrier of its own
    f.kind = 0;
    product_phase<MODE_PLAIN, COH>(a.Qx, a.Qx.xin, a.Qx.e, a.Qx.uses, a.relaxed, f, prod, red);
    grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
  }
  // ---- phase 0: x' and xbar (elementwise; any distribution over the workgroups gives the same bits)
  if (a.xbar_only) xbar_body(a.n, a.x, a.x_next, a.theta, a.xbar, w, nwg);
  else if (a.has_q) primal_body<true, true>(a.n, a.x, a.c, a.aty, a.qx, a.lb, a.ub, a.tau, a.theta, a.x_next, a.xbar, a.avg_w, a.sum_x, w, nwg);
  else primal_body<false, true, COH>(a.n, a.x, a.c, a.aty, nullptr, a.lb, a.ub, a.tau, a.theta, a.x_next, a.xbar, a.avg_w, a.sum_x, w, nwg);
  // dx for the interaction term.  After the primal step every thread reads back the x' it wrote itself (same
  // element mapping); on the Malitsky-Pock retries x' is an earlier kernel's output.
  if (a.has_q) diff_pairs_body(a.n, a.x_next, a.x, a.dx, w, nwg);
  product_prefetch(a.A, f);                                      // static data: requested before the barrier
  PDHG_STAMP(1);
  grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
  PDHG_STAMP(2);
  // ---- phase 1: y' = proj(y + sigma (b - A xbar)), sum dy^2   (K3+K4)
  product_phase<MODE_DUAL, COH>(a.A, a.A.xin, a.A.e, a.A.uses, a.relaxed, f, prod, red);
  if (a.has_q) {                                                 // Q' dx: independent of A xbar, same phase
    Prefetched none;
    none.kind = 0;
    product_phase<MODE_PLAIN, COH>(a.Qtdx, a.Qtdx.xin, a.Qtdx.e, a.Qtdx.uses, a.relaxed, none, prod, red);
  }
  product_prefetch(a.T, f);
  PDHG_STAMP(3);
  grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
  PDHG_STAMP(4);
  // ---- phase 2: A'y' and the interaction sums   (K5+K6)
  product_phase<MODE_ATY, COH>(a.T, a.T.xin, a.T.e, a.T.uses, a.relaxed, f, prod, red);
  if (a.has_q) {                                                 // partials of dx . (Q' dx), block by block as dot_kernel does
    for (int b = w; b < a.q_blocks; b += nwg) {
      __syncthreads();
      dot_body(a.n, a.qtdx, a.dx, a.pq, b, a.q_blocks, red, true);
    }
  }
  PDHG_STAMP(5);
  // ---- second stage: the workgroup that finishes last adds the block partials (K6b)
  __syncthreads();
  if (threadIdx.x == 0) {
    // every block partial of this workgroup went out as a write-through store: once they are
    // acknowledged, take the ticket
    // (two levels, like the barrier: atomics on ONE address are served at ~15-25 ns apiece, and a flat ticket over
    // 500-1000 workgroups that finish together cost 8-12 us here)
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
    // (requested before the partials so that its trip to memory overlaps theirs)
    const unsigned long long errw = threadIdx.x == 0 ? __hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#if PDHG_TRIAL_LP_SECOND_STAGE
    if (!a.has_q) final_reduce_lp<PDHG_STEPS_LP_BATCH>(a.sp, res, &red[0][0]);      // (uniform; `red` is free: 6 x 4 doubles)
    else
#endif
    final_reduce_body<TPB / WAVE>(a.sp, res);
    if (threadIdx.x == 0) {
      // Publish WITHOUT a system-scope fence (an L2 write-back, microseconds): the eight words
      // may reach host memory in any order, so word 5 carries a checksum over the others and
      // the host accepts a read only when the sequence number AND the checksum match.
      const double err = (double)errw;
      const double seq = (double)a.seq;                  // exact up to 2^53 launches
      unsigned long long ck = RESULT_CHECK_SALT ^ (unsigned long long)__double_as_longlong(err) ^ (unsigned long long)__double_as_longlong(seq);
#pragma unroll
      for (int k = 0; k < 5; ++k) ck ^= (unsigned long long)__double_as_longlong(res[k]);
#pragma unroll
      for (int k = 0; k < 5; ++k) a.res_host[k] = res[k];
      a.res_host[5] = __longlong_as_double((long long)ck);
      a.res_host[6] = err;
      a.res_host[7] = seq;
      *a.seq_dev = a.seq;                                // the graph path's device-side counter stays in step
    }
    PDHG_STAMP(6);
  }
#undef PDHG_STAMP
}


// ---- several adaptive take_steps per launch (pdhg_take_steps_adaptive) ------------------------------------------
// Between two termination evaluations the reference's loop is take_step after take_step (pdhg.jl:862-1046), and
// take_step's scalar part needs nothing but the trial's five sums: so the accept / reject decision and the
// step-size rule run on the device too (adaptive_step_rule, common.hpp: the host loop's own function, with the two
// powers of the iteration count looked up in a table the HOST computed), and one launch takes up to n_steps steps.
// Per trial that removes the launch latency (~5 us), the result's trip to host memory and back (~3 us) and the
// host's own microseconds, and adds a third grid barrier: the XCD's last arriver -- its L1 and L2 are clean after
// its acquire -- runs the second-stage reduction and the rule between the barrier's global phase and the release,
// and leaves the decision in its XCD's control line for the workgroups it releases.  All eight leaders compute
// the same bits from the same partials.
//
// THE DEFAULT for stream-layout LPs on one handle (PDHG_DEVICE_LOOP=0: one launch per trial).  Bitwise the per-trial
// launches over thousands of steps (tests/test_gpu_device_loop.py).  Measured (profiles/r03_trial_kernel.txt):
// L1-SVM (856 workgroups) 19.7k -> 23.4k it/s, random 100K x 100K 22.5k -> 28.6k, 60000 x 50000 27.2k -> 40.7k,
// 3000 x 2500 32.7k -> 52.7k; random 250K x 250K (1232 workgroups) a tie.
// What decided it was the SECOND STAGE.  With the general form (final_reduce_body: 2 x 4 x 5 doubles in flight per
// lane) the kernel, at its register limit, spills there once the trial loop lets the compiler hoist that code's
// address arithmetic: 11-22 us from "barrier 3's global phase complete" to "every workgroup knows the decision", and
// the whole loop was SLOWER than a launch per trial on large grids (L1-SVM 17.1k against 19.9k).  With one wave per
// quantity and eight load pairs in flight (final_reduce_lp) that takes 2.9 us on 40 workgroups and 6.9 us on 856.
// What the loop still pays: the phases up to the end of phase 2 take ~2 us longer than after a launch (a launch
// starts every workgroup together, the loop starts each where the previous decision reached it; coherent loads), and a
// third barrier (~2 us to its global phase); what it saves: the launch (~5 us), the completion ticket and the result's
// trip to the host and back (~6 us), the host's own ~3 us.
//
// What a multi-trial kernel must add to the single-trial one is L1 coherence ACROSS trials: x', xbar, y', A'y' are
// rewritten every trial by other compute units than those that read them, and a compute unit's L1 may still hold
// last trial's lines.  A per-workgroup L1 invalidate serialises in the XCD's L2 (~50 ns per workgroup:
// profiles/r03_grid_barrier_probe.txt, 5.7 us per trial for 856 workgroups), so instead every load of a vector that
// another compute unit may have rewritten is an agent-scope load (ldc<true>: served by the L2, which the barrier
// leaders keep coherent) -- the gathers, the epilogues' operands, A'y in the primal step.  Data a thread re-reads
// after ITS OWN store (x in the primal step, the running sums) and the static matrix and problem vectors stay plain.
struct StepsCtl {
  double slot[8][16];          // per XCD (one 128-byte line), as 64-bit words: [0] epoch << 2 | numerical_error << 1 | accept,
                               // [1] the next step size's bits, [2] word 0 ^ word 1 ^ salt
  FinalSpec sp;                // where the block partials are (read by the leaders only: as kernel arguments these 27
};                             // SGPRs' worth of pointers stay live through every phase and push the kernel into scratch)

struct StepsKernelArgs {
  int n, num_eq;
  double *xa, *xb, *ya, *yb, *atya, *atyb;       // the iterate and the trial point: flip 0 -> x = xa, x' = xb, ...
  const double *c, *lb, *ub, *b;
  double *xbar, *sum_x, *sum_y;
  TrialProduct A, T;                              // static parts (xin / e / uses are set per trial)
  double *pA, *pAt;
  int pA_slots, pAt_stride;
  double primal_weight, step_size;
  int n_steps, max_trials, table_len;
  int pend;                                       // the accept before this launch left its average update to the next trial
  double pend_w;
  double wsum_x, wsum_y;                          // the averages' weight sums (every accept adds the step size on entry)
  const double *pow_red, *pow_growth;             // entry t: the powers for the launch's t-th trial
  unsigned long long epoch, uses_a, uses_t;
  GridSync *sync;
  StepsCtl *ctl;
  volatile double *res_host;
  unsigned long long seq;
  unsigned nxcd;
  unsigned xcd_cnt[8];
  int relaxed;
  unsigned long long *trace;                      // PDHG_COOP_TRACE: stamps of the launch's last trial, as for trial_kernel ([7]: leaders, global phase done)
  int local_g;                                    // > 0: XCD-local mode -- the launch is 8 x local_g workgroups, those on XCD local_home work
  unsigned local_home;
  unsigned long long local_ticket_base;            // tickets drawn by earlier launches (local_g each)
};

// result words: [0] step size, [1] steps taken, [2] trials, [3] flip, [4] pending average update, [5] its weight,
// [6] / [7] weight sums, [8] numerical_error, [9] aborted on a barrier time-out, [10] barrier epoch, [11] the barriers'
// error word, [12] sequence number; [13] checksum over [0..12]; [14] nonzero: ended inside a take_step (table exhausted) whose step size on entry this is;
// [15] sequence number again (what the host polls)
constexpr int STEPS_RES_WORDS = 16;

#ifndef PDHG_STEPS_PREFETCH
#define PDHG_STEPS_PREFETCH 1
#endif
// LOCAL: the XCD-local mode as its own instantiation (the all-XCD kernel sits at its register limit: as a run-time flag
// the mode cost it 92 more bytes of scratch per lane)
template <bool LOCAL>
__global__ __launch_bounds__(TPB, PDHG_TRIAL_WAVES_PER_EU) void steps_kernel(StepsKernelArgs a) {
  __shared__ double prod[BLOCK_NNZ];
  __shared__ double red[6][TPB / WAVE];
  __shared__ int s_leader;
  __shared__ double s_dec[4];
  // the doubles of the loop's state live in LDS (thread 0 updates them): as registers they are ten more VGPRs in a
  // kernel that has none to spare.  [0] step size of the trial, [1] step size on entry of the take_step, [2] weight of
  // the pending average update, [3] / [4] the averages' weight sums
  __shared__ double s_st[5];
  __shared__ double s_pow[2];
  __shared__ double s_res8[8];
  int w = blockIdx.x, nwg = gridDim.x;
  constexpr bool local = LOCAL;
  if (local) {
    // XCD-local mode (small grids): one XCD's workgroups do all the work -- one L2, barriers without write-back /
    // invalidate (grid_barrier_local).  Eight times the workgroups are launched (the dispatcher deals them round the
    // XCDs); the others leave at once.  The stayers number themselves as they arrive: any bijection serves (row blocks
    // and partial slots are indexed by item, not by who worked on it).
    if (xcc_id() != a.local_home) return;
    __shared__ int s_w;
    if (threadIdx.x == 0)
      s_w = (int)(__hip_atomic_fetch_add(&a.sync->ticket[0][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.local_ticket_base);
    __syncthreads();
    w = s_w; nwg = a.local_g;
    // more workgroups on the home XCD than the census saw: the surplus leaves (exactly local_g work, the barriers' counts
    // hold); fewer: a barrier times out and the host goes back to the all-XCD kernel
    if (w >= nwg) return;
  }
  int flip = 0, pend = a.pend, steps = 0, trials = 0, num_err = 0, hw_err = 0, mid = 0;
  if (threadIdx.x == 0) { s_st[0] = a.step_size; s_st[1] = a.step_size; s_st[2] = a.pend_w; s_st[3] = a.wsum_x; s_st[4] = a.wsum_y; }
  unsigned long long epoch = a.epoch;
  Prefetched f;
  __syncthreads();
  // the trial budget ends the launch only BETWEEN take_steps (mid: the last trial was rejected -- the host API carries
  // one step size, not the pair (trial step, step on entry) of an unfinished take_step); the pow tables hold
  // table_len >= max_trials entries for that
  while (steps < a.n_steps && (trials < a.max_trials || mid) && trials < a.table_len) {
    const double step = s_st[0], pend_w = s_st[2];
#define PDHG_STAMP(k) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)w * 8 + (k)] = wall_clock64(); } while (0)
    PDHG_STAMP(0);
    double *x = flip ? a.xb : a.xa, *xn = flip ? a.xa : a.xb;
    double *y = flip ? a.yb : a.ya, *yn = flip ? a.ya : a.yb;
    double *aty = flip ? a.atyb : a.atya, *atyn = flip ? a.atya : a.atyb;
    const double tau = step / a.primal_weight, sigma = a.primal_weight * step;
    // thread 0 requests, off the critical path, what the barriers and a leader will need: the error word and the
    // trial's two powers (any workgroup may turn out to be its XCD's leader)
    unsigned long long err_pref = 0;
    double pw_r = 0.0, pw_g = 0.0;
#if PDHG_STEPS_PREFETCH
    if (threadIdx.x == 0) {
      err_pref = __hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pw_r = a.pow_red[trials]; pw_g = a.pow_growth[trials];
    }
#else
    err_pref = ~0ull;
#endif
    // ---- phase 0: x' and xbar (+ the deferred sum_x update of the previous accept)
    primal_body<false, true, true>(a.n, x, a.c, aty, nullptr, a.lb, a.ub, tau, 1.0, xn, a.xbar, pend_w,
                                   pend ? a.sum_x : nullptr, w, nwg);
    product_prefetch(a.A, f, w, nwg);
#if PDHG_STEPS_PREFETCH
    if (threadIdx.x == 0) { s_pow[0] = pw_r; s_pow[1] = pw_g; }
#endif
    PDHG_STAMP(1);
    if (local) grid_barrier_local(a.sync, ++epoch, a.local_home, (unsigned long long)a.local_g, err_pref);
    else grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt, err_pref);
    PDHG_STAMP(2);
    // ---- phase 1: y' and sum dy^2 (+ the deferred sum_y update)
#if PDHG_STEPS_PREFETCH
    if (threadIdx.x == 0) err_pref = __hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    {
      EpiArgs e{};
      e.y = y; e.b = a.b; e.y_next = yn; e.sigma = sigma; e.num_eq = a.num_eq;
      e.partials = a.pA; e.stride = a.pA_slots; e.lo_offset = a.pA_slots;
      if (pend) { e.sum_y = a.sum_y; e.avg_w = pend_w; }
      product_phase<MODE_DUAL, true>(a.A, a.xbar, e, a.uses_a + (unsigned long long)trials, a.relaxed, f, prod, red, w, nwg);
    }
    pend = 0;
    product_prefetch(a.T, f, w, nwg);
    PDHG_STAMP(3);
    if (local) grid_barrier_local(a.sync, ++epoch, a.local_home, (unsigned long long)a.local_g, err_pref);
    else grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt, err_pref);
    PDHG_STAMP(4);
    // ---- phase 2: A'y' and the interaction sums
#if PDHG_STEPS_PREFETCH
    if (threadIdx.x == 0) err_pref = __hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    {
      EpiArgs e{};
      e.x = x; e.x_next = xn; e.aty = aty; e.aty_next = atyn;
      e.partials = a.pAt; e.stride = a.pAt_stride; e.lo_offset = 3 * a.pAt_stride;
      product_phase<MODE_ATY, true>(a.T, yn, e, a.uses_t + (unsigned long long)trials, a.relaxed, f, prod, red, w, nwg);
    }
    PDHG_STAMP(5);
    // ---- third barrier; its XCD leaders reduce and decide, and the decision IS the release: three words in the
    // XCD's control line -- {epoch, accept, numerical_error}, the next step size, a check word -- that the waiting
    // workgroups poll (one trip instead of release word, then decision, then error word)
    ++epoch;
    __syncthreads();
    if (threadIdx.x == 0) {
      s_leader = 0;
#if !PDHG_STEPS_PREFETCH
      err_pref = __hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      s_dec[3] = (double)err_pref;
      if (err_pref == 0) {
        const unsigned xcd = xcc_id();
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(a.ctl->slot[xcd]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long cnt = local ? (unsigned long long)a.local_g : (unsigned long long)a.xcd_cnt[xcd];
        const unsigned long long prev = __hip_atomic_fetch_add(&a.sync->xcd_arrive[xcd][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        if (prev + 1 == cnt * epoch) {
          // (XCD-local mode: the partials are in this XCD's L2 already -- no write-back, no global phase; the invalidate
          //  below still clears this compute unit's L1, which may hold the partials of an earlier trial)
          if (!local) {
            asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&a.sync->global[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          while (!local && __hip_atomic_load(&a.sync->global[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)a.nxcd * epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > GRID_SPIN_LIMIT) { __hip_atomic_store(&a.sync->error[0], 4ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_dec[3] = 4.0; break; }
          }
          asm volatile("buffer_inv sc1" ::: "memory");
          // a leader whose global wait timed out must not reduce (the partials are incomplete) nor publish a decision
          // with the current epoch: its XCD's slot stays stale and the waiters below leave through the error word
          s_leader = (s_dec[3] == 0.0) ? 1 : 0;
          PDHG_STAMP(7);
        } else {
          for (;;) {
            const unsigned long long w0 = __hip_atomic_load(slot + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long w1 = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long w2 = __hip_atomic_load(slot + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w0 >> 2) == epoch && (w0 ^ w1 ^ RESULT_CHECK_SALT) == w2) {
              s_dec[0] = (double)(w0 & 1ull); s_dec[1] = (double)((w0 >> 1) & 1ull);
              s_dec[2] = __longlong_as_double((long long)w1);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > GRID_SPIN_LIMIT) { __hip_atomic_store(&a.sync->error[0], 5ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_dec[3] = 5.0; break; }
            if ((spins & 0xFFF) == 0) {                      // a leader that gave up publishes nothing: leave with its error
              const unsigned long long ew = __hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (ew != 0) { s_dec[3] = (double)ew; break; }
            }
          }
        }
      }
    }
    __syncthreads();
    if (s_leader) {                                          // workgroup-uniform
      double res[5];
      // one wave per quantity (an LP has four).  The general form (final_reduce_body) was kept beside it as a dev
      // option through round 3: its per-thread index arithmetic is loop-invariant, was hoisted in front of the trial loop
      // and cost 60 of the kernel's spilled registers although it never ran.
      final_reduce_lp<PDHG_STEPS_LP_BATCH>(a.ctl->sp, res, s_res8);
      if (threadIdx.x == 0) {
        res[4] *= 0.5;
#if PDHG_STEPS_PREFETCH
        const StepRule rule = adaptive_step_rule(res, a.primal_weight, s_st[0], s_pow[0], s_pow[1]);
#else
        const StepRule rule = adaptive_step_rule(res, a.primal_weight, s_st[0], a.pow_red[trials], a.pow_growth[trials]);
#endif
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(a.ctl->slot[xcc_id()]);
        const unsigned long long w0 = (epoch << 2) | (rule.accept ? 1ull : 0ull) | (rule.numerical_error ? 2ull : 0ull);
        const unsigned long long w1 = (unsigned long long)__double_as_longlong(rule.next_step);
        __hip_atomic_store(slot + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 2, w0 ^ w1 ^ RESULT_CHECK_SALT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_dec[0] = (double)rule.accept; s_dec[1] = (double)rule.numerical_error; s_dec[2] = rule.next_step;
      }
    }
    if (threadIdx.x == 0) asm volatile("s_dcache_inv" ::: "memory");
    __syncthreads();
    PDHG_STAMP(6);
#undef PDHG_STAMP
    const int bad = __builtin_amdgcn_readfirstlane((int)(s_dec[3] != 0.0));
    const int nerr = __builtin_amdgcn_readfirstlane((int)(s_dec[1] != 0.0));
    const int acc = __builtin_amdgcn_readfirstlane((int)(s_dec[0] != 0.0));
    if (bad) { hw_err = 1; break; }                          // a barrier timed out: this trial does not count (the host repeats it)
    trials += 1;
    if (nerr) { num_err = 1; mid = 0; break; }               // movement == 0: no accept, the step size stays (pdhg.jl:692-697)
    if (acc) { flip ^= 1; pend = 1; steps += 1; }
    mid = !acc;
    if (threadIdx.x == 0) {
      const double next = s_dec[2];
      if (acc) {
        const double entry = s_st[1];
        s_st[2] = entry;                                     // update_solution_in_solver_state: weight = step size on entry (pdhg.jl:512)
        s_st[3] = s_st[3] + entry; s_st[4] = s_st[4] + entry;
        s_st[1] = next;
      }
      s_st[0] = next;
    }
    __syncthreads();                                         // the state is read, s_dec / s_leader rewritten, in the next trial
  }
  if (w == 0 && threadIdx.x == 0) {
    // results, checksum, sequence number: the host accepts a read when both match (as for the single-trial kernel)
    unsigned long long ck = RESULT_CHECK_SALT;
    int k = 0;
#define PDHG_PUB(v) do { const double pv = (v); a.res_host[k] = pv; ck ^= (unsigned long long)__double_as_longlong(pv) * (2ull * k + 1ull); ++k; } while (0)
    PDHG_PUB(s_st[0]); PDHG_PUB((double)steps); PDHG_PUB((double)trials); PDHG_PUB((double)flip); PDHG_PUB((double)pend);
    PDHG_PUB(s_st[2]); PDHG_PUB(s_st[3]); PDHG_PUB(s_st[4]); PDHG_PUB((double)num_err); PDHG_PUB((double)hw_err);
    PDHG_PUB((double)epoch);
    PDHG_PUB((double)__hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    PDHG_PUB((double)a.seq);
#undef PDHG_PUB
    const double e14 = mid ? s_st[1] : 0.0;   // ended inside a take_step (table exhausted): its step size on entry, for the host to finish it
    ck ^= (unsigned long long)__double_as_longlong(e14) * 29ull;     // (word 14 is under the checksum too: steps_wait)
    a.res_host[14] = e14;
    a.res_host[13] = __longlong_as_double((long long)ck);
    a.res_host[15] = (double)a.seq;
  }
}

}  // namespace
