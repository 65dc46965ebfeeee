// dist.hpp -- part of the single translation unit pdhg_hip.hip (included there, after struct pdhg_handle).
// The row-partitioned multi-GPU form, owned by the library: shard group, the
// collective back ends (RCCL over xGMI; direct peer kernels inside one process)
// and the row partition itself.
//
// Form ("reduce-scatter -> slice -> all-gather"): rank p holds the
// row block A_p (CSR for A_p*xbar, CSR of A_p' for A_p'*y), its rows of y, b,
// sum_y, and OWNS the column slice [p*S, (p+1)*S) of every n-vector: x, x', c,
// lb, ub, A'y, A'y', sum_x are updated on that slice only.  Per trial step:
//   primal step on the slice -> all-gather xbar -> y'_p = prox(A_p xbar)
//   -> t_p = A_p' y'_p (length n) -> reduce-scatter(sum) t_p: the slice of A'y'
//   -> interaction/movement partial sums on the slice -> the 5 scalars of every
//   rank are gathered and added IN RANK ORDER on every rank, so all ranks take
//   bitwise identical accept/reject decisions whatever the collective algorithm.
// Communication volume equals the all-reduce form's (one RS + one AG of n
// doubles), vector work is 1/P per rank, and no rank needs a bitwise-consistent
// replica of A'y.
// The reference has no counterpart (single process); the arithmetic being
// distributed is src/primal_dual_hybrid_gradient.jl:442-549.
#pragma once

#include "rccl_loader.hpp"

namespace {

enum { COMM_RCCL = 0, COMM_P2P = 1 };
constexpr int DIST_MAX_WORLD = 64;     // scalar exchange buffers are sized for this
constexpr int P2P_MAX_WORLD = 16;      // peer-kernel back end (single process)
constexpr int SCAL_MAX = 32;           // scalars one shard contributes per reduction

// the run-time bound RCCL entry points (rccl_loader.hpp); 2999: RCCL unavailable / refused
#define RCCL_API(R)                                                            \
  const RcclApi *R = rccl();                                                   \
  if (!R) return 2999

#define NCCL_TRY(expr)                                                         \
  do {                                                                         \
    ncclResult_t _r = (expr);                                                  \
    if (_r != ncclSuccess) {                                                   \
      g_last_error = std::string(#expr) + ": " + rccl_loader().api.GetErrorString(_r); \
      return 2000 + (int)_r;                                                   \
    }                                                                          \
  } while (0)

// ---- host-side fan-out: one worker thread per local shard -------------------------------
// pdhg_create_multi hands one process several GPUs.  Issued from the calling thread alone, a
// trial is ~15 launches / event calls / collectives PER SHARD, one shard after the other:
// at 8 shards that is ~0.5 ms of host calls against ~0.33 ms of kernels on config S
// (DESIGN.md section 5).  With the pool every shard's sequence is issued by its own thread
// (thread 0 is the caller), so the host cost per trial is that of ONE shard.  The threads
// sleep on a condition variable between entry points and spin only briefly (the next trial
// usually follows within microseconds).  PDHG_SHARD_THREADS=0 keeps the single-thread issue.
struct ShardPool {
  int n = 0;
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> pending{0};
  std::atomic<bool> stop{false}, abort_job{false};
  std::function<int(int)> job;
  std::vector<int> rc;
  std::vector<std::string> err;
  // sense-reversing barrier among the n issuing threads (peer back end: "all events recorded")
  std::atomic<int> bar_arrived{0};
  std::atomic<uint64_t> bar_gen{0};

  explicit ShardPool(int n_) : n(n_), rc((size_t)n_, 0), err((size_t)n_) {
    for (int i = 1; i < n; ++i) th.emplace_back([this, i] { worker(i); });
  }
  ~ShardPool() {
    { std::lock_guard<std::mutex> lk(mu); stop.store(true); }
    cv.notify_all();
    for (std::thread &t : th) t.join();
  }
  void run_one(int i) {
    g_last_error.clear();
    const int r = job(i);
    rc[(size_t)i] = r;
    if (r) { err[(size_t)i] = g_last_error; abort_job.store(true); }
    pending.fetch_sub(1, std::memory_order_acq_rel);
  }
  void worker(int i) {
    uint64_t seen = 0;
    for (;;) {
      // short spin (back-to-back trials), then sleep
      bool got = false;
      for (int spin = 0; spin < 20000; ++spin) {
        if (gen.load(std::memory_order_acquire) != seen || stop.load(std::memory_order_relaxed)) { got = true; break; }
        __builtin_ia32_pause();
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen || stop.load(); });
      }
      if (stop.load()) return;
      seen = gen.load(std::memory_order_acquire);
      run_one(i);
    }
  }
  // f(i) on n threads at once (i = 0 on the caller); returns the first nonzero code in shard order
  int run(std::function<int(int)> f) {
    job = std::move(f);
    abort_job.store(false);
    bar_arrived.store(0);        // a failed job may have left arrivals behind
    pending.store(n, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(mu); gen.fetch_add(1, std::memory_order_acq_rel); }
    cv.notify_all();
    run_one(0);
    while (pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    for (int i = 0; i < n; ++i)
      if (rc[(size_t)i]) { g_last_error = err[(size_t)i]; return rc[(size_t)i]; }
    return 0;
  }
  // all n issuing threads meet; nonzero if a sibling failed (nobody may wait for it for ever)
  int barrier() {
    const uint64_t my = bar_gen.load(std::memory_order_acquire);
    if (bar_arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      bar_arrived.store(0, std::memory_order_relaxed);
      bar_gen.fetch_add(1, std::memory_order_acq_rel);
      return 0;
    }
    while (bar_gen.load(std::memory_order_acquire) == my) {
      if (abort_job.load(std::memory_order_relaxed)) return fail(997, "a sibling shard failed while issuing");
      __builtin_ia32_pause();
    }
    return 0;
  }
};

struct GroupTrialArgs;
struct GridSync;
// one device's share of a group trial taken as persistent kernels (group_kernel.hpp)
struct GroupDevLaunch {
  int device = 0;
  std::vector<int> members;             // indices into DistGroup::sh, ascending rank
  std::vector<int> base;                // first workgroup of every member (+ the total)
  int grid = 0;
  hipStream_t stream = nullptr;         // the first member's stream: the launch runs here
  GroupTrialArgs *args_dev = nullptr, *args_host = nullptr;
  GridSync **sync_dev = nullptr;
  std::vector<hipEvent_t> ev;           // "the member's own stream is drained"
  hipEvent_t ev_done = nullptr;         // "the launch is queued" (the members' streams wait for it)
};

struct DistGroup {
  ShardPool *pool = nullptr;            // several local shards: one issuing thread each (trial steps)
  // host-side cost of the trials issued so far: seconds until the last launch call returned
  // (max over the issuing threads), seconds from then until the scalars were on the host
  double t_issue = 0.0, t_wait = 0.0;
  int64_t n_trials = 0;
  int world = 1;
  int backend = COMM_RCCL;
  std::vector<pdhg_handle *> sh;        // local shards, ascending rank (multi-process: one)
  std::vector<ncclComm_t> comm;         // per local shard (COMM_RCCL)
  int64_t n = 0, m_global = 0, num_eq_global = 0;
  int64_t S = 0;                        // slice stride: rank r owns columns [r*S, min(n, (r+1)*S))
  std::vector<int64_t> row_lo;          // [world+1] global row range of every rank
  std::vector<hipEvent_t> ev[2];        // cross-stream barrier of the peer back end
  int flip = 0;
  // Exchange pattern of the trial's A_p'y_p: one reduce-scatter after the product, or P
  // per-slice reductions issued as the product's slices complete (on the comm streams,
  // overlapping the rest of the product).  Decided from (n, world, environment) only, so
  // that every rank issues the same sequence of collectives.
  bool overlap = false;
  // The all-gather of xbar overlapped with A_p xbar (round 6; SURVEY 8e(ii), pdhg.jl:472-494).  On the fully connected node
  // every slice has a link of its own and all of them arrive together, so the product can only run beside the transfer
  // if it consumes PARTS of every slice: the column space is cut into `ag_chunks` chunks -- chunk c = sub-range c (ag_sub
  // columns) of EVERY rank's slice -- xbar travels chunk by chunk (one ncclAllGather per chunk on the comm streams, into a
  // chunk-major copy of xbar: pdhg_handle::xchunk) and A_p xbar runs as one pass per chunk (a complete layout of A_p restricted to the chunk's columns, the row
  // sums carried through memory), pass c waiting only for chunk c.  0 / 1: off (one all-gather, one product).
  // ag_mode 1: as described; 2: the same passes behind ONE all-gather (nothing overlapped: the reference the overlapped
  // form must equal bit for bit, and what the peer back end inside one process runs).  A row's products are added chunk
  // by chunk, ascending columns inside a chunk: not the single pass's order (rows within 1e-13 * sum |a x| of it), the
  // same order in both modes and on every back end.  Decided from the environment only (PDHG_DIST_AG_OVERLAP, default
  // off; PDHG_DIST_AG_CHUNKS, default 4): every rank makes the same sequence of collectives.
  int ag_chunks = 0, ag_mode = 0;
  int64_t ag_sub = 0;
  // true: every rank lives in this process (scalars and vectors can be collected shard by
  // shard).  false: one rank per process -- everything other ranks hold arrives through RCCL.
  // PDHG_DIST_FORCE_REMOTE=1 takes the second route even when all ranks are local, so that
  // its collectives run (and are tested) on a 1-GPU box.
  bool force_remote = false;
  // (force_remote is honoured for a single local shard only: with several local RCCL shards the
  // one-rank-per-process route would issue a collective on one of N communicators and hang)
  // one persistent kernel per device and trial with cross-shard barriers inside (group_kernel.hpp)
  int coop_mode = -1;                   // -1 undecided, 0 off, 1 on
  struct GroupSync *gsync = nullptr;    // shared by the shards' kernels
  std::vector<GroupDevLaunch> coop_dev; // one launch per device: its shards, their places in the grid, staging for the arguments
  // The launches run on the first member's stream.  Ordering against the members' own streams is settled lazily: trials and
  // (lazy) accepts queue nothing there, so only the first trial after any OTHER entry point waits for the members' streams
  // (members_dirty), and only the first other entry point after a trial makes them wait for the launch (join_pending).
  bool members_dirty = true, join_pending = false;
  unsigned long long xepoch = 0;        // cross-shard barriers passed so far
  int coop_fallbacks = 0;
  int64_t coop_trials = 0;              // trials taken that way
  bool all_local() const { return (int)sh.size() == world && !(force_remote && backend == COMM_RCCL && sh.size() == 1); }
};

// frees what group_coop_prepare (pdhg_hip.hip) set up for the persistent group launches
inline void group_coop_release(DistGroup &g) {
  for (GroupDevLaunch &D : g.coop_dev) {
    (void)hipSetDevice(D.device);
    if (D.args_dev) (void)hipFree(D.args_dev);
    if (D.args_host) (void)hipHostFree(D.args_host);
    if (D.sync_dev) (void)hipFree(D.sync_dev);
    for (hipEvent_t e : D.ev) if (e) (void)hipEventDestroy(e);
    if (D.ev_done) (void)hipEventDestroy(D.ev_done);
  }
  g.coop_dev.clear();
}


// ---- the shard list every entry point walks: the group's local shards, or the handle itself
struct Shards {
  pdhg_handle *const *p;
  int count;
  DistGroup *g;
};
inline Shards shards_of(pdhg_handle *h) {
  if (h->grp) return Shards{h->grp->sh.data(), (int)h->grp->sh.size(), h->grp};
  return Shards{&h->self, 1, nullptr};
}
// Walks the local shards with the shard's device current.  A failing hipSetDevice is an
// ERROR of the enclosing entry point (every user returns int): skipping the shard instead
// would leave its kernel, accept or flush unissued and the call would still report success.
// (the macro ends in `else`, so that the loop body is the else-branch: intended)
#pragma clang diagnostic ignored "-Wdangling-else"
#define FOR_SHARDS(L, s)                                                       \
  for (int _si = 0; _si < (L).count; ++_si)                                     \
    if (pdhg_handle *s = (L).p[_si])                                            \
      if (hipError_t _sde = hipSetDevice(s->device); _sde != hipSuccess)        \
        return fail_hip(_sde, "hipSetDevice (shard loop)");                     \
      else

// ---- peer back end: kernels that read the other shards' buffers directly
struct PeerPtrs {
  const double *p[P2P_MAX_WORLD];
  int world;
};
// out[i] = op over ranks q = 0..world-1 (ascending: deterministic) of p[q][off + i]
template <bool MAXOP>
__global__ __launch_bounds__(TPB) void p2p_reduce_kernel(PeerPtrs pp, int64_t off, int64_t count, double *out) {
  const int64_t stride = (int64_t)gridDim.x * TPB;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < count; i += stride) {
    double v = pp.p[0][off + i];
    for (int q = 1; q < pp.world; ++q) {
      const double t = pp.p[q][off + i];
      v = MAXOP ? fmax(v, t) : v + t;
    }
    out[i] = v;
  }
}

// every local stream waits for everything queued so far on every other local stream.
// Up to 4 shards: all pairs (one hop, P^2 calls).  More: the first stream waits for all
// the others and records "go", the others wait for "go" (two hops, 3P calls) -- measured
// on 2 / 8 shards of one GPU: the hub form is 20 % slower at 2 and 6 % faster at 8.
int p2p_barrier(DistGroup &g) {
  const int k = (int)g.sh.size();
  if (k <= 1) return 0;
  g.flip ^= 1;
  std::vector<hipEvent_t> &ev = g.ev[g.flip];
  if (k <= 4) {
    for (int i = 0; i < k; ++i) {
      HIP_TRY(hipSetDevice(g.sh[i]->device));
      HIP_TRY(hipEventRecord(ev[i], g.sh[i]->stream));
    }
    for (int i = 0; i < k; ++i) {
      HIP_TRY(hipSetDevice(g.sh[i]->device));
      for (int q = 0; q < k; ++q)
        if (q != i) HIP_TRY(hipStreamWaitEvent(g.sh[i]->stream, ev[q], 0));
    }
    return 0;
  }
  for (int i = 1; i < k; ++i) {
    HIP_TRY(hipSetDevice(g.sh[i]->device));
    HIP_TRY(hipEventRecord(ev[i], g.sh[i]->stream));
  }
  HIP_TRY(hipSetDevice(g.sh[0]->device));
  for (int q = 1; q < k; ++q) HIP_TRY(hipStreamWaitEvent(g.sh[0]->stream, ev[q], 0));
  HIP_TRY(hipEventRecord(ev[0], g.sh[0]->stream));          // "go": everything queued anywhere so far is done
  for (int i = 1; i < k; ++i) {
    HIP_TRY(hipSetDevice(g.sh[i]->device));
    HIP_TRY(hipStreamWaitEvent(g.sh[i]->stream, ev[0], 0));
  }
  return 0;
}

// own[q*S .. q*S+S) = peer q's slice, for every q != rank: the all-gather's pull as ONE kernel
__global__ __launch_bounds__(TPB) void p2p_gather_kernel(PeerPtrs pp, int rank, int64_t S, double *own) {
  const int64_t stride = (int64_t)gridDim.x * TPB;
  for (int q = 0; q < pp.world; ++q) {
    if (q == rank) continue;
    const double *src = pp.p[q] + (int64_t)q * S;
    double *dst = own + (int64_t)q * S;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < S; i += stride) dst[i] = src[i];
  }
}

// A buffer selector: the same logical vector on every shard.
typedef double *(*BufSel)(pdhg_handle *);

// In-place all-gather: rank r contributes buf[r*S .. r*S+S), every rank ends with all of them.
// The buffers hold world*S doubles.
template <typename Sel>
int dist_all_gather(DistGroup &g, Sel sel, int64_t S) {
  if (g.world == 1 && g.backend == COMM_P2P) return 0;
  if (g.backend == COMM_RCCL) {
    RCCL_API(R);
    NCCL_TRY(R->GroupStart());
    for (size_t i = 0; i < g.sh.size(); ++i) {
      pdhg_handle *s = g.sh[i];
      HIP_TRY(hipSetDevice(s->device));
      double *b = sel(s);
      NCCL_TRY(R->AllGather(b + (int64_t)s->rank * S, b, (size_t)S, ncclDouble, g.comm[i], s->stream));
    }
    NCCL_TRY(R->GroupEnd());
    return 0;
  }
  int rc;
  if ((rc = p2p_barrier(g))) return rc;
  PeerPtrs pp{};
  pp.world = g.world;
  for (pdhg_handle *q : g.sh) pp.p[q->rank] = sel(q);
  for (pdhg_handle *s : g.sh) {
    HIP_TRY(hipSetDevice(s->device));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((S + TPB - 1) / TPB, EW_MAX_BLOCKS));
    hipLaunchKernelGGL(p2p_gather_kernel, dim3(grid), dim3(TPB), 0, s->stream, pp, s->rank, S, sel(s));
    HIP_TRY(hipGetLastError());
  }
  return p2p_barrier(g);
}

// In-place reduce-scatter: rank r ends with op_q buf_q[r*S .. r*S+S) in ITS buf[r*S ..).
template <typename Sel>
int dist_reduce_scatter(DistGroup &g, Sel sel, int64_t S, bool maxop = false) {
  if (g.world == 1 && g.backend == COMM_P2P) return 0;
  if (g.backend == COMM_RCCL) {
    RCCL_API(R);
    NCCL_TRY(R->GroupStart());
    for (size_t i = 0; i < g.sh.size(); ++i) {
      pdhg_handle *s = g.sh[i];
      HIP_TRY(hipSetDevice(s->device));
      double *b = sel(s);
      NCCL_TRY(R->ReduceScatter(b, b + (int64_t)s->rank * S, (size_t)S, ncclDouble, maxop ? ncclMax : ncclSum,
                                 g.comm[i], s->stream));
    }
    NCCL_TRY(R->GroupEnd());
    return 0;
  }
  int rc;
  if ((rc = p2p_barrier(g))) return rc;
  PeerPtrs pp{};
  pp.world = g.world;
  for (pdhg_handle *q : g.sh) pp.p[q->rank] = sel(q);
  for (pdhg_handle *s : g.sh) {
    HIP_TRY(hipSetDevice(s->device));
    const int64_t off = (int64_t)s->rank * S;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((S + TPB - 1) / TPB, EW_MAX_BLOCKS));
    if (maxop) hipLaunchKernelGGL(p2p_reduce_kernel<true>, dim3(grid), dim3(TPB), 0, s->stream, pp, off, S, sel(s) + off);
    else hipLaunchKernelGGL(p2p_reduce_kernel<false>, dim3(grid), dim3(TPB), 0, s->stream, pp, off, S, sel(s) + off);
    HIP_TRY(hipGetLastError());
  }
  return p2p_barrier(g);
}

// Slice k of the partial vectors (S doubles at k*S) summed over ranks into its owner's
// buffer, asynchronously on the comm streams, once every shard has recorded ev_part[k]
// on its compute stream.  dist_join_comm() makes the compute streams wait for all of them.
template <typename Sel>
int dist_reduce_slice_async(DistGroup &g, Sel sel, int64_t S, int k) {
  if (g.backend == COMM_RCCL) {
    RCCL_API(R);
    for (pdhg_handle *s : g.sh) {
      HIP_TRY(hipSetDevice(s->device));
      HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->ev_part[(size_t)k], 0));
    }
    NCCL_TRY(R->GroupStart());
    for (size_t i = 0; i < g.sh.size(); ++i) {
      pdhg_handle *s = g.sh[i];
      HIP_TRY(hipSetDevice(s->device));
      double *b = sel(s) + (int64_t)k * S;
      NCCL_TRY(R->Reduce(b, b, (size_t)S, ncclDouble, ncclSum, k, g.comm[i], s->comm_stream));
    }
    NCCL_TRY(R->GroupEnd());
    return 0;
  }
  pdhg_handle *owner = nullptr;
  PeerPtrs pp{};
  pp.world = g.world;
  for (pdhg_handle *q : g.sh) { pp.p[q->rank] = sel(q); if (q->rank == k) owner = q; }
  if (!owner) return fail(-1, "peer back end needs every rank in this process");
  HIP_TRY(hipSetDevice(owner->device));
  for (pdhg_handle *q : g.sh) HIP_TRY(hipStreamWaitEvent(owner->comm_stream, q->ev_part[(size_t)k], 0));
  const int64_t off = (int64_t)k * S;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((S + TPB - 1) / TPB, EW_MAX_BLOCKS));
  hipLaunchKernelGGL(p2p_reduce_kernel<false>, dim3(grid), dim3(TPB), 0, owner->comm_stream, pp, off, S, sel(owner) + off);
  HIP_TRY(hipGetLastError());
  return 0;
}

int dist_join_comm(DistGroup &g) {
  for (pdhg_handle *s : g.sh) {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventRecord(s->ev_comm, s->comm_stream));
  }
  for (pdhg_handle *s : g.sh) {
    HIP_TRY(hipSetDevice(s->device));
    if (g.backend == COMM_RCCL) {
      HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_comm, 0));
    } else {
      // peer kernels read the OTHER shards' buffers: nobody may reuse its partial vector
      // before every owner has finished reading
      for (pdhg_handle *q : g.sh) HIP_TRY(hipStreamWaitEvent(s->stream, q->ev_comm, 0));
    }
  }
  return 0;
}

// ---- the all-gather of xbar, chunk by chunk (DistGroup::ag_chunks) -------------------------------------------------------
// CHUNK layout of an n-vector: chunk c = [world][ag_sub] doubles, rank q's part being columns q*S + c*ag_sub ... of the
// natural vector (its sub-range c).  One ncclAllGather per chunk fills it in place -- every rank contributes ag_sub doubles at
// its own offset -- so a chunk is ONE standard collective (not a broadcast per rank), and the chunk layouts of A_p
// (host_shards.hpp: build_column_chunks) carry column indices into their chunk.
// rows of `src` [q*S, q*S + S) for q in [q0, q1) into the chunk layout `dst` ([C][world * sub])
__global__ __launch_bounds__(TPB) void chunk_pack_kernel(const double *__restrict__ src, double *__restrict__ dst, int64_t S, int64_t sub,
                                                         int C, int world, int q0, int q1, int64_t src_len) {
  const int64_t stride = (int64_t)gridDim.x * TPB, W = (int64_t)world * sub;
  for (int q = q0; q < q1; ++q)
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < S; i += stride) {
      const int64_t j = (int64_t)q * S + i;
      const int64_t c = i / sub, off = i - c * sub;
      dst[c * W + (int64_t)q * sub + off] = j < src_len ? src[j] : 0.0;
    }
}
inline int launch_chunk_pack(DistGroup &g, pdhg_handle *s, const double *src, double *dst, int q0, int q1, int64_t src_len, hipStream_t stream) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((g.S + TPB - 1) / TPB, EW_MAX_BLOCKS));
  hipLaunchKernelGGL(chunk_pack_kernel, dim3(grid), dim3(TPB), 0, stream, src, dst, g.S, g.ag_sub, g.ag_chunks, g.world, q0, q1, src_len);
  HIP_TRY(hipGetLastError());
  (void)s;
  return 0;
}
// Chunk c of every local shard's xchunk, on the COMM streams (which first wait for "the owned slice is packed": ev_xbar,
// recorded by the caller on the compute streams); ev_ag[c] is recorded behind it.
int dist_all_gather_chunk(DistGroup &g, int c) {
  RCCL_API(R);
  const int64_t W = (int64_t)g.world * g.ag_sub;
  if (c == 0)
    for (pdhg_handle *s : g.sh) {
      HIP_TRY(hipSetDevice(s->device));
      HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->ev_xbar, 0));
    }
  NCCL_TRY(R->GroupStart());
  for (size_t i = 0; i < g.sh.size(); ++i) {
    pdhg_handle *s = g.sh[i];
    HIP_TRY(hipSetDevice(s->device));
    double *b = s->xchunk + (int64_t)c * W;
    NCCL_TRY(R->AllGather(b + (int64_t)s->rank * g.ag_sub, b, (size_t)g.ag_sub, ncclDouble, g.comm[i], s->comm_stream));
  }
  NCCL_TRY(R->GroupEnd());
  for (pdhg_handle *s : g.sh) {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventRecord(s->ev_ag[(size_t)c], s->comm_stream));
  }
  return 0;
}
// the same for ONE shard, issued by its own host thread
int mt_all_gather_chunk(DistGroup &g, pdhg_handle *s, int i, int c) {
  RCCL_API(R);
  const int64_t W = (int64_t)g.world * g.ag_sub;
  if (c == 0) HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->ev_xbar, 0));
  double *b = s->xchunk + (int64_t)c * W;
  NCCL_TRY(R->AllGather(b + (int64_t)s->rank * g.ag_sub, b, (size_t)g.ag_sub, ncclDouble, g.comm[(size_t)i], s->comm_stream));
  HIP_TRY(hipEventRecord(s->ev_ag[(size_t)c], s->comm_stream));
  return 0;
}

// ---- the same collectives issued PER SHARD, each by its own host thread (ShardPool) -------
// `i` is the shard's index in g.sh, `s` the shard; every local shard's thread makes the same
// sequence of calls.  RCCL: plain per-communicator calls (one thread per device needs no
// group).  Peer back end: the cross-stream barrier becomes "record my event, meet the other
// threads, wait for their events".

int mt_stream_barrier(DistGroup &g, pdhg_handle *s, int i) {
  const int k = (int)g.sh.size();
  if (k <= 1) return 0;
  s->mt_flip ^= 1;
  std::vector<hipEvent_t> &ev = g.ev[s->mt_flip];
  HIP_TRY(hipEventRecord(ev[(size_t)i], s->stream));
  int rc = g.pool->barrier();
  if (rc) return rc;
  for (int q = 0; q < k; ++q)
    if (q != i) HIP_TRY(hipStreamWaitEvent(s->stream, ev[(size_t)q], 0));
  return 0;
}

inline PeerPtrs peer_ptrs(DistGroup &g, BufSel sel) {
  PeerPtrs pp{};
  pp.world = g.world;
  for (pdhg_handle *q : g.sh) pp.p[q->rank] = sel(q);
  return pp;
}

int mt_all_gather(DistGroup &g, pdhg_handle *s, int i, BufSel sel, int64_t S) {
  if (g.world == 1 && g.backend == COMM_P2P) return 0;
  if (g.backend == COMM_RCCL) {
    RCCL_API(R);
    double *b = sel(s);
    NCCL_TRY(R->AllGather(b + (int64_t)s->rank * S, b, (size_t)S, ncclDouble, g.comm[(size_t)i], s->stream));
    return 0;
  }
  int rc;
  if ((rc = mt_stream_barrier(g, s, i))) return rc;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((S + TPB - 1) / TPB, EW_MAX_BLOCKS));
  hipLaunchKernelGGL(p2p_gather_kernel, dim3(grid), dim3(TPB), 0, s->stream, peer_ptrs(g, sel), s->rank, S, sel(s));
  HIP_TRY(hipGetLastError());
  return mt_stream_barrier(g, s, i);
}

int mt_reduce_scatter(DistGroup &g, pdhg_handle *s, int i, BufSel sel, int64_t S) {
  if (g.world == 1 && g.backend == COMM_P2P) return 0;
  if (g.backend == COMM_RCCL) {
    RCCL_API(R);
    double *b = sel(s);
    NCCL_TRY(R->ReduceScatter(b, b + (int64_t)s->rank * S, (size_t)S, ncclDouble, ncclSum, g.comm[(size_t)i], s->stream));
    return 0;
  }
  int rc;
  if ((rc = mt_stream_barrier(g, s, i))) return rc;
  const int64_t off = (int64_t)s->rank * S;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((S + TPB - 1) / TPB, EW_MAX_BLOCKS));
  hipLaunchKernelGGL(p2p_reduce_kernel<false>, dim3(grid), dim3(TPB), 0, s->stream, peer_ptrs(g, sel), off, S, sel(s) + off);
  HIP_TRY(hipGetLastError());
  return mt_stream_barrier(g, s, i);
}

// slice k of the partial vectors to its owner, on the comm streams, after every shard has
// recorded ev_part[k] on its compute stream (the caller did so for `s` just before)
int mt_reduce_slice_async(DistGroup &g, pdhg_handle *s, int i, BufSel sel, int64_t S, int k) {
  if (g.backend == COMM_RCCL) {
    RCCL_API(R);
    HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->ev_part[(size_t)k], 0));
    double *b = sel(s) + (int64_t)k * S;
    NCCL_TRY(R->Reduce(b, b, (size_t)S, ncclDouble, ncclSum, k, g.comm[(size_t)i], s->comm_stream));
    return 0;
  }
  int rc = g.pool->barrier();            // everybody's ev_part[k] is recorded
  if (rc) return rc;
  if (s->rank != k) return 0;
  for (pdhg_handle *q : g.sh) HIP_TRY(hipStreamWaitEvent(s->comm_stream, q->ev_part[(size_t)k], 0));
  const int64_t off = (int64_t)k * S;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((S + TPB - 1) / TPB, EW_MAX_BLOCKS));
  hipLaunchKernelGGL(p2p_reduce_kernel<false>, dim3(grid), dim3(TPB), 0, s->comm_stream, peer_ptrs(g, sel), off, S, sel(s) + off);
  HIP_TRY(hipGetLastError());
  return 0;
}

int mt_join_comm(DistGroup &g, pdhg_handle *s, int i) {
  (void)i;
  HIP_TRY(hipEventRecord(s->ev_comm, s->comm_stream));
  if (g.backend == COMM_RCCL) {
    HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_comm, 0));
    return 0;
  }
  int rc = g.pool->barrier();            // every owner's ev_comm is recorded
  if (rc) return rc;
  for (pdhg_handle *q : g.sh) HIP_TRY(hipStreamWaitEvent(s->stream, q->ev_comm, 0));
  return 0;
}

// All-reduce as reduce-scatter + all-gather: every element is reduced once, at
// its owner, and then copied -- all ranks hold the same bits whatever the order.
template <typename Sel>
int dist_all_reduce(DistGroup &g, Sel sel, int64_t S, bool maxop = false) {
  int rc = dist_reduce_scatter(g, sel, S, maxop);
  if (rc) return rc;
  return dist_all_gather(g, sel, S);
}

// In-place all-gather of row-partitioned m-vectors: rank r contributes
// buf[row_lo[r] .. row_lo[r+1]).  The buffers hold m_global doubles.
template <typename Sel>
int dist_all_gather_rows(DistGroup &g, Sel sel) {
  if (g.world == 1 && g.backend == COMM_P2P) return 0;
  if (g.backend == COMM_RCCL) {
    RCCL_API(R);
    NCCL_TRY(R->GroupStart());
    for (size_t i = 0; i < g.sh.size(); ++i) {
      pdhg_handle *s = g.sh[i];
      HIP_TRY(hipSetDevice(s->device));
      double *b = sel(s);
      for (int q = 0; q < g.world; ++q) {
        const int64_t cnt = g.row_lo[q + 1] - g.row_lo[q];
        if (cnt > 0)
          NCCL_TRY(R->Broadcast(b + g.row_lo[q], b + g.row_lo[q], (size_t)cnt, ncclDouble, q, g.comm[i], s->stream));
      }
    }
    NCCL_TRY(R->GroupEnd());
    return 0;
  }
  int rc;
  if ((rc = p2p_barrier(g))) return rc;
  for (pdhg_handle *s : g.sh) {
    HIP_TRY(hipSetDevice(s->device));
    for (pdhg_handle *q : g.sh) {
      const int64_t cnt = g.row_lo[q->rank + 1] - g.row_lo[q->rank];
      if (q != s && cnt > 0)
        HIP_TRY(hipMemcpyAsync(sel(s) + g.row_lo[q->rank], sel(q) + g.row_lo[q->rank], sizeof(double) * (size_t)cnt,
                               hipMemcpyDeviceToDevice, s->stream));
    }
  }
  return p2p_barrier(g);
}

// The k scalars every shard left in its scal_dev, combined over ALL ranks in
// rank order on the host: entries [0, nsum) are added, [nsum, k) are maxed.
// Every rank computes the same bits.  Synchronises the streams.
int combine_scalars(const Shards &L, int k, int nsum, double *out) {
  if (k > SCAL_MAX) return fail(-1, "too many scalars in one reduction");
  DistGroup *g = L.g;
  if (!g || g->all_local()) {
    for (int i = 0; i < L.count; ++i) {
      pdhg_handle *s = L.p[i];
      HIP_TRY(hipSetDevice(s->device));
      HIP_TRY(hipMemcpyAsync(s->scal_host, s->scal_dev, sizeof(double) * k, hipMemcpyDeviceToHost, s->stream));
    }
    for (int i = 0; i < L.count; ++i) {
      pdhg_handle *s = L.p[i];
      HIP_TRY(hipSetDevice(s->device));
      HIP_TRY(hipStreamSynchronize(s->stream));
    }
    for (int q = 0; q < k; ++q) {
      double v = L.p[0]->scal_host[q];
      for (int i = 1; i < L.count; ++i) {
        const double t = L.p[i]->scal_host[q];
        v = (q < nsum) ? v + t : std::fmax(v, t);
      }
      out[q] = v;
    }
    return 0;
  }
  // one local shard per process: gather everybody's scalars through RCCL
  pdhg_handle *s = L.p[0];
  HIP_TRY(hipSetDevice(s->device));
  RCCL_API(R);
  NCCL_TRY(R->AllGather(s->scal_dev, s->scal_all, (size_t)SCAL_MAX, ncclDouble, g->comm[0], s->stream));
  HIP_TRY(hipMemcpyAsync(s->scal_host, s->scal_all, sizeof(double) * SCAL_MAX * g->world, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (int q = 0; q < k; ++q) {
    double v = s->scal_host[q];
    for (int r = 1; r < g->world; ++r) {
      const double t = s->scal_host[(size_t)r * SCAL_MAX + q];
      v = (q < nsum) ? v + t : std::fmax(v, t);
    }
    out[q] = v;
  }
  return 0;
}

// ---- row partition (host) -----------------------------------------------------

// Contiguous row ranges balanced by nonzeros; equalities-first order is kept
// because the ranges are contiguous.  bounds[world+1].
// prefix[r] = nonzeros in rows [0, r)
// Returns the number of entries whose row index lies outside [base, m + base) (they are not counted).
int64_t row_nnz_prefix(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int base,
                       std::vector<int64_t> &prefix) {
  const int64_t nnz = colptr[n] - base;
  prefix.assign((size_t)m + 1, 0);
  // integer counts: host threads over entry ranges with relaxed atomic increments give the same
  // numbers as the serial loop (2.2 G entries: ~5 s serial)
  int64_t *cnt = prefix.data();
  int64_t bad = 0;
  const int64_t grain = 1 << 22;
  const int parts = (int)std::min<int64_t>(1 << 20, std::max<int64_t>(1, (nnz + grain - 1) / grain));
  parallel_ranges(parts, 1, [&](int pb, int pe) {
    const int64_t kb = nnz * pb / parts, ke = nnz * pe / parts;
    for (int64_t k = kb; k < ke; ++k) {
      const int64_t r = rowval[k] - base;
      if (r >= 0 && r < m) __atomic_fetch_add(&cnt[r + 1], (int64_t)1, __ATOMIC_RELAXED);
      else __atomic_fetch_add(&bad, (int64_t)1, __ATOMIC_RELAXED);
    }
  });
  for (int64_t i = 0; i < m; ++i) prefix[(size_t)i + 1] += prefix[(size_t)i];
  return bad;
}

void partition_rows_from_prefix(const std::vector<int64_t> &prefix, int world, std::vector<int64_t> &bounds) {
  const int64_t m = (int64_t)prefix.size() - 1, nnz = prefix[(size_t)m];
  bounds.assign((size_t)world + 1, 0);
  for (int p = 1; p < world; ++p) {
    // first row index r with prefix[r] >= nnz*p/world (exact rational compare)
    const __int128 target_num = (__int128)nnz * p;     // target = target_num / world
    int64_t lo = 0, hi = m;                            // search over prefix[0..m]
    while (lo < hi) {
      const int64_t mid = (lo + hi) / 2;
      if ((__int128)prefix[(size_t)mid] * world < target_num) lo = mid + 1; else hi = mid;
    }
    bounds[(size_t)p] = std::min<int64_t>(std::max<int64_t>(lo, bounds[(size_t)p - 1]), m);
  }
  bounds[(size_t)world] = m;
}

void partition_rows_by_nnz(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int base, int world,
                           std::vector<int64_t> &bounds) {
  std::vector<int64_t> prefix;
  row_nnz_prefix(m, n, colptr, rowval, base, prefix);
  partition_rows_from_prefix(prefix, world, bounds);
}

// CSC of rows [lo, hi) of a CSC matrix (row indices rebased to 0, 0-based output).
void slice_csc_rows(int64_t n, const int64_t *colptr, const int64_t *rowval, const double *nzval, int base,
                    int64_t lo, int64_t hi, std::vector<int64_t> &cp, uvec<int64_t> &rv, dvec &nv) {
  cp.assign((size_t)n + 1, 0);
  parallel_ranges((int)std::min<int64_t>(n, INT32_MAX), 1 << 16, [&](int jb, int je) {
    for (int64_t j = jb; j < je; ++j) {
      int64_t cnt = 0;
      for (int64_t k = colptr[j] - base; k < colptr[j + 1] - base; ++k) {
        const int64_t r = rowval[k] - base;
        cnt += (r >= lo && r < hi);
      }
      cp[(size_t)j + 1] = cnt;
    }
  });
  for (int64_t j = 0; j < n; ++j) cp[(size_t)j + 1] += cp[(size_t)j];
  rv.resize((size_t)cp[(size_t)n]);
  nv.resize((size_t)cp[(size_t)n]);
  parallel_ranges((int)std::min<int64_t>(n, INT32_MAX), 1 << 16, [&](int jb, int je) {
    for (int64_t j = jb; j < je; ++j) {
      int64_t pos = cp[(size_t)j];
      for (int64_t k = colptr[j] - base; k < colptr[j + 1] - base; ++k) {
        const int64_t r = rowval[k] - base;
        if (r >= lo && r < hi) { rv[(size_t)pos] = r - lo; nv[(size_t)pos] = nzval[k]; ++pos; }
      }
    }
  });
}

}  // namespace
