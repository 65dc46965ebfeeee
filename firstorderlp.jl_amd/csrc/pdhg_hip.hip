// pdhg_hip.hip -- MI355X (gfx950 / CDNA4) PDHG inner step: kernels + C ABI.
//
// Implements include/pdhg_hip.h.  Written for gfx950 only: wave64, 256 CUs in
// 8 XCDs, 160 KiB LDS/CU, HBM3E.  The path is sparse fp64 and HBM-bound, so
// there is no MFMA here; what matters is coalesced streaming of the CSR
// arrays, LDS-staged products, wave-shuffle reductions and launch shapes that
// fill 256 CUs.
//
// Reference arithmetic being reproduced (paths relative to /root/reference/src):
//   primal step      primal_dual_hybrid_gradient.jl:442-470, saddle_point.jl:82-106,1093-1100
//   dual step        primal_dual_hybrid_gradient.jl:472-494, saddle_point.jl:110-117,1102-1107
//   interaction etc. primal_dual_hybrid_gradient.jl:527-549
//   accept/average   primal_dual_hybrid_gradient.jl:500-519, saddle_point.jl:252-301
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -shared -fPIC -pthread
// (-ffp-contract=off: elementwise updates must round like Julia's unfused
//  broadcasts; the product a*x and the sum are separate roundings).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "pdhg_hip.h"

// The kernels and layout builders live in the headers below; they form ONE
// translation unit with this file (the tiled kernel is sensitive to code
// placement, and one TU keeps every launch a direct call).
#include "common.hpp"
#include "spmv_kernels.hpp"
#include "sj_kernels.hpp"
#include "vector_kernels.hpp"
#include "eval_kernels.hpp"
#include "rescale_kernels.hpp"
#include "device_layout.hpp"
#include "layout.hpp"
#include "trial_kernel.hpp"
#include "small_lp_kernel.hpp"
#include "tr_coop_kernel.hpp"

namespace { struct DistGroup; }

struct pdhg_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int64_t m = 0, n = 0, nnz = 0, num_eq = 0;
  bool remap = true;
  // PDHG_ROW_ORDER=strict: every row sum strictly left to right (bit-exact with the CPU loops for rows
  // <= BLOCK_NNZ entries); default "relaxed": rows of more than 64 entries are summed wave-parallel in a
  // fixed order, within 1e-13 * sum |a x| of the sequential sum (spmv_kernels.hpp)
  bool relaxed = true;

  CsrDev A;    // m x n, rows = constraints   (K3)
  CsrDev At;   // n x m, rows = variables     (K5) == Julia's CSC arrays
  bool has_q = false;
  CsrDev Q;    // CSR(Q)   for Q*x
  CsrDev Qt;   // CSR(Q')  for dx'*Q

  double *c = nullptr, *b = nullptr, *lb = nullptr, *ub = nullptr;
  double *x = nullptr, *x_next = nullptr, *xbar = nullptr;
  double *y = nullptr, *y_next = nullptr;
  double *aty = nullptr, *aty_next = nullptr;  // n+1 each (slot n: exchange scalar)
  double *sum_x = nullptr, *sum_y = nullptr;
  double *qx = nullptr, *tmp_n = nullptr, *tmp_n2 = nullptr, *tmp_m = nullptr;
  int64_t sum_x_count = 0, sum_y_count = 0;
  // Lazy accept: pdhg_accept swaps the iterates and leaves K7 (sum += w * iterate) to the
  // kernels of the next trial, which read x and y anyway (primal_kernel, the dual epilogue);
  // every other entry point that reads or writes x, y or the sums settles it first
  // (flush_pending).  pend_x / pend_y: the sums do not contain pend_w * (x | y) yet.
  bool pend_x = false, pend_y = false, lazy_accept = true;
  double pend_w = 0.0;
  double sum_x_weights = 0.0, sum_y_weights = 0.0;

  double *pA = nullptr;   // partials of the A kernel (1 quantity)
  double *pAt = nullptr;  // partials of the A' kernel / interaction kernel (3 quantities)
  double *pQ = nullptr;   // partials of the QP dot
  int pAt_stride = 0;
  int ew_grid_n = 1, ew_grid_m = 1, ew_grid_nm = 1;


  // evaluation branch (N1), allocated on first use
  double *E = nullptr, *Dv = nullptr, *c_o = nullptr, *b_o = nullptr, *lb_o = nullptr, *ub_o = nullptr;
  double *x_r = nullptr, *y_r = nullptr;          // last restart point
  double *px_avg = nullptr, *py_avg = nullptr;    // materialised average
  double *ev_ax = nullptr, *ev_aty = nullptr;     // A*x (m), A'*y (n) at the evaluated point
  // The evaluation branch asks for the same products several times per check
  // (eval_point, then one or more trust-region bounds at the same point): keep
  // A*x and A'*y of the CURRENT and the AVERAGE point until the state changes.
  // slot 0 CURRENT, 1 AVERAGE (valid for one state_version), 2 RESTART point (valid until the
  // restart point or the matrix changes: every check asks for its products again)
  double *ev_cax[3] = {nullptr, nullptr, nullptr}, *ev_caty[3] = {nullptr, nullptr, nullptr};
  double *ev_cqx[3] = {nullptr, nullptr, nullptr}, *ev_qx = nullptr;   // Q*x at those points (QP only)
  uint64_t restart_version = 1, matrix_version = 1, ev_rkey = 0;
  // the scalars the rest of a check asks for next (distances of AVERAGE / CURRENT to the restart point, sum of squares of
  // the evaluated point), reduced WITH pdhg_eval_point's 22 quantities and kept until the state moves (abi_eval.hpp)
  double chk_vals[6] = {0, 0, 0, 0, 0, 0};
  uint64_t chk_state = 0, chk_restart = ~0ull;     // state_version / restart_version the values belong to
  int chk_point = -1;                              // the point whose sum of squares chk_vals[4..5] is
  bool chk_have_avg = false;
  uint64_t state_version = 1;                      // bumped by everything that moves x, y, the sums or A
  uint64_t ev_cversion[2] = {0, 0}, avg_version = 0;
  double *tr_g = nullptr, *tr_dir = nullptr, *tr_thr = nullptr;  // n+m each: g d, w d^2, breakpoint (tr_setup_kernel)
  // the trust-region search as one persistent launch (tr_coop_kernel.hpp): its own barrier words, census and partials
  int tr_coop = -1;                         // -1 not decided yet, 0 off (not wanted / does not suit / a barrier failed), 1 on
  int tr_grid = 0;
  GridSync *tr_sync = nullptr;
  unsigned long long tr_epoch = 0;
  unsigned tr_nxcd = 0, tr_xcd_cnt[8] = {};
  double *tr_partials = nullptr;
  long tr_coop_calls = 0;
  // several searches in one launch (tr_coop_batch_kernel): their own scratch vectors and partials
  double *trb_scratch = nullptr, *trb_partials = nullptr;
  long trb_calls = 0;
  double *ev_partials = nullptr;
  double *ev_xg = nullptr;                         // [n_alloc] full x at the evaluated point (group only)
  // the point being evaluated and its products (set by point_products)
  const double *pt_x = nullptr, *pt_y = nullptr;
  double *pt_ax = nullptr, *pt_aty = nullptr, *pt_qx = nullptr;
  int ev_grid = 1;
  bool has_original = false;

  bool profile = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int64_t prof_count[PDHG_K_COUNT] = {0};
  double prof_ms[PDHG_K_COUNT] = {0};

  // ---- row-partitioned form (dist.hpp).  A plain handle is rank 0 of 1: it owns
  // every column and every row, and all of the fields below keep their defaults.
  pdhg_handle *self = nullptr;     // == this (storage of the one-element shard list)
  DistGroup *grp = nullptr;        // shared by the shards of a group
  int rank = 0, world = 1;
  int64_t clo = 0, cn = 0;         // owned column slice [clo, clo + cn) of the n-vectors
  int64_t n_alloc = 0;             // length of the n-vectors that take part in collectives (world * S >= n)
  int64_t row_lo = 0;              // first GLOBAL row of this shard (m is the local row count)
  int64_t m_global = 0;
  int mt_flip = 0;                     // event set of this shard's next threaded stream barrier
  hipStream_t comm_stream = nullptr;   // group: per-slice reductions run here, beside the product that feeds them
  std::vector<hipEvent_t> ev_part;     // [world] "slice k of A_p'y_p is complete" on `stream`
  hipEvent_t ev_comm = nullptr;        // "all of this shard's reductions are done" on `comm_stream`
  // the all-gather of xbar overlapped with A_p xbar (DistGroup::ag_chunks > 1, dist.hpp): A_p cut by COLUMN CHUNK -- chunk c =
  // sub-range c of every rank's slice -- one complete layout per chunk, the row sums carried from pass to pass
  std::vector<CsrDev> Achunk;
  double *chunk_carry = nullptr;       // [m] row sums between the passes
  double *xchunk = nullptr;            // [chunks][world * ag_sub] xbar in CHUNK layout: chunk c holds, rank after rank, sub-range c of every slice
                                       // (what one ncclAllGather per chunk delivers; the chunk layouts' column indices point into it)
  std::vector<hipEvent_t> ev_ag;       // [chunks] "chunk c of xbar has arrived" on `comm_stream`
  hipEvent_t ev_xbar = nullptr;        // "the owned slice of xbar is written" on `stream`
  double *dn_buf = nullptr;        // [n_alloc] gather / partial buffer (group only)
  double *dm_buf = nullptr;        // [m_global] row-gather buffer (group only)
  // scalar results: scal_dev[SCAL_MAX] on the device, scal_all[world*SCAL_MAX] (RCCL gather), pinned scal_host
  double *scal_dev = nullptr, *scal_all = nullptr, *scal_host = nullptr;
  double *ev_host = nullptr;                // pinned result words of the evaluation reductions (ev_finish)
  unsigned long long ev_seq = 0;

  // ---- one trial step as ONE graph launch (small / medium problems: stream layouts,
  // where the ~8 launches and the result copy cost as much as the kernels).  Two
  // instances: the trial reads (x, y, A'y) and writes (x', y', A'y'), and accept swaps them.
  struct TrialGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t n_primal = nullptr, n_dual = nullptr, n_dual_long = nullptr;
    const double *x = nullptr, *y = nullptr, *aty = nullptr;   // the buffers this instance was built for
    double tau = 0.0, theta = 0.0, sigma = 0.0;                // scalars currently baked into the nodes
    bool add_x = false, add_y = false;                         // deferred K7 baked into the primal / dual nodes
    double add_wx = 0.0, add_wy = 0.0;
  } tgraph[2];
  // ---- one trial as ONE kernel launch (trial_kernel.hpp): stream layouts without column slabs, LP
  int coop_mode = -1;                   // -1 undecided, 0 off, 1 on
  int coop_grid = 0;                    // workgroups of the persistent launch (multiple of 8, all co-resident)
  unsigned coop_nxcd = 0;               // XCDs that hold workgroups of such a launch
  unsigned coop_xcd_cnt[8] = {0};       // ... and how many each
  unsigned long long coop_launches = 0, coop_epoch = 0;   // launches / grid barriers of the one-launch kernel so far
  // the multi-step kernel's XCD-local mode (small grids: every working workgroup on one XCD, trial_kernel.hpp)
  int local_mode = -1;                  // -1 not decided, 0 off, 1 on
  GridSync *lsync = nullptr;
  unsigned long long local_epoch = 0;
  long local_launches = 0;
  GridSync *gsync = nullptr;
  unsigned long long *coop_trace = nullptr;   // PDHG_COOP_TRACE=1: phase stamps of the last launch
  int graph_mode = -1;                  // -1 undecided, 0 off, 1 on
  hipStream_t graph_stream = nullptr;   // graphs launch here (== stream)
  unsigned long long *seq_dev = nullptr;   // launch counter, incremented by the final kernel
  volatile double *res_host = nullptr;     // pinned, coherent: 5 results + [7] = sequence number
  unsigned long long seq_expected = 0;
  int coop_fallbacks = 0;                   // trials repeated on the other paths after a barrier time-out
  double timeline_last_out[5] = {0, 0, 0, 0, 0};   // trial_timeline: when the last workgroup left each phase
  // several take_steps per launch (steps_kernel): control lines, pow tables (device + pinned staging), result words
  StepsCtl *steps_ctl = nullptr;
  double *steps_pow_dev = nullptr, *steps_pow_host = nullptr, *steps_res = nullptr;
  int steps_pow_cap = 0;
  unsigned long long steps_seq = 0;
  int64_t steps_launches = 0, steps_trials = 0;
  int small_lp_mode = -1;                   // -1 undecided, 0 off, 1: small_lp_steps_kernel takes the batches of steps
  int64_t small_lp_launches = 0;
  double res_error = 0.0;                   // error word of the last checked result read
  // host-side breakdown of graph trials (PDHG_VERBOSE): seconds in node updates, in hipGraphLaunch, waiting
  double t_set = 0.0, t_launch = 0.0, t_wait = 0.0;
  long n_graph_trials = 0;
};

#include "dist.hpp"
#include "group_kernel.hpp"

namespace {

void destroy_shard(pdhg_handle *h);

int ew_grid(int64_t len) {
  int64_t g = (len + TPB - 1) / TPB;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, EW_MAX_BLOCKS));
}

inline const char *prof_scope_name(int kid) {
  static const char *names[PDHG_K_COUNT] = {"pdhg:primal (K1+K2)", "pdhg:A*xbar + dual step (K3+K4)", "pdhg:A'*y' + interaction sums (K5+K6)",
                                           "pdhg:second-stage reduction (K6b)", "pdhg:accept (K7)", "pdhg:all-gather xbar",
                                           "pdhg:reduce-scatter A'y'", "pdhg:interaction on the slice"};
  return (kid >= 0 && kid < PDHG_K_COUNT && names[kid]) ? names[kid] : "pdhg:kernel";
}
struct ProfScope {
  pdhg_handle *h;
  int kid;
  RoctxRange range;
  ProfScope(pdhg_handle *h_, int kid_) : h(h_), kid(kid_), range(prof_scope_name(kid_)) {
    if (h->profile) (void)hipEventRecord(h->ev0, h->stream);
  }
  ~ProfScope() {
    if (h->profile) {
      (void)hipEventRecord(h->ev1, h->stream);
      (void)hipEventSynchronize(h->ev1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
      h->prof_count[kid] += 1;
      h->prof_ms[kid] += ms;
    }
  }
};

// The opt-in for > 64 KiB of dynamic LDS is per device and per kernel instance,
// and must only ever grow: another handle on the same device may need more than
// this one (hipFuncSetAttribute sets the limit, it does not raise it).
int ensure_lds_limit(pdhg_handle *h, int mode, int chunk_mode, size_t lds, const void *func) {
  static size_t limit[64][3][5] = {};
  static std::mutex mu;            // handles may be created / driven from several host threads (shard pool, Julia tasks)
  std::lock_guard<std::mutex> lock(mu);
  size_t &cur = limit[h->device & 63][mode][chunk_mode];
  if (cur < lds) {
    HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    cur = lds;
  }
  return 0;
}

// Dynamic LDS of one sweep workgroup: the accumulators -- padded to just over a third of the
// CU's 160 KiB, so that never more than the two workgroups per CU the layout plans for become
// resident.  Without the padding a problem with <= ~830 rows per wave gets three (the
// accumulators need < 53 KiB): the residency rounds no longer match the geometry and 24 waves
// per CU gather from more tiles at once -- 10M-nnz-per-million-rows LPs of 5.3M-6.5M rows ran
// at 10 ps per nonzero, against 7.0 at 5M and 7.7 at 7M (profiles/r02_locality.txt).
size_t tiled_lds_bytes(const CsrDev &D) {
  const size_t need = sizeof(double) * ((size_t)TW_WPB * D.tw_rows + 6 * TW_WPB + (D.tw_mode == 1 ? TW_WPB * WAVE : 0));
  return std::max(need, D.tw_lds_floor);
}

// XCD remap of the sweep's row groups (spmv_tiled_kernel): row groups per XCD, 0 = workgroup b takes row group b.
// Only for matrices whose workgroups touch a band of the tiles (10M, +-3M columns: 0.97 / 0.98 -> 0.85 / 0.87 ms); rows that
// scatter over every tile gain nothing, and where the row groups' work falls with the index (column-skewed A': the first
// eighth holds the heaviest) contiguous eighths leave one XCD with the most work: 1.11 -> 1.43 ms.
int tiled_per_xcd(const pdhg_handle *h, const CsrDev &D, int ngroups) {
  const char *ev = dev_env("PDHG_TW_REMAP");                                                  // dev knob: 0 / 1 force
  const bool on = ev ? ev[0] != '0' : D.tw_band;
  return (h->remap && on && ngroups >= 2 * NUM_XCD) ? (ngroups + NUM_XCD - 1) / NUM_XCD : 0;
}

// the tiled kernel's five chunk variants behind one call
template <int MODE>
int launch_tiled(pdhg_handle *h, const CsrDev &D, const double *xin, const EpiArgs &e, int g0, int g1) {
  const size_t lds = tiled_lds_bytes(D);
  const int w0 = g0 * TW_WPB;
  const int per_xcd = tiled_per_xcd(h, D, g1 - g0);
  const int grid = per_xcd > 0 ? per_xcd * NUM_XCD : g1 - g0;
  int rc;
#define PDHG_TILED(CH)                                                                                              \
  do {                                                                                                             \
    if ((rc = ensure_lds_limit(h, MODE, CH, lds, (const void *)spmv_tiled_kernel<MODE, CH>))) return rc;           \
    hipLaunchKernelGGL((spmv_tiled_kernel<MODE, CH>), dim3(grid), dim3(TW_WPB * WAVE), lds, h->stream,             \
                       D.wave_rows + w0, D.wave_ent, D.wave_step_off + w0, D.step_tile, D.wg_step_off + g0,        \
                       D.nwaves - w0, D.tile_shift, D.tw_rows, D.pk, D.tv, xin, e, g1 - g0, per_xcd);              \
  } while (0)
  if (D.tw_mode == 1) PDHG_TILED(1);
  else if (D.tw_mode == 2) PDHG_TILED(2);
  else if (D.tw_mode == 3) PDHG_TILED(3);
  else if (D.tw_mode == 4) PDHG_TILED(4);
  else PDHG_TILED(0);
#undef PDHG_TILED
  return 0;
}

// TAG: 0 the constraint matrix, 1 its transpose, 2 the objective matrix (profiler names)
// init != nullptr: the row sums start from init[row] -- a later COLUMN-CHUNK pass of a shard group's A_p xbar (the chunks of
// xbar arrive one after the other, dist.hpp: every chunk's product runs while the next chunk is still on the links).  For
// layouts without column slabs / row segments (the chunk matrices are built without them).
template <int MODE, int TAG>
int launch_spmv(pdhg_handle *h, const CsrDev &D, const double *xin, EpiArgs e, const double *init = nullptr) {
  if (init && (!D.segs.empty() || !D.slabs.empty())) return fail(-1, "a carried product needs a layout without segments / column slabs");
  e.init = init;
  if (!D.segs.empty()) {
    // segments of whole rows (layout.hpp): the same product segment by segment, every row-indexed operand moved to the
    // segment's first row, its block partials behind those of the segments before it
    for (const CsrDev &S : D.segs) {
      EpiArgs se = e;
      const int r0 = S.row0;
      if (MODE == MODE_PLAIN) se.out = e.out + r0;
      if (MODE == MODE_DUAL) {
        se.y = e.y + r0; se.b = e.b + r0; se.y_next = e.y_next + r0;
        se.num_eq = std::max(0, std::min(S.rows, e.num_eq - r0));
        if (e.sum_y) se.sum_y = e.sum_y + r0;
      }
      if (MODE == MODE_ATY) { se.x = e.x + r0; se.x_next = e.x_next + r0; se.aty = e.aty + r0; se.aty_next = e.aty_next + r0; }
      if (e.partials) se.partials = e.partials + S.slot0;
      const int rc = launch_spmv<MODE, TAG>(h, S, xin, se);
      if (rc) return rc;
    }
    return 0;
  }
  const int rx = h->relaxed ? 1 : 0;
  if (D.tiled) {
    if (D.grid > 0) {
      int rc = launch_tiled<MODE>(h, D, xin, e, 0, D.grid);
      if (rc) return rc;
    }
  } else if (!D.slabs.empty()) {
    // column-slab passes, ascending: partial row sums travel through slab_partial
    const int P = (int)D.slabs.size(), rm = h->remap ? 1 : 0;
    for (int p = 0; p < P; ++p) {
      const SlabDev &S = D.slabs[(size_t)p];
      if (S.sj.on()) {        // the slab in the sliced jagged layout (sj_kernels.hpp)
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        EpiArgs le = e;
        le.init = D.slab_partial;
        if (p + 1 < P && p == 0)
          launch_sj<MODE_PLAIN, false, TAG>(h->stream, S.sj, xin, rm, rx, 0, pe);
        else if (p + 1 < P)
          launch_sj<MODE_PLAIN, true, TAG>(h->stream, S.sj, xin, rm, rx, 0, pe);
        else
          launch_sj<MODE, true, TAG>(h->stream, S.sj, xin, rm, rx, S.grid, le);
        continue;
      }
      if (S.pipe_grid > 0) {  // the slab's row blocks as a persistent pipelined launch (spmv_stream_pipe_kernel)
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        EpiArgs le = e;
        le.init = D.slab_partial;
        if (p + 1 < P && p == 0)
          hipLaunchKernelGGL((spmv_stream_pipe_kernel<MODE_PLAIN, false, TAG>), dim3(S.pipe_grid), dim3(TPB), 0, h->stream, S.view(D.rows), xin,
                             (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx, 0, pe);
        else if (p + 1 < P)
          hipLaunchKernelGGL((spmv_stream_pipe_kernel<MODE_PLAIN, true, TAG>), dim3(S.pipe_grid), dim3(TPB), 0, h->stream, S.view(D.rows), xin,
                             (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx, 0, pe);
        else
          hipLaunchKernelGGL((spmv_stream_pipe_kernel<MODE, true, TAG>), dim3(S.pipe_grid), dim3(TPB), 0, h->stream, S.view(D.rows), xin,
                             (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx, S.grid, le);
        continue;
      }
      if (p + 1 < P) {
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        if (p == 0)
          hipLaunchKernelGGL((spmv_stream_kernel<MODE_PLAIN, false, TAG>), dim3(S.grid), dim3(TPB), 0, h->stream,
                             S.view(D.rows), xin, S.blks, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx | 2, pe);
        else
          hipLaunchKernelGGL((spmv_stream_kernel<MODE_PLAIN, true, TAG>), dim3(S.grid), dim3(TPB), 0, h->stream,
                             S.view(D.rows), xin, S.blks, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx | 2, pe);
      } else {
        EpiArgs le = e;
        le.init = D.slab_partial;
        hipLaunchKernelGGL((spmv_stream_kernel<MODE, true, TAG>), dim3(S.grid), dim3(TPB), 0, h->stream,
                           S.view(D.rows), xin, S.blks, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx | 2, le);
      }
    }
  } else if (D.sj.on()) {
    if (init) launch_sj<MODE, true, TAG>(h->stream, D.sj, xin, h->remap ? 1 : 0, rx, D.grid, e);
    else launch_sj<MODE, false, TAG>(h->stream, D.sj, xin, h->remap ? 1 : 0, rx, D.grid, e);
  } else if (D.pipe_grid > 0) {
    if (init)
      hipLaunchKernelGGL((spmv_stream_pipe_kernel<MODE, true, TAG>), dim3(D.pipe_grid), dim3(TPB), 0, h->stream, D.view(), xin,
                         (const int4 *)D.ext, D.nblk, D.per_xcd, h->remap ? 1 : 0, rx, D.grid, e);
    else
      hipLaunchKernelGGL((spmv_stream_pipe_kernel<MODE, false, TAG>), dim3(D.pipe_grid), dim3(TPB), 0, h->stream, D.view(), xin,
                         (const int4 *)D.ext, D.nblk, D.per_xcd, h->remap ? 1 : 0, rx, D.grid, e);
  } else if (D.grid > 0) {
    if (init)
      hipLaunchKernelGGL((spmv_stream_kernel<MODE, true, TAG>), dim3(D.grid), dim3(TPB), 0, h->stream,
                         D.view(), xin, D.blks, (const int4 *)D.ext, D.nblk, D.per_xcd, h->remap ? 1 : 0, rx, e);
    else
      hipLaunchKernelGGL((spmv_stream_kernel<MODE, false, TAG>), dim3(D.grid), dim3(TPB), 0, h->stream,
                         D.view(), xin, D.blks, (const int4 *)D.ext, D.nblk, D.per_xcd, h->remap ? 1 : 0, rx, e);
  }
  if (D.nlong > 0) {
    hipLaunchKernelGGL(spmv_long_partial_kernel<TAG>, dim3(D.nchunks), dim3(TPB), 0, h->stream,
                       D.view(), xin, D.chunk_row, D.chunk_off, D.chunk_partial);
    hipLaunchKernelGGL(spmv_long_final_kernel<MODE>, dim3(D.long_grid), dim3(TPB), 0, h->stream,
                       D.long_row, D.long_chunk_ptr, D.nlong, D.chunk_partial, e, D.grid);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// MODE_PLAIN product restricted to the tiled workgroups [g0, g1) (their rows are a
// contiguous range of the output); `with_long` also runs the long-row path.
// The kernel is the one launch_spmv uses: the per-wave / per-workgroup tables
// are simply passed from offset g0 (row numbers and entry offsets are absolute).
int launch_spmv_plain_part(pdhg_handle *h, const CsrDev &D, const double *xin, double *out,
                           int g0, int g1, bool with_long) {
  if (!D.tiled) return fail(-1, "partial launch needs the tiled layout");
  EpiArgs e{};
  e.out = out;
  if (with_long && D.nlong > 0) {
    hipLaunchKernelGGL(spmv_long_partial_kernel<1>, dim3(D.nchunks), dim3(TPB), 0, h->stream,
                       D.view(), xin, D.chunk_row, D.chunk_off, D.chunk_partial);
    hipLaunchKernelGGL(spmv_long_final_kernel<MODE_PLAIN>, dim3(D.long_grid), dim3(TPB), 0, h->stream,
                       D.long_row, D.long_chunk_ptr, D.nlong, D.chunk_partial, e, D.grid);
  }
  if (g1 > g0) {
    int rc = launch_tiled<MODE_PLAIN>(h, D, xin, e, g0, g1);
    if (rc) return rc;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// K1+K2 on this shard's column slice (the whole vector for a plain handle).
// QP: Q is replicated and acts on the full x (kept full on every shard).
int launch_primal(pdhg_handle *h, double tau, double theta, bool write_xbar) {
  ProfScope ps(h, PDHG_K_PRIMAL);
  if (h->has_q) {
    EpiArgs e{};
    e.out = h->qx;
    int rc = launch_spmv<MODE_PLAIN, 2>(h, h->Q, h->x, e);
    if (rc) return rc;
  }
  const int n = (int)h->cn;
  const int64_t o = h->clo;
  const int grid = ew_grid((h->cn + 1) / 2);
#define PK(HQ, WX)                                                                           \
  hipLaunchKernelGGL((primal_kernel<HQ, WX>), dim3(grid), dim3(TPB), 0, h->stream, n,        \
                     h->x + o, h->c + o, h->aty + o, h->has_q ? h->qx + o : nullptr, h->lb + o, h->ub + o, tau, theta, \
                     h->x_next + o, h->xbar + o, h->pend_w, h->pend_x ? h->sum_x + o : nullptr)
  if (h->has_q) { if (write_xbar) PK(true, true); else PK(true, false); }
  else          { if (write_xbar) PK(false, true); else PK(false, false); }
#undef PK
  HIP_TRY(hipGetLastError());
  h->pend_x = false;
  return 0;
}

int launch_xbar(pdhg_handle *h, double theta) {
  const int64_t o = h->clo;
  hipLaunchKernelGGL(xbar_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, h->x + o, h->x_next + o,
                     theta, h->xbar + o);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_dual(pdhg_handle *h, double sigma) {
  ProfScope ps(h, PDHG_K_SPMV_DUAL);
  EpiArgs e{};
  e.y = h->y; e.b = h->b; e.y_next = h->y_next; e.sigma = sigma; e.num_eq = (int)h->num_eq;
  e.partials = h->pA; e.stride = h->A.slots(); e.lo_offset = h->A.slots();
  if (h->pend_y) { e.sum_y = h->sum_y; e.avg_w = h->pend_w; }
  const int rc = launch_spmv<MODE_DUAL, 0>(h, h->A, h->xbar, e);
  if (!rc) h->pend_y = false;
  return rc;
}

// The same product as column-chunk passes (DistGroup::ag_chunks, dist.hpp): chunk c's layout against chunk c of xbar (xchunk:
// the chunk's columns side by side, rank after rank), the row sums carried through chunk_carry, the dual step fused into the LAST pass, whose block partials go to the
// front of pA (stride = that layout's slots: dual_chunk_slots).  wait_events: pass c first waits for "chunk c of xbar has
// arrived" (ev_ag[c], recorded on the comm stream).
int dual_chunk_slots(const pdhg_handle *h) { return std::max(h->Achunk.empty() ? 0 : h->Achunk.back().slots(), 1); }
int launch_dual_chunked(pdhg_handle *h, double sigma, bool wait_events) {
  ProfScope ps(h, PDHG_K_SPMV_DUAL);
  const int C = (int)h->Achunk.size();
  int rc = 0;
  for (int c = 0; c < C && !rc; ++c) {
    if (wait_events) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_ag[(size_t)c], 0));
    const double *init = c > 0 ? h->chunk_carry : nullptr;
    if (c + 1 < C) {
      EpiArgs e{};
      e.out = h->chunk_carry;
      rc = launch_spmv<MODE_PLAIN, 0>(h, h->Achunk[(size_t)c], h->xchunk + (size_t)c * h->Achunk[(size_t)c].cols, e, init);
    } else {
      EpiArgs e{};
      e.y = h->y; e.b = h->b; e.y_next = h->y_next; e.sigma = sigma; e.num_eq = (int)h->num_eq;
      e.partials = h->pA; e.stride = dual_chunk_slots(h); e.lo_offset = dual_chunk_slots(h);
      if (h->pend_y) { e.sum_y = h->sum_y; e.avg_w = h->pend_w; }
      rc = launch_spmv<MODE_DUAL, 0>(h, h->Achunk[(size_t)c], h->xchunk + (size_t)c * h->Achunk[(size_t)c].cols, e, init);
      if (!rc) h->pend_y = false;
    }
  }
  return rc;
}

int launch_aty_fused(pdhg_handle *h) {
  ProfScope ps(h, PDHG_K_SPMV_ATY);
  EpiArgs e{};
  e.x = h->x; e.x_next = h->x_next; e.aty = h->aty; e.aty_next = h->aty_next;
  e.partials = h->pAt; e.stride = h->pAt_stride; e.lo_offset = 3 * h->pAt_stride;
  return launch_spmv<MODE_ATY, 1>(h, h->At, h->y_next, e);
}

int launch_aty_plain(pdhg_handle *h, const double *yin, double *out) {
  ProfScope ps(h, PDHG_K_SPMV_ATY);
  EpiArgs e{};
  e.out = out;
  return launch_spmv<MODE_PLAIN, 1>(h, h->At, yin, e);
}

// 0.5 * dx' Q dx partials into pQ (QP only; full vectors -- replicated in a group)
int launch_q_interaction(pdhg_handle *h, int *count) {
  *count = 0;
  if (!h->has_q) return 0;
  const int n = (int)h->n;
  hipLaunchKernelGGL(diff_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, h->x_next, h->x, h->tmp_n);
  EpiArgs e{};
  e.out = h->tmp_n2;
  int rc = launch_spmv<MODE_PLAIN, 2>(h, h->Qt, h->tmp_n, e);  // (dx' Q)' = Q' dx
  if (rc) return rc;
  hipLaunchKernelGGL(dot_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, h->tmp_n2, h->tmp_n, h->pQ);
  HIP_TRY(hipGetLastError());
  *count = h->ew_grid_n;
  return 0;
}

// second stage of the block partials straight into pinned host memory, then the launch's
// sequence number: the host polls that word instead of a device-to-host copy + stream
// synchronisation (the copy alone is a 4 us kernel on this runtime).
__global__ __launch_bounds__(FINAL_TPB) void final_reduce_host_kernel(FinalSpec sp, unsigned long long *seq_dev,
                                                                      volatile double *res_host) {
  double res[5];
  final_reduce_body<FINAL_TPB / WAVE>(sp, res);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) res_host[k] = res[k];
    const unsigned long long s = *seq_dev + 1ull;
    *seq_dev = s;
    __threadfence_system();
    res_host[7] = (double)s;      // exact up to 2^53 launches
  }
}

// second-stage reduction of the trial's block partials into scal_dev[0..5)
// (dy_lo: where the low parts of the dy^2 partials start behind p_dy; < 0: the handle's own A.slots())
int launch_final(pdhg_handle *h, const double *p_int, int n_int, int stride_int, const double *p_dy, int n_dy,
                 int q_count, bool to_host = false, int dy_lo = -1) {
  ProfScope ps(h, PDHG_K_FINAL);
  FinalSpec sp{};
  sp.ptr[0] = p_int;                  sp.count[0] = n_int;
  sp.ptr[1] = p_int + stride_int;     sp.count[1] = n_int;
  sp.ptr[2] = p_dy;                   sp.count[2] = n_dy;
  sp.ptr[3] = p_int + 2 * stride_int; sp.count[3] = n_int;
  sp.ptr[4] = h->pQ;                  sp.count[4] = q_count;
  for (int q : {0, 1, 3}) sp.ptr_lo[q] = sp.ptr[q] + 3 * stride_int;
  sp.ptr_lo[2] = p_dy + (dy_lo >= 0 ? dy_lo : h->A.slots());
  sp.ptr_lo[4] = h->pQ + h->ew_grid_n;
  sp.out = h->scal_dev;
  if (to_host) {       // results straight into the pinned result word (the caller polls it: wait_result_word)
    hipLaunchKernelGGL(final_reduce_host_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, sp, h->seq_dev, h->res_host);
    h->seq_expected += 1;
  } else {
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, sp);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

#include "host_trial_graph.hpp"
#include "host_trial_coop.hpp"
#include "host_small_lp.hpp"
#include "host_trial_graph_build.hpp"
#include "host_shards.hpp"
}  // namespace

// ================================================================== C ABI

extern "C" {

const char *pdhg_last_error(void) { return g_last_error.c_str(); }
int pdhg_abi_version(void) { return 11; }

// The kernels behind one fused product, as rocprofv3 prints them (template arguments <MODE,
// INIT, TAG> / <MODE, CH> / <TAG>; MODE 0 plain, 1 dual epilogue, 2 A'y epilogue; TAG 0 = A,
// 1 = A'), joined by " + ": a profile of the separate launches adds up to the product by
// summing these names (tools/rocprof_summary.py does).
static std::string product_kernels(const CsrDev &D, int mode, int tag) {
  if (!D.segs.empty())
    return product_kernels(D.segs.front(), mode, tag) + " (x " + std::to_string(D.segs.size()) + " row segments)";
  std::string out;
  auto add = [&](const std::string &k) { out += (out.empty() ? "" : " + ") + k; };
  const std::string m = std::to_string(mode), t = std::to_string(tag);
  if (D.tiled) {
    if (D.grid > 0) add("spmv_tiled_kernel<" + m + ", " + std::to_string(D.tw_mode) + ">");
  } else if (!D.slabs.empty()) {
    const SjDev &J = D.slabs.front().sj;
    const std::string k = J.on() ? "spmv_sj_kernel<" : (D.slabs.front().pipe_grid > 0 ? "spmv_stream_pipe_kernel<" : "spmv_stream_kernel<");
    const std::string g = J.on() ? ", " + std::to_string(J.G) + ">" : ">";      // the sliced jagged kernel's fourth argument: its form (sj_kernels.hpp)
    add(k + "0, false, " + t + g);
    if (D.slabs.size() > 2) add(k + "0, true, " + t + g);
    add(k + m + ", true, " + t + g);
  } else if (D.sj.on()) {
    add("spmv_sj_kernel<" + m + ", false, " + t + ", " + std::to_string(D.sj.G) + ">");
  } else if (D.pipe_grid > 0) {
    add("spmv_stream_pipe_kernel<" + m + ", false, " + t + ">");
  } else if (D.grid > 0) {
    add("spmv_stream_kernel<" + m + ", false, " + t + ">");
  }
  if (D.nlong > 0) {
    add("spmv_long_partial_kernel<" + t + ">");
    add("spmv_long_final_kernel<" + m + ">");
  }
  return out.empty() ? "(no kernel: empty matrix)" : out;
}

const char *pdhg_kernel_name(pdhg_handle *h, int kernel_id) {
  static thread_local std::string buf;
  const bool fused = !(h && h->grp);
  switch (kernel_id) {
    case PDHG_K_PRIMAL: return "primal_kernel";
    case PDHG_K_SPMV_DUAL:
      if (!h) return "spmv_stream_kernel<1, false, 0>";
      if (h->grp && h->grp->ag_chunks > 1 && !h->has_q && !h->Achunk.empty()) {
        // a shard group with the all-gather in column chunks: one carried pass per chunk (the stream kernels' later passes
        // are their INIT instances, <.., true, ..>: a label for people, not for tools/rocprof_summary.py)
        buf = std::to_string(h->Achunk.size()) + " column-chunk passes with carried row sums: " + product_kernels(h->Achunk.front(), MODE_PLAIN, 0) +
              " ... " + product_kernels(h->Achunk.back(), MODE_DUAL, 0);
        return buf.c_str();
      }
      buf = product_kernels(h->A, MODE_DUAL, 0);
      return buf.c_str();
    case PDHG_K_SPMV_ATY:
      if (!h) return "spmv_stream_kernel<2, false, 1>";
      buf = product_kernels(h->At, fused ? MODE_ATY : MODE_PLAIN, 1);
      return buf.c_str();
    case PDHG_K_FINAL: return "final_reduce_kernel";
    case PDHG_K_ACCEPT: return "accept_kernel";
    case PDHG_K_ALLGATHER: return "all_gather(xbar)";
    case PDHG_K_REDUCE_SCATTER: return "reduce_scatter(A_p'y_p)";
    case PDHG_K_INTERACTION: return "interaction_kernel";
    default: return "?";
  }
}

static int create_multi_impl(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                             const int64_t *colptr, const int64_t *rowval, const double *nzval,
                             int index_base, const double *c, const double *b, const double *lb,
                             const double *ub, int64_t num_equalities, int n_devices, const int *device_ids,
                             const int64_t *row_bounds);

int pdhg_create(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                const int64_t *colptr, const int64_t *rowval, const double *nzval,
                int index_base, const double *c, const double *b, const double *lb,
                const double *ub, int64_t num_equalities, int device_id, void *stream) {
  if (!out) return fail(-1, "out == NULL");
  // The device layouts index nonzeros with 32 bits.  The reference's matrices are
  // SparseMatrixCSC{Float64,Int64} (quadratic_programming.jl:64): a matrix with more
  // nonzeros than that is held as SEGMENTS of whole rows inside this one handle (64-bit
  // extents = segment base pointer + 32-bit local offsets; build_segments) -- the default
  // since round 4: no exchange, every row sum in its reference order.  PDHG_HUGE=shards keeps
  // rounds 2-3's form: row shards on this ONE device behind a group handle (the row-partitioned
  // form with the peer-kernel exchange, dist.hpp).  m and n stay below 2^31.
  // PDHG_MAX_SHARD_NNZ lowers the limit (tests).
  int64_t cap = (int64_t)INT32_MAX - 1;
  if (const char *ev = getenv("PDHG_MAX_SHARD_NNZ")) cap = std::max<int64_t>(1, atoll(ev));
  const char *huge = dev_env("PDHG_HUGE");
  if (nnz > cap && !(huge && !strcmp(huge, "shards"))) {
    // host-side validation first (no device needed, nothing indexed with an unchecked row index later on)
    *out = nullptr;
    if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
    if (!colptr || !rowval || !nzval || !c || !b || !lb || !ub) return fail(-1, "null input array");
    if (colptr[0] != index_base || colptr[n] - index_base != nnz) return fail(-1, "colptr does not match nnz / index_base");
    {
      std::vector<int64_t> prefix;
      if (row_nnz_prefix(m, n, colptr, rowval, index_base, prefix) != 0) return fail(-1, "row index out of range");
    }
    return create_shard(out, m, n, nnz, colptr, rowval, nzval, index_base, c, b, lb, ub, num_equalities, device_id, stream, n, cap);
  }
  if (nnz > cap && m > 1) {
    *out = nullptr;
    // The shards run on private streams and synchronise among themselves: work the caller
    // orders against ITS stream would silently lose that ordering.
    if (stream) return fail(-1, "a matrix beyond the 32-bit nonzero limit is sharded on the device and cannot run on a "
                                "caller-supplied stream: pass stream = NULL");
    if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
    if (!colptr || !rowval || !nzval || !c || !b || !lb || !ub) return fail(-1, "null input array");
    if (colptr[0] != index_base || colptr[n] - index_base != nnz) return fail(-1, "colptr does not match nnz / index_base");
    int dev = device_id;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    // shards balanced by nonzeros, aiming at 80 % of the limit; the partition works on whole
    // rows, so every shard is checked against the limit and the count raised until all fit
    const int64_t target = std::max<int64_t>(1, (cap / 10) * 8);
    int64_t shards = std::min<int64_t>(m, (nnz + target - 1) / target);
    std::vector<int64_t> prefix, bounds;
    // (every row index is range-checked here, before anything is indexed with it: create_shard's own validation
    //  only runs after the partition has walked rowval)
    if (row_nnz_prefix(m, n, colptr, rowval, index_base, prefix) != 0) return fail(-1, "row index out of range");
    for (int64_t r = 0; r < m; ++r)
      if (prefix[(size_t)r + 1] - prefix[(size_t)r] > cap)
        return fail(-2, "row " + std::to_string(r) + " alone holds " + std::to_string(prefix[(size_t)r + 1] - prefix[(size_t)r]) +
                            " nonzeros, more than a shard can index (" + std::to_string(cap) + ")");
    for (;; ++shards) {
      if (shards > P2P_MAX_WORLD) return fail(-2, "more than 16 x 2^31 nonzeros on one device are not supported");
      partition_rows_from_prefix(prefix, (int)shards, bounds);
      int64_t worst = 0;
      for (int64_t p = 0; p < shards; ++p)
        worst = std::max(worst, prefix[(size_t)bounds[(size_t)p + 1]] - prefix[(size_t)bounds[(size_t)p]]);
      if (worst <= cap) break;
    }
    std::vector<int> ids((size_t)shards, dev);
    return create_multi_impl(out, m, n, nnz, colptr, rowval, nzval, index_base, c, b, lb, ub, num_equalities,
                             (int)shards, ids.data(), bounds.data());
  }
  return create_shard(out, m, n, nnz, colptr, rowval, nzval, index_base, c, b, lb, ub, num_equalities,
                      device_id, stream, n, 0);
}

#include "abi_dist.hpp"
#include "abi_trial.hpp"
#include "abi_eval.hpp"
#include "abi_rescale.hpp"
#include "abi_measure.hpp"
}  // extern "C"
