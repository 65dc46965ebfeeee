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
  unsigned long long local_epoch = 0, local_tickets = 0;
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
  static size_t limit[64][3][3] = {};
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

// the tiled kernel's three chunk variants behind one call
template <int MODE>
int launch_tiled(pdhg_handle *h, const CsrDev &D, const double *xin, const EpiArgs &e, int g0, int g1) {
  const size_t lds = tiled_lds_bytes(D);
  const int w0 = g0 * TW_WPB;
  int rc;
#define PDHG_TILED(CH)                                                                                              \
  do {                                                                                                             \
    if ((rc = ensure_lds_limit(h, MODE, CH, lds, (const void *)spmv_tiled_kernel<MODE, CH>))) return rc;           \
    hipLaunchKernelGGL((spmv_tiled_kernel<MODE, CH>), dim3(g1 - g0), dim3(TW_WPB * WAVE), lds, h->stream,          \
                       D.wave_rows + w0, D.wave_ent, D.wave_step_off + w0, D.step_tile, D.wg_step_off + g0,        \
                       D.nwaves - w0, D.tile_shift, D.tw_rows, D.pk, D.tv, xin, e);                                \
  } while (0)
  if (D.tw_mode == 1) PDHG_TILED(1);
  else if (D.tw_mode == 2) PDHG_TILED(2);
  else PDHG_TILED(0);
#undef PDHG_TILED
  return 0;
}

// TAG: 0 the constraint matrix, 1 its transpose, 2 the objective matrix (profiler names)
template <int MODE, int TAG>
int launch_spmv(pdhg_handle *h, const CsrDev &D, const double *xin, EpiArgs e) {
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
          hipLaunchKernelGGL((spmv_sj_kernel<MODE_PLAIN, false, TAG>), dim3(S.sj.grid), dim3(TPB), 0, h->stream, sj_view(S.sj), xin, rm, 0, pe);
        else if (p + 1 < P)
          hipLaunchKernelGGL((spmv_sj_kernel<MODE_PLAIN, true, TAG>), dim3(S.sj.grid), dim3(TPB), 0, h->stream, sj_view(S.sj), xin, rm, 0, pe);
        else
          hipLaunchKernelGGL((spmv_sj_kernel<MODE, true, TAG>), dim3(S.sj.grid), dim3(TPB), 0, h->stream, sj_view(S.sj), xin, rm, S.grid, le);
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
                             S.view(D.rows), xin, S.blks, S.nblk, S.per_xcd, rm, rx, pe);
        else
          hipLaunchKernelGGL((spmv_stream_kernel<MODE_PLAIN, true, TAG>), dim3(S.grid), dim3(TPB), 0, h->stream,
                             S.view(D.rows), xin, S.blks, S.nblk, S.per_xcd, rm, rx, pe);
      } else {
        EpiArgs le = e;
        le.init = D.slab_partial;
        hipLaunchKernelGGL((spmv_stream_kernel<MODE, true, TAG>), dim3(S.grid), dim3(TPB), 0, h->stream,
                           S.view(D.rows), xin, S.blks, S.nblk, S.per_xcd, rm, rx, le);
      }
    }
  } else if (D.sj.on()) {
    hipLaunchKernelGGL((spmv_sj_kernel<MODE, false, TAG>), dim3(D.sj.grid), dim3(TPB), 0, h->stream, sj_view(D.sj), xin,
                       h->remap ? 1 : 0, D.grid, e);
  } else if (D.pipe_grid > 0) {
    hipLaunchKernelGGL((spmv_stream_pipe_kernel<MODE, false, TAG>), dim3(D.pipe_grid), dim3(TPB), 0, h->stream, D.view(), xin,
                       (const int4 *)D.ext, D.nblk, D.per_xcd, h->remap ? 1 : 0, rx, D.grid, e);
  } else if (D.grid > 0) {
    hipLaunchKernelGGL((spmv_stream_kernel<MODE, false, TAG>), dim3(D.grid), dim3(TPB), 0, h->stream,
                       D.view(), xin, D.blks, D.nblk, D.per_xcd, h->remap ? 1 : 0, rx, e);
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
int launch_final(pdhg_handle *h, const double *p_int, int n_int, int stride_int, const double *p_dy, int n_dy,
                 int q_count, bool to_host = false) {
  ProfScope ps(h, PDHG_K_FINAL);
  FinalSpec sp{};
  sp.ptr[0] = p_int;                  sp.count[0] = n_int;
  sp.ptr[1] = p_int + stride_int;     sp.count[1] = n_int;
  sp.ptr[2] = p_dy;                   sp.count[2] = n_dy;
  sp.ptr[3] = p_int + 2 * stride_int; sp.count[3] = n_int;
  sp.ptr[4] = h->pQ;                  sp.count[4] = q_count;
  for (int q : {0, 1, 3}) sp.ptr_lo[q] = sp.ptr[q] + 3 * stride_int;
  sp.ptr_lo[2] = p_dy + h->A.slots();
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

// ---- the trial step as a HIP graph ---------------------------------------------------

template <typename... Args>
hipError_t graph_add_kernel_lds(hipGraph_t g, hipGraphNode_t *node, const std::vector<hipGraphNode_t> &deps,
                                const void *func, dim3 grid, dim3 block, size_t lds, Args... args) {
  void *params[] = {(void *)&args...};
  hipKernelNodeParams p{};
  p.func = const_cast<void *>(func);
  p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)lds;
  p.kernelParams = params; p.extra = nullptr;
  return hipGraphAddKernelNode(node, g, deps.empty() ? nullptr : deps.data(), deps.size(), &p);
}
template <typename... Args>
hipError_t graph_add_kernel(hipGraph_t g, hipGraphNode_t *node, const std::vector<hipGraphNode_t> &deps,
                            const void *func, dim3 grid, dim3 block, Args... args) {
  return graph_add_kernel_lds(g, node, deps, func, grid, block, 0, args...);
}
template <typename... Args>
hipError_t graph_set_kernel_lds(hipGraphExec_t exec, hipGraphNode_t node, const void *func, dim3 grid, dim3 block,
                                size_t lds, Args... args) {
  void *params[] = {(void *)&args...};
  hipKernelNodeParams p{};
  p.func = const_cast<void *>(func);
  p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)lds;
  p.kernelParams = params; p.extra = nullptr;
  return hipGraphExecKernelNodeSetParams(exec, node, &p);
}
template <typename... Args>
hipError_t graph_set_kernel(hipGraphExec_t exec, hipGraphNode_t node, const void *func, dim3 grid, dim3 block,
                            Args... args) {
  return graph_set_kernel_lds(exec, node, func, grid, block, 0, args...);
}

// pinned, host-coherent result word of the one-launch paths: [0..5) sums, [6] error, [7] sequence number
int ensure_result_word(pdhg_handle *h) {
  if (h->seq_dev) return 0;
  HIP_TRY(hipMalloc((void **)&h->seq_dev, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(h->seq_dev, 0, sizeof(unsigned long long), nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));   // the null stream does not order against h->stream
  HIP_TRY(hipHostMalloc((void **)&h->res_host, 8 * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
  for (int q = 0; q < 8; ++q) h->res_host[q] = 0.0;
  return 0;
}

// wait for launch number seq_expected's results in pinned memory (bounded spin, then the stream).
// checked: the trial kernel publishes without a system-scope fence -- a read counts only when
// the sequence number AND the checksum over the eight words match (trial_kernel.hpp).
int wait_result_word(pdhg_handle *h, double out[5], bool checked = false) {
  const double want = (double)h->seq_expected;
  const volatile unsigned long long *bits = reinterpret_cast<const volatile unsigned long long *>(h->res_host);
  auto ready = [&]() -> bool {
    if (h->res_host[7] != want) return false;
    if (!checked) return true;
    unsigned long long w[8];
    for (int q = 0; q < 8; ++q) w[q] = bits[q];
    unsigned long long ck = RESULT_CHECK_SALT ^ w[6] ^ w[7];
    for (int q = 0; q < 5; ++q) ck ^= w[q];
    if (ck != w[5]) return false;
    for (int q = 0; q < 5; ++q) memcpy(&out[q], &w[q], 8);
    memcpy(&h->res_error, &w[6], 8);
    return true;
  };
  bool seen = false;
  for (long spin = 0; spin < 40000000L; ++spin) {
    if (ready()) { seen = true; break; }
    if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(h->stream) != hipErrorNotReady) break;
  }
  if (!seen) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!ready()) {
      h->seq_expected = (unsigned long long)h->res_host[7];   // resynchronise: the next launch can succeed
      return fail(998, "one-launch trial finished without publishing its results");
    }
  }
  if (!checked) for (int q = 0; q < 5; ++q) out[q] = h->res_host[q];
  out[4] *= 0.5;
  return 0;
}

// ---- the trial step as ONE persistent kernel (trial_kernel.hpp) -----------------------------

int coop_prepare(pdhg_handle *h, int cap_limit = 0, bool several_items = false);
bool graph_eligible(pdhg_handle *h);

bool coop_eligible(pdhg_handle *h) {
  if (h->coop_mode < 0) {
    const char *ev = getenv("PDHG_COOP");
    bool on = !h->grp && !h->A.tiled && !h->At.tiled && h->A.slabs.empty() && h->At.slabs.empty() &&
              h->A.segs.empty() && h->At.segs.empty() && h->n > 0 && h->m > 0;
    if (h->has_q) on = on && !h->Q.tiled && !h->Qt.tiled && h->Q.slabs.empty() && h->Qt.slabs.empty();
    if (ev) on = on && ev[0] != '0';
    const char *gv = getenv("PDHG_GRAPH");             // PDHG_GRAPH=0: separate launches, no one-launch path of either kind
    if (gv) on = on && gv[0] != '0';
    h->coop_mode = on ? 1 : 0;
    if (on && coop_prepare(h) != 0) h->coop_mode = 0;  // too many items for one co-resident grid, or no census: graph / plain path
  }
  return h->coop_mode == 1 && !h->profile;
}

// grid of the persistent launch + the census of workgroups per XCD (once per handle)
// cap_limit: at most this many workgroups (several shards share a device); several_items: accept more items than
// workgroups (the phases then walk several row blocks per workgroup)
int coop_prepare(pdhg_handle *h, int cap_limit, bool several_items) {
  if (h->gsync) return 0;
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_result_word(h);
  if (rc) return rc;
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trial_kernel<false>, TPB, 0));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, h->device));
  int cap = std::max(8, per_cu * prop.multiProcessorCount / 8 * 8);
  if (const char *ev = dev_env("PDHG_COOP_WGS")) cap = std::max(8, std::min(cap, atoi(ev) / 8 * 8));
  if (cap_limit > 0) cap = std::max(8, std::min(cap, cap_limit / 8 * 8));
  // test knob: pretend the device holds this many workgroups (more than it does: the barriers cannot complete)
  const char *pretend = dev_env("PDHG_COOP_TEST_PRETEND_WGS");
  if (pretend) cap = std::max(8, atoi(pretend) / 8 * 8);
  // one item per workgroup and phase where the device can hold that many: row blocks from the front, long-row chunks from the end
  int items = std::max(h->A.grid + h->A.nchunks, h->At.grid + h->At.nchunks);
  if (h->has_q) items = std::max(items, std::max(h->Q.grid + h->Q.nchunks, h->A.grid + h->A.nchunks + h->Qt.grid + h->Qt.nchunks));
  h->coop_grid = pretend ? cap : std::min(cap, std::max(8, (items + 7) / 8 * 8));
  // More items than co-resident workgroups: the persistent kernel would walk several row blocks per workgroup at
  // 5 workgroups per CU, where the separate stream kernels keep 8 per CU in flight -- measured slower (PageRank-1M,
  // 4 552 items on 1 280 workgroups: 4 380 it/s against 4 620 as a graph of slab passes).  Leave those to the graph.
  if (items > cap && !several_items && !dev_env("PDHG_COOP_FORCE")) {
    h->coop_mode = 0;
    return 1;       // not an error: the caller falls through to the graph / plain path
  }
  HIP_TRY(hipMalloc((void **)&h->gsync, sizeof(GridSync)));
  HIP_TRY(hipMemsetAsync(h->gsync, 0, sizeof(GridSync), h->stream));
  if (getenv("PDHG_COOP_TRACE")) {
    HIP_TRY(hipMalloc((void **)&h->coop_trace, sizeof(unsigned long long) * 8 * (size_t)h->coop_grid));
    HIP_TRY(hipMemsetAsync(h->coop_trace, 0, sizeof(unsigned long long) * 8 * (size_t)h->coop_grid, h->stream));
  }
  hipLaunchKernelGGL(xcd_register_kernel, dim3(h->coop_grid), dim3(TPB), 0, h->stream, h->gsync);
  HIP_TRY(hipGetLastError());
  GridSync host;
  HIP_TRY(hipMemcpyAsync(&host, h->gsync, sizeof(GridSync), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned long long total = 0;
  h->coop_nxcd = 0;
  for (int x = 0; x < 8; ++x) { total += host.xcd_count[x][0]; h->coop_nxcd += host.xcd_count[x][0] > 0; h->coop_xcd_cnt[x] = (unsigned)host.xcd_count[x][0]; }
  if (total != (unsigned long long)h->coop_grid || h->coop_nxcd == 0) {
    h->coop_mode = 0;
    return fail(996, "one-launch trial: workgroup census does not add up");
  }
  if (getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] one-launch trial: %d workgroups (%d per CU possible) on %u XCDs, %d + %d / %d + %d row blocks + long chunks\n",
            h->coop_grid, per_cu, h->coop_nxcd, h->A.grid, h->A.nchunks, h->At.grid, h->At.nchunks);
  return 0;
}

TrialProduct trial_product(pdhg_handle *h, CsrDev &D, const double *xin, const EpiArgs &e) {
  TrialProduct P{};
  P.M = D.view();
  P.blks = D.blks; P.nblk = D.nblk; P.per_xcd = D.per_xcd; P.grid = D.grid; P.remap = h->remap ? 1 : 0;
  P.nchunks = D.nchunks; P.nlong = D.nlong; P.long_grid = D.long_grid;
  P.chunk_row = D.chunk_row; P.chunk_off = D.chunk_off; P.chunk_lidx = D.chunk_lidx; P.chunk_partial = D.chunk_partial;
  P.long_ticket = D.long_ticket;
  P.long_row = D.long_row; P.long_chunk_ptr = D.long_chunk_ptr;
  P.xin = xin; P.e = e;
  P.uses = D.coop_uses;
  D.coop_uses += 1;
  return P;
}

// returns 1 when the handle turned out not to suit the one-launch kernel (nothing was launched)
// Two persistent launches that are both only PARTLY resident would wait for each other's workgroups for ever
// (until the spin limit): one such kernel at a time per device, from launch until its results are back.
std::mutex &coop_device_mutex(int device) {
  static std::mutex mu[64];
  return mu[device & 63];
}

int coop_trial(pdhg_handle *h, double step_size, double primal_weight, double theta, bool xbar_only, double out[5]) {
  int rc = coop_prepare(h);
  if (rc) return rc;
  std::lock_guard<std::mutex> one_at_a_time(coop_device_mutex(h->device));
  const auto c0 = std::chrono::steady_clock::now();
  TrialKernelArgs a{};
  a.n = (int)h->n; a.xbar_only = xbar_only ? 1 : 0;
  a.x = h->x; a.c = h->c; a.aty = h->aty; a.lb = h->lb; a.ub = h->ub;
  a.tau = step_size / primal_weight; a.theta = theta;
  a.x_next = h->x_next; a.xbar = h->xbar;
  a.avg_w = h->pend_w; a.sum_x = (h->pend_x && !xbar_only) ? h->sum_x : nullptr;
  EpiArgs de{};
  de.y = h->y; de.b = h->b; de.y_next = h->y_next; de.sigma = primal_weight * step_size; de.num_eq = (int)h->num_eq;
  de.partials = h->pA; de.stride = h->A.slots(); de.lo_offset = h->A.slots();
  if (h->pend_y) { de.sum_y = h->sum_y; de.avg_w = h->pend_w; }
  a.A = trial_product(h, h->A, h->xbar, de);
  EpiArgs te{};
  te.x = h->x; te.x_next = h->x_next; te.aty = h->aty; te.aty_next = h->aty_next;
  te.partials = h->pAt; te.stride = h->pAt_stride; te.lo_offset = 3 * h->pAt_stride;
  a.T = trial_product(h, h->At, h->y_next, te);
  a.sp.ptr[0] = h->pAt;                       a.sp.count[0] = h->At.slots();
  a.sp.ptr[1] = h->pAt + h->pAt_stride;       a.sp.count[1] = h->At.slots();
  a.sp.ptr[2] = h->pA;                        a.sp.count[2] = h->A.slots();
  a.sp.ptr[3] = h->pAt + 2 * h->pAt_stride;   a.sp.count[3] = h->At.slots();
  a.sp.ptr[4] = h->pQ;                        a.sp.count[4] = 0;
  for (int q : {0, 1, 3}) a.sp.ptr_lo[q] = a.sp.ptr[q] + 3 * h->pAt_stride;
  a.sp.ptr_lo[2] = h->pA + h->A.slots();
  a.sp.ptr_lo[4] = h->pQ + h->ew_grid_n;
  a.sp.out = nullptr;
  a.has_q = h->has_q ? 1 : 0;
  a.epoch = h->coop_epoch;
  h->coop_epoch += 2;
  if (h->has_q) {
    a.q_blocks = h->ew_grid_n;
    a.sp.count[4] = h->ew_grid_n;
    a.qx = h->qx; a.dx = h->tmp_n; a.qtdx = h->tmp_n2; a.pq = h->pQ;
    EpiArgs qe{};
    qe.out = h->tmp_n2;
    a.Qtdx = trial_product(h, h->Qt, h->tmp_n, qe);
    if (!xbar_only) {
      qe.out = h->qx;
      a.Qx = trial_product(h, h->Q, h->x, qe);
      h->coop_epoch += 1;
    }
  }
  a.seq_dev = h->seq_dev; a.res_host = h->res_host; a.sync = h->gsync;
  h->seq_expected += 1;
  a.launch = h->coop_launches; a.seq = h->seq_expected; a.nxcd = h->coop_nxcd; a.relaxed = h->relaxed ? 1 : 0;
  a.trace = h->coop_trace;
  for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->coop_xcd_cnt[x];
  // test knob: raise the barriers' error word in front of launch number k, as a time-out in it would (the launch
  // then runs without synchronisation and reports the error; the recovery below is what is being tested)
  static const long break_at = dev_env("PDHG_COOP_TEST_BREAK_AT") ? atol(dev_env("PDHG_COOP_TEST_BREAK_AT")) : -1;
  if (break_at >= 0 && (long)h->coop_launches == break_at) {
    static const unsigned long long nine = 9ull;
    HIP_TRY(hipMemcpyAsync(&h->gsync->error[0], &nine, sizeof nine, hipMemcpyHostToDevice, h->stream));
  }
  h->coop_launches += 1;
  const auto c1 = std::chrono::steady_clock::now();
  static const bool coh_single = dev_env("PDHG_COOP_COH") && dev_env("PDHG_COOP_COH")[0] == '1';   // dev: L1-bypassing loads in the single-trial kernel too
  if (coh_single) hipLaunchKernelGGL(trial_kernel<true>, dim3(h->coop_grid), dim3(TPB), 0, h->stream, a);
  else hipLaunchKernelGGL(trial_kernel<false>, dim3(h->coop_grid), dim3(TPB), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  const auto c2 = std::chrono::steady_clock::now();
  h->t_set += std::chrono::duration<double>(c1 - c0).count();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  h->n_graph_trials += 1;
  if (!xbar_only) h->pend_x = false;
  h->pend_y = false;                  // the launch carries the deferred average update
  rc = wait_result_word(h, out, true);
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  if (rc) return rc;
  if (h->res_error != 0.0) {
    // A grid barrier ran into its spin limit: the workgroups were not all co-resident (another process runs a
    // persistent kernel on this device, or a debugger / profiler serialises dispatch).  Once the error word is up no
    // workgroup waits any more, every workgroup still runs every phase, so the launch has ended and the elementwise
    // work that does not depend on the barriers -- the deferred average update it carried -- is applied exactly once.
    // x', y', A'y' and the sums are not trustworthy: the caller repeats the trial on the graph / plain path (its
    // inputs x, y, A'y are untouched), and this handle stays there (the barrier counters are out of step now).
    h->coop_mode = 0;
    h->coop_fallbacks += 1;
    fprintf(stderr, "[pdhg_hip] one-launch trial: a grid barrier timed out (code %g; is the device shared with another "
                    "persistent kernel?) -- this handle uses the %s path from here on\n", h->res_error,
            graph_eligible(h) ? "graph" : "separate-launch");
    return 1;
  }
  return 0;
}

// ---- what the two multi-step launchers (coop_steps, small_lp_steps) share ---------------------------------------
// Trial budget of a launch of n steps, the tables of (total_number_iterations + 1)^-exponent for its trials (host pow,
// uploaded), the pinned result words.  Budget: the steps asked for plus room for rejections (a launch that runs out
// returns at a take_step boundary and the caller launches again); 64 more table entries for finishing the take_step
// the budget ends in.  The tables cost two pow() per entry on the host: sized to the batch, not to the worst case.
static int steps_prepare(pdhg_handle *h, int n, int64_t total_number_iterations, double reduction_exponent,
                         double growth_exponent, int *max_trials_out, int *table_len_out) {
  int max_trials = n + n / 8 + 16, table_len = max_trials + 64;
  if (const char *tv = dev_env("PDHG_STEPS_TEST_TABLE")) max_trials = table_len = std::max(1, atoi(tv));   // test knob: launches end inside take_steps
  if (!h->steps_res) {
    HIP_TRY(hipHostMalloc((void **)&h->steps_res, STEPS_RES_WORDS * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
    memset(h->steps_res, 0, STEPS_RES_WORDS * sizeof(double));
  }
  if (h->steps_pow_cap < table_len) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->steps_pow_dev) (void)hipFree(h->steps_pow_dev);
    if (h->steps_pow_host) (void)hipHostFree(h->steps_pow_host);
    h->steps_pow_dev = h->steps_pow_host = nullptr;
    h->steps_pow_cap = std::max(table_len, 512);
    HIP_TRY(hipMalloc((void **)&h->steps_pow_dev, sizeof(double) * 2 * (size_t)h->steps_pow_cap));
    // (room behind the tables: coop_steps stages its FinalSpec there)
    HIP_TRY(hipHostMalloc((void **)&h->steps_pow_host, sizeof(double) * 2 * (size_t)h->steps_pow_cap + sizeof(FinalSpec) + 64, hipHostMallocDefault));
  }
  // the t-th trial of the launch runs with total_number_iterations = total + t + 1 and uses k1 = that + 1 (pdhg.jl:713-714)
  for (int t = 0; t < table_len; ++t) {
    const double k1 = (double)(total_number_iterations + t + 2);
    h->steps_pow_host[t] = pow(k1, -reduction_exponent);
    h->steps_pow_host[table_len + t] = pow(k1, -growth_exponent);
  }
  HIP_TRY(hipMemcpyAsync(h->steps_pow_dev, h->steps_pow_host, sizeof(double) * 2 * (size_t)table_len, hipMemcpyHostToDevice, h->stream));
  *max_trials_out = max_trials;
  *table_len_out = table_len;
  return 0;
}
// wait for a multi-step launch's result words: r[0..12] once sequence number and checksum match (bounded spin, then the stream)
// Every word is read through the volatile pointer (a plain read in the spin loop may be hoisted).  r14: the step size
// on entry of a take_step the launch ended inside (0: none); it is under the checksum like the other words.
static int steps_wait(pdhg_handle *h, unsigned long long seq, double r[13], double *r14) {
  const volatile unsigned long long *bits = reinterpret_cast<const volatile unsigned long long *>(h->steps_res);
  const double seq_d = (double)seq;
  unsigned long long seq_bits;
  memcpy(&seq_bits, &seq_d, 8);
  auto ready = [&]() -> bool {
    if (bits[15] != seq_bits) return false;
    unsigned long long w[13], ck = RESULT_CHECK_SALT;
    for (int k = 0; k < 13; ++k) { w[k] = bits[k]; ck ^= w[k] * (2ull * (unsigned long long)k + 1ull); }
    const unsigned long long w14 = bits[14];
    ck ^= w14 * 29ull;
    if (ck != bits[13]) return false;
    for (int k = 0; k < 13; ++k) memcpy(&r[k], &w[k], 8);
    memcpy(r14, &w14, 8);
    return r[12] == seq_d;
  };
  for (long spin = 0; spin < 400000000L; ++spin) {
    if (ready()) return 0;
    if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(h->stream) != hipErrorNotReady) break;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (ready()) return 0;
  return fail(998, "multi-step kernel finished without publishing its results");
}

// Up to n_steps adaptive take_steps in ONE launch (steps_kernel, trial_kernel.hpp).  On return *steps_done take_steps
// have been taken (fewer when the launch ran out of its trial budget, met numerical_error, or a barrier timed out: the
// caller goes on from the state left).  Returns 1 when nothing could be launched (not eligible).
// The multi-step kernel's XCD-local mode: LPs whose products are at most PDHG_COOP_LOCAL_MAX (default 32: one per compute
// unit of an XCD) items.  8 x coop_grid workgroups are launched, the dispatcher deals them round the XCDs, those on XCD 0
// work -- the census must find exactly coop_grid of them there.  Own barrier words and epoch (the single-trial kernel
// keeps the handle's all-XCD census).  PDHG_COOP_LOCAL=0 turns it off.  Returns 0 when the mode is on.
static int steps_local_prepare(pdhg_handle *h) {
  if (h->local_mode >= 0) return h->local_mode ? 0 : 1;
  h->local_mode = 0;
  const char *ev = dev_env("PDHG_COOP_LOCAL");
  if (ev && ev[0] == '0') return 1;
  const int cap = dev_env("PDHG_COOP_LOCAL_MAX") ? atoi(dev_env("PDHG_COOP_LOCAL_MAX")) : 32;
  if (h->coop_grid <= 0 || h->coop_grid > cap || dev_env("PDHG_COOP_TEST_PRETEND_WGS")) return 1;
  HIP_TRY(hipMalloc((void **)&h->lsync, sizeof(GridSync)));
  HIP_TRY(hipMemsetAsync(h->lsync, 0, sizeof(GridSync), h->stream));
  hipLaunchKernelGGL(xcd_register_kernel, dim3(8 * h->coop_grid), dim3(TPB), 0, h->stream, h->lsync);
  HIP_TRY(hipGetLastError());
  GridSync host;
  HIP_TRY(hipMemcpyAsync(&host, h->lsync, sizeof(GridSync), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (host.xcd_count[0][0] != (unsigned long long)h->coop_grid) return 1;      // the dispatcher dealt them otherwise: all-XCD mode
  h->local_epoch = 0;
  h->local_mode = 1;
  if (getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] multi-step kernel: XCD-local mode, %d workgroups on XCD 0 (of %d launched)\n", h->coop_grid, 8 * h->coop_grid);
  return 0;
}

int coop_steps(pdhg_handle *h, int64_t n_steps, double reduction_exponent, double growth_exponent, double *step_size_io,
               double primal_weight, int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
               int *numerical_error_out, int64_t *steps_done, double *unfinished_entry) {
  *steps_done = 0;
  *unfinished_entry = 0.0;
  if (!coop_eligible(h) || h->has_q || !h->lazy_accept || h->pend_x != h->pend_y) return 1;
  int rc = coop_prepare(h);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  const int n = (int)std::min<int64_t>(n_steps, 1 << 20);
  int max_trials = 0, table_len = 0;
  if (!h->steps_ctl) {
    HIP_TRY(hipMalloc((void **)&h->steps_ctl, sizeof(StepsCtl)));
    HIP_TRY(hipMemsetAsync(h->steps_ctl, 0, sizeof(StepsCtl), h->stream));
  }
  std::lock_guard<std::mutex> one_at_a_time(coop_device_mutex(h->device));
  if ((rc = steps_prepare(h, n, *total_number_iterations_io, reduction_exponent, growth_exponent, &max_trials, &table_len))) return rc;
  StepsKernelArgs a{};
  a.n = (int)h->n; a.num_eq = (int)h->num_eq;
  a.xa = h->x; a.xb = h->x_next; a.ya = h->y; a.yb = h->y_next; a.atya = h->aty; a.atyb = h->aty_next;
  a.c = h->c; a.lb = h->lb; a.ub = h->ub; a.b = h->b;
  a.xbar = h->xbar; a.sum_x = h->sum_x; a.sum_y = h->sum_y;
  EpiArgs none{};
  a.A = trial_product(h, h->A, nullptr, none);
  a.T = trial_product(h, h->At, nullptr, none);
  h->A.coop_uses -= 1; h->At.coop_uses -= 1;       // (trial_product counted one use: the launch's own count comes back with the results)
  a.uses_a = a.A.uses; a.uses_t = a.T.uses;
  a.pA = h->pA; a.pAt = h->pAt; a.pA_slots = h->A.slots(); a.pAt_stride = h->pAt_stride;
  FinalSpec sp{};
  sp.ptr[0] = h->pAt;                       sp.count[0] = h->At.slots();
  sp.ptr[1] = h->pAt + h->pAt_stride;       sp.count[1] = h->At.slots();
  sp.ptr[2] = h->pA;                        sp.count[2] = h->A.slots();
  sp.ptr[3] = h->pAt + 2 * h->pAt_stride;   sp.count[3] = h->At.slots();
  sp.ptr[4] = h->pQ;                        sp.count[4] = 0;
  for (int q : {0, 1, 3}) sp.ptr_lo[q] = sp.ptr[q] + 3 * h->pAt_stride;
  sp.ptr_lo[2] = h->pA + h->A.slots();
  sp.ptr_lo[4] = h->pQ + h->ew_grid_n;
  sp.out = nullptr;
  memcpy(h->steps_pow_host + 2 * (size_t)table_len, &sp, sizeof sp);        // staged behind the pow tables (pinned)
  HIP_TRY(hipMemcpyAsync(&h->steps_ctl->sp, h->steps_pow_host + 2 * (size_t)table_len, sizeof sp, hipMemcpyHostToDevice, h->stream));
  a.primal_weight = primal_weight; a.step_size = *step_size_io;
  a.n_steps = n; a.max_trials = max_trials; a.table_len = table_len;
  a.pend = h->pend_x ? 1 : 0; a.pend_w = h->pend_w;
  a.wsum_x = h->sum_x_weights; a.wsum_y = h->sum_y_weights;
  a.pow_red = h->steps_pow_dev; a.pow_growth = h->steps_pow_dev + table_len;
  const bool local = steps_local_prepare(h) == 0;
  a.epoch = local ? h->local_epoch : h->coop_epoch;
  a.sync = local ? h->lsync : h->gsync; a.ctl = h->steps_ctl; a.res_host = h->steps_res;
  a.local_g = local ? h->coop_grid : 0; a.local_home = 0;
  // test knob: the kernel expects eight workgroups more than are launched on the home XCD -- its first barrier times out
  if (local && dev_env("PDHG_COOP_LOCAL_TEST_BAD")) a.local_g += 8;
  if (local) { a.local_ticket_base = h->local_tickets; h->local_tickets += (unsigned long long)h->coop_grid; }
  a.seq = ++h->steps_seq;
  a.nxcd = h->coop_nxcd; a.relaxed = h->relaxed ? 1 : 0;
  a.trace = h->coop_trace;
  for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->coop_xcd_cnt[x];
  const auto c1 = std::chrono::steady_clock::now();
  if (local) hipLaunchKernelGGL(steps_kernel<true>, dim3(8 * h->coop_grid), dim3(TPB), 0, h->stream, a);
  else hipLaunchKernelGGL(steps_kernel<false>, dim3(h->coop_grid), dim3(TPB), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  const auto c2 = std::chrono::steady_clock::now();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  double r[13], r14 = 0.0;
  if ((rc = steps_wait(h, a.seq, r, &r14))) return rc;
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  const int64_t steps = (int64_t)r[1], trials = (int64_t)r[2];
  const bool flip = r[3] != 0.0, aborted = r[9] != 0.0 || r[11] != 0.0;
  h->steps_launches += 1; h->steps_trials += trials; h->n_graph_trials += trials;
  if (local) { h->local_epoch = (unsigned long long)r[10]; h->local_launches += 1; }
  else h->coop_epoch = (unsigned long long)r[10];
  h->A.coop_uses += (unsigned long long)trials + (aborted ? 1ull : 0ull);
  h->At.coop_uses += (unsigned long long)trials + (aborted ? 1ull : 0ull);
  if (flip) { std::swap(h->x, h->x_next); std::swap(h->y, h->y_next); std::swap(h->aty, h->aty_next); }
  h->pend_x = h->pend_y = r[4] != 0.0;
  h->pend_w = r[5];
  h->sum_x_count += steps; h->sum_y_count += steps;
  h->sum_x_weights = r[6]; h->sum_y_weights = r[7];
  if (trials > 0 || aborted) h->state_version += 1;     // (bump_version of a single handle)
  *step_size_io = r[0];
  *total_number_iterations_io += trials;
  *cumulative_kkt_passes_io += (double)trials;
  *steps_done = steps;
  // (also after a barrier time-out: the launch may have aborted inside a take_step whose earlier trials were rejected,
  //  and the word is written by the same thread as the other result words)
  *unfinished_entry = r14;
  if (r[8] != 0.0) { *numerical_error_out = 1; *steps_done = steps + 1; }   // the failing take_step counts as taken (it is not repeated)
  if (aborted && local) {
    // the XCD-local form failed (a workgroup of the launch was not where the census saw it): the all-XCD form from here on
    h->local_mode = 0;
    fprintf(stderr, "[pdhg_hip] multi-step trial kernel, XCD-local mode: a barrier timed out (code %g) -- all-XCD mode from here on\n", r[11]);
  } else if (aborted) {
    h->coop_mode = 0;
    h->coop_fallbacks += 1;
    fprintf(stderr, "[pdhg_hip] multi-step trial kernel: a grid barrier timed out (code %g; is the device shared with another "
                    "persistent kernel?) -- this handle uses the %s path from here on\n", r[11],
            graph_eligible(h) ? "graph" : "separate-launch");
  }
  return 0;
}

// ---- small LPs: a batch of take_steps in one workgroup, vectors in LDS (small_lp_kernel.hpp) ---------------------
int flush_pending(const Shards &L);
bool small_lp_eligible(pdhg_handle *h) {
  if (h->small_lp_mode < 0) {
    const char *ev = getenv("PDHG_SMALL_LP");
    const size_t lds = sizeof(double) * (9 * (size_t)h->n + 4 * (size_t)h->m);
    bool on = !h->grp && !h->has_q && h->n > 0 && h->m > 0 && h->A.segs.empty() && h->At.segs.empty() && !h->A.tiled && !h->At.tiled && h->A.slabs.empty() &&
              h->At.slabs.empty() && h->A.max_row_nnz <= SMALL_MAX_ROW && h->At.max_row_nnz <= SMALL_MAX_ROW &&
              lds <= (size_t)144 * 1024;
    if (ev) on = on && ev[0] != '0';
    h->small_lp_mode = on ? 1 : 0;
  }
  return h->small_lp_mode == 1 && !h->profile;
}

// returns 1 when not eligible (nothing launched)
int small_lp_steps(pdhg_handle *h, int64_t n_steps, double reduction_exponent, double growth_exponent, double *step_size_io,
                   double primal_weight, int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
                   int *numerical_error_out, int64_t *steps_done, double *unfinished_entry) {
  *steps_done = 0;
  *unfinished_entry = 0.0;
  if (!small_lp_eligible(h)) return 1;
  HIP_TRY(hipSetDevice(h->device));
  int rc;
  if (h->pend_x != h->pend_y) { Shards L = shards_of(h); if ((rc = flush_pending(L))) return rc; }
  const int n = (int)std::min<int64_t>(n_steps, 1 << 20);
  int max_trials = 0, table_len = 0;
  if ((rc = steps_prepare(h, n, *total_number_iterations_io, reduction_exponent, growth_exponent, &max_trials, &table_len))) return rc;
  const size_t lds = sizeof(double) * (9 * (size_t)h->n + 4 * (size_t)h->m);
  {
    static size_t limit[64] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    size_t &cur = limit[h->device & 63];
    if (cur < lds) {
      HIP_TRY(hipFuncSetAttribute((const void *)small_lp_steps_kernel<SMALL_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(hipFuncSetAttribute((const void *)small_lp_steps_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      cur = lds;
    }
  }

  SmallLpArgs a{};
  a.n = (int)h->n; a.m = (int)h->m; a.num_eq = (int)h->num_eq;
  a.A = h->A.view(); a.T = h->At.view();
  a.x = h->x; a.y = h->y; a.aty = h->aty; a.sum_x = h->sum_x; a.sum_y = h->sum_y;
  a.c = h->c; a.lb = h->lb; a.ub = h->ub; a.b = h->b;
  a.primal_weight = primal_weight; a.step_size = *step_size_io;
  a.n_steps = n; a.max_trials = max_trials; a.table_len = table_len;
  a.pend = h->pend_x ? 1 : 0; a.pend_w = h->pend_w;
  a.wsum_x = h->sum_x_weights; a.wsum_y = h->sum_y_weights;
  a.pow_red = h->steps_pow_dev; a.pow_growth = h->steps_pow_dev + table_len;
  a.res_host = h->steps_res;
  a.seq = ++h->steps_seq;
  h->pend_x = h->pend_y = false;             // the launch applies it
  const auto c1 = std::chrono::steady_clock::now();
  static const int few_env = dev_env("PDHG_SMALL_FEW_ROWS") ? atoi(dev_env("PDHG_SMALL_FEW_ROWS")) : SMALL_FEW_ROWS;   // dev knob
  if (std::max(h->n, h->m) <= few_env) hipLaunchKernelGGL(small_lp_steps_kernel<256>, dim3(1), dim3(256), lds, h->stream, a);
  else hipLaunchKernelGGL(small_lp_steps_kernel<SMALL_TPB>, dim3(1), dim3(SMALL_TPB), lds, h->stream, a);
  HIP_TRY(hipGetLastError());
  const auto c2 = std::chrono::steady_clock::now();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  double r[13], r14 = 0.0;
  if ((rc = steps_wait(h, a.seq, r, &r14))) return rc;
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  const int64_t steps = (int64_t)r[1], trials = (int64_t)r[2];
  h->small_lp_launches += 1; h->n_graph_trials += trials;
  h->sum_x_count += steps; h->sum_y_count += steps;
  h->sum_x_weights = r[6]; h->sum_y_weights = r[7];
  h->state_version += 1;
  *step_size_io = r[0];
  *total_number_iterations_io += trials;
  *cumulative_kkt_passes_io += (double)trials;
  *steps_done = steps;
  *unfinished_entry = r14;
  if (r[8] != 0.0) { *numerical_error_out = 1; *steps_done = steps + 1; }
  return 0;
}

bool graph_eligible(pdhg_handle *h) {
  if (h->graph_mode < 0) {
    const char *ev = getenv("PDHG_GRAPH");
    // stream layouts only.  The sweep can run as graph nodes too (PDHG_GRAPH_TILED=1) but gains nothing: with the
    // take_step loop in C the separate launches already overlap the kernels -- random 1M x 1M 5 709 it/s as a graph
    // against 5 637, 4M x 4M 1 667 / 1 672, config S 611 / 613 (profiles/r03_trial_kernel.txt).
    const bool tiled_ok = dev_env("PDHG_GRAPH_TILED") != nullptr;
    bool on = !h->grp && !h->has_q && h->n > 0 && h->A.segs.empty() && h->At.segs.empty() && (tiled_ok || (!h->A.tiled && !h->At.tiled));
    if (ev) on = on && ev[0] != '0';
    h->graph_mode = on ? 1 : 0;
  }
  return h->graph_mode == 1 && !h->has_q && !h->profile;
}

void graph_destroy(pdhg_handle::TrialGraph &G) {
  if (G.exec) (void)hipGraphExecDestroy(G.exec);
  if (G.graph) (void)hipGraphDestroy(G.graph);
  G = pdhg_handle::TrialGraph();
}

// the argument packs of the three nodes whose scalars change from trial to trial
struct GraphArgs {
  pdhg_handle *h;
  int n;
  dim3 primal_grid;
  EpiArgs dual_epi;
  GraphArgs(pdhg_handle *h_, double sigma) : h(h_), n((int)h_->n), primal_grid(ew_grid((h_->n + 1) / 2)) {
    dual_epi = EpiArgs{};
    dual_epi.y = h->y; dual_epi.b = h->b; dual_epi.y_next = h->y_next; dual_epi.sigma = sigma;
    dual_epi.num_eq = (int)h->num_eq; dual_epi.partials = h->pA; dual_epi.stride = h->A.slots(); dual_epi.lo_offset = h->A.slots();
    if (h->pend_y) { dual_epi.sum_y = h->sum_y; dual_epi.avg_w = h->pend_w; }
  }
};

// the sweep kernel of a layout for graph nodes: function pointer (with the dynamic-LDS opt-in done)
template <int MODE>
int tiled_node_func(pdhg_handle *h, const CsrDev &D, const void **fn, size_t *lds) {
  *lds = tiled_lds_bytes(D);
  *fn = D.tw_mode == 1 ? (const void *)spmv_tiled_kernel<MODE, 1>
                       : (D.tw_mode == 2 ? (const void *)spmv_tiled_kernel<MODE, 2> : (const void *)spmv_tiled_kernel<MODE, 0>);
  return ensure_lds_limit(h, MODE, D.tw_mode, *lds, *fn);
}

// nodes of one fused SpMV: stream kernel (one node) or its column-slab passes (a chain),
// beside the long-row pair.  `done` receives the nodes the next stage must wait for;
// main_node / long_node (optional) receive the nodes that carry the epilogue's scalars.
template <int MODE, int TAG>
int graph_add_spmv(pdhg_handle *h, hipGraph_t graph, const CsrDev &D, const double *xin, const EpiArgs &e,
                   const std::vector<hipGraphNode_t> &deps, std::vector<hipGraphNode_t> &done,
                   hipGraphNode_t *main_node, hipGraphNode_t *long_node) {
  const int rm = h->remap ? 1 : 0, rx = h->relaxed ? 1 : 0;
  if (D.tiled) {
    if (D.grid > 0) {
      const void *fn;
      size_t lds;
      int rc = tiled_node_func<MODE>(h, D, &fn, &lds);
      if (rc) return rc;
      hipGraphNode_t nd = nullptr;
      HIP_TRY(graph_add_kernel_lds(graph, &nd, deps, fn, dim3(D.grid), dim3(TW_WPB * WAVE), lds, (const int2 *)D.wave_rows,
                                   (const int *)D.wave_ent, (const int *)D.wave_step_off, (const int *)D.step_tile,
                                   (const int *)D.wg_step_off, D.nwaves, D.tile_shift, D.tw_rows, (const unsigned *)D.pk,
                                   (const double *)D.tv, xin, e));
      if (main_node) *main_node = nd;
      done.push_back(nd);
    }
  } else if (!D.slabs.empty()) {
    const int P = (int)D.slabs.size();
    std::vector<hipGraphNode_t> prev = deps;
    for (int p = 0; p < P; ++p) {
      const SlabDev &S = D.slabs[(size_t)p];
      hipGraphNode_t nd = nullptr;
      if (S.sj.on()) {
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        EpiArgs le = e;
        le.init = D.slab_partial;
        if (p + 1 < P) {
          const void *fn = p == 0 ? (const void *)spmv_sj_kernel<MODE_PLAIN, false, TAG> : (const void *)spmv_sj_kernel<MODE_PLAIN, true, TAG>;
          HIP_TRY(graph_add_kernel(graph, &nd, prev, fn, dim3(S.sj.grid), dim3(TPB), sj_view(S.sj), xin, rm, 0, pe));
        } else {
          HIP_TRY(graph_add_kernel(graph, &nd, prev, (const void *)spmv_sj_kernel<MODE, true, TAG>, dim3(S.sj.grid), dim3(TPB),
                                   sj_view(S.sj), xin, rm, S.grid, le));
          if (main_node) *main_node = nd;
        }
        prev.assign(1, nd);
        continue;
      }
      if (S.pipe_grid > 0) {
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        EpiArgs le = e;
        le.init = D.slab_partial;
        if (p + 1 < P) {
          const void *fn = p == 0 ? (const void *)spmv_stream_pipe_kernel<MODE_PLAIN, false, TAG> : (const void *)spmv_stream_pipe_kernel<MODE_PLAIN, true, TAG>;
          HIP_TRY(graph_add_kernel(graph, &nd, prev, fn, dim3(S.pipe_grid), dim3(TPB), S.view(D.rows), xin, (const int4 *)S.ext, S.nblk,
                                   S.per_xcd, rm, rx, 0, pe));
        } else {
          HIP_TRY(graph_add_kernel(graph, &nd, prev, (const void *)spmv_stream_pipe_kernel<MODE, true, TAG>, dim3(S.pipe_grid), dim3(TPB),
                                   S.view(D.rows), xin, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx, S.grid, le));
          if (main_node) *main_node = nd;
        }
        prev.assign(1, nd);
        continue;
      }
      if (p + 1 < P) {
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        const void *fn = p == 0 ? (const void *)spmv_stream_kernel<MODE_PLAIN, false, TAG> : (const void *)spmv_stream_kernel<MODE_PLAIN, true, TAG>;
        HIP_TRY(graph_add_kernel(graph, &nd, prev, fn, dim3(S.grid), dim3(TPB), S.view(D.rows), xin, (const int2 *)S.blks,
                                 S.nblk, S.per_xcd, rm, rx, pe));
      } else {
        EpiArgs le = e;
        le.init = D.slab_partial;
        HIP_TRY(graph_add_kernel(graph, &nd, prev, (const void *)spmv_stream_kernel<MODE, true, TAG>, dim3(S.grid), dim3(TPB),
                                 S.view(D.rows), xin, (const int2 *)S.blks, S.nblk, S.per_xcd, rm, rx, le));
        if (main_node) *main_node = nd;
      }
      prev.assign(1, nd);
    }
    done.push_back(prev[0]);
  } else if (D.sj.on()) {
    hipGraphNode_t nd = nullptr;
    HIP_TRY(graph_add_kernel(graph, &nd, deps, (const void *)spmv_sj_kernel<MODE, false, TAG>, dim3(D.sj.grid), dim3(TPB),
                             sj_view(D.sj), xin, rm, D.grid, e));
    if (main_node) *main_node = nd;
    done.push_back(nd);
  } else if (D.pipe_grid > 0) {
    hipGraphNode_t nd = nullptr;
    HIP_TRY(graph_add_kernel(graph, &nd, deps, (const void *)spmv_stream_pipe_kernel<MODE, false, TAG>, dim3(D.pipe_grid), dim3(TPB),
                             D.view(), xin, (const int4 *)D.ext, D.nblk, D.per_xcd, rm, rx, D.grid, e));
    if (main_node) *main_node = nd;
    done.push_back(nd);
  } else if (D.grid > 0) {
    hipGraphNode_t nd = nullptr;
    HIP_TRY(graph_add_kernel(graph, &nd, deps, (const void *)spmv_stream_kernel<MODE, false, TAG>, dim3(D.grid), dim3(TPB),
                             D.view(), xin, (const int2 *)D.blks, D.nblk, D.per_xcd, rm, rx, e));
    if (main_node) *main_node = nd;
    done.push_back(nd);
  }
  if (D.nlong > 0) {
    hipGraphNode_t part = nullptr, fin = nullptr;
    HIP_TRY(graph_add_kernel(graph, &part, deps, (const void *)spmv_long_partial_kernel<TAG>, dim3(D.nchunks), dim3(TPB),
                             D.view(), xin, (const int *)D.chunk_row, (const int *)D.chunk_off, D.chunk_partial));
    HIP_TRY(graph_add_kernel(graph, &fin, {part}, (const void *)spmv_long_final_kernel<MODE>, dim3(D.long_grid), dim3(TPB),
                             (const int *)D.long_row, (const int *)D.long_chunk_ptr, D.nlong,
                             (const double *)D.chunk_partial, e, D.grid));
    if (long_node) *long_node = fin;
    done.push_back(fin);
  }
  return 0;
}

// the dual stream node's parameters again, with a new sigma
int graph_set_dual(pdhg_handle *h, pdhg_handle::TrialGraph &G, const EpiArgs &dual_epi) {
  const CsrDev &A = h->A;
  const int rm = h->remap ? 1 : 0, rx = h->relaxed ? 1 : 0;
  if (G.n_dual) {
    if (A.tiled) {
      const void *fn;
      size_t lds;
      int rc = tiled_node_func<MODE_DUAL>(h, A, &fn, &lds);
      if (rc) return rc;
      HIP_TRY(graph_set_kernel_lds(G.exec, G.n_dual, fn, dim3(A.grid), dim3(TW_WPB * WAVE), lds, (const int2 *)A.wave_rows,
                                   (const int *)A.wave_ent, (const int *)A.wave_step_off, (const int *)A.step_tile,
                                   (const int *)A.wg_step_off, A.nwaves, A.tile_shift, A.tw_rows, (const unsigned *)A.pk,
                                   (const double *)A.tv, (const double *)h->xbar, dual_epi));
    } else if (!A.slabs.empty()) {
      const SlabDev &S = A.slabs.back();
      EpiArgs le = dual_epi;
      le.init = A.slab_partial;
      if (S.sj.on())
        HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_sj_kernel<MODE_DUAL, true, 0>, dim3(S.sj.grid), dim3(TPB),
                                 sj_view(S.sj), (const double *)h->xbar, rm, S.grid, le));
      else if (S.pipe_grid > 0)
        HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_pipe_kernel<MODE_DUAL, true, 0>, dim3(S.pipe_grid), dim3(TPB),
                                 S.view(A.rows), (const double *)h->xbar, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx, S.grid, le));
      else
        HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_kernel<MODE_DUAL, true, 0>, dim3(S.grid), dim3(TPB),
                                 S.view(A.rows), (const double *)h->xbar, (const int2 *)S.blks, S.nblk, S.per_xcd, rm, rx, le));
    } else if (A.sj.on()) {
      HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_sj_kernel<MODE_DUAL, false, 0>, dim3(A.sj.grid), dim3(TPB),
                               sj_view(A.sj), (const double *)h->xbar, rm, A.grid, dual_epi));
    } else if (A.pipe_grid > 0) {
      HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_pipe_kernel<MODE_DUAL, false, 0>, dim3(A.pipe_grid), dim3(TPB),
                               A.view(), (const double *)h->xbar, (const int4 *)A.ext, A.nblk, A.per_xcd, rm, rx, A.grid, dual_epi));
    } else {
      HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_kernel<MODE_DUAL, false, 0>, dim3(A.grid), dim3(TPB),
                               A.view(), (const double *)h->xbar, (const int2 *)A.blks, A.nblk, A.per_xcd, rm, rx, dual_epi));
    }
  }
  if (G.n_dual_long)
    HIP_TRY(graph_set_kernel(G.exec, G.n_dual_long, (const void *)spmv_long_final_kernel<MODE_DUAL>,
                             dim3(A.long_grid), dim3(TPB), (const int *)A.long_row, (const int *)A.long_chunk_ptr,
                             A.nlong, (const double *)A.chunk_partial, dual_epi, A.grid));
  return 0;
}

int graph_build(pdhg_handle *h, pdhg_handle::TrialGraph &G, double tau, double theta, double sigma) {
  graph_destroy(G);
  int rcw = ensure_result_word(h);
  if (rcw) return rcw;
  HIP_TRY(hipGraphCreate(&G.graph, 0));
  GraphArgs a(h, sigma);
  const double *nullq = nullptr;
  // K1+K2
  HIP_TRY(graph_add_kernel(G.graph, &G.n_primal, {}, (const void *)primal_kernel<false, true>, a.primal_grid, dim3(TPB),
                           a.n, (const double *)h->x, (const double *)h->c, (const double *)h->aty, nullq,
                           (const double *)h->lb, (const double *)h->ub, tau, theta, h->x_next, h->xbar,
                           h->pend_w, h->pend_x ? h->sum_x : (double *)nullptr));
  // K3+K4 on CSR(A), K5+K6 on CSR(A'): the stream kernel (or its column-slab passes, a
  // chain) and the long-row pair are independent branches
  std::vector<hipGraphNode_t> dual_done, aty_done;
  {
    int rc = graph_add_spmv<MODE_DUAL, 0>(h, G.graph, h->A, h->xbar, a.dual_epi, {G.n_primal}, dual_done, &G.n_dual, &G.n_dual_long);
    if (rc) return rc;
  }
  if (dual_done.empty()) dual_done.push_back(G.n_primal);
  const CsrDev &A = h->A;
  const CsrDev &T = h->At;
  EpiArgs te{};
  te.x = h->x; te.x_next = h->x_next; te.aty = h->aty; te.aty_next = h->aty_next;
  te.partials = h->pAt; te.stride = h->pAt_stride; te.lo_offset = 3 * h->pAt_stride;
  {
    int rc = graph_add_spmv<MODE_ATY, 1>(h, G.graph, T, h->y_next, te, dual_done, aty_done, nullptr, nullptr);
    if (rc) return rc;
  }
  if (aty_done.empty()) aty_done = dual_done;
  // K6b -> pinned host memory + sequence number
  FinalSpec sp{};
  sp.ptr[0] = h->pAt;                       sp.count[0] = T.slots();
  sp.ptr[1] = h->pAt + h->pAt_stride;       sp.count[1] = T.slots();
  sp.ptr[2] = h->pA;                        sp.count[2] = A.slots();
  sp.ptr[3] = h->pAt + 2 * h->pAt_stride;   sp.count[3] = T.slots();
  sp.ptr[4] = h->pQ;                        sp.count[4] = 0;
  for (int q : {0, 1, 3}) sp.ptr_lo[q] = sp.ptr[q] + 3 * h->pAt_stride;
  sp.ptr_lo[2] = h->pA + A.slots();
  sp.ptr_lo[4] = h->pQ + h->ew_grid_n;
  sp.out = nullptr;
  hipGraphNode_t fin = nullptr;
  HIP_TRY(graph_add_kernel(G.graph, &fin, aty_done, (const void *)final_reduce_host_kernel, dim3(1), dim3(FINAL_TPB), sp,
                           h->seq_dev, h->res_host));
  HIP_TRY(hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0));
  G.x = h->x; G.y = h->y; G.aty = h->aty;
  G.tau = tau; G.theta = theta; G.sigma = sigma;
  G.add_x = h->pend_x; G.add_wx = h->pend_w;
  G.add_y = h->pend_y; G.add_wy = h->pend_w;
  return 0;
}

int graph_trial(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  const double tau = step_size / primal_weight, sigma = primal_weight * step_size;
  pdhg_handle::TrialGraph *G = nullptr;
  for (int k = 0; k < 2; ++k)
    if (h->tgraph[k].exec && h->tgraph[k].x == h->x && h->tgraph[k].y == h->y && h->tgraph[k].aty == h->aty)
      G = &h->tgraph[k];
  if (!G) {
    G = !h->tgraph[0].exec ? &h->tgraph[0] : (!h->tgraph[1].exec ? &h->tgraph[1] : &h->tgraph[0]);
    int rc = graph_build(h, *G, tau, theta, sigma);
    if (rc) return rc;
  } else {
    const auto c0 = std::chrono::steady_clock::now();
    GraphArgs a(h, sigma);
    if (G->tau != tau || G->theta != theta || G->add_x != h->pend_x || (h->pend_x && G->add_wx != h->pend_w)) {
      const double *nullq = nullptr;
      HIP_TRY(graph_set_kernel(G->exec, G->n_primal, (const void *)primal_kernel<false, true>, a.primal_grid, dim3(TPB),
                               a.n, (const double *)h->x, (const double *)h->c, (const double *)h->aty, nullq,
                               (const double *)h->lb, (const double *)h->ub, tau, theta, h->x_next, h->xbar,
                               h->pend_w, h->pend_x ? h->sum_x : (double *)nullptr));
      G->tau = tau; G->theta = theta;
      G->add_x = h->pend_x; G->add_wx = h->pend_w;
    }
    if (G->sigma != sigma || G->add_y != h->pend_y || (h->pend_y && G->add_wy != h->pend_w)) {
      int rc = graph_set_dual(h, *G, a.dual_epi);
      if (rc) return rc;
      G->sigma = sigma;
      G->add_y = h->pend_y; G->add_wy = h->pend_w;
    }
    h->t_set += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
  }
  h->seq_expected += 1;
  const auto c1 = std::chrono::steady_clock::now();
  HIP_TRY(hipGraphLaunch(G->exec, h->stream));
  const auto c2 = std::chrono::steady_clock::now();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  h->n_graph_trials += 1;
  h->pend_x = h->pend_y = false;     // the launch carries the deferred average update
  int rcw = wait_result_word(h, out);
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  return rcw;
}

// queues_work: the entry point may put work on the shards' own streams (everything but a trial step and a lazy accept)
int check_handle(pdhg_handle *h, bool queues_work = true) {
  if (!h) return fail(-1, "null handle");
  if (queues_work && h->grp && !h->grp->coop_dev.empty()) {
    DistGroup &g = *h->grp;
    if (g.join_pending) {          // the members' streams wait for the last persistent group launch (group_kernel.hpp)
      for (GroupDevLaunch &D : g.coop_dev) {
        HIP_TRY(hipSetDevice(D.device));
        for (int i : D.members)
          if (g.sh[(size_t)i]->stream != D.stream) HIP_TRY(hipStreamWaitEvent(g.sh[(size_t)i]->stream, D.ev_done, 0));
      }
      g.join_pending = false;
    }
    g.members_dirty = true;
  }
  HIP_TRY(hipSetDevice(h->device));
  return 0;
}

int sync_all(const Shards &L) {
  FOR_SHARDS(L, s) HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

// Settle a deferred K7 (lazy accept): sum_x += w * x and / or sum_y += w * y on the
// iterate that is current now.  Called by every entry point other than the trial itself.
int flush_pending(const Shards &L) {
  FOR_SHARDS(L, h) {
    if (!h->pend_x && !h->pend_y) continue;
    ProfScope ps(h, PDHG_K_ACCEPT);
    const int64_t o = h->clo;
    const int nn = h->pend_x ? (int)h->cn : 0, mm = h->pend_y ? (int)h->m : 0;
    hipLaunchKernelGGL(accept_kernel, dim3(ew_grid(std::max<int64_t>(std::max(nn, mm), 1))), dim3(TPB), 0, h->stream, nn, mm,
                       h->pend_w, h->x + o, h->sum_x + o, h->y, h->sum_y);
    HIP_TRY(hipGetLastError());
    h->pend_x = h->pend_y = false;
  }
  return 0;
}

void bump_version(const Shards &L) {   // x, y, the running sums or A change: cached A*x / A'*y are stale
  for (int i = 0; i < L.count; ++i) L.p[i]->state_version += 1;
}

// ---- moving distributed vectors -------------------------------------------------
// "column vectors": n-vectors whose valid part on a shard is its own slice;
// "row vectors": m-vectors, each shard holds its rows.

// full copy of a column vector on every shard's device: dst[0..n) (dst holds n_alloc)
template <typename Src, typename Dst>
int gather_cols_device(const Shards &L, Src src, Dst dst) {
  if (!L.g) {
    pdhg_handle *s = L.p[0];
    if (src(s) != dst(s))
      HIP_TRY(hipMemcpyAsync(dst(s), src(s), sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice, s->stream));
    return 0;
  }
  FOR_SHARDS(L, s) {
    if (src(s) != dst(s) && s->cn > 0)
      HIP_TRY(hipMemcpyAsync(dst(s) + s->clo, src(s) + s->clo, sizeof(double) * (size_t)s->cn,
                             hipMemcpyDeviceToDevice, s->stream));
  }
  return dist_all_gather(*L.g, dst, L.g->S);
}

// a column vector to a host array of length n (every process gets all of it); caller syncs
template <typename Src>
int cols_to_host(const Shards &L, Src src, double *host) {
  if (!L.g || L.g->all_local()) {
    FOR_SHARDS(L, s) {
      if (s->cn > 0)
        HIP_TRY(hipMemcpyAsync(host + s->clo, src(s) + s->clo, sizeof(double) * (size_t)s->cn, hipMemcpyDeviceToHost, s->stream));
    }
    return 0;
  }
  int rc = gather_cols_device(L, src, [](pdhg_handle *s) { return s->dn_buf; });
  if (rc) return rc;
  pdhg_handle *s = L.p[0];
  HIP_TRY(hipMemcpyAsync(host, s->dn_buf, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToHost, s->stream));
  return 0;
}

// a row vector to a host array of length m_global; caller syncs
template <typename Src>
int rows_to_host(const Shards &L, Src src, double *host) {
  if (!L.g || L.g->all_local()) {
    FOR_SHARDS(L, s) {
      if (s->m > 0)
        HIP_TRY(hipMemcpyAsync(host + s->row_lo, src(s), sizeof(double) * (size_t)s->m, hipMemcpyDeviceToHost, s->stream));
    }
    return 0;
  }
  pdhg_handle *s = L.p[0];
  HIP_TRY(hipSetDevice(s->device));
  if (s->m > 0)
    HIP_TRY(hipMemcpyAsync(s->dm_buf + s->row_lo, src(s), sizeof(double) * (size_t)s->m, hipMemcpyDeviceToDevice, s->stream));
  int rc = dist_all_gather_rows(*L.g, [](pdhg_handle *q) { return q->dm_buf; });
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(host, s->dm_buf, sizeof(double) * (size_t)s->m_global, hipMemcpyDeviceToHost, s->stream));
  return 0;
}

// host arrays (global length) to the shards: column vectors are stored in full, row vectors by rows
template <typename Dst>
int cols_from_host(const Shards &L, const double *host, Dst dst) {
  FOR_SHARDS(L, s) {
    if (s->n > 0) HIP_TRY(hipMemcpyAsync(dst(s), host, sizeof(double) * (size_t)s->n, hipMemcpyHostToDevice, s->stream));
  }
  return 0;
}
template <typename Dst>
int rows_from_host(const Shards &L, const double *host, Dst dst) {
  FOR_SHARDS(L, s) {
    if (s->m > 0)
      HIP_TRY(hipMemcpyAsync(dst(s), host + s->row_lo, sizeof(double) * (size_t)s->m, hipMemcpyHostToDevice, s->stream));
  }
  return 0;
}

// A'y for a row vector y (each shard its rows) into the column vector `out`
// (valid on the owned slice; out holds n_alloc in a group).
template <typename Yin, typename Out>
int dual_product(const Shards &L, Yin yin, Out out) {
  int rc;
  FOR_SHARDS(L, s) { if ((rc = launch_aty_plain(s, yin(s), out(s)))) return rc; }
  if (L.g) {
    ProfScope ps(L.p[0], PDHG_K_REDUCE_SCATTER);
    if ((rc = dist_reduce_scatter(*L.g, out, L.g->S))) return rc;
  }
  return 0;
}

// Layout choice for one CSR: the tiled sweep pays off when the gathered vector
// (cols doubles) is comparable to or larger than an XCD's 4 MiB L2.
// PDHG_SPMV=stream|tiled forces a layout; PDHG_TILE_SHIFT sets log2(tile cols).
int choose_tile_cols(int64_t cols, int64_t nnz, int64_t rows) {
  const char *mode = getenv("PDHG_SPMV");
  const char *ts = dev_env("PDHG_TILE_SHIFT"), *tc = getenv("PDHG_TILE_COLS");
  int64_t tile = tc ? atoll(tc) : (ts ? (1LL << std::min(22, std::max(6, atoi(ts)))) : 65536);
  bool thin = false;
  if (!ts && !tc && rows > 0) {
    // One step of the sweep costs about the same for any cell of up to TW_U x 64
    // entries, and a smaller tile keeps the gathered vector in L2 more reliably, so
    // the tile is as narrow as a cell of ~100-110 entries allows (entries per wave /
    // number of tiles; tile widths are multiples of 4096 columns, not powers of two).
    // Measured on MI355X (profiles/r02_tile_rule.txt), time per nonzero against
    // entries per cell on the same matrix: 40 -> +20 %, 48-64 -> +10-18 %,
    // 88-120 -> best, 160 -> +15-20 %.  The width is capped where the L2 stops
    // holding the tile against the entry stream: 76K columns (608 KiB) when several
    // residency rounds are in flight (config S: 0.73 ms at 72-80K, 0.96 ms at 96K;
    // 16M x 16M: 1.33 ms at 80K, 1.58 ms at 96K), 144K columns for a single round
    // (row shards of a multi-GPU run).
    const int64_t slots = 256LL * 2 * TW_WPB;
    const int64_t rounds = std::max<int64_t>(1, (rows + slots * TW_MAX_ROWS - 1) / (slots * TW_MAX_ROWS));
    const int64_t rpw = std::max<int64_t>(64, (rows + slots * rounds - 1) / (slots * rounds));
    const double per_wave = (double)nnz / (double)rows * (double)rpw;
    const double target = dev_env("PDHG_TILE_FILL") ? atof(dev_env("PDHG_TILE_FILL")) : (rounds == 1 ? 110.0 : 100.0);
    // Many rounds (>= 5, i.e. beyond ~21M rows): the cells thin out at the 76K cap and the balance tips
    // towards wider tiles -- 24M x 24M 2.51 ms at 76K columns / 2.39 at 96K, 30M x 30M 3.47 / 3.08 / 2.92 at
    // 76K / 96K / 112K (128K: 3.46), 20M and below indifferent or worse -- so the cap stretches to what
    // gives a cell ~45 entries, up to 112K.
    const int64_t unit = 4096;
    int64_t cap = tile_width_cap(rows);
    if (rounds >= 2) {
      const int64_t stretch = ((int64_t)((double)cols * 45.0 / std::max(per_wave, 1.0)) + unit - 1) / unit * unit;
      cap = std::max(cap, std::min<int64_t>(112 * 1024, stretch));
    }
    const int64_t want = (int64_t)((double)cols * target / std::max(per_wave, 1.0));
    tile = std::min(cap, std::max<int64_t>(2 * unit, (want + unit / 2) / unit * unit));
    // Few rows against a very long vector: even the widest tile leaves a wave a handful of
    // entries per step and the sweep is all barriers (100K x 10M, 9 per cell: 0.081 ms swept,
    // 0.029 ms streamed; 1M x 30M, 12 per cell: 0.248 / 0.220; 500K x 10M, 18 per cell: 0.099 / 0.109).
    thin = per_wave * (double)tile / (double)std::max<int64_t>(cols, 1) < 15.0;
  }
  tile = std::min<int64_t>(std::max<int64_t>(tile, 64), 1LL << 22);   // leave >= 10 bits for row_local
  if ((cols + tile - 1) / tile > 65536) return 0;        // tile table would be huge
  if (mode && !strcmp(mode, "stream")) return 0;
  if (mode && !strcmp(mode, "tiled")) return (int)tile;
  // The sweep is tried whenever the gathered vector is beyond ~3 MiB; build_tiled() then
  // declines for matrices it does not suit (a row with a long run inside one tile, rows
  // that stay inside a band of columns) and the stream layout takes over.  Measured
  // crossover on square 10-per-row LPs, whole iterations: 250K columns stream 13.9k / swept
  // 13.3k it/s (the stream layout's trial is one graph launch), 500K columns 8.3k / 8.9k,
  // 1M 4.6k / 5.6k.  Row length is no criterion: 100 per row (10M x 1M transposed) streams
  // at 1.91 ms and sweeps at 0.62 ms; 1 000 per row 1.85 / 1.33 ms (profiles/r02_locality.txt).
  const bool big_vector = cols * 8 > (3LL << 20);
  return (big_vector && rows > 0 && !thin) ? (int)tile : 0;
}

int host_threads() {
  int threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *ev = getenv("PDHG_HOST_THREADS")) threads = std::max(1, atoi(ev));
  return threads;
}

template <typename F>
void run_threads(int T, F f) {
  if (T <= 1) { f(0); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t) pool.emplace_back([=, &f] { f(t); });
  for (std::thread &th : pool) th.join();
}

// CSC (any int64 base) -> int32 CSR of the transpose (direct) and CSR (stable sort by row).
// The sort is a two-level bucket sort on host threads, O(nnz) work in total:
// thread t scans ITS column range and appends every entry to the bucket of the
// entry's row range (T buckets; per-(thread, bucket) output segments come from a
// small T x T count table, so bucket b holds its entries in ascending column
// order); thread b then counting-sorts bucket b by row.  Each row receives its
// entries in ascending column order -- what the sequential loop produces -- and
// the result does not depend on T.
int csc_to_both(int64_t rows, int64_t cols, int64_t nnz, const int64_t *colptr,
                const int64_t *rowval, const double *nzval, int base,
                std::vector<int> &t_rowptr, ivec &t_col, dvec &t_val,
                std::vector<int> &rowptr, ivec &col, dvec &val) {
  if (rows < 0 || cols < 0 || nnz < 0) return fail(-1, "negative dimension");
  if (rows >= INT32_MAX || cols >= INT32_MAX || nnz >= INT32_MAX || rows + cols >= INT32_MAX)
    return fail(-2, "m, n, m + n or nnz >= 2^31 need the 64-bit index path (not built)");
  if (colptr[0] != base) return fail(-1, "colptr[0] != index_base");
  if (colptr[cols] - base != nnz) return fail(-1, "colptr[n] - base != nnz");
  t_rowptr.resize(cols + 1);
  for (int64_t j = 0; j <= cols; ++j) {
    const int64_t v = colptr[j] - base;
    if (v < 0 || v > nnz || (j > 0 && v < t_rowptr[j - 1])) return fail(-1, "colptr not monotone");
    t_rowptr[j] = (int)v;
  }
  t_col.resize(nnz);
  t_val.resize(nnz);
  rowptr.assign(rows + 1, 0);
  col.resize(nnz);
  val.resize(nnz);
  const int T = (nnz >= (1 << 22) && rows >= 1024 && cols >= 1024) ? host_threads() : 1;
  std::atomic<int> bad{0};
  if (T == 1) {
    for (int64_t k = 0; k < nnz; ++k) {
      const int64_t r = rowval[k] - base;
      if (r < 0 || r >= rows) return fail(-1, "rowval out of range");
      t_col[k] = (int)r;
      t_val[k] = nzval[k];
      rowptr[r + 1] += 1;
    }
    for (int64_t i = 0; i < rows; ++i) rowptr[i + 1] += rowptr[i];
    std::vector<int> next(rowptr.begin(), rowptr.end() - 1);
    for (int64_t j = 0; j < cols; ++j)
      for (int k = t_rowptr[j]; k < t_rowptr[j + 1]; ++k) {
        const int p = next[t_col[k]]++;
        col[p] = (int)j;
        val[p] = t_val[k];
      }
    return 0;
  }
  const int64_t rpb = (rows + T - 1) / T;                 // rows per bucket
  auto col_begin = [&](int t) { return (int64_t)cols * t / T; };
  std::vector<int64_t> cnt((size_t)T * T, 0);             // cnt[t*T + b]
  // pass 1: CSR(A') = the CSC input narrowed to 32 bits; bucket counts
  run_threads(T, [&](int t) {
    int64_t *c = cnt.data() + (size_t)t * T;
    for (int64_t k = t_rowptr[col_begin(t)]; k < t_rowptr[col_begin(t + 1)]; ++k) {
      const int64_t r = rowval[k] - base;
      if (r < 0 || r >= rows) { bad.store(1); return; }
      t_col[k] = (int)r;
      t_val[k] = nzval[k];
      c[r / rpb] += 1;
    }
  });
  if (bad.load()) return fail(-1, "rowval out of range");
  std::vector<int64_t> off((size_t)T * T), bstart((size_t)T + 1, 0);
  for (int b = 0; b < T; ++b) {
    int64_t run = bstart[b];
    for (int t = 0; t < T; ++t) { off[(size_t)t * T + b] = run; run += cnt[(size_t)t * T + b]; }
    bstart[b + 1] = run;
  }
  // pass 2: scatter (row, col, val) into the buckets
  ivec brow((size_t)nnz), bcol((size_t)nnz);
  dvec bval((size_t)nnz);
  run_threads(T, [&](int t) {
    int64_t *o = off.data() + (size_t)t * T;
    for (int64_t j = col_begin(t); j < col_begin(t + 1); ++j)
      for (int k = t_rowptr[j]; k < t_rowptr[j + 1]; ++k) {
        const int r = t_col[k];
        const int64_t p = o[r / rpb]++;
        brow[p] = r; bcol[p] = (int)j; bval[p] = t_val[k];
      }
  });
  // pass 3: row counts (every bucket owns its rows), serial prefix, placement
  run_threads(T, [&](int b) {
    for (int64_t p = bstart[b]; p < bstart[b + 1]; ++p) rowptr[brow[p] + 1] += 1;
  });
  for (int64_t i = 0; i < rows; ++i) rowptr[i + 1] += rowptr[i];
  run_threads(T, [&](int b) {
    const int64_t r0 = std::min<int64_t>(rows, rpb * b), r1 = std::min<int64_t>(rows, rpb * (b + 1));
    std::vector<int> next(rowptr.begin() + r0, rowptr.begin() + r1);
    for (int64_t p = bstart[b]; p < bstart[b + 1]; ++p) {
      const int q = next[brow[p] - r0]++;
      col[q] = bcol[p];
      val[q] = bval[p];
    }
  });
  return 0;
}

// Large matrices: upload the caller's CSC arrays as they are and build CSR(A'), CSR(A) in HBM
// (device_layout.hpp).  On return A / At hold rowptr, col, val on the device and the two host vectors the
// row pointers (the only per-row data the host-side planning needs).  Validation as csc_to_both's.
int ingest_on_device(CsrDev &A, CsrDev &At, int64_t rows, int64_t cols, int64_t nnz, const int64_t *colptr,
                     const int64_t *rowval, const double *nzval, int base, std::vector<int> &rowptr,
                     std::vector<int> &t_rowptr) {
  if (rows >= INT32_MAX || cols >= INT32_MAX || nnz >= INT32_MAX || rows + cols >= INT32_MAX)
    return fail(-2, "m, n, m + n or nnz >= 2^31 need the 64-bit index path (not built)");
  if (colptr[0] != base) return fail(-1, "colptr[0] != index_base");
  if (colptr[cols] - base != nnz) return fail(-1, "colptr[n] - base != nnz");
  hipStream_t st = nullptr;
  int64_t *d_colptr = nullptr, *d_rowval = nullptr;
  int *key = nullptr, *key2 = nullptr, *col2 = nullptr, *row_cnt = nullptr, *flags = nullptr;
  double *val2 = nullptr;
  auto cleanup = [&]() {
    for (void *p : {(void *)d_colptr, (void *)d_rowval, (void *)key, (void *)key2, (void *)col2, (void *)row_cnt, (void *)flags, (void *)val2})
      if (p) (void)hipFree(p);
  };
#define DL(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail_hip(_e, #expr); } } while (0)
  const size_t nz = (size_t)std::max<int64_t>(nnz, 1);
  DL(hipMalloc((void **)&d_colptr, sizeof(int64_t) * (size_t)(cols + 1)));
  DL(hipMalloc((void **)&d_rowval, sizeof(int64_t) * nz));
  DL(hipMalloc((void **)&At.rowptr, sizeof(int) * (size_t)(cols + 1)));
  DL(hipMalloc((void **)&At.col, sizeof(int) * nz));
  DL(hipMalloc((void **)&At.val, sizeof(double) * nz));
  DL(hipMalloc((void **)&A.rowptr, sizeof(int) * (size_t)(rows + 1)));
  DL(hipMalloc((void **)&A.col, sizeof(int) * nz));
  DL(hipMalloc((void **)&A.val, sizeof(double) * nz));
  DL(hipMalloc((void **)&key, sizeof(int) * nz));
  DL(hipMalloc((void **)&key2, sizeof(int) * nz));
  DL(hipMalloc((void **)&col2, sizeof(int) * nz));
  DL(hipMalloc((void **)&val2, sizeof(double) * nz));
  DL(hipMalloc((void **)&row_cnt, sizeof(int) * (size_t)(rows + 1)));
  DL(hipMalloc((void **)&flags, sizeof(int)));
  DL(hipMemcpy(d_colptr, colptr, sizeof(int64_t) * (size_t)(cols + 1), hipMemcpyHostToDevice));
  DL(hipMemcpy(d_rowval, rowval, sizeof(int64_t) * (size_t)nnz, hipMemcpyHostToDevice));
  DL(hipMemcpy(At.val, nzval, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
  DL(hipMemsetAsync(row_cnt, 0, sizeof(int) * (size_t)(rows + 1), st));
  DL(hipMemsetAsync(flags, 0, sizeof(int), st));
  hipLaunchKernelGGL(ingest_colptr_kernel, dim3((unsigned)((cols + 1 + TPB - 1) / TPB)), dim3(TPB), 0, st,
                     (const int64_t *)d_colptr, cols, nnz, base, At.rowptr, flags);
  hipLaunchKernelGGL(ingest_entries_kernel, dim3(4096), dim3(TPB), 0, st, (const int64_t *)d_rowval, (const int *)At.rowptr, nnz,
                     rows, cols, base, At.col, key, A.col, row_cnt, flags);
  int hflags = 0;
  DL(hipMemcpy(&hflags, flags, sizeof(int), hipMemcpyDeviceToHost));
  if (hflags) { cleanup(); return fail(-1, (hflags & 2) ? "colptr not monotone" : "rowval out of range"); }
  int rc = device_exclusive_scan(row_cnt, A.rowptr, rows + 1, nullptr, st);
  if (rc) { cleanup(); return rc; }
  DL(hipMemcpyAsync(A.val, At.val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, st));
  int key_bits = 1;
  while ((1LL << key_bits) < rows) ++key_bits;
  bool in_scratch = false;
  rc = device_radix_sort(key, A.col, A.val, key2, col2, val2, nnz, key_bits, &in_scratch, st);
  if (rc) { cleanup(); return rc; }
  if (in_scratch) { std::swap(A.col, col2); std::swap(A.val, val2); }
  rowptr.resize((size_t)rows + 1);
  t_rowptr.resize((size_t)cols + 1);
  DL(hipMemcpy(rowptr.data(), A.rowptr, sizeof(int) * (size_t)(rows + 1), hipMemcpyDeviceToHost));
  DL(hipMemcpy(t_rowptr.data(), At.rowptr, sizeof(int) * (size_t)(cols + 1), hipMemcpyDeviceToHost));
#undef DL
  cleanup();
  return 0;
}

// Both device layouts of one CSC matrix (either may be skipped: a row segment of a matrix beyond the 32-bit entry limit
// needs only one of them, create_segmented below): ingest -- on the device from 8M nonzeros, host threads below --
// then the row blocks, long-row tables and, where chosen, the sweep's tile-major copy or the column slabs.
int build_layout_pair(int dev, bool remap, bool relaxed, int64_t m, int64_t n, int64_t nnz, const int64_t *colptr,
                      const int64_t *rowval, const double *nzval, int index_base, CsrDev *A_out, CsrDev *At_out) {
  const bool verbose = getenv("PDHG_VERBOSE") != nullptr;   // phase timings of the set-up on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
    return std::chrono::duration<double>(b2 - a).count();
  };
  const auto t_start = now();
  std::vector<int> t_rowptr, rowptr;
  ivec t_col, col;
  dvec t_val, val;
  // Large matrices are laid out ON THE DEVICE (device_layout.hpp): bit-identical layouts, a fraction of
  // the host builders' time.  PDHG_DEVICE_LAYOUT=0 keeps the host builders, =1 forces the device path.
  bool device_layout = nnz >= (8 << 20) && m > 0 && n > 0;
  if (const char *ev = getenv("PDHG_DEVICE_LAYOUT")) device_layout = ev[0] != '0' && nnz > 0 && m > 0 && n > 0;
  int rc = 0;
  CsrDev dev_A, dev_At;
  if (device_layout) {
    rc = ingest_on_device(dev_A, dev_At, m, n, nnz, colptr, rowval, nzval, index_base, rowptr, t_rowptr);
    if (rc) { free_csr_dev(dev_A); free_csr_dev(dev_At); return rc; }
  } else {
    rc = csc_to_both(m, n, nnz, colptr, rowval, nzval, index_base, t_rowptr, t_col, t_val, rowptr, col, val);
    if (rc) return rc;
  }
  const auto t_conv = now();
  // the two layouts are independent: CSR(A) is built on a second host thread while this one builds CSR(A')
  // (each fans out over host_threads() workers for the per-nonzero passes; uploads are synchronous copies)
  int rc_a = 0;
  std::string err_a;
  double t_a = 0.0, t_at = 0.0;
  const int tile_a = choose_tile_cols(n, nnz, m), tile_at = choose_tile_cols(m, nnz, n);
  // Off by default: measured on the 2 x 64-core host of the GPU box at config S, the two builds side by side took
  // 0.92 s against 0.83 s one after the other (0.78 || 0.60 s against 0.47 + 0.36 s) -- the per-nonzero passes are bound
  // by host memory bandwidth, not by threads (32 threads per pass instead of 16 changed nothing either).
  const bool two = A_out && At_out && nnz >= (1 << 22) && dev_env("PDHG_PARALLEL_LAYOUTS") != nullptr;
  if (device_layout) {
    if (A_out) *A_out = dev_A; else free_csr_dev(dev_A);
    if (At_out) *At_out = dev_At; else free_csr_dev(dev_At);
    dev_A = CsrDev(); dev_At = CsrDev();
  }
  auto build_a = [&]() {
    if (!A_out) return;
    const auto t0 = now();
    if (hipSetDevice(dev) != hipSuccess) { rc_a = 999; err_a = "hipSetDevice failed on the layout thread"; return; }
    rc_a = device_layout ? build_csr_dev_resident(*A_out, (int)m, (int)n, rowptr, remap, tile_a, relaxed)
                         : build_csr_dev(*A_out, (int)m, (int)n, rowptr, col, val, remap, tile_a, relaxed);
    if (rc_a) err_a = g_last_error;
    t_a = secs(t0, now());
  };
  std::thread worker;
  if (two) worker = std::thread(build_a); else build_a();
  const auto t0 = now();
  int rc_t = 0;
  if (At_out)
    rc_t = device_layout ? build_csr_dev_resident(*At_out, (int)n, (int)m, t_rowptr, remap, tile_at, relaxed)
                         : build_csr_dev(*At_out, (int)n, (int)m, t_rowptr, t_col, t_val, remap, tile_at, relaxed);
  t_at = secs(t0, now());
  if (two) worker.join();
  if (rc_a) { g_last_error = err_a; return rc_a; }
  if (rc_t) return rc_t;
  if (verbose)
    fprintf(stderr, "pdhg_create (%s): CSC -> CSR(A), CSR(A') %.2fs; layouts%s A %.2fs %s A' %.2fs (%.2fs elapsed)\n",
            device_layout ? "device layout construction" : "host layout construction", secs(t_start, t_conv),
            device_layout ? "" : " + upload", t_a, two ? "beside" : "then", t_at, secs(t_conv, now()));
  return 0;
}

// A matrix with more entries than the layouts' 32-bit offsets index (quadratic_programming.jl:64: Int64 in the reference):
// both copies are built as SEGMENTS of whole rows (layout.hpp, CsrDev::segs) -- CSR(A) from row ranges of the matrix,
// CSR(A') from column ranges (= row ranges of A'), every range below `cap` entries -- inside ONE ordinary handle: no
// shards, no exchange, every row sum in its reference order.  Each range is ingested like a matrix of its own; the
// side of the pair that the range does not need is not built.
int build_segments(int dev, pdhg_handle *h, int64_t m, int64_t n, int64_t nnz, const int64_t *colptr, const int64_t *rowval,
                   const double *nzval, int base, int64_t cap) {
  if (!colptr || !rowval || !nzval) return fail(-1, "null input array");
  if (colptr[0] != base || colptr[n] - base != nnz) return fail(-1, "colptr does not match nnz / index_base");
  for (int64_t j = 0; j < n; ++j)
    if (colptr[j + 1] < colptr[j]) return fail(-1, "colptr not monotone");
  std::vector<int64_t> prefix;
  // (every row index is range-checked here, before anything is indexed with it)
  if (row_nnz_prefix(m, n, colptr, rowval, base, prefix) != 0) return fail(-1, "row index out of range");
  const int64_t target = std::max<int64_t>(1, (cap / 10) * 8);         // aim at 80 % of the limit
  auto cut = [&](int64_t count, auto extent, const char *what, std::vector<int64_t> &bounds) -> int {
    bounds.assign(1, 0);
    int64_t i = 0;
    while (i < count) {
      const int64_t i0 = i;
      if (extent(i0, i0 + 1) > cap)
        return fail(-2, std::string(what) + " " + std::to_string(i0) + " alone holds " + std::to_string(extent(i0, i0 + 1)) +
                            " nonzeros, more than 32-bit offsets can index (" + std::to_string(cap) + ")");
      ++i;
      while (i < count && extent(i0, i + 1) <= target) ++i;
      bounds.push_back(i);
    }
    return 0;
  };
  std::vector<int64_t> rb, cb;
  int rc;
  if ((rc = cut(m, [&](int64_t a, int64_t b) { return prefix[(size_t)b] - prefix[(size_t)a]; }, "row", rb))) return rc;
  if ((rc = cut(n, [&](int64_t a, int64_t b) { return colptr[b] - colptr[a]; }, "column", cb))) return rc;
  const bool verbose = getenv("PDHG_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "[pdhg_hip] %lld nonzeros exceed the 32-bit entry limit (%lld): CSR(A) in %zu row segments, CSR(A') in %zu\n",
            (long long)nnz, (long long)cap, rb.size() - 1, cb.size() - 1);
  h->A.rows = (int)m; h->A.cols = (int)n; h->A.nnz = nnz;
  h->At.rows = (int)n; h->At.cols = (int)m; h->At.nnz = nnz;
  int slot = 0;
  for (size_t k = 0; k + 1 < rb.size(); ++k) {
    std::vector<int64_t> cp;
    uvec<int64_t> rv;
    dvec nv;
    slice_csc_rows(n, colptr, rowval, nzval, base, rb[k], rb[k + 1], cp, rv, nv);        // 0-based CSC of the row range
    CsrDev S;
    static const int64_t none_i = 0;
    static const double none_d = 0.0;
    if ((rc = build_layout_pair(dev, h->remap, h->relaxed, rb[k + 1] - rb[k], n, cp[(size_t)n], cp.data(), rv.empty() ? &none_i : rv.data(),
                                nv.empty() ? &none_d : nv.data(), 0, &S, nullptr))) { free_csr_dev(S); return rc; }
    S.row0 = (int)rb[k];
    S.slot0 = slot;
    slot += S.slots();
    h->A.max_row_nnz = std::max(h->A.max_row_nnz, S.max_row_nnz);
    h->A.segs.push_back(S);
  }
  slot = 0;
  for (size_t k = 0; k + 1 < cb.size(); ++k) {
    const int64_t c0 = cb[k], c1 = cb[k + 1], k0 = colptr[c0] - base;
    std::vector<int64_t> cp((size_t)(c1 - c0) + 1);
    for (int64_t j = c0; j <= c1; ++j) cp[(size_t)(j - c0)] = colptr[j] - colptr[c0] + base;
    CsrDev S;
    if ((rc = build_layout_pair(dev, h->remap, h->relaxed, m, c1 - c0, colptr[c1] - colptr[c0], cp.data(), rowval + k0, nzval + k0, base,
                                nullptr, &S))) { free_csr_dev(S); return rc; }
    S.row0 = (int)c0;
    S.slot0 = slot;
    slot += S.slots();
    h->At.max_row_nnz = std::max(h->At.max_row_nnz, S.max_row_nnz);
    h->At.segs.push_back(S);
  }
  return 0;
}

// One shard: device layouts + vectors for the rows it is given.  n_alloc >= n is the
// allocation length of the n-vectors that take part in collectives.
int create_shard(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                 const int64_t *colptr, const int64_t *rowval, const double *nzval,
                 int index_base, const double *c, const double *b, const double *lb,
                 const double *ub, int64_t num_equalities, int device_id, void *stream, int64_t n_alloc, int64_t seg_cap) {
  *out = nullptr;
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m) return fail(-1, "num_equalities out of range");
  if (!colptr || !c || !lb || !ub || (m > 0 && !b) || (nnz > 0 && (!rowval || !nzval)))
    return fail(-1, "null input array");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(-3, "no HIP device visible");
  int dev = device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return fail(-1, "device_id out of range");
  HIP_TRY(hipSetDevice(dev));
  n_alloc = std::max(n_alloc, n);

  pdhg_handle *h = new pdhg_handle();
  h->self = h;
  h->device = dev;
  h->m = m; h->n = n; h->nnz = nnz; h->num_eq = num_equalities;
  h->cn = n; h->n_alloc = n_alloc; h->m_global = m;
  const char *env = getenv("PDHG_XCD_REMAP");
  h->remap = !(env && env[0] == '0');
  env = getenv("PDHG_ROW_ORDER");         // strict: every row sum strictly left to right; relaxed (default): long rows wave-parallel
  h->relaxed = !(env && !strcmp(env, "strict"));
  env = getenv("PDHG_LAZY_ACCEPT");       // 0: pdhg_accept runs K7 itself (one more launch and n + m more words per iteration)
  h->lazy_accept = !(env && env[0] == '0');
  if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail((int)e, "hipStreamCreate failed"); }
    h->own_stream = true;
  }
#define CK(expr) do { int _rc = (expr); if (_rc) { destroy_shard(h); return _rc; } } while (0)
  if (seg_cap > 0 && nnz > seg_cap) CK(build_segments(dev, h, m, n, nnz, colptr, rowval, nzval, index_base, seg_cap));   // 64-bit extents
  else CK(build_layout_pair(dev, h->remap, h->relaxed, m, n, nnz, colptr, rowval, nzval, index_base, &h->A, &h->At));
  auto up = [&](double **dst, const double *src, int64_t len) -> int {
    int r2 = alloc_zero(dst, len);
    if (r2) return r2;
    if (len > 0) { HIP_TRY(hipMemcpy(*dst, src, sizeof(double) * (size_t)len, hipMemcpyHostToDevice)); HIP_TRY(hipStreamSynchronize(nullptr)); }
    return 0;
  };
  CK(up(&h->c, c, n)); CK(up(&h->b, b, m)); CK(up(&h->lb, lb, n)); CK(up(&h->ub, ub, n));
  CK(alloc_zero(&h->x, n_alloc)); CK(alloc_zero(&h->x_next, n_alloc)); CK(alloc_zero(&h->xbar, n_alloc));
  CK(alloc_zero(&h->y, m)); CK(alloc_zero(&h->y_next, m));
  CK(alloc_zero(&h->aty, n_alloc + 1)); CK(alloc_zero(&h->aty_next, n_alloc + 1));
  CK(alloc_zero(&h->sum_x, n)); CK(alloc_zero(&h->sum_y, m));
  CK(alloc_zero(&h->tmp_n, n_alloc)); CK(alloc_zero(&h->tmp_m, m));
  h->ew_grid_n = ew_grid(n); h->ew_grid_m = ew_grid(m); h->ew_grid_nm = ew_grid(std::max(n, m));
  h->pAt_stride = std::max(h->At.slots(), h->ew_grid_n);
  // block partials are double-double: hi parts, then lo parts
  CK(alloc_zero(&h->pA, 2 * (int64_t)std::max(h->A.slots(), 1)));
  CK(alloc_zero(&h->pAt, 6 * (int64_t)std::max(h->pAt_stride, 1)));
  CK(alloc_zero(&h->pQ, 2 * (int64_t)h->ew_grid_n));
  CK(alloc_zero(&h->scal_dev, SCAL_MAX));
  {
    hipError_t e = hipHostMalloc((void **)&h->scal_host, sizeof(double) * SCAL_MAX * DIST_MAX_WORLD, hipHostMallocDefault);
    if (e != hipSuccess) { destroy_shard(h); return fail((int)e, "hipHostMalloc failed"); }
    e = hipEventCreate(&h->ev0); if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { destroy_shard(h); return fail((int)e, "hipEventCreate failed"); }
  }
#undef CK
  HIP_TRY(hipDeviceSynchronize());
  *out = h;
  return 0;
}

// Phase timeline of the LAST one-launch trial (PDHG_COOP_TRACE=1: the persistent kernels stamp the 100 MHz wall clock at
// every phase boundary, per workgroup).  out[0..4]: mean duration (us) over the workgroups of phase 0, barrier 1, phase 1,
// barrier 2, phase 2; out[5..9]: the slowest workgroup's; out[10]: when the last workgroup left phase 2 (us after the
// trial's first stamp); out[11]: barrier 3's global phase complete (multi-step kernel; 0: single-trial kernel);
// out[12]: decision known to the last workgroup / results published; out[13]: workgroups.
int trial_timeline(pdhg_handle *h, double out[14]) {
  if (!h->coop_trace || h->coop_grid <= 0) return 1;
  std::vector<unsigned long long> t((size_t)8 * h->coop_grid);
  if (hipMemcpy(t.data(), h->coop_trace, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < h->coop_grid; ++w) t0 = std::min(t0, t[(size_t)w * 8]);
  for (int k = 0; k < 5; ++k) {
    double sum = 0, mx = 0, last_end = 0;
    for (int w = 0; w < h->coop_grid; ++w) {
      const double d = 0.01 * (double)(t[(size_t)w * 8 + k + 1] - t[(size_t)w * 8 + k]);
      sum += d; mx = std::max(mx, d);
      last_end = std::max(last_end, 0.01 * (double)(t[(size_t)w * 8 + k + 1] - t0));
    }
    out[k] = sum / h->coop_grid;
    out[5 + k] = mx;
    h->timeline_last_out[k] = last_end;
  }
  out[10] = h->timeline_last_out[4];
  unsigned long long fin = 0, lead = 0;
  for (int w = 0; w < h->coop_grid; ++w) { fin = std::max(fin, t[(size_t)w * 8 + 6]); lead = std::max(lead, t[(size_t)w * 8 + 7]); }
  out[11] = lead ? 0.01 * (double)(lead - t0) : 0.0;
  out[12] = fin ? 0.01 * (double)(fin - t0) : 0.0;
  out[13] = (double)h->coop_grid;
  return 0;
}

void destroy_shard(pdhg_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->n_graph_trials > 0 && getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] %ld graph trials: host us per trial: node updates %.2f, hipGraphLaunch %.2f, wait for the result %.2f\n",
            h->n_graph_trials, 1e6 * h->t_set / h->n_graph_trials, 1e6 * h->t_launch / h->n_graph_trials,
            1e6 * h->t_wait / h->n_graph_trials);
  if (h->coop_trace && (h->coop_launches > 0 || h->steps_launches > 0)) {
    // phase timeline of the LAST one-launch trial: per phase, mean and max over the workgroups of its duration (us)
    double t[14];
    if (trial_timeline(h, t) == 0) {
      const char *names[5] = {"phase 0 (x', xbar) + entry prefetch", "barrier 1", "phase 1 (A xbar, y')", "barrier 2", "phase 2 (A'y', sums)"};
      fprintf(stderr, "[pdhg_hip] one-launch trial timeline (last launch, %d workgroups, 100 MHz clock):\n", h->coop_grid);
      for (int k = 0; k < 5; ++k)
        fprintf(stderr, "    %-38s mean %6.2f us, max %6.2f us; last workgroup out at %6.2f us\n", names[k], t[k], t[5 + k], h->timeline_last_out[k]);
      if (t[11] > 0) fprintf(stderr, "    barrier 3: global phase complete at %6.2f us; decision known to the last workgroup at %6.2f us (multi-step kernel, last trial)\n",
                             t[11], t[12]);
      else fprintf(stderr, "    second-stage reduction published at %6.2f us\n", t[12]);
    }
  }
  free_csr_dev(h->A); free_csr_dev(h->At); free_csr_dev(h->Q); free_csr_dev(h->Qt);
  double *bufs[] = {h->c, h->b, h->lb, h->ub, h->x, h->x_next, h->xbar, h->y, h->y_next,
                    h->aty, h->aty_next, h->sum_x, h->sum_y, h->qx, h->tmp_n, h->tmp_n2,
                    h->tmp_m, h->pA, h->pAt, h->pQ, h->scal_dev, h->scal_all, h->dn_buf, h->dm_buf,
                    h->E, h->Dv, h->c_o, h->b_o, h->lb_o,
                    h->ub_o, h->x_r, h->y_r, h->px_avg, h->py_avg, h->ev_ax, h->ev_aty, h->tr_g,
                    h->tr_dir, h->tr_thr, h->ev_partials, h->ev_cax[0], h->ev_cax[1], h->ev_cax[2],
                    h->ev_caty[0], h->ev_caty[1], h->ev_caty[2], h->ev_cqx[0], h->ev_cqx[1], h->ev_cqx[2],
                    h->ev_qx, h->ev_xg};
  for (double *p : bufs) if (p) (void)hipFree(p);
  graph_destroy(h->tgraph[0]); graph_destroy(h->tgraph[1]);
  if (h->comm_stream) { (void)hipStreamSynchronize(h->comm_stream); (void)hipStreamDestroy(h->comm_stream); }
  for (hipEvent_t ev : h->ev_part) if (ev) (void)hipEventDestroy(ev);
  if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
  if (h->seq_dev) (void)hipFree(h->seq_dev);
  if (h->gsync) (void)hipFree(h->gsync);
  if (h->tr_sync) (void)hipFree(h->tr_sync);
  if (h->lsync) (void)hipFree(h->lsync);
  if (h->tr_partials) (void)hipFree(h->tr_partials);
  if (h->trb_scratch) (void)hipFree(h->trb_scratch);
  if (h->trb_partials) (void)hipFree(h->trb_partials);
  if (h->coop_trace) (void)hipFree(h->coop_trace);
  if (h->res_host) (void)hipHostFree((void *)h->res_host);
  if (h->scal_host) (void)hipHostFree(h->scal_host);
  if (h->ev_host) (void)hipHostFree(h->ev_host);
  if (h->steps_ctl) (void)hipFree(h->steps_ctl);
  if (h->steps_pow_dev) (void)hipFree(h->steps_pow_dev);
  if (h->steps_pow_host) (void)hipHostFree(h->steps_pow_host);
  if (h->steps_res) (void)hipHostFree(h->steps_res);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

void destroy_group(DistGroup *g) {
  if (!g) return;
  if (g->n_trials > 0 && getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] %lld group trials over %zu local shards (%s): host us per trial: issuing %.1f, waiting for the scalars %.1f\n",
            (long long)g->n_trials, g->sh.size(), g->pool ? "one issuing thread per shard" : "issued by the calling thread",
            1e6 * g->t_issue / g->n_trials, 1e6 * g->t_wait / g->n_trials);
  delete g->pool;
  g->pool = nullptr;
  for (size_t i = 0; i < g->sh.size(); ++i) {
    (void)hipSetDevice(g->sh[i]->device);
    (void)hipStreamSynchronize(g->sh[i]->stream);
  }
  for (ncclComm_t c : g->comm) if (c) (void)rccl_loader().api.CommDestroy(c);   // a communicator exists only if RCCL was bound
  for (int f = 0; f < 2; ++f)
    for (size_t i = 0; i < g->ev[f].size(); ++i) {
      (void)hipSetDevice(g->sh[i]->device);
      if (g->ev[f][i]) (void)hipEventDestroy(g->ev[f][i]);
    }
  group_coop_release(*g);
  if (g->gsync) { (void)hipSetDevice(g->sh.empty() ? 0 : g->sh[0]->device); (void)hipFree(g->gsync); }
  for (pdhg_handle *s : g->sh) destroy_shard(s);
  delete g;
}

int create_rank_shard_local(DistGroup *g, int rank, int64_t n, const int64_t *colptr, const int64_t *rowval,
                            const double *nzval, int base, const double *c, const double *b_local, const double *lb,
                            const double *ub, int device_id, void *stream, pdhg_handle **out);

// Build rank `rank`'s shard of the GLOBAL problem: rows row_lo[rank]..row_lo[rank+1), all columns.
int create_rank_shard(DistGroup *g, int rank, int64_t n, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, int base, const double *c, const double *b, const double *lb,
                      const double *ub, int device_id, void *stream, pdhg_handle **out) {
  const int64_t lo = g->row_lo[(size_t)rank], hi = g->row_lo[(size_t)rank + 1];
  std::vector<int64_t> cp;
  uvec<int64_t> rv;
  dvec nv;
  slice_csc_rows(n, colptr, rowval, nzval, base, lo, hi, cp, rv, nv);
  return create_rank_shard_local(g, rank, n, cp.data(), rv.data(), nv.data(), 0, c, b ? b + lo : nullptr, lb, ub,
                                 device_id, stream, out);
}

// The same from the rank's OWN rows: (colptr, rowval, nzval) is the CSC of rows lo..hi of the
// global matrix with row indices rebased to 0, b_local its hi - lo right-hand sides.
int create_rank_shard_local(DistGroup *g, int rank, int64_t n, const int64_t *colptr, const int64_t *rowval,
                            const double *nzval, int base, const double *c, const double *b_local, const double *lb,
                            const double *ub, int device_id, void *stream, pdhg_handle **out) {
  const int64_t lo = g->row_lo[(size_t)rank], hi = g->row_lo[(size_t)rank + 1];
  const int64_t ne = std::min<int64_t>(std::max<int64_t>(g->num_eq_global - lo, 0), hi - lo);
  pdhg_handle *s = nullptr;
  int rc = create_shard(&s, hi - lo, n, colptr[n] - base, colptr, rowval, nzval, base, c, b_local,
                        lb, ub, ne, device_id, stream, g->world * g->S, 0);
  if (rc) return rc;
  if (hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_comm, hipEventDisableTiming) != hipSuccess) {
    destroy_shard(s);
    return fail(999, "comm stream / event creation failed");
  }
  s->ev_part.assign((size_t)g->world, nullptr);
  for (int k = 0; k < g->world; ++k)
    if (hipEventCreateWithFlags(&s->ev_part[(size_t)k], hipEventDisableTiming) != hipSuccess) {
      destroy_shard(s);
      return fail(999, "event creation failed");
    }
  s->grp = g;
  s->rank = rank;
  s->world = g->world;
  s->row_lo = lo;
  s->m_global = g->m_global;
  s->clo = std::min<int64_t>(n, (int64_t)rank * g->S);
  s->cn = std::min<int64_t>(n, (int64_t)(rank + 1) * g->S) - s->clo;
  if ((rc = alloc_zero(&s->dn_buf, s->n_alloc))) { destroy_shard(s); return rc; }
  if ((rc = alloc_zero(&s->dm_buf, g->m_global))) { destroy_shard(s); return rc; }
  if ((rc = alloc_zero(&s->scal_all, (int64_t)SCAL_MAX * g->world))) { destroy_shard(s); return rc; }
  *out = s;
  return 0;
}

// One reduce-scatter after the product, or per-slice reductions overlapped with it
// (DistGroup::overlap).  Decided from (n, world, back end, environment) only.  Default: on
// for the peer back end when the exchanged vector is large (a slice is one small kernel of
// its owner); OFF for RCCL, where P reductions to P roots cost P collective latencies and
// may not reach the bandwidth of one reduce-scatter over all links -- the product they
// could hide behind is 0.1 ms at P = 8.  PDHG_DIST_OVERLAP=0/1 forces.
void choose_exchange_pattern(DistGroup *g) {
  const char *ov = getenv("PDHG_DIST_OVERLAP");
  g->overlap = ov ? (ov[0] != '0') : (g->backend == COMM_P2P && g->world > 1 && g->n * 8 > (4LL << 20));
}

// row_bounds != nullptr: the caller's partition ([world + 1], ascending, 0 .. m) instead of the
// library's nnz-balanced one (colptr / rowval may then be null)
int init_group_geometry(DistGroup *g, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int base,
                        int64_t num_equalities, int world, const int64_t *row_bounds = nullptr) {
  if (world < 1 || world > DIST_MAX_WORLD) return fail(-1, "world size out of range (1..64)");
  if (!colptr && !row_bounds) return fail(-1, "null input array");
  g->world = world;
  g->n = n;
  g->m_global = m;
  g->num_eq_global = num_equalities;
  const int64_t per = (n + world - 1) / world;
  g->S = std::max<int64_t>(16, (per + 15) / 16 * 16);      // slice stride: whole 128-byte lines
  if (row_bounds) {
    if (row_bounds[0] != 0 || row_bounds[world] != m) return fail(-1, "row_bounds must run from 0 to m");
    for (int p = 0; p < world; ++p)
      if (row_bounds[p + 1] < row_bounds[p]) return fail(-1, "row_bounds not ascending");
    g->row_lo.assign(row_bounds, row_bounds + world + 1);
  } else {
    partition_rows_by_nnz(m, n, colptr, rowval, base, world, g->row_lo);
  }
  const char *fr = dev_env("PDHG_DIST_FORCE_REMOTE");
  g->force_remote = fr && fr[0] == '1';
  return 0;
}

}  // namespace

// ================================================================== C ABI

extern "C" {

const char *pdhg_last_error(void) { return g_last_error.c_str(); }
int pdhg_abi_version(void) { return 10; }

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
    const std::string k = D.slabs.front().sj.on() ? "spmv_sj_kernel<" : (D.slabs.front().pipe_grid > 0 ? "spmv_stream_pipe_kernel<" : "spmv_stream_kernel<");
    add(k + "0, false, " + t + ">");
    if (D.slabs.size() > 2) add(k + "0, true, " + t + ">");
    add(k + m + ", true, " + t + ">");
  } else if (D.sj.on()) {
    add("spmv_sj_kernel<" + m + ", false, " + t + ">");
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

// ---- row-partitioned multi-GPU handles -------------------------------------------

int pdhg_dist_get_unique_id(void *id) {
  if (!id) return fail(-1, "id == NULL");
  static_assert(sizeof(ncclUniqueId) <= PDHG_UNIQUE_ID_BYTES, "unique id does not fit the ABI's buffer");
  RCCL_API(R);
  ncclUniqueId u;
  NCCL_TRY(R->GetUniqueId(&u));
  memset(id, 0, PDHG_UNIQUE_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return 0;
}

int pdhg_create_dist(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                     const int64_t *colptr, const int64_t *rowval, const double *nzval,
                     int index_base, const double *c, const double *b, const double *lb,
                     const double *ub, int64_t num_equalities, int device_id, void *stream,
                     const void *unique_id, int rank, int world) {
  if (!out) return fail(-1, "out == NULL");
  *out = nullptr;
  if (!unique_id) return fail(-1, "unique_id == NULL");
  if (rank < 0 || rank >= world) return fail(-1, "rank out of range");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m) return fail(-1, "num_equalities out of range");
  if (!colptr || (nnz > 0 && (!rowval || !nzval))) return fail(-1, "null input array");
  if (colptr[0] != index_base || colptr[n] - index_base != nnz) return fail(-1, "colptr does not match nnz / index_base");
  DistGroup *g = new DistGroup();
  int rc = init_group_geometry(g, m, n, colptr, rowval, index_base, num_equalities, world);
  if (rc) { delete g; return rc; }
  g->backend = COMM_RCCL;
  choose_exchange_pattern(g);
  pdhg_handle *s = nullptr;
  rc = create_rank_shard(g, rank, n, colptr, rowval, nzval, index_base, c, b, lb, ub, device_id, stream, &s);
  if (rc) { delete g; return rc; }
  g->sh.push_back(s);
  g->comm.assign(1, nullptr);
  ncclUniqueId u;
  memcpy(&u, unique_id, sizeof(u));
  const RcclApi *R = rccl();
  if (!R) { destroy_group(g); return 2999; }
  ncclResult_t nr = R->CommInitRank(&g->comm[0], world, u, rank);
  if (nr != ncclSuccess) {
    g_last_error = std::string("ncclCommInitRank: ") + R->GetErrorString(nr);
    destroy_group(g);
    return 2000 + (int)nr;
  }
  *out = s;
  return 0;
}

int pdhg_partition_rows(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int index_base,
                        int world, int64_t *row_bounds) {
  if (!colptr || !row_bounds || m < 0 || n < 0) return fail(-1, "null / negative argument");
  if (world < 1 || world > DIST_MAX_WORLD) return fail(-1, "world size out of range (1..64)");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (colptr[0] != index_base) return fail(-1, "colptr[0] != index_base");
  if (colptr[n] - index_base > 0 && !rowval) return fail(-1, "null input array");
  std::vector<int64_t> b;
  partition_rows_by_nnz(m, n, colptr, rowval, index_base, world, b);
  for (int p = 0; p <= world; ++p) row_bounds[p] = b[(size_t)p];
  return 0;
}

int pdhg_create_dist_rows(pdhg_handle **out, int64_t m_global, int64_t n, const int64_t *row_bounds,
                          int64_t local_nnz, const int64_t *colptr, const int64_t *rowval, const double *nzval,
                          int index_base, const double *c, const double *b_local, const double *lb,
                          const double *ub, int64_t num_equalities, int device_id, void *stream,
                          const void *unique_id, int rank, int world) {
  if (!out) return fail(-1, "out == NULL");
  *out = nullptr;
  if (!unique_id) return fail(-1, "unique_id == NULL");
  if (rank < 0 || rank >= world) return fail(-1, "rank out of range");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m_global) return fail(-1, "num_equalities out of range");
  if (!row_bounds || !colptr || (local_nnz > 0 && (!rowval || !nzval))) return fail(-1, "null input array");
  if (colptr[0] != index_base || colptr[n] - index_base != local_nnz) return fail(-1, "colptr does not match local_nnz / index_base");
  DistGroup *g = new DistGroup();
  int rc = init_group_geometry(g, m_global, n, nullptr, nullptr, index_base, num_equalities, world, row_bounds);
  if (rc) { delete g; return rc; }
  g->backend = COMM_RCCL;
  choose_exchange_pattern(g);
  pdhg_handle *s = nullptr;
  rc = create_rank_shard_local(g, rank, n, colptr, rowval, nzval, index_base, c, b_local, lb, ub, device_id, stream, &s);
  if (rc) { delete g; return rc; }
  g->sh.push_back(s);
  g->comm.assign(1, nullptr);
  ncclUniqueId u;
  memcpy(&u, unique_id, sizeof(u));
  const RcclApi *R = rccl();
  if (!R) { destroy_group(g); return 2999; }
  ncclResult_t nr = R->CommInitRank(&g->comm[0], world, u, rank);
  if (nr != ncclSuccess) {
    g_last_error = std::string("ncclCommInitRank: ") + R->GetErrorString(nr);
    destroy_group(g);
    return 2000 + (int)nr;
  }
  *out = s;
  return 0;
}

int pdhg_rccl_info(int *compiled_version, int *runtime_version, char *path, int path_len) {
  if (compiled_version) *compiled_version = NCCL_VERSION_CODE;
  if (runtime_version) *runtime_version = 0;
  if (path && path_len > 0) path[0] = 0;
  RcclLoader &L = rccl_loader();
  if (runtime_version) *runtime_version = L.api.runtime_version;
  if (path && path_len > 0) snprintf(path, (size_t)path_len, "%s", L.api.path.c_str());
  if (!L.ok) { g_last_error = L.api.error; return 2999; }
  return 0;
}

int pdhg_host_issue_stats(pdhg_handle *h, int64_t *trials, double *issue_seconds, double *wait_seconds) {
  if (!h || !trials || !issue_seconds || !wait_seconds) return fail(-1, "null argument");
  if (h->grp) { *trials = h->grp->n_trials; *issue_seconds = h->grp->t_issue; *wait_seconds = h->grp->t_wait; }
  else { *trials = h->n_graph_trials; *issue_seconds = h->t_set + h->t_launch; *wait_seconds = h->t_wait; }
  return 0;
}

int pdhg_create_multi(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                      const int64_t *colptr, const int64_t *rowval, const double *nzval,
                      int index_base, const double *c, const double *b, const double *lb,
                      const double *ub, int64_t num_equalities, int n_devices, const int *device_ids) {
  return create_multi_impl(out, m, n, nnz, colptr, rowval, nzval, index_base, c, b, lb, ub, num_equalities,
                           n_devices, device_ids, nullptr);
}

static int create_multi_impl(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                             const int64_t *colptr, const int64_t *rowval, const double *nzval,
                             int index_base, const double *c, const double *b, const double *lb,
                             const double *ub, int64_t num_equalities, int n_devices, const int *device_ids,
                             const int64_t *row_bounds) {
  if (!out) return fail(-1, "out == NULL");
  *out = nullptr;
  if (n_devices < 1 || !device_ids) return fail(-1, "n_devices < 1 or device_ids == NULL");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m) return fail(-1, "num_equalities out of range");
  if (!colptr || (nnz > 0 && (!rowval || !nzval))) return fail(-1, "null input array");
  if (colptr[0] != index_base || colptr[n] - index_base != nnz) return fail(-1, "colptr does not match nnz / index_base");
  DistGroup *g = new DistGroup();
  int rc = init_group_geometry(g, m, n, colptr, rowval, index_base, num_equalities, n_devices, row_bounds);
  if (rc) { delete g; return rc; }
  // Back end: RCCL (ncclCommInitAll) when every shard has its own device; direct peer
  // kernels when devices repeat (several shards on one GPU: tests, oversubscription)
  // or when PDHG_COMM=p2p asks for them.
  bool distinct = true;
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < i; ++j) if (device_ids[i] == device_ids[j]) distinct = false;
  const char *cm = getenv("PDHG_COMM");
  g->backend = (!distinct || (cm && !strcmp(cm, "p2p"))) ? COMM_P2P : COMM_RCCL;
  if (g->backend == COMM_P2P && n_devices > P2P_MAX_WORLD) { delete g; return fail(-1, "peer back end supports at most 16 shards"); }
  choose_exchange_pattern(g);
  for (int r = 0; r < n_devices; ++r) {
    pdhg_handle *s = nullptr;
    rc = create_rank_shard(g, r, n, colptr, rowval, nzval, index_base, c, b, lb, ub, device_ids[r], nullptr, &s);
    if (rc) { destroy_group(g); return rc; }
    g->sh.push_back(s);
  }
  if (g->backend == COMM_RCCL) {
    g->comm.assign((size_t)n_devices, nullptr);
    const RcclApi *R = rccl();
    if (!R) { destroy_group(g); return 2999; }
    ncclResult_t nr = R->CommInitAll(g->comm.data(), n_devices, device_ids);
    if (nr != ncclSuccess) {
      g_last_error = std::string("ncclCommInitAll: ") + R->GetErrorString(nr);
      destroy_group(g);
      return 2000 + (int)nr;
    }
  } else {
    for (int f = 0; f < 2; ++f) g->ev[f].assign((size_t)n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
      (void)hipSetDevice(device_ids[i]);
      for (int j = 0; j < n_devices; ++j)
        if (device_ids[j] != device_ids[i]) {
          hipError_t e = hipDeviceEnablePeerAccess(device_ids[j], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            destroy_group(g);
            return fail((int)e, "hipDeviceEnablePeerAccess failed");
          }
          (void)hipGetLastError();
        }
      for (int f = 0; f < 2; ++f)
        if (hipEventCreateWithFlags(&g->ev[f][(size_t)i], hipEventDisableTiming) != hipSuccess) {
          destroy_group(g);
          return fail(999, "hipEventCreate failed");
        }
    }
  }
  // one issuing host thread per shard for the trial steps (dist.hpp, ShardPool)
  {
    const char *ev = getenv("PDHG_SHARD_THREADS");
    if (n_devices > 1 && !(ev && ev[0] == '0')) g->pool = new ShardPool(n_devices);
  }
  *out = g->sh[0];
  return 0;
}

int pdhg_dist_info(pdhg_handle *h, int64_t info[8]) {
  if (!h || !info) return fail(-1, "null argument");
  info[0] = h->world;
  info[1] = h->grp ? (int64_t)h->grp->sh.size() : 1;
  info[2] = h->rank;
  info[3] = h->grp ? h->grp->backend : -1;
  info[4] = h->row_lo;
  info[5] = h->row_lo + h->m;
  info[6] = h->clo;
  info[7] = h->clo + h->cn;
  return 0;
}

int pdhg_set_objective_matrix(pdhg_handle *h0, int64_t q_nnz, const int64_t *q_colptr,
                              const int64_t *q_rowval, const double *q_nzval, int index_base) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  bump_version(L);
  bool all_zero = true;
  for (int64_t k = 0; k < q_nnz; ++k) if (q_nzval[k] != 0.0) all_zero = false;
  std::vector<int> t_rowptr, rowptr;
  ivec t_col, col;
  dvec t_val, val;
  if (!all_zero) {
    rc = csc_to_both(h0->n, h0->n, q_nnz, q_colptr, q_rowval, q_nzval, index_base, t_rowptr, t_col, t_val, rowptr, col, val);
    if (rc) return rc;
  }
  for (int i = 0; i < L.count; ++i) L.p[i]->matrix_version += 1;
  FOR_SHARDS(L, h) {   // the objective matrix is replicated on every shard (it acts on full n-vectors)
    // the launch paths were decided for the problem without (or with another) Q: decide again at the next trial
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->gsync) { (void)hipFree(h->gsync); h->gsync = nullptr; }
    if (h->coop_trace) { (void)hipFree(h->coop_trace); h->coop_trace = nullptr; }
    h->coop_mode = -1; h->coop_launches = 0; h->coop_epoch = 0;
    h->graph_mode = -1;
    h->small_lp_mode = -1;
    graph_destroy(h->tgraph[0]); graph_destroy(h->tgraph[1]);
    if (h->has_q) { free_csr_dev(h->Q); free_csr_dev(h->Qt); h->has_q = false; }
    if (all_zero) continue;  // iszero(objective_matrix): LP path (pdhg.jl:536)
    if ((rc = build_csr_dev(h->Q, (int)h->n, (int)h->n, rowptr, col, val, h->remap))) return rc;
    if ((rc = build_csr_dev(h->Qt, (int)h->n, (int)h->n, t_rowptr, t_col, t_val, h->remap))) return rc;
    if (!h->qx) { if ((rc = alloc_zero(&h->qx, h->n))) return rc; }
    if (!h->tmp_n2) { if ((rc = alloc_zero(&h->tmp_n2, h->n))) return rc; }
    h->has_q = true;
  }
  return 0;
}

void pdhg_destroy(pdhg_handle *h) {
  if (!h) return;
  if (h->grp) destroy_group(h->grp);
  else destroy_shard(h);
}

// ---- the trial step -----------------------------------------------------------------

// Single GPU: K1+K2, K3+K4, K5+K6 (fused epilogues), second-stage reduction.
static int trial_dual_single(pdhg_handle *h, double step_size, double primal_weight, double out[5]) {
  int rc;
  if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  if ((rc = launch_aty_fused(h))) return rc;
  int qcount = 0;
  if ((rc = launch_q_interaction(h, &qcount))) return rc;
  // The five sums go straight into pinned host memory and the host polls the launch's sequence number there
  // (as on the graph path) instead of a device-to-host copy + stream synchronisation: ~10 us per trial, which
  // is 5 % of a 1M x 1M LP's iteration.  While profiling: the copy, so that the event brackets stay simple.
  static const bool host_word = !(dev_env("PDHG_TRIAL_HOST_WORD") && dev_env("PDHG_TRIAL_HOST_WORD")[0] == '0');
  if (host_word && !h->profile) {
    if ((rc = ensure_result_word(h))) return rc;
    if ((rc = launch_final(h, h->pAt, h->At.slots(), h->pAt_stride, h->pA, h->A.slots(), qcount, true))) return rc;
    return wait_result_word(h, out);
  }
  if ((rc = launch_final(h, h->pAt, h->At.slots(), h->pAt_stride, h->pA, h->A.slots(), qcount))) return rc;
  HIP_TRY(hipMemcpyAsync(h->scal_host, h->scal_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int q = 0; q < 5; ++q) out[q] = h->scal_host[q];
  out[4] *= 0.5;
  return 0;
}

// One shard's whole trial, issued by that shard's own host thread (ShardPool): the same
// launches, in the same order, as trial_dual_group issues for it from the calling thread.
struct TrialArgs {
  double step_size, primal_weight, theta;
  bool primal;      // K1+K2 first (pdhg_trial_step); false: xbar only (pdhg_trial_dual)
};
static int trial_shard_mt(DistGroup &g, pdhg_handle *s, int i, const TrialArgs &a, double *t_issued) {
  HIP_TRY(hipSetDevice(s->device));
  int rc;
  if (a.primal) { if ((rc = launch_primal(s, a.step_size / a.primal_weight, a.theta, true))) return rc; }
  else if ((rc = launch_xbar(s, a.theta))) return rc;
  if ((rc = mt_all_gather(g, s, i, [](pdhg_handle *q) { return q->xbar; }, g.S))) return rc;
  if (s->has_q && (rc = mt_all_gather(g, s, i, [](pdhg_handle *q) { return q->x_next; }, g.S))) return rc;
  const double sigma = a.primal_weight * a.step_size;
  if ((rc = launch_dual(s, sigma))) return rc;
  if (!g.overlap) {
    if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc;
    if ((rc = mt_reduce_scatter(g, s, i, [](pdhg_handle *q) { return q->aty_next; }, g.S))) return rc;
  } else {
    // see trial_dual_group: the product in residency rounds, slice k reduced as soon as its rows are complete
    const char *rw_env = dev_env("PDHG_DIST_ROUND_WGS");
    const int round_wgs = rw_env ? std::max(1, atoi(rw_env)) : 256 * 2;
    const CsrDev &T = s->At;
    int issued = 0, next_wg = 0;
    for (int k = 0; k < g.world; ++k) {
      const int64_t need = std::min<int64_t>(s->n, (int64_t)(k + 1) * g.S);
      if (!T.tiled) {
        if (!issued) { if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc; issued = 1; }
      } else {
        while (next_wg < T.grid || !issued) {
          const int g0 = next_wg;
          const bool covered = g0 >= T.grid || (int64_t)T.wg_first_row[(size_t)g0] >= need;
          if (covered && issued) break;
          int g1 = std::min(T.grid, g0 + round_wgs);
          if (T.grid - g1 < round_wgs / 2) g1 = T.grid;
          if ((rc = launch_spmv_plain_part(s, T, s->y_next, s->aty_next, g0, g1, !issued))) return rc;
          issued = 1;
          next_wg = g1;
        }
      }
      HIP_TRY(hipEventRecord(s->ev_part[(size_t)k], s->stream));
      if ((rc = mt_reduce_slice_async(g, s, i, [](pdhg_handle *q) { return q->aty_next; }, g.S, k))) return rc;
    }
    if ((rc = mt_join_comm(g, s, i))) return rc;
  }
  {
    const int64_t o = s->clo;
    hipLaunchKernelGGL(interaction_kernel, dim3(ew_grid(s->cn)), dim3(TPB), 0, s->stream, (int)s->cn, s->x + o,
                       s->x_next + o, s->aty + o, s->aty_next + o, s->pAt, s->pAt_stride);
    HIP_TRY(hipGetLastError());
  }
  int qcount = 0;
  if ((rc = launch_q_interaction(s, &qcount))) return rc;
  if ((rc = launch_final(s, s->pAt, ew_grid(s->cn), s->pAt_stride, s->pA, s->A.slots(), qcount))) return rc;
  HIP_TRY(hipMemcpyAsync(s->scal_host, s->scal_dev, sizeof(double) * 5, hipMemcpyDeviceToHost, s->stream));
  *t_issued = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

static int trial_group_mt(const Shards &L, const TrialArgs &a, double out[5]) {
  DistGroup &g = *L.g;
  const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  std::vector<double> issued((size_t)L.count, t0);
  int rc = g.pool->run([&](int i) { return trial_shard_mt(g, L.p[i], i, a, &issued[(size_t)i]); });
  if (rc) return rc;
  const double t2 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  double t1 = t0;
  for (double v : issued) t1 = std::max(t1, v);
  g.t_issue += t1 - t0; g.t_wait += t2 - t1; g.n_trials += 1;
  // the shards' scalars, added in rank order ([4], dx'Q dx, is replicated: maxed) -- as combine_scalars does
  for (int q = 0; q < 5; ++q) {
    double v = L.p[0]->scal_host[q];
    for (int i = 1; i < L.count; ++i) {
      const double t = L.p[i]->scal_host[q];
      v = (q < 4) ? v + t : std::fmax(v, t);
    }
    out[q] = v;
  }
  out[4] *= 0.5;
  return 0;
}

// ---- a group's trial as ONE persistent kernel per device (group_kernel.hpp) -----------------------------------------
// Eligible: every shard of the group lives in this process on the peer back end, LP, stream layouts without slabs, and
// the shards' grids fit their device side by side.  Default: on when all shards share ONE device (the configuration this
// environment can test -- bitwise the ordinary group path); for shards on distinct devices the protocol has never run,
// so it waits for PDHG_GROUP_COOP=1.  PDHG_GROUP_COOP=0: off.
static int group_coop_prepare(const Shards &L) {
  DistGroup &g = *L.g;
  std::vector<int> devs;
  for (int i = 0; i < L.count; ++i) devs.push_back(L.p[i]->device);
  std::sort(devs.begin(), devs.end());
  devs.erase(std::unique(devs.begin(), devs.end()), devs.end());
  const char *pretend = dev_env("PDHG_COOP_TEST_PRETEND_WGS");       // test knob: a grid the device cannot hold
  for (int dev : devs) {
    HIP_TRY(hipSetDevice(dev));
    // (the record enters g.coop_dev FIRST: whatever fails below, the caller's group_coop_release frees what it holds by then)
    g.coop_dev.emplace_back();
    GroupDevLaunch &D = g.coop_dev.back();
    D.device = dev;
    for (int i = 0; i < L.count; ++i) if (L.p[i]->device == dev) D.members.push_back(i);
    D.stream = L.p[D.members[0]]->stream;
    int per_cu = 0, per_cu_inline = 0;
    hipDeviceProp_t prop;
    // co-residency of the kernel that WILL be launched: up to GROUP_INLINE_SHARDS members per device take the inline-argument
    // variant (group_coop_trial), whose registers and kernel arguments differ -- size for the smaller of the two
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, group_trial_kernel, TPB, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_inline, group_trial_inline_kernel, TPB, 0));
    per_cu = std::min(per_cu, per_cu_inline);
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const int cap = std::max(8, per_cu * prop.multiProcessorCount / 8 * 8);
    // every shard one workgroup per item where the device holds that many side by side, else in proportion
    std::vector<int> items, grid;
    int64_t total_items = 0;
    for (int i : D.members) {
      const pdhg_handle *s = L.p[i];
      items.push_back(std::max(8, (std::max(s->A.grid + s->A.nchunks, s->At.grid + s->At.nchunks) + 7) / 8 * 8));
      total_items += items.back();
    }
    int total = 0;
    for (size_t k = 0; k < items.size(); ++k) {
      int gk = total_items <= cap ? items[k] : std::max(8, (int)((int64_t)cap * items[k] / total_items) / 8 * 8);
      if (pretend) gk = std::max(8, atoi(pretend) / 8 * 8);
      if (items[k] > 2 * gk) { g_last_error = "too many row blocks for one persistent launch per device"; return 1; }
      grid.push_back(gk);
      total += gk;
    }
    if (total > cap && !pretend) return 1;
    D.base.assign(1, 0);
    for (int gk : grid) D.base.push_back(D.base.back() + gk);
    D.grid = total;
    const size_t k_n = D.members.size();
    HIP_TRY(hipMalloc((void **)&D.args_dev, sizeof(GroupTrialArgs) * k_n));
    HIP_TRY(hipHostMalloc((void **)&D.args_host, sizeof(GroupTrialArgs) * k_n, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&D.sync_dev, sizeof(GridSync *) * k_n));
    HIP_TRY(hipEventCreateWithFlags(&D.ev_done, hipEventDisableTiming));
    D.ev.assign(k_n, nullptr);
    std::vector<GridSync *> syncs;
    for (size_t k = 0; k < k_n; ++k) {
      pdhg_handle *s = L.p[D.members[k]];
      HIP_TRY(hipEventCreateWithFlags(&D.ev[k], hipEventDisableTiming));
      int rc = ensure_result_word(s);
      if (rc) return rc;
      if (!s->gsync) HIP_TRY(hipMalloc((void **)&s->gsync, sizeof(GridSync)));
      HIP_TRY(hipMemset(s->gsync, 0, sizeof(GridSync)));
      s->coop_grid = grid[k];
      s->coop_epoch = 0; s->coop_launches = 0;
      if (s->coop_grid > s->pAt_stride) {          // the interaction partials take one slot per workgroup
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (s->pAt) (void)hipFree(s->pAt);
        s->pAt = nullptr;
        s->pAt_stride = s->coop_grid;
        if ((rc = alloc_zero(&s->pAt, 6 * (int64_t)s->pAt_stride))) return rc;
      }
      syncs.push_back(s->gsync);
    }
    HIP_TRY(hipMemcpy(D.sync_dev, syncs.data(), sizeof(GridSync *) * k_n, hipMemcpyHostToDevice));
    // census of the merged launch shape: workgroups of every shard per XCD
    GroupDeviceArgs da{};
    da.shard = D.args_dev; da.nshards = (int)k_n;
    for (size_t k = 0; k <= k_n; ++k) da.base[k] = D.base[k];
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(group_register_kernel, dim3(D.grid), dim3(TPB), 0, D.stream, da, (GridSync *const *)D.sync_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(D.stream));
    for (size_t k = 0; k < k_n; ++k) {
      pdhg_handle *s = L.p[D.members[k]];
      GridSync host;
      HIP_TRY(hipMemcpy(&host, s->gsync, sizeof(GridSync), hipMemcpyDeviceToHost));
      unsigned long long seen = 0;
      s->coop_nxcd = 0;
      for (int x = 0; x < 8; ++x) { seen += host.xcd_count[x][0]; s->coop_nxcd += host.xcd_count[x][0] > 0; s->coop_xcd_cnt[x] = (unsigned)host.xcd_count[x][0]; }
      if (seen != (unsigned long long)grid[k] || s->coop_nxcd == 0) return fail(996, "group trial kernel: workgroup census does not add up");
    }
  }
  if (!g.gsync) {
    HIP_TRY(hipSetDevice(L.p[0]->device));
    void *p = nullptr;
    // fine-grained device memory when the runtime offers it: the devices poll these words with system-scope atomics
    if (hipExtMallocWithFlags(&p, sizeof(GroupSync), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      HIP_TRY(hipMalloc(&p, sizeof(GroupSync)));
    }
    HIP_TRY(hipMemset(p, 0, sizeof(GroupSync)));
    HIP_TRY(hipDeviceSynchronize());
    g.gsync = reinterpret_cast<GroupSync *>(p);
  }
  return 0;
}

static bool group_coop_eligible(const Shards &L) {
  DistGroup &g = *L.g;
  if (g.coop_mode < 0) {
    const char *ev = getenv("PDHG_GROUP_COOP");
    bool on = g.all_local() && g.backend == COMM_P2P && L.count == g.world && g.world >= 2 && g.world <= P2P_MAX_WORLD &&
              !(ev && ev[0] == '0') && !(getenv("PDHG_GRAPH") && getenv("PDHG_GRAPH")[0] == '0');
    bool one_device = true;
    for (int i = 0; i < L.count && on; ++i) {
      const pdhg_handle *s = L.p[i];
      one_device = one_device && s->device == L.p[0]->device;
      on = !s->has_q && s->lazy_accept && s->n > 0 && s->cn > 0 && !s->A.tiled && !s->At.tiled && s->A.slabs.empty() &&
           s->At.slabs.empty() && s->A.segs.empty() && s->At.segs.empty() && s->coop_mode != 1 && !s->gsync;
    }
    if (on && !one_device && !(ev && ev[0] == '1')) on = false;
    g.coop_mode = 0;
    if (on) {
      const int rc = group_coop_prepare(L);
      if (rc == 0) g.coop_mode = 1;
      else { (void)hipGetLastError(); group_coop_release(g); }
      if (getenv("PDHG_VERBOSE")) {
        fprintf(stderr, "[pdhg_hip] group of %d shards: one persistent kernel per device and trial %s", g.world, rc == 0 ? "ON" : "not possible");
        for (const GroupDevLaunch &D : g.coop_dev) fprintf(stderr, " [device %d: %zu shards, %d workgroups]", D.device, D.members.size(), D.grid);
        fprintf(stderr, "\n");
      }
    }
  }
  return g.coop_mode == 1 && !L.p[0]->profile;
}

// returns 1 when the trial was not taken here (the caller runs the ordinary group path)
static int group_coop_trial(const Shards &L, const TrialArgs &ta, double out[5]) {
  DistGroup &g = *L.g;
  // one persistent launch set at a time per device (two half-resident sets would wait for each other)
  std::vector<std::unique_lock<std::mutex>> locks;
  for (GroupDevLaunch &D : g.coop_dev) locks.emplace_back(coop_device_mutex(D.device));      // (ascending device ids)
  const auto t_begin = std::chrono::steady_clock::now();
  const double sigma = ta.primal_weight * ta.step_size;
  for (GroupDevLaunch &D : g.coop_dev) {
    HIP_TRY(hipSetDevice(D.device));
    for (size_t k = 0; k < D.members.size(); ++k) {
      pdhg_handle *s = L.p[D.members[k]];
      GroupTrialArgs a{};
      a.rank = s->rank; a.world = g.world;
      const int64_t o = s->clo;
      a.cn = (int)s->cn; a.clo = o; a.xbar_only = ta.primal ? 0 : 1;
      a.x = s->x + o; a.c = s->c + o; a.aty = s->aty + o; a.lb = s->lb + o; a.ub = s->ub + o;
      a.tau = ta.step_size / ta.primal_weight; a.theta = ta.theta;
      a.x_next = s->x_next + o;
      a.avg_w = s->pend_w; a.sum_x = (s->pend_x && ta.primal) ? s->sum_x + o : nullptr;
      for (int q = 0; q < L.count; ++q) {
        a.xbar_peer[L.p[q]->rank] = L.p[q]->xbar;
        a.part_peer[L.p[q]->rank] = L.p[q]->aty_next;
      }
      EpiArgs de{};
      de.y = s->y; de.b = s->b; de.y_next = s->y_next; de.sigma = sigma; de.num_eq = (int)s->num_eq;
      de.partials = s->pA; de.stride = s->A.slots(); de.lo_offset = s->A.slots();
      if (s->pend_y) { de.sum_y = s->sum_y; de.avg_w = s->pend_w; }
      a.A = trial_product(s, s->A, s->xbar, de);
      EpiArgs te{};
      te.out = s->aty_next;
      a.T = trial_product(s, s->At, s->y_next, te);
      a.off = o; a.aty_next = s->aty_next;
      a.pAt = s->pAt; a.pAt_stride = s->pAt_stride;
      a.sp.ptr[0] = s->pAt;                         a.sp.count[0] = s->coop_grid;
      a.sp.ptr[1] = s->pAt + s->pAt_stride;         a.sp.count[1] = s->coop_grid;
      a.sp.ptr[2] = s->pA;                          a.sp.count[2] = s->A.slots();
      a.sp.ptr[3] = s->pAt + 2 * s->pAt_stride;     a.sp.count[3] = s->coop_grid;
      a.sp.ptr[4] = s->pQ;                          a.sp.count[4] = 0;
      for (int q : {0, 1, 3}) a.sp.ptr_lo[q] = a.sp.ptr[q] + 3 * s->pAt_stride;
      a.sp.ptr_lo[2] = s->pA + s->A.slots();
      a.sp.ptr_lo[4] = s->pQ + s->ew_grid_n;
      a.sp.out = nullptr;
      a.sync = s->gsync; a.gsync = g.gsync;
      a.epoch = s->coop_epoch; s->coop_epoch += 3;
      a.xepoch = g.xepoch;
      a.launch = s->coop_launches; s->coop_launches += 1;
      s->seq_expected += 1;
      a.seq = s->seq_expected;
      a.nxcd = s->coop_nxcd;
      for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = s->coop_xcd_cnt[x];
      a.seq_dev = s->seq_dev; a.res_host = s->res_host; a.relaxed = s->relaxed ? 1 : 0;
      D.args_host[k] = a;                       // (the previous launch has returned its results: the staging copy is free)
      if (ta.primal) s->pend_x = false;
      s->pend_y = false;                        // the launch carries the deferred average update
      // whatever the shard's own stream has queued since the last trial (a flush, a set_current, an evaluation) comes first
      if (g.members_dirty && s->stream != D.stream) {
        HIP_TRY(hipEventRecord(D.ev[k], s->stream));
        HIP_TRY(hipStreamWaitEvent(D.stream, D.ev[k], 0));
      }
    }
    if ((int)D.members.size() <= GROUP_INLINE_SHARDS) {            // the argument blocks by value: nothing to upload
      const int k_n = (int)D.members.size();
      hipLaunchKernelGGL(group_trial_inline_kernel, dim3(D.grid), dim3(TPB), 0, D.stream, D.args_host[0], D.args_host[k_n > 1 ? 1 : 0],
                         k_n, k_n > 1 ? D.base[1] : D.grid, D.grid);
    } else {
      HIP_TRY(hipMemcpyAsync(D.args_dev, D.args_host, sizeof(GroupTrialArgs) * D.members.size(), hipMemcpyHostToDevice, D.stream));
      GroupDeviceArgs da{};
      da.shard = D.args_dev; da.nshards = (int)D.members.size();
      for (size_t k = 0; k <= D.members.size(); ++k) da.base[k] = D.base[k];
      hipLaunchKernelGGL(group_trial_kernel, dim3(D.grid), dim3(TPB), 0, D.stream, da);
    }
    HIP_TRY(hipGetLastError());
    // ... and whatever is queued on the members' streams next comes after this launch (check_handle: lazily)
    HIP_TRY(hipEventRecord(D.ev_done, D.stream));
  }
  g.members_dirty = false;
  g.join_pending = true;
  g.xepoch += 2;
  const auto t_issued = std::chrono::steady_clock::now();
  bool failed = false;
  double sums[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < L.count; ++i) {
    pdhg_handle *s = L.p[i];
    HIP_TRY(hipSetDevice(s->device));
    double r[5];
    const int rc = wait_result_word(s, r, true);
    if (rc) return rc;
    failed = failed || s->res_error != 0.0;
    for (int q = 0; q < 4; ++q) sums[q] = (i == 0) ? r[q] : sums[q] + r[q];      // rank order (L.p is ascending in rank)
  }
  g.t_issue += std::chrono::duration<double>(t_issued - t_begin).count();
  g.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_issued).count();
  g.n_trials += 1;
  if (failed) {
    // a barrier ran into its spin limit (the launches were not all co-resident): every workgroup still ran every phase, so
    // the deferred average updates are applied exactly once; x', y', A'y' and the sums are not trustworthy -- the caller
    // repeats the trial on the ordinary group path (its inputs are untouched) and the group stays there
    g.coop_mode = 0;
    g.coop_fallbacks += 1;
    fprintf(stderr, "[pdhg_hip] group trial kernel: a barrier timed out -- this group uses the per-launch path from here on\n");
    return check_handle(L.p[0]) ? -1 : 1;        // (the members' streams wait for the failed launches before the repeat)
  }
  for (int q = 0; q < 4; ++q) out[q] = sums[q];
  out[4] = 0.0;
  g.coop_trials += 1;
  return 0;
}

// Row-partitioned group: the dual half of a trial.  xbar's owned slices are ready.
static int trial_dual_group(const Shards &L, double step_size, double primal_weight, double out[5]) {
  DistGroup &g = *L.g;
  pdhg_handle *lead = L.p[0];
  int rc;
  const auto t_begin = std::chrono::steady_clock::now();
  {
    ProfScope ps(lead, PDHG_K_ALLGATHER);
    if ((rc = dist_all_gather(g, [](pdhg_handle *s) { return s->xbar; }, g.S))) return rc;
    // QP: Q acts on full vectors, so x' is kept full as well (x becomes x' at accept)
    if (lead->has_q && (rc = dist_all_gather(g, [](pdhg_handle *s) { return s->x_next; }, g.S))) return rc;
  }
  if (!g.overlap) {
    FOR_SHARDS(L, s) {
      if ((rc = launch_dual(s, primal_weight * step_size))) return rc;
      if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc;      // t_p = A_p' y'_p, all n columns
    }
    ProfScope ps(lead, PDHG_K_REDUCE_SCATTER);
    if ((rc = dist_reduce_scatter(g, [](pdhg_handle *s) { return s->aty_next; }, g.S))) return rc;
  } else {
    // t_p in parts: a shard whose A_p' uses the tiled layout launches it one residency
    // round at a time (256 CUs x 2 workgroups: a smaller launch would idle CUs for the whole
    // sweep); as soon as the rows of slice k are complete, slice k is reduced to rank k on
    // the comm stream while the next round computes.  The sequence of collectives (slice
    // 0, 1, ..., P-1) is the same on every rank however the local product is cut.
    FOR_SHARDS(L, s) { if ((rc = launch_dual(s, primal_weight * step_size))) return rc; }
    const char *rw_env = dev_env("PDHG_DIST_ROUND_WGS");            // tests use a finer granule on small problems
    const int round_wgs = rw_env ? std::max(1, atoi(rw_env)) : 256 * 2;
    std::vector<int> issued((size_t)L.count, 0), next_wg((size_t)L.count, 0);
    int k_issued = 0;
    while (k_issued < g.world) {
      // every local shard advances until slice k_issued is complete on it
      for (int i = 0; i < L.count; ++i) {
        pdhg_handle *s = L.p[i];
        HIP_TRY(hipSetDevice(s->device));
        const CsrDev &T = s->At;
        const int64_t need = std::min<int64_t>(s->n, (int64_t)(k_issued + 1) * g.S);   // rows [0, need) must be done
        if (!T.tiled) {
          if (issued[(size_t)i] == 0) {
            if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc;
            issued[(size_t)i] = 1;
          }
        } else {
          while (next_wg[(size_t)i] < T.grid || issued[(size_t)i] == 0) {
            const int g0 = next_wg[(size_t)i];
            const bool covered = g0 >= T.grid || (int64_t)T.wg_first_row[(size_t)g0] >= need;
            if (covered && issued[(size_t)i] != 0) break;
            int g1 = std::min(T.grid, g0 + round_wgs);
            if (T.grid - g1 < round_wgs / 2) g1 = T.grid;       // no runt round at the end
            ProfScope ps(s, PDHG_K_SPMV_ATY);
            if ((rc = launch_spmv_plain_part(s, T, s->y_next, s->aty_next, g0, g1, issued[(size_t)i] == 0))) return rc;
            issued[(size_t)i] = 1;
            next_wg[(size_t)i] = g1;
          }
        }
        HIP_TRY(hipEventRecord(s->ev_part[(size_t)k_issued], s->stream));
      }
      if ((rc = dist_reduce_slice_async(g, [](pdhg_handle *s) { return s->aty_next; }, g.S, k_issued))) return rc;
      ++k_issued;
    }
    ProfScope ps(lead, PDHG_K_REDUCE_SCATTER);     // what is left of the exchange after the product
    if ((rc = dist_join_comm(g))) return rc;
  }
  FOR_SHARDS(L, s) {
    {
      ProfScope ps(s, PDHG_K_INTERACTION);
      const int64_t o = s->clo;
      hipLaunchKernelGGL(interaction_kernel, dim3(ew_grid(s->cn)), dim3(TPB), 0, s->stream, (int)s->cn, s->x + o,
                         s->x_next + o, s->aty + o, s->aty_next + o, s->pAt, s->pAt_stride);
      HIP_TRY(hipGetLastError());
    }
    int qcount = 0;
    if ((rc = launch_q_interaction(s, &qcount))) return rc;   // replicated: identical on every shard
    if ((rc = launch_final(s, s->pAt, ew_grid(s->cn), s->pAt_stride, s->pA, s->A.slots(), qcount))) return rc;
  }
  double r[5];
  const auto t_issued = std::chrono::steady_clock::now();
  // [0..4) are added in rank order; [4] (dx'Q dx, replicated: the same value on every rank) is "maxed"
  if ((rc = combine_scalars(L, 5, 4, r))) return rc;
  g.t_issue += std::chrono::duration<double>(t_issued - t_begin).count();
  g.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_issued).count();
  g.n_trials += 1;
  for (int q = 0; q < 4; ++q) out[q] = r[q];
  out[4] = 0.5 * r[4];
  return 0;
}

int pdhg_trial_primal(pdhg_handle *h, double step_size, double primal_weight) {
  int rc = check_handle(h);
  if (rc) return rc;
  const Shards L = shards_of(h);
  FOR_SHARDS(L, s) { if ((rc = launch_primal(s, step_size / primal_weight, 0.0, false))) return rc; }
  return 0;
}

// true when the trial will be taken as persistent group launches (group_kernel.hpp), which queue nothing on the members' own
// streams -- the same predicate group_coop_eligible() ends on (a profiled group takes the per-launch path, which DOES)
static bool trial_stays_off_member_streams(const pdhg_handle *h) {
  return h && h->grp && h->grp->coop_mode == 1 && !h->grp->sh.empty() && !h->grp->sh[0]->profile;
}

int pdhg_trial_dual(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  int rc = check_handle(h, !trial_stays_off_member_streams(h));
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  const Shards L = shards_of(h);
  if (!L.g && coop_eligible(h)) {         // Malitsky-Pock retries: xbar + the dual half
    if ((rc = coop_trial(h, step_size, primal_weight, theta, true, out)) != 1) return rc;    // 1: not run / timed out, repeat below
  }
  if (L.g && group_coop_eligible(L)) {
    if ((rc = group_coop_trial(L, TrialArgs{step_size, primal_weight, theta, false}, out)) != 1) return rc;
  }
  if (L.g && L.g->pool && !L.p[0]->profile) return trial_group_mt(L, TrialArgs{step_size, primal_weight, theta, false}, out);
  FOR_SHARDS(L, s) { if ((rc = launch_xbar(s, theta))) return rc; }
  if (L.g) return trial_dual_group(L, step_size, primal_weight, out);
  return trial_dual_single(h, step_size, primal_weight, out);
}

int pdhg_trial_step(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  RoctxRange roctx_range("pdhg_trial_step");
  int rc = check_handle(h, !trial_stays_off_member_streams(h));
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  const Shards L = shards_of(h);
  if (!L.g && coop_eligible(h)) {
    if ((rc = coop_trial(h, step_size, primal_weight, theta, false, out)) != 1) return rc;   // 1: not run / timed out, repeat below
  }
  if (!L.g && graph_eligible(h)) return graph_trial(h, step_size, primal_weight, theta, out);
  if (L.g && group_coop_eligible(L)) {
    if ((rc = group_coop_trial(L, TrialArgs{step_size, primal_weight, theta, true}, out)) != 1) return rc;
  }
  if (L.g && L.g->pool && !L.p[0]->profile) return trial_group_mt(L, TrialArgs{step_size, primal_weight, theta, true}, out);
  FOR_SHARDS(L, s) { if ((rc = launch_primal(s, step_size / primal_weight, theta, true))) return rc; }
  if (L.g) return trial_dual_group(L, step_size, primal_weight, out);
  return trial_dual_single(h, step_size, primal_weight, out);
}

int pdhg_accept(pdhg_handle *h0, double avg_weight) {
  RoctxRange roctx_range("pdhg_accept");
  // (a lazy accept with nothing pending queues no work: the iterates are swapped on the host)
  int rc = check_handle(h0, !(trial_stays_off_member_streams(h0) && h0->lazy_accept && !h0->pend_x && !h0->pend_y));
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;   // two accepts without a trial in between
  bump_version(L);
  FOR_SHARDS(L, h) {
    if (h->lazy_accept) {
      h->pend_x = h->pend_y = true;         // K7 rides on the next trial's kernels
      h->pend_w = avg_weight;
    } else {
      ProfScope ps(h, PDHG_K_ACCEPT);
      const int64_t o = h->clo;
      hipLaunchKernelGGL(accept_kernel, dim3(ew_grid(std::max(h->cn, h->m))), dim3(TPB), 0, h->stream, (int)h->cn,
                         (int)h->m, avg_weight, h->x_next + o, h->sum_x + o, h->y_next, h->sum_y);
      HIP_TRY(hipGetLastError());
    }
    std::swap(h->x, h->x_next);
    std::swap(h->y, h->y_next);
    std::swap(h->aty, h->aty_next);
    h->sum_x_count += 1; h->sum_y_count += 1;
    h->sum_x_weights += avg_weight; h->sum_y_weights += avg_weight;
  }
  return 0;
}

/* take_step(::AdaptiveStepsizeParams, ...) -- src/primal_dual_hybrid_gradient.jl:653-731 --
 * with its host part in C: the retry loop, compute_interaction_and_movement's scalar
 * arithmetic (:527-549), the step-size rule (:713-729) and the accept.  The same
 * statements as primal_dual_hybrid_gradient.py::take_step_adaptive (bitwise equal
 * results; tests/test_gpu_native_take_step.py); what it removes is the host
 * language's per-call overhead between the trial and the accept. */
// step_on_entry: the step size the take_step was entered with (the average's weight, pdhg.jl:512) -- equal to
// *step_size_io except when a multi-step kernel handed back a take_step it had begun (some trials already rejected)
static int take_step_adaptive_from(pdhg_handle *h, double reduction_exponent, double growth_exponent,
                                   double *step_size_io, double step_on_entry, double primal_weight,
                                   int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
                                   int *numerical_error_out);
int pdhg_take_step_adaptive(pdhg_handle *h, double reduction_exponent, double growth_exponent,
                            double *step_size_io, double primal_weight, int64_t *total_number_iterations_io,
                            double *cumulative_kkt_passes_io, int *numerical_error_out) {
  if (!h || !step_size_io || !total_number_iterations_io || !cumulative_kkt_passes_io || !numerical_error_out)
    return fail(-1, "null argument");
  return take_step_adaptive_from(h, reduction_exponent, growth_exponent, step_size_io, *step_size_io, primal_weight,
                                 total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out);
}
static int take_step_adaptive_from(pdhg_handle *h, double reduction_exponent, double growth_exponent,
                                   double *step_size_io, double step_on_entry, double primal_weight,
                                   int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
                                   int *numerical_error_out) {
  double step_size = *step_size_io;
  *numerical_error_out = 0;
  bool done = false;
  while (!done) {
    *total_number_iterations_io += 1;
    double raw[5];
    int rc = pdhg_trial_step(h, step_size, primal_weight, 1.0, raw);
    if (rc) return rc;
    *cumulative_kkt_passes_io += 1;
    const double k1 = (double)(*total_number_iterations_io + 1);
    const StepRule rule = adaptive_step_rule(raw, primal_weight, step_size, pow(k1, -reduction_exponent), pow(k1, -growth_exponent));
    if (rule.numerical_error) {
      *numerical_error_out = 1;
      break;
    }
    if (rule.accept) {
      if ((rc = pdhg_accept(h, step_on_entry))) return rc;   // weight = step size on entry (pdhg.jl:512)
      done = true;
    }
    step_size = rule.next_step;
  }
  *step_size_io = step_size;
  return 0;
}

// Does pdhg_take_steps_adaptive take this handle's batches with the multi-step kernel (steps_kernel)?
static bool device_loop_for(pdhg_handle *h) {
  // Several take_steps per launch (steps_kernel: the rule on the device; stream-layout LPs on one handle).  Bitwise the
  // per-trial launches (tests/test_gpu_device_loop.py) and faster on every grid measured but one tie: L1-SVM 19.7k ->
  // 23.4k it/s, random 100K 22.5k -> 28.6k, 3000 x 2500 32.7k -> 52.7k (trial_kernel.hpp, profiles/r03_trial_kernel.txt).
  // PDHG_DEVICE_LOOP=0 / 1: never / whenever eligible; PDHG_DEVICE_LOOP_MAX_WGS: largest grid it is the default for.
  const char *dl_env = getenv("PDHG_DEVICE_LOOP");
  bool device_loop = dl_env && dl_env[0] == '1';
  if (!dl_env && !h->grp && !h->profile && !h->has_q && check_handle(h) == 0 && coop_eligible(h)) {
    static const int max_wgs = dev_env("PDHG_DEVICE_LOOP_MAX_WGS") ? atoi(dev_env("PDHG_DEVICE_LOOP_MAX_WGS")) : (1 << 30);
    device_loop = h->coop_grid <= max_wgs;
  }
  return device_loop;
}

/* `n_steps` consecutive take_steps (the iterations optimize() runs between two termination
 * evaluations, pdhg.jl:862-1046: nothing but take_step happens there).  Stops after the step that
 * raised numerical_error, like the reference's loop does at the top of the next iteration. */
int pdhg_take_steps_adaptive(pdhg_handle *h, int64_t n_steps, double reduction_exponent, double growth_exponent,
                             double *step_size_io, double primal_weight, int64_t *total_number_iterations_io,
                             double *cumulative_kkt_passes_io, int *numerical_error_out, int64_t *steps_done_out) {
  RoctxRange roctx_range("pdhg_take_steps_adaptive");
  if (!steps_done_out) return fail(-1, "null argument");
  if (n_steps < 0) return fail(-2, "pdhg_take_steps_adaptive: n_steps < 0");
  *steps_done_out = 0;
  if (!h || !step_size_io || !total_number_iterations_io || !cumulative_kkt_passes_io || !numerical_error_out)
    return fail(-1, "null argument");
  *numerical_error_out = 0;
  const bool device_loop = device_loop_for(h);
  int64_t s = 0;
  while (s < n_steps) {
    double entry = 0.0;         // nonzero: a multi-step kernel ended inside a take_step (its table of powers ran out)
    if (n_steps - s >= 2 && !h->grp && check_handle(h) == 0 && small_lp_eligible(h)) {
      // a small LP: the batch in one workgroup with the vectors in LDS (small_lp_kernel.hpp)
      int64_t k = 0;
      const int rc = small_lp_steps(h, n_steps - s, reduction_exponent, growth_exponent, step_size_io, primal_weight,
                                    total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out, &k, &entry);
      if (rc != 0 && rc != 1) return rc;
      if (rc == 0) {
        s += k;
        *steps_done_out = s;
        if (*numerical_error_out) break;
        if (k > 0 && entry == 0.0) continue;
      }
    }
    // (entry != 0: the small-LP launch above ended inside a take_step -- its step size on entry must reach the accept
    //  of THAT take_step, so it is finished launch by launch below, never handed to a fresh multi-step launch)
    if (entry == 0.0 && device_loop && n_steps - s >= 2 && !h->grp && !h->profile && check_handle(h) == 0) {
      int64_t k = 0;
      const int rc = coop_steps(h, n_steps - s, reduction_exponent, growth_exponent, step_size_io, primal_weight,
                                total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out, &k, &entry);
      if (rc != 0 && rc != 1) return rc;
      if (rc == 0) {
        s += k;
        *steps_done_out = s;
        if (*numerical_error_out) break;
        if (k > 0 && entry == 0.0) continue;   // (k == 0: trial budget spent on rejections, or a time-out: take the next step singly)
      }
    }
    if (s >= n_steps) break;
    // one take_step, launch by launch -- or the rest of one that a multi-step kernel began (entry: its step size on entry)
    const int rc = take_step_adaptive_from(h, reduction_exponent, growth_exponent, step_size_io,
                                           entry != 0.0 ? entry : *step_size_io, primal_weight,
                                           total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out);
    if (rc) return rc;
    *steps_done_out = ++s;
    if (*numerical_error_out) break;
  }
  return 0;
}

int pdhg_add_current_primal_to_average(pdhg_handle *h0, double weight) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(accept_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, 0, weight,
                       h->x + o, h->sum_x + o, h->y, h->sum_y);
    HIP_TRY(hipGetLastError());
    h->sum_x_count += 1;
    h->sum_x_weights += weight;
  }
  return 0;
}

int pdhg_get_average_info(pdhg_handle *h, int64_t counts[2], double weights[2]) {
  if (!h) return fail(-1, "null handle");
  counts[0] = h->sum_x_count; counts[1] = h->sum_y_count;
  weights[0] = h->sum_x_weights; weights[1] = h->sum_y_weights;
  return 0;
}

int pdhg_get_average(pdhg_handle *h0, double *x_avg, double *y_avg) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  if (x_avg) {
    FOR_SHARDS(L, h) {
      const int64_t o = h->clo;
      hipLaunchKernelGGL(div_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, h->sum_x + o,
                         h->sum_x_weights, h->tmp_n + o);
      HIP_TRY(hipGetLastError());
    }
    if ((rc = cols_to_host(L, [](pdhg_handle *s) { return s->tmp_n; }, x_avg))) return rc;
  }
  if (y_avg) {
    FOR_SHARDS(L, h) {
      hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->tmp_m);
      HIP_TRY(hipGetLastError());
    }
    if ((rc = rows_to_host(L, [](pdhg_handle *s) { return s->tmp_m; }, y_avg))) return rc;
  }
  return sync_all(L);
}

int pdhg_reset_average(pdhg_handle *h0) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  bump_version(L);
  FOR_SHARDS(L, h) {
    HIP_TRY(hipMemsetAsync(h->sum_x, 0, sizeof(double) * (size_t)std::max<int64_t>(h->n, 1), h->stream));
    HIP_TRY(hipMemsetAsync(h->sum_y, 0, sizeof(double) * (size_t)std::max<int64_t>(h->m, 1), h->stream));
    h->pend_x = h->pend_y = false;          // a deferred update belongs to the sums being discarded
    h->sum_x_count = h->sum_y_count = 0;
    h->sum_x_weights = h->sum_y_weights = 0.0;
  }
  return 0;
}

// after x (owned slices) changed outside a trial: QP groups keep x full on every shard
static int refresh_full_x(const Shards &L) {
  if (!L.g || !L.p[0]->has_q) return 0;
  return dist_all_gather(*L.g, [](pdhg_handle *s) { return s->x; }, L.g->S);
}

int pdhg_restart_to_average(pdhg_handle *h0) {
  RoctxRange roctx_range("pdhg_restart_to_average");
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  if (h0->sum_x_count == 0 || h0->sum_y_count == 0) return fail(-1, "average is empty");
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(div_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, h->sum_x + o, h->sum_x_weights, h->x + o);
    hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->y);
    HIP_TRY(hipGetLastError());
  }
  if ((rc = refresh_full_x(L))) return rc;
  return dual_product(L, [](pdhg_handle *s) { return (const double *)s->y; }, [](pdhg_handle *s) { return s->aty; });
}

int pdhg_get_current(pdhg_handle *h0, double *x, double *y, double *aty) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if (x && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->x; }, x))) return rc;
  if (y && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->y; }, y))) return rc;
  if (aty && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->aty; }, aty))) return rc;
  return sync_all(L);
}

int pdhg_get_trial(pdhg_handle *h0, double *x_next, double *y_next, double *aty_next) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if (x_next && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->x_next; }, x_next))) return rc;
  if (y_next && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->y_next; }, y_next))) return rc;
  if (aty_next && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->aty_next; }, aty_next))) return rc;
  return sync_all(L);
}

int pdhg_set_current(pdhg_handle *h0, const double *x, const double *y) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  if (x && (rc = cols_from_host(L, x, [](pdhg_handle *s) { return s->x; }))) return rc;
  if (y && (rc = rows_from_host(L, y, [](pdhg_handle *s) { return s->y; }))) return rc;
  if ((rc = dual_product(L, [](pdhg_handle *s) { return (const double *)s->y; }, [](pdhg_handle *s) { return s->aty; }))) return rc;
  return sync_all(L);
}

int pdhg_spmv(pdhg_handle *h0, const double *x, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!x || !out) return fail(-1, "null vector");
  const Shards L = shards_of(h0);
  if ((rc = cols_from_host(L, x, [](pdhg_handle *s) { return s->tmp_n; }))) return rc;
  FOR_SHARDS(L, h) {
    EpiArgs e{};
    e.out = h->tmp_m;
    if ((rc = launch_spmv<MODE_PLAIN, 0>(h, h->A, h->tmp_n, e))) return rc;
  }
  if ((rc = rows_to_host(L, [](pdhg_handle *s) { return s->tmp_m; }, out))) return rc;
  return sync_all(L);
}

int pdhg_spmv_t(pdhg_handle *h0, const double *y, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!y || !out) return fail(-1, "null vector");
  const Shards L = shards_of(h0);
  if ((rc = rows_from_host(L, y, [](pdhg_handle *s) { return s->tmp_m; }))) return rc;
  // the partial products go through tmp_n (n_alloc long); a group's gather-to-host then uses dn_buf
  if ((rc = dual_product(L, [](pdhg_handle *s) { return (const double *)s->tmp_m; }, [](pdhg_handle *s) { return s->tmp_n; }))) return rc;
  if ((rc = cols_to_host(L, [](pdhg_handle *s) { return s->tmp_n; }, out))) return rc;
  return sync_all(L);
}

// ---- evaluation branch on the device (N1) -----------------------------------

static int ev_alloc(pdhg_handle *h) {
  if (h->ev_partials) return 0;
  int rc;
  {
    // the evaluation kernels reduce up to 30 quantities per workgroup and a second stage reads every workgroup's
    // partials: two elements per thread and at most 1024 workgroups measured best (L1-SVM 229K elements: 448
    // workgroups 67 us per trust-region call against 82 with 895; 2M elements: 977 workgroups 123 us against 147 with 2048)
    static const int per_thread = dev_env("PDHG_EV_ELEMS") ? std::max(1, atoi(dev_env("PDHG_EV_ELEMS"))) : 2;
    h->ev_grid = std::min(1024, ew_grid((h->n + h->m + per_thread) / per_thread));
  }
  if ((rc = alloc_zero(&h->ev_partials, (int64_t)EV_MAXQ * h->ev_grid))) return rc;
  if ((rc = alloc_zero(&h->ev_ax, h->m))) return rc;
  if ((rc = alloc_zero(&h->ev_aty, h->n_alloc))) return rc;
  for (int k = 0; k < 3; ++k) {
    if ((rc = alloc_zero(&h->ev_cax[k], h->m))) return rc;
    if ((rc = alloc_zero(&h->ev_caty[k], h->n_alloc))) return rc;
  }
  if (h->grp && (rc = alloc_zero(&h->ev_xg, h->n_alloc))) return rc;
  if ((rc = alloc_zero(&h->px_avg, h->n))) return rc;
  if ((rc = alloc_zero(&h->py_avg, h->m))) return rc;
  if ((rc = alloc_zero(&h->x_r, h->n))) return rc;   // zeros == the initial restart point (pdhg.jl:869)
  if ((rc = alloc_zero(&h->y_r, h->m))) return rc;
  return 0;
}

// second stage of every shard's block partials (ns sums then nm maxes), then the
// combination over ranks in rank order
// the pinned result words of the evaluation reductions (multi_final_kernel, tr_small_kernel): k values, checksum, sequence number
static int ev_ensure_host(pdhg_handle *h) {
  if (!h->ev_host) {
    HIP_TRY(hipHostMalloc((void **)&h->ev_host, (EV_HOST_SLOTS + 2) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
    memset(h->ev_host, 0, (EV_HOST_SLOTS + 2) * sizeof(double));
  }
  return 0;
}
static int ev_wait_host(pdhg_handle *h, int k, unsigned long long seq, double *out) {
  const volatile unsigned long long *bits = reinterpret_cast<const volatile unsigned long long *>(h->ev_host);
  auto ready = [&]() -> bool {
    if (bits[EV_HOST_SEQ] != seq) return false;
    unsigned long long w[EV_HOST_SLOTS];
    unsigned long long ck = EV_CHECK_SALT ^ seq ^ ((unsigned long long)k << 56);
    for (int q = 0; q < k; ++q) { w[q] = bits[q]; ck ^= w[q] * (2ull * (unsigned long long)q + 1ull); }
    if (ck != bits[EV_HOST_CK]) return false;
    for (int q = 0; q < k; ++q) memcpy(&out[q], &w[q], 8);
    return true;
  };
  for (long spin = 0; spin < 40000000L; ++spin) {
    if (ready()) return 0;
    if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(h->stream) != hipErrorNotReady) break;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (ready()) return 0;
  return fail(998, "evaluation reduction finished without publishing its results");
}

// The evaluation reductions publish into pinned host memory (one handle) unless PDHG_EVAL_HOST_WORD=0 (dev): decided in
// ONE place and per call -- pdhg_eval_point's combined 22-quantity reduction (which needs the max mask) and ev_finish
// must never disagree (a cached copy here once could: sums where maxima belong).
static bool eval_host_word() {
  const char *hw = dev_env("PDHG_EVAL_HOST_WORD");
  return !(hw && hw[0] == '0');
}

static int ev_finish(const Shards &L, int ns, int nm, double *out, unsigned max_mask = 0) {
  if (ns + nm > EV_MAXQ) return fail(-1, "too many scalars in one reduction");
  const bool host_word = eval_host_word();
  if (max_mask != 0 && (L.g || !host_word)) return fail(-1, "a mixed sum / max reduction needs the host-word form");
  if (!L.g && host_word) {
    // one handle: the second stage publishes into pinned memory and the host polls (see multi_final_kernel)
    pdhg_handle *h = L.p[0];
    const int k = ns + nm;
    if (k > EV_HOST_SLOTS) return fail(-1, "too many scalars in one reduction");
    HIP_TRY(hipSetDevice(h->device));
    int rc0 = ev_ensure_host(h);
    if (rc0) return rc0;
    const unsigned long long seq = ++h->ev_seq;
    hipLaunchKernelGGL(multi_final_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->ev_partials, h->ev_grid,
                       h->ev_grid, ns, nm, h->scal_dev, h->ev_host, seq, max_mask);
    HIP_TRY(hipGetLastError());
    return ev_wait_host(h, k, seq, out);
  }
  FOR_SHARDS(L, h) {
    hipLaunchKernelGGL(multi_final_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->ev_partials, h->ev_grid,
                       h->ev_grid, ns, nm, h->scal_dev, (double *)nullptr, 0ull, 0u);
    HIP_TRY(hipGetLastError());
  }
  return combine_scalars(L, ns + nm, ns, out);
}

// px: column vector (valid on the owned slice), py: this shard's rows
static int select_point(pdhg_handle *h, int point, const double **px, const double **py) {
  int rc = ev_alloc(h);
  if (rc) return rc;
  if (point == PDHG_POINT_CURRENT) { *px = h->x; *py = h->y; return 0; }
  if (point == PDHG_POINT_RESTART) { *px = h->x_r; *py = h->y_r; return 0; }
  if (point == PDHG_POINT_AVERAGE) {
    if (h->sum_x_count == 0 || h->sum_y_count == 0) return fail(-1, "average is empty");
    if (h->avg_version != h->state_version) {
      const int64_t o = h->clo;
      hipLaunchKernelGGL(div_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, h->sum_x + o, h->sum_x_weights, h->px_avg + o);
      hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->py_avg);
      HIP_TRY(hipGetLastError());
      h->avg_version = h->state_version;
    }
    *px = h->px_avg; *py = h->py_avg;
    return 0;
  }
  return fail(-1, "unknown point selector");
}

// A*x (this shard's rows), A'*y (owned slice) and, for a QP, Q*x (full) at `point`
// on every shard; cached for CURRENT / AVERAGE until the state changes.  Results in
// h->pt_ax / pt_aty / pt_qx together with the point itself in h->pt_x / pt_y.
static int point_products(const Shards &L, int point) {
  int rc;
  static const bool cache_off = dev_env("PDHG_NO_EVAL_CACHE") != nullptr;   // debugging aid
  const bool cached = !cache_off && (point == PDHG_POINT_CURRENT || point == PDHG_POINT_AVERAGE ||
                                     point == PDHG_POINT_RESTART);
  bool fresh = true;
  FOR_SHARDS(L, h) {
    if ((rc = select_point(h, point, &h->pt_x, &h->pt_y))) return rc;
    h->pt_ax = h->ev_ax; h->pt_aty = h->ev_aty;
    double **dqx = &h->ev_qx;
    bool f = true;
    if (cached) {
      const int k = point == PDHG_POINT_CURRENT ? 0 : (point == PDHG_POINT_AVERAGE ? 1 : 2);
      h->pt_ax = h->ev_cax[k]; h->pt_aty = h->ev_caty[k]; dqx = &h->ev_cqx[k];
      if (k < 2) {
        f = h->ev_cversion[k] != h->state_version;
        h->ev_cversion[k] = h->state_version;
      } else {
        const uint64_t key = (h->matrix_version << 32) + h->restart_version;
        f = h->ev_rkey != key;
        h->ev_rkey = key;
      }
    }
    if (h->has_q && !*dqx) {
      if ((rc = alloc_zero(dqx, h->n))) return rc;
      f = true;
    }
    h->pt_qx = h->has_q ? *dqx : nullptr;
    fresh = f;              // shards move in lock step: the same answer on all of them
  }
  if (!fresh) return 0;
  // full x at the point on every shard (a plain handle's vectors are full already)
  if (L.g) {
    if ((rc = gather_cols_device(L, [](pdhg_handle *s) { return s->pt_x; }, [](pdhg_handle *s) { return s->ev_xg; }))) return rc;
  }
  FOR_SHARDS(L, h) {
    const double *xfull = L.g ? h->ev_xg : h->pt_x;
    EpiArgs e{};
    e.out = h->pt_ax;
    if ((rc = launch_spmv<MODE_PLAIN, 0>(h, h->A, xfull, e))) return rc;
    if (h->has_q) {
      e.out = h->pt_qx;
      if ((rc = launch_spmv<MODE_PLAIN, 2>(h, h->Q, xfull, e))) return rc;
    }
  }
  return dual_product(L, [](pdhg_handle *s) { return s->pt_y; }, [](pdhg_handle *s) { return s->pt_aty; });
}

int pdhg_set_original_problem(pdhg_handle *h0, const double *constraint_rescaling,
                              const double *variable_rescaling, const double *c_o, const double *b_o,
                              const double *lb_o, const double *ub_o) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!constraint_rescaling || !variable_rescaling || !c_o || !lb_o || !ub_o || (h0->m_global > 0 && !b_o))
    return fail(-1, "null input array");
  const Shards L = shards_of(h0);
  FOR_SHARDS(L, h) {
    auto up = [&](double **dst, const double *src, int64_t len) -> int {
      if (!*dst) { int r2 = alloc_zero(dst, len); if (r2) return r2; }
      if (len > 0) { HIP_TRY(hipMemcpy(*dst, src, sizeof(double) * (size_t)len, hipMemcpyHostToDevice)); HIP_TRY(hipStreamSynchronize(nullptr)); }
      return 0;
    };
    // row vectors arrive with their GLOBAL length: a shard keeps its rows
    if ((rc = up(&h->E, constraint_rescaling + h->row_lo, h->m))) return rc;
    if ((rc = up(&h->b_o, b_o ? b_o + h->row_lo : nullptr, h->m))) return rc;
    if ((rc = up(&h->Dv, variable_rescaling, h->n))) return rc;
    if ((rc = up(&h->c_o, c_o, h->n))) return rc;
    if ((rc = up(&h->lb_o, lb_o, h->n))) return rc;
    if ((rc = up(&h->ub_o, ub_o, h->n))) return rc;
    h->has_original = true;
    if ((rc = ev_alloc(h))) return rc;
  }
  return 0;
}

int pdhg_eval_point(pdhg_handle *h0, int point, double out[24]) {
  RoctxRange roctx_range("pdhg_eval_point");
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!h0->has_original) return fail(-1, "pdhg_set_original_problem has not been called");
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  if ((rc = point_products(L, point))) return rc;
  if (!L.g && eval_host_word()) {                              // (read per call: tests compare the two forms in one process)
    // one handle: the row and the column kernels leave their block partials side by side (8 + 14 quantities), ONE second
    // stage reduces all 22 and the host makes one round trip instead of two.  Same partials, same order per quantity:
    // the same bits as the two-round form below.
    pdhg_handle *h = L.p[0];
    hipLaunchKernelGGL(eval_rows_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->m, (int)h->num_eq,
                       h->pt_ax, h->pt_y, h->E, h->b_o, h->ev_partials, h->ev_grid);
    hipLaunchKernelGGL(eval_cols_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, h->pt_aty,
                       h->pt_qx, h->pt_x, h->Dv, h->c_o, h->lb_o, h->ub_o, h->ev_partials + (size_t)8 * h->ev_grid, h->ev_grid);
    HIP_TRY(hipGetLastError());
    double r[22];
    // quantities 0-3 sums, 4-7 maxes (rows); 8-14 sums, 15-21 maxes (columns)
    if ((rc = ev_finish(L, 22, 0, r, 0xF0u | (0x7Fu << 15)))) return rc;
    for (int q = 0; q < 8; ++q) out[q] = r[q];
    for (int q = 0; q < 6; ++q) { out[8 + q] = r[8 + q]; out[14 + q] = r[8 + 7 + q]; }
    out[20] = r[8 + 6]; out[21] = r[8 + 13]; out[22] = out[23] = 0.0;
    return 0;
  }
  FOR_SHARDS(L, h) {
    hipLaunchKernelGGL(eval_rows_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->m, (int)h->num_eq,
                       h->pt_ax, h->pt_y, h->E, h->b_o, h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  if ((rc = ev_finish(L, 4, 4, out))) return rc;
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(eval_cols_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, h->pt_aty + o,
                       h->pt_qx ? h->pt_qx + o : nullptr, h->pt_x + o, h->Dv + o, h->c_o + o, h->lb_o + o,
                       h->ub_o + o, h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  double r[14];
  if ((rc = ev_finish(L, 7, 7, r))) return rc;
  for (int q = 0; q < 6; ++q) { out[8 + q] = r[q]; out[14 + q] = r[7 + q]; }
  out[20] = r[6]; out[21] = r[13]; out[22] = out[23] = 0.0;
  return 0;
}

int pdhg_save_restart_point(pdhg_handle *h0) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  FOR_SHARDS(L, h) {
    if ((rc = ev_alloc(h))) return rc;
    h->restart_version += 1;
    if (h->cn > 0)
      HIP_TRY(hipMemcpyAsync(h->x_r + h->clo, h->x + h->clo, sizeof(double) * (size_t)h->cn, hipMemcpyDeviceToDevice, h->stream));
    if (h->m > 0)
      HIP_TRY(hipMemcpyAsync(h->y_r, h->y, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToDevice, h->stream));
  }
  return 0;
}

static int dist2_common(pdhg_handle *h0, int point, bool to_restart, double out[2]) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  FOR_SHARDS(L, h) {
    const double *px, *py;
    if ((rc = select_point(h, point, &px, &py))) return rc;
    const int64_t o = h->clo;
    hipLaunchKernelGGL(dist2_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)h->m, px + o,
                       to_restart ? (const double *)(h->x_r + o) : (const double *)nullptr, py,
                       to_restart ? (const double *)h->y_r : (const double *)nullptr, h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  return ev_finish(L, 2, 0, out);
}

int pdhg_distance_to_restart(pdhg_handle *h, int point, double out[2]) { return dist2_common(h, point, true, out); }
int pdhg_point_sumsq(pdhg_handle *h, int point, double out[2]) { return dist2_common(h, point, false, out); }

int pdhg_get_point(pdhg_handle *h0, int point, double *x, double *y) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  FOR_SHARDS(L, h) { if ((rc = select_point(h, point, &h->pt_x, &h->pt_y))) return rc; }
  if (x && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->pt_x; }, x))) return rc;
  if (y && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->pt_y; }, y))) return rc;
  return sync_all(L);
}

static inline uint64_t d2bits(double v) { uint64_t b; memcpy(&b, &v, 8); return b; }
static inline double bits2d(uint64_t b) { double v; memcpy(&v, &b, 8); return v; }

// ---- the trust-region problem as ONE persistent launch (tr_coop_kernel.hpp) ----
// Decided once per handle: a single handle (no shard group) whose n + m elements fit PDHG_TR_COOP_MAX (default 1M;
// measured per call, 5 passes: n + m = 40K 94 -> 73 us, 229K (L1-SVM) 104 -> 80, 500K 138 -> 90, 1M 143 -> 110, 2M 167 -> 160:
// beyond that a pass is bandwidth, not latency, and the multi-launch kernels' 1 024 workgroups stream it as fast as 256 do).
// PDHG_TR_COOP=0 turns it off.  Returns 0 (prepared), 1 (does not apply) or an error code.
static int tr_coop_prepare(pdhg_handle *h) {
  if (h->tr_coop >= 0) return h->tr_coop ? 0 : 1;
  h->tr_coop = 0;
  const char *ev = getenv("PDHG_TR_COOP");
  if (ev && ev[0] == '0') return 1;
  const int64_t total = h->n + h->m;
  const int64_t cap = dev_env("PDHG_TR_COOP_MAX") ? atoll(dev_env("PDHG_TR_COOP_MAX")) : 1000000;
  if (total > cap || total < 1) return 1;
  HIP_TRY(hipSetDevice(h->device));
  int grid = (int)std::min<int64_t>(TRC_MAX_WGS, (total + TPB * 4 - 1) / (TPB * 4));
  grid = std::max(8, (grid + 7) / 8 * 8);
  if (const char *g = dev_env("PDHG_TR_COOP_WGS")) grid = std::max(8, std::min(TRC_MAX_WGS, atoi(g) / 8 * 8));
  HIP_TRY(hipMalloc((void **)&h->tr_sync, sizeof(GridSync)));
  HIP_TRY(hipMemsetAsync(h->tr_sync, 0, sizeof(GridSync), h->stream));
  HIP_TRY(hipMalloc((void **)&h->tr_partials, sizeof(double) * 2 * EV_MAXQ * (size_t)grid));
  HIP_TRY(hipMemsetAsync(h->tr_partials, 0, sizeof(double) * 2 * EV_MAXQ * (size_t)grid, h->stream));
  hipLaunchKernelGGL(xcd_register_kernel, dim3(grid), dim3(TPB), 0, h->stream, h->tr_sync);
  HIP_TRY(hipGetLastError());
  GridSync host;
  HIP_TRY(hipMemcpyAsync(&host, h->tr_sync, sizeof(GridSync), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned long long seen = 0;
  h->tr_nxcd = 0;
  for (int x = 0; x < 8; ++x) { seen += host.xcd_count[x][0]; h->tr_nxcd += host.xcd_count[x][0] > 0; h->tr_xcd_cnt[x] = (unsigned)host.xcd_count[x][0]; }
  if (seen != (unsigned long long)grid || h->tr_nxcd == 0) return 1;     // no census: the multi-launch form
  // test knob: a census that expects one workgroup too many -- the first barrier cannot complete (spin limit, error word)
  if (dev_env("PDHG_TR_COOP_TEST_BAD_CENSUS")) h->tr_xcd_cnt[0] += 1;
  h->tr_grid = grid;
  h->tr_epoch = 0;
  h->tr_coop = 1;
  if (getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] trust-region search: one persistent launch of %d workgroups on %u XCDs per call\n", grid, h->tr_nxcd);
  return 0;
}

// one call; returns 0 with out[] filled, 1 when a barrier could not complete (the caller repeats the call launch by
// launch, and this handle stays with that form), or an error code
static int tr_coop_call(pdhg_handle *h, double wp, double wd, double radius, int range, int approximate, double out[8]) {
  int rc = ev_ensure_host(h);
  if (rc) return rc;
  TrCoopArgs a{};
  a.n = (int)h->n; a.m = (int)h->m; a.ne = (int)h->num_eq; a.range = range; a.approximate = approximate ? 1 : 0;
  a.px = h->pt_x; a.py = h->pt_y; a.aty = h->pt_aty; a.qx = h->pt_qx; a.ax = h->pt_ax;
  a.c = h->c; a.b = h->b; a.lb = h->lb; a.ub = h->ub;
  a.wp = wp; a.wd = wd; a.radius = radius;
  a.gdv = h->tr_g; a.wd2v = h->tr_dir; a.thr = h->tr_thr;
  a.partials = h->tr_partials;
  a.sync = h->tr_sync;
  a.epoch = h->tr_epoch;
  a.nxcd = h->tr_nxcd;
  for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->tr_xcd_cnt[x];
  a.host_out = h->ev_host;
  a.seq = ++h->ev_seq;
  double r[10];
  {
    // one partly resident persistent kernel at a time per device (as the trial kernels): from launch to results
    std::lock_guard<std::mutex> lock(coop_device_mutex(h->device));
    hipLaunchKernelGGL(tr_coop_kernel, dim3(h->tr_grid), dim3(TPB), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    if ((rc = ev_wait_host(h, 10, a.seq, r))) return rc;
  }
  h->tr_epoch = (unsigned long long)r[9];
  if (r[8] != 0.0) {
    h->tr_coop = 0;
    if (getenv("PDHG_VERBOSE")) fprintf(stderr, "[pdhg_hip] trust-region search: a grid barrier timed out (code %g); back to one launch per pass\n", r[8]);
    return 1;
  }
  for (int q = 0; q < 8; ++q) out[q] = r[q];
  h->tr_coop_calls += 1;
  return 0;
}

int pdhg_trust_region_bound(pdhg_handle *h0, int point, double primal_weight_norm, double dual_weight_norm,
                            double radius, int range, int approximate, double out[8]) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (range < 0 || range > 2) return fail(-1, "range must be 0, 1 or 2");
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  const double wp = primal_weight_norm, wd = dual_weight_norm;
  if ((rc = point_products(L, point))) return rc;
  {
    // small problems on one handle: set-up, search and results in ONE workgroup and one launch (tr_small_kernel)
    const char *se = dev_env("PDHG_SMALL_EVAL");
    pdhg_handle *h = L.p[0];
    if (!L.g && h->n + h->m <= TRS_MAX && !(se && se[0] == '0') && !h->profile) {
      HIP_TRY(hipSetDevice(h->device));
      if ((rc = ev_ensure_host(h))) return rc;
      const size_t lds = sizeof(double) * 3 * (size_t)(h->n + h->m);
      {
        static size_t limit[64] = {};
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        size_t &cur = limit[h->device & 63];
        if (cur < lds) {
          HIP_TRY(hipFuncSetAttribute((const void *)tr_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          cur = lds;
        }
      }
      TrSmallArgs a{};
      a.n = (int)h->n; a.m = (int)h->m; a.ne = (int)h->num_eq; a.range = range; a.approximate = approximate ? 1 : 0;
      a.px = h->pt_x; a.py = h->pt_y; a.aty = h->pt_aty; a.qx = h->pt_qx; a.ax = h->pt_ax;
      a.c = h->c; a.b = h->b; a.lb = h->lb; a.ub = h->ub;
      a.wp = wp; a.wd = wd; a.radius = radius;
      a.host_out = h->ev_host;
      a.seq = ++h->ev_seq;
      hipLaunchKernelGGL(tr_small_kernel, dim3(1), dim3(TRS_TPB), lds, h->stream, a);
      HIP_TRY(hipGetLastError());
      return ev_wait_host(h, 8, a.seq, out);
    }
  }
  // every shard works on the concatenation [its column slice ; its rows]
  FOR_SHARDS(L, h) {
    if (!h->tr_g) {
      const int64_t total = h->n + h->m;
      if ((rc = alloc_zero(&h->tr_g, total))) return rc;
      if ((rc = alloc_zero(&h->tr_dir, total))) return rc;
      if ((rc = alloc_zero(&h->tr_thr, total))) return rc;
    }
  }
  if (!L.g && !L.p[0]->profile) {
    // medium problems on one handle: set-up, every probe pass and the results in ONE persistent launch (tr_coop_kernel.hpp)
    pdhg_handle *h = L.p[0];
    rc = tr_coop_prepare(h);
    if (rc > 1 || rc < 0) return rc;
    if (rc == 0) {
      rc = tr_coop_call(h, wp, wd, radius, range, approximate, out);
      if (rc != 1) return rc;
    }
  }
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(tr_setup_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)h->m,
                       (int)h->num_eq, h->pt_x + o, h->pt_y, h->pt_aty + o, h->pt_qx ? h->pt_qx + o : nullptr, h->pt_ax,
                       h->c + o, h->b, h->lb + o, h->ub + o, wp, wd, range,
                       h->tr_g, h->tr_dir, h->tr_thr, h->ev_partials, h->ev_grid);   // tr_g: g d, tr_dir: w d^2
    HIP_TRY(hipGetLastError());
  }
  double r[EV_MAXQ];
  if ((rc = ev_finish(L, TR_SETUP_NS, 1, r))) return rc;
  // compute_lagrangian_value (saddle_point.jl:1109-1120) without objective_constant
  out[0] = 0.5 * r[10] + r[0] - r[1] + r[2];
  out[1] = out[2] = 0.0;
  out[3] = r[8]; out[4] = r[9];
  out[5] = 0.0; out[6] = 0.0; out[7] = 0.0;
  const double hinf = r[3], g2 = r[4], wd2_all = r[5], tmax = r[TR_SETUP_NS];
  const double r2 = radius * radius;
  if (approximate) {
    // approximately_solve_bound_constrained_trust_region (trust_region_utils.jl:194-224)
    const double dn = sqrt(wd2_all);
    const double sc = dn > 0.0 ? radius / dn : 1.0;
    out[1] = sc * r[6]; out[2] = sc * r[7];
    return 0;
  }
  if (radius == 0.0 || g2 == 0.0) return 0;   // trust_region_utils.jl:81-83
  // Find t* with radius^2(t*) = r2, radius^2(t) = low(t) + t^2 high(t).  The
  // reference eliminates breakpoints by repeated medians (trust_region_utils.jl:112-165);
  // here: TR_K-ary search over the IEEE bit patterns of t in [0, max finite
  // breakpoint] until no breakpoint lies strictly inside the bracket, then the
  // same closed form (trust_region_utils.jl:167-175).  Every probe carries the value sums of its t
  // (tr_probe_kernel), and the set-up pass those of t = tmax, so t* needs no pass of its own:
  //   value(t*) = vlow + t* vhigh  at the bracket's lower end (no breakpoint lies in between).
  auto probe = [&](const TrProbes &pr, double *sums) -> int {
    FOR_SHARDS(L, h) {
      hipLaunchKernelGGL(tr_probe_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)(h->cn + h->m),
                         h->tr_thr, h->tr_dir, h->tr_g, pr, h->ev_partials, h->ev_grid);
      HIP_TRY(hipGetLastError());
    }
    return ev_finish(L, TR_Q * TR_K, 0, sums);
  };
  double lh[TR_Q * TR_K];
  TrProbes pr;
  TrSearch S;                                // the search itself: eval_kernels.hpp (shared with the one-workgroup kernel)
  tr_search_begin(S, r2, tmax, hinf, TrEnd{r[11], hinf, {r[12], r[14], r[13], r[15]}});
  while (tr_search_next(S, pr)) {
    if ((rc = probe(pr, lh))) return rc;
    tr_search_feed(S, pr, lh);
  }
  out[1] = S.at.v[0] + S.tstar * S.at.v[1];
  out[2] = S.at.v[2] + S.tstar * S.at.v[3];
  out[5] = S.tstar; out[6] = (double)S.passes;     // probe passes (the set-up pass evaluates t = tmax itself)
  return 0;
}

int pdhg_trust_region_bounds(pdhg_handle *h0, int count, const int *points, double primal_weight_norm, double dual_weight_norm,
                             const double *radii, const int *ranges, int approximate, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (count < 1 || count > TRB_MAX || !points || !radii || !ranges || !out) return fail(-1, "count must be 1..3 and the arrays non-null");
  for (int p = 0; p < count; ++p) if (ranges[p] < 0 || ranges[p] > 2) return fail(-1, "range must be 0, 1 or 2");
  const Shards L = shards_of(h0);
  pdhg_handle *h = L.p[0];
  const char *se = dev_env("PDHG_SMALL_EVAL");
  const bool small = !L.g && h->n + h->m <= TRS_MAX && !(se && se[0] == '0');
  const char *be = dev_env("PDHG_TR_BATCH");
  bool batch = count > 1 && !L.g && !h->profile && !small && !(be && be[0] == '0');
  if (batch) {
    if ((rc = flush_pending(L))) return rc;
    rc = tr_coop_prepare(h);
    if (rc > 1 || rc < 0) return rc;
    batch = rc == 0;
  }
  if (batch) {
    if ((rc = ev_ensure_host(h))) return rc;
    const int64_t total = h->n + h->m;
    if (!h->trb_scratch) {
      if ((rc = alloc_zero(&h->trb_scratch, 3 * (int64_t)TRB_MAX * total))) return rc;
      if ((rc = alloc_zero(&h->trb_partials, 2 * (int64_t)TRB_MAX * EV_MAXQ * h->tr_grid))) return rc;
    }
    TrBatchArgs a{};
    a.n = (int)h->n; a.m = (int)h->m; a.ne = (int)h->num_eq; a.approximate = approximate ? 1 : 0; a.count = count;
    a.c = h->c; a.b = h->b; a.lb = h->lb; a.ub = h->ub;
    a.wp = primal_weight_norm; a.wd = dual_weight_norm;
    for (int p = 0; p < count; ++p) {
      if ((rc = point_products(L, points[p]))) return rc;       // (cached per point: nothing is recomputed for a point seen before)
      TrBatchProblem &q = a.pb[p];
      q.range = ranges[p]; q.radius = radii[p];
      q.px = h->pt_x; q.py = h->pt_y; q.aty = h->pt_aty; q.qx = h->pt_qx; q.ax = h->pt_ax;
      q.gdv = h->trb_scratch + (3 * (int64_t)p + 0) * total;
      q.wd2v = h->trb_scratch + (3 * (int64_t)p + 1) * total;
      q.thr = h->trb_scratch + (3 * (int64_t)p + 2) * total;
    }
    a.partials = h->trb_partials;
    a.sync = h->tr_sync;
    a.epoch = h->tr_epoch;
    a.nxcd = h->tr_nxcd;
    for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->tr_xcd_cnt[x];
    a.host_out = h->ev_host;
    a.seq = ++h->ev_seq;
    double r[8 * TRB_MAX + 2];
    {
      std::lock_guard<std::mutex> lock(coop_device_mutex(h->device));
      hipLaunchKernelGGL(tr_coop_batch_kernel, dim3(h->tr_grid), dim3(TPB), 0, h->stream, a);
      HIP_TRY(hipGetLastError());
      if ((rc = ev_wait_host(h, 8 * count + 2, a.seq, r))) return rc;
    }
    h->tr_epoch = (unsigned long long)r[8 * count + 1];
    if (r[8 * count] == 0.0) {
      for (int q = 0; q < 8 * count; ++q) out[q] = r[q];
      h->trb_calls += 1;
      h->tr_coop_calls += count;
      return 0;
    }
    h->tr_coop = 0;             // a barrier timed out: this handle goes back to one launch per pass, starting with these problems
    if (getenv("PDHG_VERBOSE")) fprintf(stderr, "[pdhg_hip] trust-region batch: a grid barrier timed out (code %g); back to one launch per pass\n", r[8 * count]);
  }
  for (int p = 0; p < count; ++p)
    if ((rc = pdhg_trust_region_bound(h0, points[p], primal_weight_norm, dual_weight_norm, radii[p], ranges[p], approximate, out + 8 * p))) return rc;
  return 0;
}

// ---- rescaling on the device (N2) --------------------------------------------

static int row_grid(int rows) { return std::max(1, (rows + (TPB / WAVE) - 1) / (TPB / WAVE)); }

// scratch vectors of one pdhg_rescale call, per shard: row factors have the
// shard's m entries, column factors all n (n_alloc: they are reduced over ranks)
struct RescaleTmp {
  double *ev = nullptr, *dv = nullptr, *inv_e = nullptr, *inv_d = nullptr, *cum_e = nullptr, *cum_d = nullptr;
  double *tmp_e = nullptr, *tmp_d = nullptr;
};

// a row statistic of one resident CSR: short rows one wave each, long rows chunk by chunk
extern "C++" {
template <int OP>
static void launch_row_op(pdhg_handle *h, const CsrDev &D, int cols, double pexp, const double *inv_scale, double *out) {
  if (!D.segs.empty()) {        // row segments (layout.hpp): the statistic is per row, segment by segment
    for (const CsrDev &S : D.segs) launch_row_op<OP>(h, S, cols, pexp, inv_scale ? inv_scale + S.row0 : inv_scale, out + S.row0);
    return;
  }
  hipLaunchKernelGGL(row_op_kernel<OP>, dim3(row_grid(D.rows)), dim3(TPB), 0, h->stream, D.view(), cols, pexp, inv_scale, out,
                     D.long_thr);
  if (D.nlong > 0) {
    hipLaunchKernelGGL(row_op_long_partial_kernel<OP>, dim3(D.nchunks), dim3(TPB), 0, h->stream, D.view(),
                       (const int *)D.chunk_row, (const int *)D.chunk_off, pexp, inv_scale, D.chunk_partial);
    hipLaunchKernelGGL(row_op_long_final_kernel<OP>, dim3(D.long_grid), dim3(TPB), 0, h->stream, D.view(),
                       (const int *)D.long_row, (const int *)D.long_chunk_ptr, D.nlong,
                       (const double *)D.chunk_partial, cols, pexp, out);
  }
}
}  // extern "C++"

// one scale_problem step on every resident layout + the vectors of one shard
static int apply_scaling(pdhg_handle *h, RescaleTmp &t) {
  const int n = (int)h->n, m = (int)h->m;
  hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev, t.inv_e, 0);
  hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv, t.inv_d, 0);
  // every resident copy of one matrix (k = 0: CSR(A), rows -> E; k = 1: CSR(A'), rows -> D); a row segment takes the
  // row-indexed factor from its first row on
  std::function<void(CsrDev &, const double *, const double *, int)> scale_one =
      [&](CsrDev &D, const double *inv_e, const double *inv_d, int k) {
    for (CsrDev &S : D.segs) scale_one(S, k == 0 ? inv_e + S.row0 : inv_e, k == 1 ? inv_d + S.row0 : inv_d, k);
    if (!D.segs.empty() || D.nnz == 0) return;
    hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(D.rows)), dim3(TPB), 0, h->stream, D.rows, D.rowptr,
                       D.col, D.val, inv_e, inv_d, k, D.long_thr);
    if (D.nlong > 0)
      hipLaunchKernelGGL(scale_long_kernel, dim3(D.nchunks), dim3(TPB), 0, h->stream, (const int *)D.rowptr,
                         (const int *)D.col, D.val, (const int *)D.chunk_row, (const int *)D.chunk_off,
                         inv_e, inv_d, k);
    if (D.tiled && D.nwaves > 0)
      hipLaunchKernelGGL(scale_tiled_kernel, dim3(row_grid(D.nwaves)), dim3(TPB), 0, h->stream, D.wave_rows,
                         D.wave_ent, D.wave_step_off, D.step_tile, D.wg_step_off, D.nwaves, D.tile_shift,
                         D.pk, D.tv, inv_e, inv_d, k);
    for (const SlabDev &S : D.slabs)
      if (S.nnz > 0)
        hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(D.rows)), dim3(TPB), 0, h->stream, D.rows, S.rowptr,
                           S.col, S.val, inv_e, inv_d, k, 0);
    // the sliced jagged copies (sj_kernels.hpp) hold the same entries in another order: copied again from the CSR arrays
    // just scaled, so that they carry the same bits (two multiplications in a fixed order per entry, done once)
    auto refill = [&](const SjDev &J, const int *rowptr, const int *col, const double *val) {
      if (J.on() && J.nnz > 0)
        hipLaunchKernelGGL(sj_fill_kernel, dim3((J.nslices + TPB / WAVE - 1) / (TPB / WAVE)), dim3(TPB), 0, h->stream, J.nslices,
                           (const unsigned *)J.meta, (const int *)J.slice_off, rowptr, col, val, J.col, J.val);
    };
    refill(D.sj, D.rowptr, D.col, D.val);
    for (const SlabDev &S : D.slabs) refill(S.sj, S.rowptr, S.col, S.val);
  };
  scale_one(h->A, t.inv_e, t.inv_d, 0);
  scale_one(h->At, t.inv_e, t.inv_d, 1);
  if (h->has_q) {
    // objective_matrix = (D^-1 Q) D^-1 (preprocess.jl:562-564); Qt holds Q' entry by entry, so the
    // "transposed" order reproduces the same two roundings on it
    hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(h->Q.rows)), dim3(TPB), 0, h->stream, h->Q.rows, h->Q.rowptr,
                       h->Q.col, h->Q.val, t.inv_d, t.inv_d, 0, 0);
    hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(h->Qt.rows)), dim3(TPB), 0, h->stream, h->Qt.rows, h->Qt.rowptr,
                       h->Qt.col, h->Qt.val, t.inv_d, t.inv_d, 1, 0);
    for (const SlabDev &S : h->Q.slabs)
      if (S.nnz > 0)
        hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(h->Q.rows)), dim3(TPB), 0, h->stream, h->Q.rows, S.rowptr,
                           S.col, S.val, t.inv_d, t.inv_d, 0, 0);
    for (const SlabDev &S : h->Qt.slabs)
      if (S.nnz > 0)
        hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(h->Qt.rows)), dim3(TPB), 0, h->stream, h->Qt.rows, S.rowptr,
                           S.col, S.val, t.inv_d, t.inv_d, 1, 0);
  }
  hipLaunchKernelGGL(resc_apply_vectors_kernel, dim3(h->ew_grid_nm), dim3(TPB), 0, h->stream, n, m, t.dv, t.ev,
                     h->c, h->lb, h->ub, h->b, t.cum_d, t.cum_e);
  HIP_TRY(hipGetLastError());
  return 0;
}

int pdhg_rescale(pdhg_handle *h0, int l_inf_ruiz_iterations, int l2_norm_rescaling,
                 int use_pock_chambolle, double pock_chambolle_alpha,
                 double *constraint_rescaling_out, double *variable_rescaling_out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (use_pock_chambolle && !(pock_chambolle_alpha >= 0.0 && pock_chambolle_alpha <= 2.0))
    return fail(-1, "pock_chambolle_alpha must be in [0, 2]");
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  for (int i = 0; i < L.count; ++i) L.p[i]->matrix_version += 1;
  std::vector<RescaleTmp> T((size_t)L.count);
  auto cleanup = [&]() {
    for (int i = 0; i < L.count; ++i) {
      (void)hipSetDevice(L.p[i]->device);
      RescaleTmp &t = T[(size_t)i];
      for (double *p : {t.ev, t.dv, t.inv_e, t.inv_d, t.cum_e, t.cum_d, t.tmp_e, t.tmp_d}) if (p) (void)hipFree(p);
    }
  };
#define RS(expr) do { int _r = (expr); if (_r) { cleanup(); return _r; } } while (0)
#define EACH(h, t) for (int _i = 0; _i < L.count; ++_i) if (pdhg_handle *h = L.p[_i]) \
    if (hipError_t _sde = hipSetDevice(h->device); _sde != hipSuccess) { cleanup(); return fail_hip(_sde, "hipSetDevice (rescale)"); } \
    else if (RescaleTmp *_tp = &T[(size_t)_i]) if (RescaleTmp &t = *_tp; true)
  EACH(h, t) {
    RS(alloc_zero(&t.ev, h->m)); RS(alloc_zero(&t.dv, h->n_alloc)); RS(alloc_zero(&t.inv_e, h->m)); RS(alloc_zero(&t.inv_d, h->n));
    RS(alloc_zero(&t.cum_e, h->m)); RS(alloc_zero(&t.cum_d, h->n)); RS(alloc_zero(&t.tmp_e, h->m)); RS(alloc_zero(&t.tmp_d, h->n_alloc));
    hipLaunchKernelGGL(fill_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, 1.0, t.cum_e);
    hipLaunchKernelGGL(fill_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, 1.0, t.cum_d);
  }
  // Column statistics of A are reductions over the row shards: every shard reduces its
  // rows, then max / sum over ranks (reduce-scatter + all-gather: the same bits everywhere).
  // Row statistics are complete on the shard that owns the row.
  auto reduce_cols = [&](bool use_tmp, bool maxop) -> int {
    if (!L.g) return 0;
    std::vector<double *> ptr((size_t)L.g->world, nullptr);
    for (int i = 0; i < L.count; ++i) ptr[(size_t)L.p[i]->rank] = use_tmp ? T[(size_t)i].tmp_d : T[(size_t)i].dv;
    return dist_all_reduce(*L.g, [&](pdhg_handle *s) { return ptr[(size_t)s->rank]; }, L.g->S, maxop);
  };
  // ruiz_rescaling, p = Inf (preprocess.jl:412-477): sqrt of the row / column max |a|, zeros -> 1
  for (int it = 0; it < l_inf_ruiz_iterations; ++it) {
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      launch_row_op<ROP_MAXABS>(h, h->At, m, 0.0, (const double *)nullptr, t.dv);
      launch_row_op<ROP_MAXABS>(h, h->A, n, 0.0, (const double *)nullptr, t.ev);
    }
    RS(reduce_cols(false, true));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      if (h->has_q) {   // QP: column max over the constraint AND the objective matrix (preprocess.jl:425-433)
        launch_row_op<ROP_MAXABS>(h, h->Qt, n, 0.0, (const double *)nullptr, t.tmp_d);
        hipLaunchKernelGGL(resc_max_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv, t.tmp_d);
      }
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv);
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev);
      RS(apply_scaling(h, t));
    }
  }
  // l2_norm_rescaling (preprocess.jl:358-372): sqrt of the row / column L2 norms, zeros -> 1
  if (l2_norm_rescaling) {
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      launch_row_op<ROP_MAXABS>(h, h->At, m, 0.0, (const double *)nullptr, t.tmp_d);
      launch_row_op<ROP_MAXABS>(h, h->A, n, 0.0, (const double *)nullptr, t.tmp_e);
    }
    RS(reduce_cols(true, true));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.tmp_d, t.inv_d, 1);
      hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.tmp_e, t.inv_e, 1);
      launch_row_op<ROP_SUMSQ_SCALED>(h, h->At, m, 0.0, t.inv_d, t.dv);
      launch_row_op<ROP_SUMSQ_SCALED>(h, h->A, n, 0.0, t.inv_e, t.ev);
    }
    RS(reduce_cols(false, false));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      hipLaunchKernelGGL(resc_l2norm_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.tmp_d, t.dv);
      hipLaunchKernelGGL(resc_l2norm_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.tmp_e, t.ev);
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv);   // norm 0 -> sqrt 0 -> 1
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev);
      RS(apply_scaling(h, t));
    }
  }
  // pock_chambolle_rescaling (preprocess.jl:508-539)
  if (use_pock_chambolle) {
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      launch_row_op<ROP_SUMPOW>(h, h->At, m, 2.0 - pock_chambolle_alpha, (const double *)nullptr, t.dv);
      launch_row_op<ROP_SUMPOW>(h, h->A, n, pock_chambolle_alpha, (const double *)nullptr, t.ev);
    }
    RS(reduce_cols(false, false));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv);
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev);
      RS(apply_scaling(h, t));
    }
  }
  EACH(h, t) {
    (void)t;
    hipError_t e1 = hipGetLastError();
    if (e1 != hipSuccess) { cleanup(); return fail((int)e1, hipGetErrorString(e1)); }
  }
  if (constraint_rescaling_out && h0->m_global > 0) {
    std::vector<double *> ptr((size_t)h0->world, nullptr);
    for (int i = 0; i < L.count; ++i) ptr[(size_t)L.p[i]->rank] = T[(size_t)i].cum_e;
    RS(rows_to_host(L, [&](pdhg_handle *s) { return ptr[(size_t)s->rank]; }, constraint_rescaling_out));
  }
  if (variable_rescaling_out && h0->n > 0) {
    (void)hipSetDevice(h0->device);
    (void)hipMemcpyAsync(variable_rescaling_out, T[0].cum_d, sizeof(double) * (size_t)h0->n, hipMemcpyDeviceToHost, h0->stream);
  }
  rc = sync_all(L);
  cleanup();
#undef RS
#undef EACH
  return rc;
}

int pdhg_get_problem_vectors(pdhg_handle *h0, double *c, double *b, double *lb, double *ub) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  pdhg_handle *h = h0;   // column vectors of the problem are stored in full on every shard
  if (c) HIP_TRY(hipMemcpyAsync(c, h->c, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (lb) HIP_TRY(hipMemcpyAsync(lb, h->lb, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (ub) HIP_TRY(hipMemcpyAsync(ub, h->ub, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (b && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->b; }, b))) return rc;
  return sync_all(L);
}

int pdhg_matrix_max_abs(pdhg_handle *h0, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  FOR_SHARDS(L, h) {
    if ((rc = ev_alloc(h))) return rc;
    if (!h->At.segs.empty()) {            // row segments: the max over the segments' maxima (single handle: L is this one)
      double best = 0.0;
      for (const CsrDev &S : h->At.segs) {
        hipLaunchKernelGGL(maxabs_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int64_t)S.nnz, S.val, h->ev_partials, h->ev_grid);
        HIP_TRY(hipGetLastError());
        double part = 0.0;
        if ((rc = ev_finish(L, 0, 1, &part))) return rc;
        best = std::max(best, part);
      }
      *out = best;
      return 0;
    }
    hipLaunchKernelGGL(maxabs_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int64_t)h->At.nnz, h->At.val,
                       h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  return ev_finish(L, 0, 1, out);
}

// ---- measurement ------------------------------------------------------------

int pdhg_profile_enable(pdhg_handle *h, int enable) {
  // (through check_handle: switching the profile flag moves a group between the persistent launches and the per-launch path,
  //  so the members' streams must first wait for the last persistent launch)
  int rc0 = check_handle(h);
  if (rc0) return rc0;
  h->profile = enable != 0;     // a group is profiled through its first local shard
  if (enable) for (int k = 0; k < PDHG_K_COUNT; ++k) { h->prof_count[k] = 0; h->prof_ms[k] = 0.0; }
  return 0;
}

int pdhg_profile_read(pdhg_handle *h, int kernel_id, int64_t *launches, double *total_ms) {
  if (!h || kernel_id < 0 || kernel_id >= PDHG_K_COUNT) return fail(-1, "bad kernel id");
  *launches = h->prof_count[kernel_id];
  *total_ms = h->prof_ms[kernel_id];
  return 0;
}

int64_t pdhg_kernel_algorithmic_bytes(pdhg_handle *h, int kernel_id) {
  if (!h) return -1;
  // sizes of THIS shard: m rows, nnz nonzeros, cn owned columns of n
  const int64_t m = h->m, n = h->n, nnz = h->nnz, cn = h->cn;
  const bool group = h->grp != nullptr;
  switch (kernel_id) {
    // lazy accept: K7's sums are read and written where x and y are read anyway
    case PDHG_K_PRIMAL: return 8 * (7 + (h->lazy_accept ? 2 : 0)) * cn;         // r: x,c,aty,lb,ub  w: x',xbar  (+ r/w sum_x)
    case PDHG_K_SPMV_DUAL:                                                      // + r: y,b  w: y'  (+ r/w sum_y)
      return nnz * 12 + (m + 1) * 4 + n * 8 + (3 + (h->lazy_accept ? 2 : 0)) * m * 8;
    case PDHG_K_SPMV_ATY:                                                // fused: + r: x,x',aty  w: aty'
      return nnz * 12 + (n + 1) * 4 + m * 8 + (group ? 1 : 4) * n * 8;
    case PDHG_K_FINAL: return 8 * (int64_t)(3 * h->At.slots() + h->A.slots());
    case PDHG_K_ACCEPT: return 8 * 3 * (cn + m);
    case PDHG_K_ALLGATHER: return group ? 8 * (h->n_alloc - h->grp->S) : 0;        // bytes received per rank
    case PDHG_K_REDUCE_SCATTER: return group ? 8 * (h->n_alloc - h->grp->S) : 0;
    case PDHG_K_INTERACTION: return group ? 8 * 4 * cn : 0;
    default: return -1;
  }
}

namespace {
// a[i] = b[i] + s*c[i], 32 bytes per lane and pass (two 16-byte loads per stream in
// flight), one workgroup of 256 threads per 8 KiB of each stream: the access shape
// that reaches the chip's streaming rate (MI355X_MICROARCH.md: float4 copy 6.29 TB/s).
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(TPB) void triad_kernel(int64_t len4, const dbl2_t *__restrict__ b,
                                                    const dbl2_t *__restrict__ c, double s,
                                                    dbl2_t *__restrict__ a) {
  const int64_t stride = (int64_t)gridDim.x * TPB;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < len4; i += stride) {
    const int64_t k = 2 * i;
    const dbl2_t b0 = __builtin_nontemporal_load(b + k), b1 = __builtin_nontemporal_load(b + k + 1);
    const dbl2_t c0 = __builtin_nontemporal_load(c + k), c1 = __builtin_nontemporal_load(c + k + 1);
    __builtin_nontemporal_store(b0 + s * c0, a + k);
    __builtin_nontemporal_store(b1 + s * c1, a + k + 1);
  }
}
}  // namespace

int pdhg_measure_triad(pdhg_handle *h, int64_t len, int reps, double *gbps) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (len <= 0 || (len & 3) || reps <= 0 || !gbps) return fail(-1, "bad triad arguments (len must be a multiple of 4)");
  double *buf = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, sizeof(double) * 3 * (size_t)len));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float best = 1e30f;
  hipError_t err = hipMemsetAsync(buf, 0, sizeof(double) * 3 * (size_t)len, h->stream);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  const int64_t len4 = len / 4;
  const int64_t full = (len4 + TPB - 1) / TPB;                 // one pass per thread
  const int64_t grids[4] = {full, std::max<int64_t>(1, full / 2), 256 * 32, 256 * 64};
  for (int g = 0; g < 4 && err == hipSuccess; ++g) {
    const int grid = (int)std::min<int64_t>(grids[g], 1 << 30);
    for (int r = 0; r <= reps && err == hipSuccess; ++r) {   // pass 0 warms up
      (void)hipEventRecord(e0, h->stream);
      hipLaunchKernelGGL(triad_kernel, dim3(grid), dim3(TPB), 0, h->stream, len4,
                         reinterpret_cast<const dbl2_t *>(buf + len), reinterpret_cast<const dbl2_t *>(buf + 2 * len),
                         0.5, reinterpret_cast<dbl2_t *>(buf));
      (void)hipEventRecord(e1, h->stream);
      err = hipEventSynchronize(e1);
      float ms = 0.f;
      if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  HIP_TRY(err);
  *gbps = 24.0 * (double)(4 * len4) / ((double)best * 1e-3) / 1e9;
  return 0;
}

namespace {
__global__ void noop_kernel(int *sink) { if (sink && threadIdx.x == 1024) *sink = 0; }
}  // namespace

extern "C++" {
namespace {
// ---- what the tiled sweep's ACCESS PATTERN can reach on this chip, with nothing else in the kernel ------------------
// The geometry of spmv_tiled_kernel -- 8-wave workgroups, two per CU (the dynamic LDS of the product is reserved, unused),
// every wave walking the same column tiles in lock step with one pacing barrier per tile, the entries of a (wave, tile)
// cell streamed as 4-byte packed offsets + 8-byte values with non-temporal loads one tile ahead, one 8-byte gather per
// entry from the tile's window of the vector -- but no accumulators, no row logic, no epilogue: the products are added
// into a register.  Its time for the same number of gathers is the floor of this design on this matrix shape; bench.py
// reports the product kernel's time against it (roofline.ceiling_frac) next to the 8 TB/s figure.  FLAT: no tiles, no
// barrier, every gather of every wave falls into ONE window of tile_cols columns -- the same wave geometry with a perfect
// cache (what tile switches and pacing cost), not the chip's all-hit rate at full occupancy.
template <bool FLAT>
__global__ __launch_bounds__(TW_WPB * WAVE) void sweep_ceiling_kernel(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                                      const double *__restrict__ x, double *__restrict__ out,
                                                                      int nwaves, int ntiles, int tile_cols, int cnt) {
  extern __shared__ double ceiling_lds[];
  constexpr int C = 3;                                   // 64-entry chunks per cell held in registers (TW_U)
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int w = blockIdx.x * TW_WPB + wid;
  const bool live = w < nwaves;
  const size_t cell = (size_t)C * WAVE;
  const unsigned *my = pk + (size_t)(live ? w : 0) * ntiles * cell;
  const double *myv = tv + (size_t)(live ? w : 0) * ntiles * cell;
  double s = 0.0;
  unsigned p[2][C];
  double v[2][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const bool ok = live && c * WAVE + lane < cnt;
    p[0][c] = ok ? __builtin_nontemporal_load(my + c * WAVE + lane) : 0u;
    v[0][c] = ok ? __builtin_nontemporal_load(myv + c * WAVE + lane) : 0.0;
  }
  for (int t0 = 0; t0 < ntiles; t0 += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int t = t0 + b;
      if (t < ntiles) {                                  // workgroup-uniform
        const double *xt = FLAT ? x : x + (size_t)t * tile_cols;
        double g[C];
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] = (live && c * WAVE + lane < cnt) ? xt[p[b][c]] : 0.0;
        if (t + 1 < ntiles) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const bool ok = live && c * WAVE + lane < cnt;
            p[b ^ 1][c] = ok ? __builtin_nontemporal_load(my + (size_t)(t + 1) * cell + c * WAVE + lane) : 0u;
            v[b ^ 1][c] = ok ? __builtin_nontemporal_load(myv + (size_t)(t + 1) * cell + c * WAVE + lane) : 0.0;
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) s = s + v[b][c] * g[c];
        if (!FLAT) __syncthreads();                      // the sweep's pacing barrier
      }
    }
  }
  if (s == 0.123456789) out[0] = s + ceiling_lds[0];     // keeps the sum (and the LDS reservation) alive
}
__global__ __launch_bounds__(TPB) void ceiling_fill_kernel(unsigned *pk, double *tv, size_t len, unsigned tile_cols) {
  const size_t stride = (size_t)gridDim.x * TPB;
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < len; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    pk[i] = (unsigned)(z % tile_cols);
    tv[i] = 1.0 + (double)(z >> 40) * 1e-9;
  }
}
}  // namespace
}  // extern "C++"

/* out[0]: G gathers/s of the sweep's pattern (tiles + pacing barriers + entry streams, no accumulation) for `rows` rows,
 * `cols` columns and `nnz` entries in this handle's geometry (its constraint matrix's sweep layout when it has one:
 * waves, tiles, tile width; otherwise 1221 rows per wave and the tile width the library would choose);
 * out[1]: milliseconds of one such pass; out[2]: G gathers/s when every gather falls into ONE window of the tile's width
 * and nothing synchronises (the chip's all-hit rate for 8-byte gathers beside the entry streams); out[3]: entries per
 * (wave, tile) cell; out[4] / out[5]: waves and tiles of the probe. */
int pdhg_measure_sweep_ceiling(pdhg_handle *h, int64_t rows, int64_t cols, int64_t nnz, int reps, double out[6]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (rows <= 0 || cols <= 0 || nnz <= 0 || reps <= 0 || !out) return fail(-1, "bad ceiling-probe arguments");
  HIP_TRY(hipSetDevice(h->device));
  auto fits = [&](const CsrDev &M) { return M.tiled && M.nwaves > 0 && M.ntiles > 0 && M.rows == rows && M.cols == cols; };
  const CsrDev &D = (!fits(h->A) && fits(h->At)) ? h->At : h->A;      // the sweep layout of the product with these extents
  const bool have = fits(D);
  const int tile_cols = have ? D.tile_cols : std::max(4096, choose_tile_cols(cols, nnz, rows) > 0 ? choose_tile_cols(cols, nnz, rows) : 65536);
  const int ntiles = have ? D.ntiles : (int)((cols + tile_cols - 1) / tile_cols);
  const int tw_rows = have ? D.tw_rows : 1221;
  const int nwaves = have ? D.nwaves : (int)((rows + tw_rows - 1) / tw_rows);
  const int cnt = (int)std::min<int64_t>(3 * WAVE, std::max<int64_t>(1, (nnz + (int64_t)nwaves * ntiles / 2) / ((int64_t)nwaves * ntiles)));
  const size_t len = (size_t)nwaves * ntiles * 3 * WAVE;
  if (len > ((size_t)1 << 32)) return fail(-2, "ceiling probe: geometry too large");
  unsigned *pk = nullptr;
  double *tv = nullptr, *x = nullptr, *o = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipMalloc((void **)&pk, sizeof(unsigned) * len);
  if (err == hipSuccess) err = hipMalloc((void **)&tv, sizeof(double) * len);
  if (err == hipSuccess) err = hipMalloc((void **)&x, sizeof(double) * ((size_t)ntiles * tile_cols + 16));
  if (err == hipSuccess) err = hipMalloc((void **)&o, 64);
  if (err == hipSuccess) err = hipMemsetAsync(x, 0, sizeof(double) * ((size_t)ntiles * tile_cols + 16), h->stream);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  double best[2] = {1e30, 1e30};
  if (err == hipSuccess) {
    hipLaunchKernelGGL(ceiling_fill_kernel, dim3(4096), dim3(TPB), 0, h->stream, pk, tv, len, (unsigned)tile_cols);
    const size_t lds = have ? tiled_lds_bytes(D) : (size_t)78 * 1024;     // two workgroups per CU, as the product runs
    err = hipFuncSetAttribute((const void *)sweep_ceiling_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err == hipSuccess) err = hipFuncSetAttribute((const void *)sweep_ceiling_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = (nwaves + TW_WPB - 1) / TW_WPB;
    for (int flat = 0; flat < 2 && err == hipSuccess; ++flat) {
      for (int r = 0; r <= reps && err == hipSuccess; ++r) {               // pass 0 warms up
        (void)hipEventRecord(e0, h->stream);
        if (flat) hipLaunchKernelGGL(sweep_ceiling_kernel<true>, dim3(grid), dim3(TW_WPB * WAVE), lds, h->stream, pk, tv, x, o, nwaves, ntiles, tile_cols, cnt);
        else hipLaunchKernelGGL(sweep_ceiling_kernel<false>, dim3(grid), dim3(TW_WPB * WAVE), lds, h->stream, pk, tv, x, o, nwaves, ntiles, tile_cols, cnt);
        (void)hipEventRecord(e1, h->stream);
        err = hipEventSynchronize(e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best[flat]) best[flat] = ms;
      }
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  for (void *q : {(void *)pk, (void *)tv, (void *)x, (void *)o}) if (q) (void)hipFree(q);
  HIP_TRY(err);
  const double gathers = (double)nwaves * ntiles * cnt;
  out[0] = gathers / (best[0] * 1e-3) / 1e9;
  out[1] = best[0];
  out[2] = gathers / (best[1] * 1e-3) / 1e9;
  out[3] = (double)cnt;
  out[4] = (double)nwaves;
  out[5] = (double)ntiles;
  return 0;
}

/* Phase timeline of the last one-launch trial (needs PDHG_COOP_TRACE=1 in the environment when the handle takes its
 * first one-launch trial): see trial_timeline.  Returns 1 when no trace was recorded.  Measurement only. */
int pdhg_trial_timeline(pdhg_handle *h, double out[14]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return trial_timeline(h, out) ? fail(1, "no one-launch trial has been traced (PDHG_COOP_TRACE=1, stream-layout LP on one handle)") : 0;
}

int pdhg_measure_launch_overhead(pdhg_handle *h, int reps, double out[2]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (reps <= 0 || !out) return fail(-1, "bad arguments");
  HIP_TRY(hipStreamSynchronize(h->stream));
  double best[2] = {1e30, 1e30};
  for (int k = 1; k <= 2; ++k)
    for (int r = 0; r <= reps; ++r) {          // pass 0 warms up
      HIP_TRY(hipEventRecord(h->ev0, h->stream));
      for (int q = 0; q < (k == 1 ? 1 : 5); ++q) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, h->stream, (int *)nullptr);
      HIP_TRY(hipEventRecord(h->ev1, h->stream));
      HIP_TRY(hipEventSynchronize(h->ev1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
      if (r > 0 && ms < best[k - 1]) best[k - 1] = ms;
    }
  out[0] = best[0];                              // one empty launch between two events
  out[1] = (best[1] - best[0]) / 4.0;            // every further launch inside the same bracket
  return 0;
}

int pdhg_layout_checksums(pdhg_handle *h, uint64_t out[32]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  if (!h->A.segs.empty() || !h->At.segs.empty()) return fail(-2, "layout checksums are per piece: not defined for a matrix held as row segments");
  HIP_TRY(hipStreamSynchronize(h->stream));
  const CsrDev *Ls[2] = {&h->A, &h->At};
  for (int k = 0; k < 2; ++k) {
    const CsrDev &D = *Ls[k];
    const struct { const void *p; int64_t words; } parts[16] = {
        {D.rowptr, (int64_t)D.rows + 1}, {D.col, D.nnz}, {D.val, 2 * D.nnz}, {D.blks, 2 * (int64_t)D.nblk},
        {D.long_row, D.nlong}, {D.long_chunk_ptr, (int64_t)D.nlong + 1}, {D.chunk_row, D.nchunks}, {D.chunk_off, D.nchunks},
        {D.tiled ? D.pk : nullptr, D.tw_entries}, {D.tiled ? D.tv : nullptr, 2 * D.tw_entries},
        {D.tiled ? D.wave_rows : nullptr, 2 * (int64_t)D.nwaves}, {D.tiled ? D.wave_ent : nullptr, D.step_ptr_len},
        {D.tiled ? D.wave_step_off : nullptr, D.nwaves}, {D.tiled ? D.step_tile : nullptr, D.total_steps},
        {D.tiled ? D.wg_step_off : nullptr, D.tiled ? (int64_t)D.grid + 1 : 0}, {nullptr, 0}};
    for (int q = 0; q < 16; ++q) {
      unsigned long long v = 0;
      if ((rc = device_checksum(parts[q].p, parts[q].words, &v, h->stream))) return rc;
      out[16 * k + q] = v;
    }
    // a stream layout's column slabs ride in the sweep's (then unused) slots: row pointers, columns, values, row blocks
    if (!D.tiled) {
      for (size_t s = 0; s < D.slabs.size(); ++s) {
        const SlabDev &S = D.slabs[s];
        const struct { const void *p; int64_t words; } sp4[4] = {
            {S.rowptr, (int64_t)D.rows + 1}, {S.col, S.nnz}, {S.val, 2 * S.nnz}, {S.blks, 2 * (int64_t)S.nblk}};
        for (int q = 0; q < 4; ++q) {
          unsigned long long v = 0;
          if ((rc = device_checksum(sp4[q].p, sp4[q].words, &v, h->stream))) return rc;
          out[16 * k + 8 + q] += v * (2ull * s + 3ull);
        }
      }
    }
    // the plan's scalars ride in the last slot
    out[16 * k + 15] = (uint64_t)D.tiled + 2ull * (uint64_t)D.tw_mode + 8ull * (uint64_t)D.tile_shift + 1024ull * (uint64_t)D.tw_rows +
                       (1ull << 32) * (uint64_t)D.grid;
  }
  return 0;
}

int pdhg_layout_info(pdhg_handle *h, int64_t info[16]) {
  if (!h) return fail(-1, "null handle");
  info[12] = (int64_t)h->A.slabs.size(); info[13] = (int64_t)h->At.slabs.size();
  // 2: one persistent kernel per trial (trial_kernel.hpp), 1: one graph launch, 0: separate launches
  info[14] = (coop_eligible(h) || h->coop_mode == 1) ? 2 : ((graph_eligible(h) || (h->graph_mode == 1 && !h->has_q)) ? 1 : 0);
  info[15] = ((h->A.tiled && h->A.var_tiles) || (!h->A.segs.empty() && h->A.segs.front().tiled && h->A.segs.front().var_tiles) ? 1 : 0) +
             ((h->At.tiled && h->At.var_tiles) || (!h->At.segs.empty() && h->At.segs.front().tiled && h->At.segs.front().var_tiles) ? 2 : 0) +
             (small_lp_eligible(h) ? 4 : 0) +
             (!h->grp && !h->has_q && !small_lp_eligible(h) && device_loop_for(h) && coop_eligible(h) ? 8 : 0) +
             (h->local_mode == 1 && h->local_launches > 0 ? 16 : 0);      // the multi-step kernel runs in its XCD-local mode
  // a matrix held as row segments (64-bit extents, layout.hpp) reports the sums over its segments, the first segment's
  // tile width, and the segment counts in bits 8-15 (A) and 16-23 (A') of info[15]
  auto total = [](const CsrDev &D, auto f) { int64_t t = 0; if (D.segs.empty()) return (int64_t)f(D); for (const CsrDev &S : D.segs) t += f(S); return t; };
  auto first = [](const CsrDev &D) -> const CsrDev & { return D.segs.empty() ? D : D.segs.front(); };
  const CsrDev *Ms[2] = {&h->A, &h->At};
  for (int k = 0; k < 2; ++k) {
    const CsrDev &D = *Ms[k];
    info[4 * k + 0] = total(D, [](const CsrDev &S) { return S.nblk; });
    info[4 * k + 1] = total(D, [](const CsrDev &S) { return S.nlong; });
    info[4 * k + 2] = total(D, [](const CsrDev &S) { return S.nchunks; });
    info[4 * k + 3] = D.max_row_nnz;
    info[8 + k] = total(D, [](const CsrDev &S) { return S.tiled ? S.nwaves : 0; });
    info[10 + k] = first(D).tiled ? first(D).tile_cols : 0;
    info[15] += (int64_t)std::min<size_t>(D.segs.size(), 255) << (8 + 8 * k);
  }
  if (h->grp)     // bits 24-39: trials this group took as one persistent kernel per shard (group_kernel.hpp); 40-47: its fallbacks
    info[15] += (std::min<int64_t>(h->grp->coop_trials, 65535) << 24) + ((int64_t)std::min(h->grp->coop_fallbacks, 255) << 40);
  info[15] += std::min<int64_t>(h->tr_coop_calls, 16383) << 48;      // trust-region calls taken as one persistent launch
  if (!h->A.segs.empty()) info[12] = (int64_t)first(h->A).slabs.size();
  if (!h->At.segs.empty()) info[13] = (int64_t)first(h->At).slabs.size();
  // bit 8 of the slab counts: the product runs on the sliced jagged layout (sj_kernels.hpp)
  for (int k = 0; k < 2; ++k) {
    const CsrDev &D = first(*Ms[k]);
    if (D.sj.on() || (!D.slabs.empty() && D.slabs.front().sj.on())) info[12 + k] += 256;
    if (D.pipe_grid > 0 || (!D.slabs.empty() && D.slabs.front().pipe_grid > 0)) info[12 + k] += 512;   // bit 9: spmv_stream_pipe_kernel
  }
  return 0;
}

}  // extern "C"
