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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pdhg_hip.h"

// The kernels and layout builders live in the headers below; they form ONE
// translation unit with this file (the tiled kernel is sensitive to code
// placement, and one TU keeps every launch a direct call).
#include "common.hpp"
#include "spmv_kernels.hpp"
#include "vector_kernels.hpp"
#include "eval_kernels.hpp"
#include "rescale_kernels.hpp"
#include "layout.hpp"

struct pdhg_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int64_t m = 0, n = 0, nnz = 0, num_eq = 0;
  bool remap = true;

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
  double sum_x_weights = 0.0, sum_y_weights = 0.0;

  double *pA = nullptr;   // partials of the A kernel (1 quantity)
  double *pAt = nullptr;  // partials of the A' kernel / interaction kernel (3 quantities)
  double *pQ = nullptr;   // partials of the QP dot
  int pAt_stride = 0;
  int ew_grid_n = 1, ew_grid_m = 1, ew_grid_nm = 1;

  double *h_out = nullptr;  // pinned, device-visible, 8 doubles
  double *d_out = nullptr;

  // evaluation branch (N1), allocated on first use
  double *E = nullptr, *Dv = nullptr, *c_o = nullptr, *b_o = nullptr, *lb_o = nullptr, *ub_o = nullptr;
  double *x_r = nullptr, *y_r = nullptr;          // last restart point
  double *px_avg = nullptr, *py_avg = nullptr;    // materialised average
  double *ev_ax = nullptr, *ev_aty = nullptr;     // A*x (m), A'*y (n) at the evaluated point
  // The evaluation branch asks for the same products several times per check
  // (eval_point, then one or more trust-region bounds at the same point): keep
  // A*x and A'*y of the CURRENT and the AVERAGE point until the state changes.
  double *ev_cax[2] = {nullptr, nullptr}, *ev_caty[2] = {nullptr, nullptr};
  double *ev_cqx[2] = {nullptr, nullptr}, *ev_qx = nullptr;   // Q*x at those points (QP only)
  uint64_t state_version = 1;                      // bumped by everything that moves x, y, the sums or A
  uint64_t ev_cversion[2] = {0, 0}, avg_version = 0;
  double *tr_g = nullptr, *tr_dir = nullptr, *tr_thr = nullptr;  // n+m each
  double *ev_partials = nullptr, *ev_out = nullptr, *ev_host = nullptr;
  int ev_grid = 1;
  bool has_original = false;

  bool profile = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int64_t prof_count[PDHG_K_COUNT] = {0};
  double prof_ms[PDHG_K_COUNT] = {0};
  bool dist_pending = false;
  std::vector<int> dist_part_wg;   // workgroup boundaries of the parts handed out by pdhg_dist_parts
};

namespace {

int ew_grid(int64_t len) {
  int64_t g = (len + TPB - 1) / TPB;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, EW_MAX_BLOCKS));
}

struct ProfScope {
  pdhg_handle *h;
  int kid;
  ProfScope(pdhg_handle *h_, int kid_) : h(h_), kid(kid_) {
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
int ensure_lds_limit(pdhg_handle *h, int mode, bool scratch, size_t lds, const void *func) {
  static size_t limit[64][3][2] = {};
  size_t &cur = limit[h->device & 63][mode][scratch ? 1 : 0];
  if (cur < lds) {
    HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    cur = lds;
  }
  return 0;
}

template <int MODE>
int launch_spmv(pdhg_handle *h, const CsrDev &D, const double *xin, EpiArgs e) {
  if (D.tiled) {
    if (D.grid > 0) {
      const size_t lds = sizeof(double) * ((size_t)TW_WPB * D.tw_rows + 3 * TW_WPB + (D.tw_scratch ? TW_WPB * WAVE : 0));
      int rc;
      if (D.tw_scratch) {
        if ((rc = ensure_lds_limit(h, MODE, true, lds, (const void *)spmv_tiled_kernel<MODE, true>))) return rc;
        hipLaunchKernelGGL((spmv_tiled_kernel<MODE, true>), dim3(D.grid), dim3(TW_WPB * WAVE), lds, h->stream,
                           D.wave_rows, D.wave_ent, D.wave_step_off, D.step_tile, D.wg_step_off, D.nwaves,
                           D.tile_shift, D.tw_rows, D.pk, D.tv, xin, e);
      } else {
        if ((rc = ensure_lds_limit(h, MODE, false, lds, (const void *)spmv_tiled_kernel<MODE, false>))) return rc;
        hipLaunchKernelGGL((spmv_tiled_kernel<MODE, false>), dim3(D.grid), dim3(TW_WPB * WAVE), lds, h->stream,
                           D.wave_rows, D.wave_ent, D.wave_step_off, D.step_tile, D.wg_step_off, D.nwaves,
                           D.tile_shift, D.tw_rows, D.pk, D.tv, xin, e);
      }
    }
  } else if (D.grid > 0) {
    hipLaunchKernelGGL(spmv_stream_kernel<MODE>, dim3(D.grid), dim3(TPB), 0, h->stream,
                       D.view(), xin, D.blks, D.nblk, D.per_xcd, h->remap ? 1 : 0, e);
  }
  if (D.nlong > 0) {
    hipLaunchKernelGGL(spmv_long_partial_kernel, dim3(D.nchunks), dim3(TPB), 0, h->stream,
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
    hipLaunchKernelGGL(spmv_long_partial_kernel, dim3(D.nchunks), dim3(TPB), 0, h->stream,
                       D.view(), xin, D.chunk_row, D.chunk_off, D.chunk_partial);
    hipLaunchKernelGGL(spmv_long_final_kernel<MODE_PLAIN>, dim3(D.long_grid), dim3(TPB), 0, h->stream,
                       D.long_row, D.long_chunk_ptr, D.nlong, D.chunk_partial, e, D.grid);
  }
  if (g1 > g0) {
    const size_t lds = sizeof(double) * ((size_t)TW_WPB * D.tw_rows + 3 * TW_WPB + (D.tw_scratch ? TW_WPB * WAVE : 0));
    const int w0 = g0 * TW_WPB;
    if (D.tw_scratch) {
      int rc = ensure_lds_limit(h, MODE_PLAIN, true, lds, (const void *)spmv_tiled_kernel<MODE_PLAIN, true>);
      if (rc) return rc;
      hipLaunchKernelGGL((spmv_tiled_kernel<MODE_PLAIN, true>), dim3(g1 - g0), dim3(TW_WPB * WAVE), lds, h->stream,
                         D.wave_rows + w0, D.wave_ent, D.wave_step_off + w0, D.step_tile, D.wg_step_off + g0,
                         D.nwaves - w0, D.tile_shift, D.tw_rows, D.pk, D.tv, xin, e);
    } else {
      int rc = ensure_lds_limit(h, MODE_PLAIN, false, lds, (const void *)spmv_tiled_kernel<MODE_PLAIN, false>);
      if (rc) return rc;
      hipLaunchKernelGGL((spmv_tiled_kernel<MODE_PLAIN, false>), dim3(g1 - g0), dim3(TW_WPB * WAVE), lds, h->stream,
                         D.wave_rows + w0, D.wave_ent, D.wave_step_off + w0, D.step_tile, D.wg_step_off + g0,
                         D.nwaves - w0, D.tile_shift, D.tw_rows, D.pk, D.tv, xin, e);
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_primal(pdhg_handle *h, double tau, double theta, bool write_xbar) {
  ProfScope ps(h, PDHG_K_PRIMAL);
  const int n = (int)h->n;
  if (h->has_q) {
    EpiArgs e{};
    e.out = h->qx;
    int rc = launch_spmv<MODE_PLAIN>(h, h->Q, h->x, e);
    if (rc) return rc;
  }
  const int grid = ew_grid((h->n + 1) / 2);
#define PK(HQ, WX)                                                                  \
  hipLaunchKernelGGL((primal_kernel<HQ, WX>), dim3(grid), dim3(TPB), 0, h->stream, n, \
                     h->x, h->c, h->aty, h->qx, h->lb, h->ub, tau, theta, h->x_next, h->xbar)
  if (h->has_q) { if (write_xbar) PK(true, true); else PK(true, false); }
  else          { if (write_xbar) PK(false, true); else PK(false, false); }
#undef PK
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_dual(pdhg_handle *h, double sigma) {
  ProfScope ps(h, PDHG_K_SPMV_DUAL);
  EpiArgs e{};
  e.y = h->y; e.b = h->b; e.y_next = h->y_next; e.sigma = sigma; e.num_eq = (int)h->num_eq;
  e.partials = h->pA; e.stride = h->A.slots();
  return launch_spmv<MODE_DUAL>(h, h->A, h->xbar, e);
}

int launch_aty_fused(pdhg_handle *h) {
  ProfScope ps(h, PDHG_K_SPMV_ATY);
  EpiArgs e{};
  e.x = h->x; e.x_next = h->x_next; e.aty = h->aty; e.aty_next = h->aty_next;
  e.partials = h->pAt; e.stride = h->pAt_stride;
  return launch_spmv<MODE_ATY>(h, h->At, h->y_next, e);
}

int launch_aty_plain(pdhg_handle *h, const double *yin, double *out) {
  ProfScope ps(h, PDHG_K_SPMV_ATY);
  EpiArgs e{};
  e.out = out;
  return launch_spmv<MODE_PLAIN>(h, h->At, yin, e);
}

// 0.5 * dx' Q dx partials into pQ (QP only)
int launch_q_interaction(pdhg_handle *h, int *count) {
  *count = 0;
  if (!h->has_q) return 0;
  const int n = (int)h->n;
  hipLaunchKernelGGL(diff_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, h->x_next, h->x, h->tmp_n);
  EpiArgs e{};
  e.out = h->tmp_n2;
  int rc = launch_spmv<MODE_PLAIN>(h, h->Qt, h->tmp_n, e);  // (dx' Q)' = Q' dx
  if (rc) return rc;
  hipLaunchKernelGGL(dot_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, h->tmp_n2, h->tmp_n, h->pQ);
  HIP_TRY(hipGetLastError());
  *count = h->ew_grid_n;
  return 0;
}

int finish_scalars(pdhg_handle *h, const double *p_int, int n_int, int stride_int,
                   const double *p_dy, int n_dy, int q_count, double out[5]) {
  {
    ProfScope ps(h, PDHG_K_FINAL);
    FinalSpec sp{};
    sp.ptr[0] = p_int;                  sp.count[0] = n_int;
    sp.ptr[1] = p_int + stride_int;     sp.count[1] = n_int;
    sp.ptr[2] = p_dy;                   sp.count[2] = n_dy;
    sp.ptr[3] = p_int + 2 * stride_int; sp.count[3] = n_int;
    sp.ptr[4] = h->pQ;                  sp.count[4] = q_count;
    sp.out = h->d_out;
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, sp);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int q = 0; q < 5; ++q) out[q] = h->h_out[q];
  out[4] *= 0.5;
  return 0;
}

int check_handle(pdhg_handle *h) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  return 0;
}

// Layout choice for one CSR: the tiled sweep pays off when the gathered vector
// (cols doubles) is far larger than an XCD's 4 MiB L2 and rows are short.
// PDHG_SPMV=stream|tiled forces a layout; PDHG_TILE_SHIFT sets log2(tile cols).
int choose_tile_shift(int64_t cols, int64_t nnz, int64_t rows) {
  const char *mode = getenv("PDHG_SPMV");
  const char *ts = getenv("PDHG_TILE_SHIFT");
  int shift = ts ? atoi(ts) : 16;   // 64K columns = 512 KiB of the gathered vector (best of 14..19 on MI355X)
  if (!ts && rows > 0) {
    // Few rows (a row shard of a multi-GPU run): a wave then owns few rows and a
    // 64K-column tile gives it well under one 64-entry chunk per step.  128K-column
    // tiles double the chunk fill (config S, 1/8 row shard: 0.200 -> 0.137 ms;
    // 1/4 shard 0.242 -> 0.222 ms; full matrix and the transposes stay at 64K).
    const int64_t slots = 256LL * 2 * TW_WPB;
    const int64_t rounds = std::max<int64_t>(1, (rows + slots * TW_MAX_ROWS - 1) / (slots * TW_MAX_ROWS));
    const int64_t rpw = std::max<int64_t>(64, (rows + slots * rounds - 1) / (slots * rounds));
    const int64_t ntiles16 = std::max<int64_t>(1, (cols + 65535) >> 16);
    const double per_step = (double)nnz / (double)rows * (double)rpw / (double)ntiles16;
    if (per_step < 48.0 && ntiles16 >= 4) shift = 17;
  }
  if (shift < 6) shift = 6;
  if (shift > 22) shift = 22;                            // leave >= 10 bits for row_local
  if (((cols + (1LL << shift) - 1) >> shift) > 65536) return 0;  // tile table would be huge
  if (mode && !strcmp(mode, "stream")) return 0;
  if (mode && !strcmp(mode, "tiled")) return shift;
  const bool big_vector = cols * 8 > (4LL << 20);        // larger than one XCD's 4 MiB L2
  const bool short_rows = rows > 0 && nnz / rows <= 64;
  return (big_vector && short_rows) ? shift : 0;
}

// CSC (any int64 base) -> int32 CSR of the transpose (direct) and CSR (counting sort).
int csc_to_both(int64_t rows, int64_t cols, int64_t nnz, const int64_t *colptr,
                const int64_t *rowval, const double *nzval, int base,
                std::vector<int> &t_rowptr, std::vector<int> &t_col, std::vector<double> &t_val,
                std::vector<int> &rowptr, std::vector<int> &col, std::vector<double> &val) {
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
  // CSR(A') is the CSC input itself (32-bit, 0-based); copied on host threads.
  std::atomic<int> bad{0};
  if (nnz > 0) parallel_ranges((int)std::min<int64_t>(nnz, 1 << 20), 1 << 15, [&](int cb, int ce) {
    const int64_t chunks = std::min<int64_t>(nnz, 1 << 20);
    const int64_t kb = nnz * cb / chunks, ke = nnz * ce / chunks;
    for (int64_t k = kb; k < ke; ++k) {
      const int64_t r = rowval[k] - base;
      if (r < 0 || r >= rows) { bad.store(1); return; }
      t_col[k] = (int)r;
      t_val[k] = nzval[k];
    }
  });
  if (bad.load()) return fail(-1, "rowval out of range");
  // CSR(A): a counting sort by row.  Every thread owns a contiguous range of ROWS
  // and walks all columns in ascending order, counting and then placing only the
  // entries of its rows -- so each row receives its entries in ascending column
  // order (what the sequential loop produces) and the writes of a thread stay
  // inside its own slice of col/val.
  rowptr.assign(rows + 1, 0);
  const int row_grain = nnz >= (1 << 22) ? 1 : INT32_MAX;   // below ~4M nonzeros one thread is faster
  parallel_ranges((int)rows, row_grain, [&](int rb, int re) {
    for (int64_t k = 0; k < nnz; ++k) {
      const int r = t_col[k];
      if (r >= rb && r < re) rowptr[r + 1] += 1;
    }
  });
  for (int64_t i = 0; i < rows; ++i) rowptr[i + 1] += rowptr[i];
  col.resize(nnz);
  val.resize(nnz);
  std::vector<int> next(rowptr.begin(), rowptr.end() - 1);
  parallel_ranges((int)rows, row_grain, [&](int rb, int re) {
    for (int64_t j = 0; j < cols; ++j) {
      for (int k = t_rowptr[j]; k < t_rowptr[j + 1]; ++k) {
        const int r = t_col[k];
        if (r >= rb && r < re) {
          const int p = next[r]++;
          col[p] = (int)j;
          val[p] = t_val[k];
        }
      }
    }
  });
  return 0;
}

}  // namespace

// ================================================================== C ABI

extern "C" {

const char *pdhg_last_error(void) { return g_last_error.c_str(); }
int pdhg_abi_version(void) { return 4; }

const char *pdhg_kernel_name(int kernel_id) {
  switch (kernel_id) {
    case PDHG_K_PRIMAL: return "primal_kernel";
    case PDHG_K_SPMV_DUAL: return "spmv_stream_kernel<MODE_DUAL>";
    case PDHG_K_SPMV_ATY: return "spmv_stream_kernel<MODE_ATY>";
    case PDHG_K_FINAL: return "final_reduce_kernel";
    case PDHG_K_ACCEPT: return "accept_kernel";
    default: return "?";
  }
}

int pdhg_create(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                const int64_t *colptr, const int64_t *rowval, const double *nzval,
                int index_base, const double *c, const double *b, const double *lb,
                const double *ub, int64_t num_equalities, int device_id, void *stream) {
  if (!out) return fail(-1, "out == NULL");
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

  const bool verbose = getenv("PDHG_VERBOSE") != nullptr;   // phase timings of the set-up on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  const auto t_start = now();
  std::vector<int> t_rowptr, t_col, rowptr, col;
  std::vector<double> t_val, val;
  int rc = csc_to_both(m, n, nnz, colptr, rowval, nzval, index_base, t_rowptr, t_col, t_val, rowptr, col, val);
  if (rc) return rc;
  const auto t_conv = now();

  pdhg_handle *h = new pdhg_handle();
  h->device = dev;
  h->m = m; h->n = n; h->nnz = nnz; h->num_eq = num_equalities;
  const char *env = getenv("PDHG_XCD_REMAP");
  h->remap = !(env && env[0] == '0');
  if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail((int)e, "hipStreamCreate failed"); }
    h->own_stream = true;
  }
#define CK(expr) do { int _rc = (expr); if (_rc) { pdhg_destroy(h); return _rc; } } while (0)
  CK(build_csr_dev(h->A, (int)m, (int)n, rowptr, col, val, h->remap, choose_tile_shift(n, nnz, m)));
  const auto t_a = now();
  CK(build_csr_dev(h->At, (int)n, (int)m, t_rowptr, t_col, t_val, h->remap, choose_tile_shift(m, nnz, n)));
  const auto t_at = now();
  if (verbose)
    fprintf(stderr, "pdhg_create: CSC -> CSR(A), CSR(A') %.2fs; layouts + upload A %.2fs, A' %.2fs\n",
            secs(t_start, t_conv), secs(t_conv, t_a), secs(t_a, t_at));
  auto up = [&](double **dst, const double *src, int64_t len) -> int {
    int r2 = alloc_zero(dst, len);
    if (r2) return r2;
    if (len > 0) HIP_TRY(hipMemcpy(*dst, src, sizeof(double) * (size_t)len, hipMemcpyHostToDevice));
    return 0;
  };
  CK(up(&h->c, c, n)); CK(up(&h->b, b, m)); CK(up(&h->lb, lb, n)); CK(up(&h->ub, ub, n));
  CK(alloc_zero(&h->x, n)); CK(alloc_zero(&h->x_next, n)); CK(alloc_zero(&h->xbar, n));
  CK(alloc_zero(&h->y, m)); CK(alloc_zero(&h->y_next, m));
  CK(alloc_zero(&h->aty, n + 1)); CK(alloc_zero(&h->aty_next, n + 1));
  CK(alloc_zero(&h->sum_x, n)); CK(alloc_zero(&h->sum_y, m));
  CK(alloc_zero(&h->tmp_n, n)); CK(alloc_zero(&h->tmp_m, m));
  h->ew_grid_n = ew_grid(n); h->ew_grid_m = ew_grid(m); h->ew_grid_nm = ew_grid(std::max(n, m));
  h->pAt_stride = std::max(h->At.slots(), h->ew_grid_n);
  CK(alloc_zero(&h->pA, std::max(h->A.slots(), 1)));
  CK(alloc_zero(&h->pAt, 3 * (int64_t)std::max(h->pAt_stride, 1)));
  CK(alloc_zero(&h->pQ, h->ew_grid_n));
  CK(alloc_zero(&h->d_out, 8));
  {
    hipError_t e = hipHostMalloc((void **)&h->h_out, 8 * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) { pdhg_destroy(h); return fail((int)e, "hipHostMalloc failed"); }
    e = hipEventCreate(&h->ev0); if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { pdhg_destroy(h); return fail((int)e, "hipEventCreate failed"); }
  }
#undef CK
  HIP_TRY(hipDeviceSynchronize());
  *out = h;
  return 0;
}

int pdhg_set_objective_matrix(pdhg_handle *h, int64_t q_nnz, const int64_t *q_colptr,
                              const int64_t *q_rowval, const double *q_nzval, int index_base) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  if (h->has_q) { free_csr_dev(h->Q); free_csr_dev(h->Qt); h->has_q = false; }
  bool all_zero = true;
  for (int64_t k = 0; k < q_nnz; ++k) if (q_nzval[k] != 0.0) all_zero = false;
  if (all_zero) return 0;  // iszero(objective_matrix): LP path (pdhg.jl:536)
  std::vector<int> t_rowptr, t_col, rowptr, col;
  std::vector<double> t_val, val;
  rc = csc_to_both(h->n, h->n, q_nnz, q_colptr, q_rowval, q_nzval, index_base, t_rowptr, t_col, t_val, rowptr, col, val);
  if (rc) return rc;
  if ((rc = build_csr_dev(h->Q, (int)h->n, (int)h->n, rowptr, col, val, h->remap))) return rc;
  if ((rc = build_csr_dev(h->Qt, (int)h->n, (int)h->n, t_rowptr, t_col, t_val, h->remap))) return rc;
  if (!h->qx) { if ((rc = alloc_zero(&h->qx, h->n))) return rc; }
  if (!h->tmp_n2) { if ((rc = alloc_zero(&h->tmp_n2, h->n))) return rc; }
  h->has_q = true;
  return 0;
}

void pdhg_destroy(pdhg_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  free_csr_dev(h->A); free_csr_dev(h->At); free_csr_dev(h->Q); free_csr_dev(h->Qt);
  double *bufs[] = {h->c, h->b, h->lb, h->ub, h->x, h->x_next, h->xbar, h->y, h->y_next,
                    h->aty, h->aty_next, h->sum_x, h->sum_y, h->qx, h->tmp_n, h->tmp_n2,
                    h->tmp_m, h->pA, h->pAt, h->pQ, h->d_out, h->E, h->Dv, h->c_o, h->b_o, h->lb_o,
                    h->ub_o, h->x_r, h->y_r, h->px_avg, h->py_avg, h->ev_ax, h->ev_aty, h->tr_g,
                    h->tr_dir, h->tr_thr, h->ev_partials, h->ev_out, h->ev_cax[0], h->ev_cax[1],
                    h->ev_caty[0], h->ev_caty[1], h->ev_cqx[0], h->ev_cqx[1], h->ev_qx};
  for (double *p : bufs) if (p) (void)hipFree(p);
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->ev_host) (void)hipHostFree(h->ev_host);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int pdhg_trial_primal(pdhg_handle *h, double step_size, double primal_weight) {
  int rc = check_handle(h);
  if (rc) return rc;
  return launch_primal(h, step_size / primal_weight, 0.0, false);
}

static int trial_dual_from(pdhg_handle *h, double step_size, double primal_weight, double out[5]) {
  int rc;
  if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  if ((rc = launch_aty_fused(h))) return rc;
  int qcount = 0;
  if ((rc = launch_q_interaction(h, &qcount))) return rc;
  return finish_scalars(h, h->pAt, h->At.slots(), h->pAt_stride, h->pA, h->A.slots(), qcount, out);
}

int pdhg_trial_dual(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  hipLaunchKernelGGL(xbar_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->x, h->x_next, theta, h->xbar);
  HIP_TRY(hipGetLastError());
  return trial_dual_from(h, step_size, primal_weight, out);
}

int pdhg_trial_step(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  if ((rc = launch_primal(h, step_size / primal_weight, theta, true))) return rc;
  return trial_dual_from(h, step_size, primal_weight, out);
}

int pdhg_accept(pdhg_handle *h, double avg_weight) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  {
    ProfScope ps(h, PDHG_K_ACCEPT);
    hipLaunchKernelGGL(accept_kernel, dim3(h->ew_grid_nm), dim3(TPB), 0, h->stream, (int)h->n, (int)h->m,
                       avg_weight, h->x_next, h->sum_x, h->y_next, h->sum_y);
    HIP_TRY(hipGetLastError());
  }
  std::swap(h->x, h->x_next);
  std::swap(h->y, h->y_next);
  std::swap(h->aty, h->aty_next);
  h->sum_x_count += 1; h->sum_y_count += 1;
  h->sum_x_weights += avg_weight; h->sum_y_weights += avg_weight;
  return 0;
}

int pdhg_add_current_primal_to_average(pdhg_handle *h, double weight) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  hipLaunchKernelGGL(accept_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, 0, weight,
                     h->x, h->sum_x, h->y, h->sum_y);
  HIP_TRY(hipGetLastError());
  h->sum_x_count += 1;
  h->sum_x_weights += weight;
  return 0;
}

int pdhg_get_average_info(pdhg_handle *h, int64_t counts[2], double weights[2]) {
  if (!h) return fail(-1, "null handle");
  counts[0] = h->sum_x_count; counts[1] = h->sum_y_count;
  weights[0] = h->sum_x_weights; weights[1] = h->sum_y_weights;
  return 0;
}

int pdhg_get_average(pdhg_handle *h, double *x_avg, double *y_avg) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (x_avg) {
    hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->sum_x, h->sum_x_weights, h->tmp_n);
    HIP_TRY(hipMemcpyAsync(x_avg, h->tmp_n, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  }
  if (y_avg) {
    hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->tmp_m);
    HIP_TRY(hipMemcpyAsync(y_avg, h->tmp_m, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int pdhg_reset_average(pdhg_handle *h) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  HIP_TRY(hipMemsetAsync(h->sum_x, 0, sizeof(double) * (size_t)std::max<int64_t>(h->n, 1), h->stream));
  HIP_TRY(hipMemsetAsync(h->sum_y, 0, sizeof(double) * (size_t)std::max<int64_t>(h->m, 1), h->stream));
  h->sum_x_count = h->sum_y_count = 0;
  h->sum_x_weights = h->sum_y_weights = 0.0;
  return 0;
}

int pdhg_restart_to_average(pdhg_handle *h) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  if (h->sum_x_count == 0 || h->sum_y_count == 0) return fail(-1, "average is empty");
  hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->sum_x, h->sum_x_weights, h->x);
  hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->y);
  HIP_TRY(hipGetLastError());
  return launch_aty_plain(h, h->y, h->aty);
}

int pdhg_get_current(pdhg_handle *h, double *x, double *y, double *aty) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (x) HIP_TRY(hipMemcpyAsync(x, h->x, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (y) HIP_TRY(hipMemcpyAsync(y, h->y, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
  if (aty) HIP_TRY(hipMemcpyAsync(aty, h->aty, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int pdhg_get_trial(pdhg_handle *h, double *x_next, double *y_next, double *aty_next) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (x_next) HIP_TRY(hipMemcpyAsync(x_next, h->x_next, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (y_next) HIP_TRY(hipMemcpyAsync(y_next, h->y_next, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
  if (aty_next) HIP_TRY(hipMemcpyAsync(aty_next, h->aty_next, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int pdhg_set_current(pdhg_handle *h, const double *x, const double *y) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  if (x) HIP_TRY(hipMemcpyAsync(h->x, x, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  if (y) HIP_TRY(hipMemcpyAsync(h->y, y, sizeof(double) * (size_t)h->m, hipMemcpyHostToDevice, h->stream));
  rc = launch_aty_plain(h, h->y, h->aty);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int pdhg_spmv(pdhg_handle *h, const double *x, double *out) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!x || !out) return fail(-1, "null vector");
  HIP_TRY(hipMemcpyAsync(h->tmp_n, x, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  EpiArgs e{};
  e.out = h->tmp_m;
  if ((rc = launch_spmv<MODE_PLAIN>(h, h->A, h->tmp_n, e))) return rc;
  HIP_TRY(hipMemcpyAsync(out, h->tmp_m, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int pdhg_spmv_t(pdhg_handle *h, const double *y, double *out) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!y || !out) return fail(-1, "null vector");
  HIP_TRY(hipMemcpyAsync(h->tmp_m, y, sizeof(double) * (size_t)h->m, hipMemcpyHostToDevice, h->stream));
  EpiArgs e{};
  e.out = h->tmp_n;
  if ((rc = launch_spmv<MODE_PLAIN>(h, h->At, h->tmp_m, e))) return rc;
  HIP_TRY(hipMemcpyAsync(out, h->tmp_n, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

// ---- row-partitioned form ---------------------------------------------------

int pdhg_dist_trial_begin(pdhg_handle *h, double step_size, double primal_weight, double theta) {
  int rc = check_handle(h);
  if (rc) return rc;
  if ((rc = launch_primal(h, step_size / primal_weight, theta, true))) return rc;   // Q (if any) is replicated
  if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  if ((rc = launch_aty_plain(h, h->y_next, h->aty_next))) return rc;
  hipLaunchKernelGGL(final_to_slot_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->pA, h->A.slots(), h->aty_next + h->n);
  HIP_TRY(hipGetLastError());
  h->dist_pending = true;
  return 0;
}

int pdhg_dist_trial_dual_begin(pdhg_handle *h, double step_size, double primal_weight, double theta) {
  int rc = check_handle(h);
  if (rc) return rc;
  hipLaunchKernelGGL(xbar_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->x, h->x_next, theta, h->xbar);
  HIP_TRY(hipGetLastError());
  if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  if ((rc = launch_aty_plain(h, h->y_next, h->aty_next))) return rc;
  hipLaunchKernelGGL(final_to_slot_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->pA, h->A.slots(), h->aty_next + h->n);
  HIP_TRY(hipGetLastError());
  h->dist_pending = true;
  return 0;
}

// ---- the same trial in parts, so that the caller can all-reduce finished column
// ranges of the exchange buffer while later ones are still being computed --------

int pdhg_dist_parts(pdhg_handle *h, int max_parts, int64_t *bounds) {
  int rc = check_handle(h);
  if (rc) return rc < 0 ? rc : -rc;
  if (max_parts < 1 || !bounds) return fail(-1, "max_parts < 1 or bounds == NULL");
  const CsrDev &D = h->At;
  // A part is a whole number of residency rounds (256 CUs x 2 workgroups): a
  // smaller launch would leave CUs idle and cost more than the overlap buys.
  const char *rw = getenv("PDHG_DIST_ROUND_WGS");   // tests use a finer granule on small problems
  const int round_wgs = rw ? std::max(1, atoi(rw)) : 256 * 2;
  const int rounds = D.tiled ? D.grid / round_wgs : 0;
  const int parts = std::max(1, std::min(max_parts, rounds));
  h->dist_part_wg.assign((size_t)parts + 1, 0);
  for (int k = 0; k <= parts; ++k) {
    const int g = (k == parts) ? D.grid : round_wgs * (int)(((int64_t)rounds * k) / parts);
    h->dist_part_wg[k] = D.tiled ? g : 0;
    bounds[k] = (k == 0) ? 0 : (k == parts ? h->n : (int64_t)D.wg_first_row[g]);
  }
  h->dist_part_wg[parts] = D.tiled ? D.grid : 0;
  return parts;
}

static int dist_part_common(pdhg_handle *h, int part, int nparts) {
  if (nparts != (int)h->dist_part_wg.size() - 1) return fail(-1, "nparts does not match pdhg_dist_parts");
  if (part < 0 || part >= nparts) return fail(-1, "part out of range");
  int rc;
  if (nparts == 1) {
    if ((rc = launch_aty_plain(h, h->y_next, h->aty_next))) return rc;
  } else {
    ProfScope ps(h, PDHG_K_SPMV_ATY);
    if ((rc = launch_spmv_plain_part(h, h->At, h->y_next, h->aty_next, h->dist_part_wg[part],
                                     h->dist_part_wg[part + 1], part == 0))) return rc;
  }
  if (part == nparts - 1) {
    hipLaunchKernelGGL(final_to_slot_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->pA, h->A.slots(), h->aty_next + h->n);
    HIP_TRY(hipGetLastError());
    h->dist_pending = true;
  }
  return 0;
}

int pdhg_dist_trial_begin_part(pdhg_handle *h, double step_size, double primal_weight, double theta,
                               int part, int nparts) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (part == 0) {
    if ((rc = launch_primal(h, step_size / primal_weight, theta, true))) return rc;
    if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  }
  return dist_part_common(h, part, nparts);
}

int pdhg_dist_trial_dual_begin_part(pdhg_handle *h, double step_size, double primal_weight, double theta,
                                    int part, int nparts) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (part == 0) {
    hipLaunchKernelGGL(xbar_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->x, h->x_next, theta, h->xbar);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  }
  return dist_part_common(h, part, nparts);
}

void *pdhg_dist_exchange_ptr(pdhg_handle *h) { return h ? (void *)h->aty_next : nullptr; }

int pdhg_dist_trial_end(pdhg_handle *h, double out[5]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!h->dist_pending) return fail(-1, "pdhg_dist_trial_end without begin");
  h->dist_pending = false;
  hipLaunchKernelGGL(interaction_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->x, h->x_next,
                     h->aty, h->aty_next, h->pAt, h->pAt_stride);
  HIP_TRY(hipGetLastError());
  int qcount = 0;
  if ((rc = launch_q_interaction(h, &qcount))) return rc;   // 0.5 dx'Q dx on the replicated vectors (QP)
  return finish_scalars(h, h->pAt, h->ew_grid_n, h->pAt_stride, h->aty_next + h->n, 1, qcount, out);
}

// A'y recompute in two halves: local partial into the exchange buffer
// (aty_next, since aty may still be needed), then adopt it after all-reduce.
int pdhg_dist_dual_product_begin(pdhg_handle *h) {
  int rc = check_handle(h);
  if (rc) return rc;
  if ((rc = launch_aty_plain(h, h->y, h->aty_next))) return rc;
  HIP_TRY(hipMemsetAsync(h->aty_next + h->n, 0, sizeof(double), h->stream));
  return 0;
}
int pdhg_dist_dual_product_end(pdhg_handle *h) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  std::swap(h->aty, h->aty_next);
  return 0;
}

// ---- evaluation branch on the device (N1) -----------------------------------

static int ev_alloc(pdhg_handle *h) {
  if (h->ev_partials) return 0;
  int rc;
  h->ev_grid = ew_grid(std::max(h->n, h->m) + 1);
  if ((rc = alloc_zero(&h->ev_partials, (int64_t)EV_MAXQ * h->ev_grid))) return rc;
  if ((rc = alloc_zero(&h->ev_out, EV_MAXQ))) return rc;
  HIP_TRY(hipHostMalloc((void **)&h->ev_host, EV_MAXQ * sizeof(double), hipHostMallocDefault));
  if ((rc = alloc_zero(&h->ev_ax, h->m))) return rc;
  if ((rc = alloc_zero(&h->ev_aty, h->n))) return rc;
  for (int k = 0; k < 2; ++k) {
    if ((rc = alloc_zero(&h->ev_cax[k], h->m))) return rc;
    if ((rc = alloc_zero(&h->ev_caty[k], h->n))) return rc;
  }
  if ((rc = alloc_zero(&h->px_avg, h->n))) return rc;
  if ((rc = alloc_zero(&h->py_avg, h->m))) return rc;
  if ((rc = alloc_zero(&h->x_r, h->n))) return rc;   // zeros == the initial restart point (pdhg.jl:869)
  if ((rc = alloc_zero(&h->y_r, h->m))) return rc;
  return 0;
}

static int ev_finish(pdhg_handle *h, int ns, int nm, double *out) {
  hipLaunchKernelGGL(multi_final_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->ev_partials, h->ev_grid,
                     h->ev_grid, ns, nm, h->ev_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h->ev_host, h->ev_out, (ns + nm) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int q = 0; q < ns + nm; ++q) out[q] = h->ev_host[q];
  return 0;
}

static int select_point(pdhg_handle *h, int point, const double **px, const double **py) {
  int rc = ev_alloc(h);
  if (rc) return rc;
  if (point == PDHG_POINT_CURRENT) { *px = h->x; *py = h->y; return 0; }
  if (point == PDHG_POINT_RESTART) { *px = h->x_r; *py = h->y_r; return 0; }
  if (point == PDHG_POINT_AVERAGE) {
    if (h->sum_x_count == 0 || h->sum_y_count == 0) return fail(-1, "average is empty");
    if (h->avg_version != h->state_version) {
      hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->sum_x, h->sum_x_weights, h->px_avg);
      hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->py_avg);
      HIP_TRY(hipGetLastError());
      h->avg_version = h->state_version;
    }
    *px = h->px_avg; *py = h->py_avg;
    return 0;
  }
  return fail(-1, "unknown point selector");
}

// A*x, A'*y and (QP) Q*x at a point selected by select_point (cached for CURRENT / AVERAGE).
static int point_products(pdhg_handle *h, int point, const double *px, const double *py,
                          const double **ax, const double **aty, const double **qx) {
  int rc;
  double *dax = h->ev_ax, *daty = h->ev_aty;
  double **dqx = &h->ev_qx;
  static const bool cache_off = getenv("PDHG_NO_EVAL_CACHE") != nullptr;   // debugging aid
  const bool cached = !cache_off && (point == PDHG_POINT_CURRENT || point == PDHG_POINT_AVERAGE);
  bool fresh = true;
  if (cached) {
    const int k = point == PDHG_POINT_CURRENT ? 0 : 1;
    dax = h->ev_cax[k]; daty = h->ev_caty[k]; dqx = &h->ev_cqx[k];
    fresh = h->ev_cversion[k] != h->state_version;
    h->ev_cversion[k] = h->state_version;
  }
  if (h->has_q && !*dqx) {
    if ((rc = alloc_zero(dqx, h->n))) return rc;
    fresh = true;
  }
  if (fresh) {
    EpiArgs e{};
    e.out = dax;
    if ((rc = launch_spmv<MODE_PLAIN>(h, h->A, px, e))) return rc;
    e.out = daty;
    if ((rc = launch_spmv<MODE_PLAIN>(h, h->At, py, e))) return rc;
    if (h->has_q) {
      e.out = *dqx;
      if ((rc = launch_spmv<MODE_PLAIN>(h, h->Q, px, e))) return rc;
    }
  }
  *ax = dax; *aty = daty;
  *qx = h->has_q ? *dqx : nullptr;
  return 0;
}

int pdhg_set_original_problem(pdhg_handle *h, const double *constraint_rescaling,
                              const double *variable_rescaling, const double *c_o, const double *b_o,
                              const double *lb_o, const double *ub_o) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!constraint_rescaling || !variable_rescaling || !c_o || !lb_o || !ub_o || (h->m > 0 && !b_o))
    return fail(-1, "null input array");
  auto up = [&](double **dst, const double *src, int64_t len) -> int {
    if (!*dst) { int r2 = alloc_zero(dst, len); if (r2) return r2; }
    if (len > 0) HIP_TRY(hipMemcpy(*dst, src, sizeof(double) * (size_t)len, hipMemcpyHostToDevice));
    return 0;
  };
  if ((rc = up(&h->E, constraint_rescaling, h->m))) return rc;
  if ((rc = up(&h->Dv, variable_rescaling, h->n))) return rc;
  if ((rc = up(&h->c_o, c_o, h->n))) return rc;
  if ((rc = up(&h->b_o, b_o, h->m))) return rc;
  if ((rc = up(&h->lb_o, lb_o, h->n))) return rc;
  if ((rc = up(&h->ub_o, ub_o, h->n))) return rc;
  h->has_original = true;
  return ev_alloc(h);
}

int pdhg_eval_point(pdhg_handle *h, int point, double out[24]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!h->has_original) return fail(-1, "pdhg_set_original_problem has not been called");
  const double *px, *py;
  if ((rc = select_point(h, point, &px, &py))) return rc;
  const double *ax, *aty, *qx;
  if ((rc = point_products(h, point, px, py, &ax, &aty, &qx))) return rc;
  hipLaunchKernelGGL(eval_rows_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->m, (int)h->num_eq,
                     ax, py, h->E, h->b_o, h->ev_partials, h->ev_grid);
  if ((rc = ev_finish(h, 4, 4, out))) return rc;
  hipLaunchKernelGGL(eval_cols_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, aty, qx, px,
                     h->Dv, h->c_o, h->lb_o, h->ub_o, h->ev_partials, h->ev_grid);
  double r[14];
  if ((rc = ev_finish(h, 7, 7, r))) return rc;
  for (int q = 0; q < 6; ++q) { out[8 + q] = r[q]; out[14 + q] = r[7 + q]; }
  out[20] = r[6]; out[21] = r[13]; out[22] = out[23] = 0.0;
  return 0;
}

int pdhg_save_restart_point(pdhg_handle *h) {
  int rc = check_handle(h);
  if (rc) return rc;
  if ((rc = ev_alloc(h))) return rc;
  HIP_TRY(hipMemcpyAsync(h->x_r, h->x, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->y_r, h->y, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToDevice, h->stream));
  return 0;
}

int pdhg_distance_to_restart(pdhg_handle *h, int point, double out[2]) {
  int rc = check_handle(h);
  if (rc) return rc;
  const double *px, *py;
  if ((rc = select_point(h, point, &px, &py))) return rc;
  hipLaunchKernelGGL(dist2_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, (int)h->m, px, h->x_r,
                     py, h->y_r, h->ev_partials, h->ev_grid);
  return ev_finish(h, 2, 0, out);
}

int pdhg_point_sumsq(pdhg_handle *h, int point, double out[2]) {
  int rc = check_handle(h);
  if (rc) return rc;
  const double *px, *py;
  if ((rc = select_point(h, point, &px, &py))) return rc;
  hipLaunchKernelGGL(dist2_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, (int)h->m, px,
                     (const double *)nullptr, py, (const double *)nullptr, h->ev_partials, h->ev_grid);
  return ev_finish(h, 2, 0, out);
}

int pdhg_get_point(pdhg_handle *h, int point, double *x, double *y) {
  int rc = check_handle(h);
  if (rc) return rc;
  const double *px, *py;
  if ((rc = select_point(h, point, &px, &py))) return rc;
  if (x) HIP_TRY(hipMemcpyAsync(x, px, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (y) HIP_TRY(hipMemcpyAsync(y, py, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

static inline uint64_t d2bits(double v) { uint64_t b; memcpy(&b, &v, 8); return b; }
static inline double bits2d(uint64_t b) { double v; memcpy(&v, &b, 8); return v; }

int pdhg_trust_region_bound(pdhg_handle *h, int point, double primal_weight_norm, double dual_weight_norm,
                            double radius, int range, int approximate, double out[8]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (range < 0 || range > 2) return fail(-1, "range must be 0, 1 or 2");
  const double *px, *py;
  if ((rc = select_point(h, point, &px, &py))) return rc;
  const int64_t total = h->n + h->m;
  if (!h->tr_g) {
    if ((rc = alloc_zero(&h->tr_g, total))) return rc;
    if ((rc = alloc_zero(&h->tr_dir, total))) return rc;
    if ((rc = alloc_zero(&h->tr_thr, total))) return rc;
  }
  const double wp = primal_weight_norm, wd = dual_weight_norm;
  const double *ax, *aty, *qx;
  if ((rc = point_products(h, point, px, py, &ax, &aty, &qx))) return rc;
  hipLaunchKernelGGL(tr_setup_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, (int)h->m,
                     (int)h->num_eq, px, py, aty, qx, ax, h->c, h->b, h->lb, h->ub, wp, wd, range,
                     h->tr_g, h->tr_dir, h->tr_thr, h->ev_partials, h->ev_grid);
  double r[EV_MAXQ];
  if ((rc = ev_finish(h, 11, 1, r))) return rc;
  // compute_lagrangian_value (saddle_point.jl:1109-1120) without objective_constant
  out[0] = 0.5 * r[10] + r[0] - r[1] + r[2];
  out[1] = out[2] = 0.0;
  out[3] = r[8]; out[4] = r[9];
  out[5] = 0.0; out[6] = 0.0; out[7] = 0.0;
  const double hinf = r[3], g2 = r[4], wd2_all = r[5], tmax = r[11];
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
  // same closed form (trust_region_utils.jl:167-175).
  auto probe = [&](const TrProbes &pr, double *lowhigh) -> int {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, (int)total,
                       h->tr_dir, h->tr_thr, wp, wd, pr, h->ev_partials, h->ev_grid);
    return ev_finish(h, 2 * TR_K, 0, lowhigh);
  };
  double lh[2 * TR_K];
  TrProbes pr;
  for (int q = 0; q < TR_K; ++q) pr.t[q] = tmax;
  if ((rc = probe(pr, lh))) return rc;
  int passes = 1;
  double tstar;
  if (lh[0] + tmax * tmax * lh[1] <= r2) {
    // every finite breakpoint is reached before the radius
    if (hinf <= 0.0) tstar = tmax;                       // "all bounds hit" special case
    else tstar = sqrt((r2 - lh[0]) / hinf);
  } else {
    uint64_t lo = 0, hi = d2bits(tmax);
    double low_lo = 0.0, high_lo = 0.0;
    bool have_lo = false;
    bool exact = false;
    tstar = 0.0;
    while (hi - lo > 1) {
      uint64_t pb[TR_K];
      const uint64_t span = hi - lo;
      int q0 = 0;
      // Probe 0: the closed-form candidate from the current lower end,
      // t' = sqrt((r2 - low)/high).  If no breakpoint lies in (lo, t'] the
      // probe returns the same (low, high) and t' is the exact answer -- this
      // fixed-point step usually lands within a few passes; the remaining
      // probes keep a guaranteed 8-ary bracket in IEEE bit space.
      if (have_lo && high_lo > 0.0) {
        const double cand = sqrt(fmax(r2 - low_lo, 0.0) / high_lo);
        const uint64_t cb = d2bits(cand);
        if (cb > lo && cb < hi) { pb[0] = cb; pr.t[0] = cand; q0 = 1; }
      }
      for (int q = q0; q < TR_K; ++q) {
        uint64_t off = (uint64_t)(((__uint128_t)span * (uint64_t)(q - q0 + 1)) / (uint64_t)(TR_K - q0 + 1));
        if (off == 0) off = 1;
        if (off >= span) off = span - 1;
        pb[q] = lo + off;
        pr.t[q] = bits2d(pb[q]);
      }
      if ((rc = probe(pr, lh))) return rc;
      ++passes;
      if (q0 == 1 && lh[0] == low_lo && lh[1] == high_lo) { tstar = pr.t[0]; exact = true; break; }
      uint64_t nlo = lo, nhi = hi;
      for (int q = 0; q < TR_K; ++q) {
        const double f = lh[2 * q] + pr.t[q] * pr.t[q] * lh[2 * q + 1];
        if (f <= r2) { if (pb[q] > nlo) { nlo = pb[q]; low_lo = lh[2 * q]; high_lo = lh[2 * q + 1]; have_lo = true; } }
        else { if (pb[q] < nhi) nhi = pb[q]; }
      }
      lo = nlo; hi = nhi;
    }
    if (!exact) {
      if (!have_lo) {  // bracket collapsed at t = 0: evaluate low/high there
        for (int q = 0; q < TR_K; ++q) pr.t[q] = 0.0;
        if ((rc = probe(pr, lh))) return rc;
        ++passes;
        low_lo = lh[0]; high_lo = lh[1];
      }
      tstar = high_lo > 0.0 ? sqrt(fmax(r2 - low_lo, 0.0) / high_lo) : bits2d(lo);
    }
  }
  hipLaunchKernelGGL(tr_value_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, (int)h->m,
                     (int)h->num_eq, px, py, h->lb, h->ub, h->tr_g, h->tr_dir, tstar, h->ev_partials, h->ev_grid);
  double vv[2];
  if ((rc = ev_finish(h, 2, 0, vv))) return rc;
  out[1] = vv[0]; out[2] = vv[1]; out[5] = tstar; out[6] = (double)passes;
  return 0;
}

// ---- rescaling on the device (N2) --------------------------------------------

static int row_grid(int rows) { return std::max(1, (rows + (TPB / WAVE) - 1) / (TPB / WAVE)); }

// one scale_problem step on every resident layout + the vectors
static int apply_scaling(pdhg_handle *h, double *ev, double *dv, double *inv_e, double *inv_d,
                         double *cum_e, double *cum_d) {
  const int n = (int)h->n, m = (int)h->m;
  hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, ev, inv_e, 0);
  hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, dv, inv_d, 0);
  CsrDev *L[2] = {&h->A, &h->At};
  for (int t = 0; t < 2; ++t) {
    CsrDev &D = *L[t];
    if (D.nnz == 0) continue;
    hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(D.rows)), dim3(TPB), 0, h->stream, D.rows, D.rowptr,
                       D.col, D.val, inv_e, inv_d, t);
    if (D.tiled && D.nwaves > 0)
      hipLaunchKernelGGL(scale_tiled_kernel, dim3(row_grid(D.nwaves)), dim3(TPB), 0, h->stream, D.wave_rows,
                         D.wave_ent, D.wave_step_off, D.step_tile, D.wg_step_off, D.nwaves, D.tile_shift,
                         D.pk, D.tv, inv_e, inv_d, t);
  }
  if (h->has_q) {
    // objective_matrix = (D^-1 Q) D^-1 (preprocess.jl:562-564); Qt holds Q' entry by entry, so the
    // "transposed" order reproduces the same two roundings on it
    hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(h->Q.rows)), dim3(TPB), 0, h->stream, h->Q.rows, h->Q.rowptr,
                       h->Q.col, h->Q.val, inv_d, inv_d, 0);
    hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(h->Qt.rows)), dim3(TPB), 0, h->stream, h->Qt.rows, h->Qt.rowptr,
                       h->Qt.col, h->Qt.val, inv_d, inv_d, 1);
  }
  hipLaunchKernelGGL(resc_apply_vectors_kernel, dim3(h->ew_grid_nm), dim3(TPB), 0, h->stream, n, m, dv, ev,
                     h->c, h->lb, h->ub, h->b, cum_d, cum_e);
  HIP_TRY(hipGetLastError());
  return 0;
}

int pdhg_rescale(pdhg_handle *h, int l_inf_ruiz_iterations, int l2_norm_rescaling,
                 int use_pock_chambolle, double pock_chambolle_alpha,
                 double *constraint_rescaling_out, double *variable_rescaling_out) {
  int rc = check_handle(h);
  if (rc) return rc;
  h->state_version += 1;   // x, y, the running sums or A change: cached A*x / A'*y are stale
  if (use_pock_chambolle && !(pock_chambolle_alpha >= 0.0 && pock_chambolle_alpha <= 2.0))
    return fail(-1, "pock_chambolle_alpha must be in [0, 2]");
  const int n = (int)h->n, m = (int)h->m;
  double *ev = nullptr, *dv = nullptr, *inv_e = nullptr, *inv_d = nullptr, *cum_e = nullptr, *cum_d = nullptr;
  double *tmp_e = nullptr, *tmp_d = nullptr;
  auto cleanup = [&]() { for (double *p : {ev, dv, inv_e, inv_d, cum_e, cum_d, tmp_e, tmp_d}) if (p) (void)hipFree(p); };
#define RS(expr) do { int _r = (expr); if (_r) { cleanup(); return _r; } } while (0)
  RS(alloc_zero(&ev, m)); RS(alloc_zero(&dv, n)); RS(alloc_zero(&inv_e, m)); RS(alloc_zero(&inv_d, n));
  RS(alloc_zero(&cum_e, m)); RS(alloc_zero(&cum_d, n)); RS(alloc_zero(&tmp_e, m)); RS(alloc_zero(&tmp_d, n));
  hipLaunchKernelGGL(fill_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, 1.0, cum_e);
  hipLaunchKernelGGL(fill_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, 1.0, cum_d);
  const CsrView Av = h->A.view(), Atv = h->At.view();
  // ruiz_rescaling, p = Inf (preprocess.jl:412-477): sqrt of the row / column max |a|, zeros -> 1
  for (int it = 0; it < l_inf_ruiz_iterations; ++it) {
    hipLaunchKernelGGL(row_op_kernel<ROP_MAXABS>, dim3(row_grid(n)), dim3(TPB), 0, h->stream, Atv, m, 0.0, (const double *)nullptr, dv);
    hipLaunchKernelGGL(row_op_kernel<ROP_MAXABS>, dim3(row_grid(m)), dim3(TPB), 0, h->stream, Av, n, 0.0, (const double *)nullptr, ev);
    if (h->has_q) {   // QP: column max over the constraint AND the objective matrix (preprocess.jl:425-433)
      hipLaunchKernelGGL(row_op_kernel<ROP_MAXABS>, dim3(row_grid(n)), dim3(TPB), 0, h->stream, h->Qt.view(), n, 0.0, (const double *)nullptr, tmp_d);
      hipLaunchKernelGGL(resc_max_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, dv, tmp_d);
    }
    hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, dv);
    hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, ev);
    RS(apply_scaling(h, ev, dv, inv_e, inv_d, cum_e, cum_d));
  }
  // l2_norm_rescaling (preprocess.jl:358-372): sqrt of the row / column L2 norms, zeros -> 1
  if (l2_norm_rescaling) {
    hipLaunchKernelGGL(row_op_kernel<ROP_MAXABS>, dim3(row_grid(n)), dim3(TPB), 0, h->stream, Atv, m, 0.0, (const double *)nullptr, tmp_d);
    hipLaunchKernelGGL(row_op_kernel<ROP_MAXABS>, dim3(row_grid(m)), dim3(TPB), 0, h->stream, Av, n, 0.0, (const double *)nullptr, tmp_e);
    hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, tmp_d, inv_d, 1);
    hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, tmp_e, inv_e, 1);
    hipLaunchKernelGGL(row_op_kernel<ROP_SUMSQ_SCALED>, dim3(row_grid(n)), dim3(TPB), 0, h->stream, Atv, m, 0.0, inv_d, dv);
    hipLaunchKernelGGL(row_op_kernel<ROP_SUMSQ_SCALED>, dim3(row_grid(m)), dim3(TPB), 0, h->stream, Av, n, 0.0, inv_e, ev);
    hipLaunchKernelGGL(resc_l2norm_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, tmp_d, dv);
    hipLaunchKernelGGL(resc_l2norm_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, tmp_e, ev);
    hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, dv);   // norm 0 -> sqrt 0 -> 1
    hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, ev);
    RS(apply_scaling(h, ev, dv, inv_e, inv_d, cum_e, cum_d));
  }
  // pock_chambolle_rescaling (preprocess.jl:508-539)
  if (use_pock_chambolle) {
    hipLaunchKernelGGL(row_op_kernel<ROP_SUMPOW>, dim3(row_grid(n)), dim3(TPB), 0, h->stream, Atv, m, 2.0 - pock_chambolle_alpha, (const double *)nullptr, dv);
    hipLaunchKernelGGL(row_op_kernel<ROP_SUMPOW>, dim3(row_grid(m)), dim3(TPB), 0, h->stream, Av, n, pock_chambolle_alpha, (const double *)nullptr, ev);
    hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, dv);
    hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, ev);
    RS(apply_scaling(h, ev, dv, inv_e, inv_d, cum_e, cum_d));
  }
  hipError_t e1 = hipGetLastError();
  if (e1 != hipSuccess) { cleanup(); return fail((int)e1, hipGetErrorString(e1)); }
  if (constraint_rescaling_out && m > 0)
    (void)hipMemcpyAsync(constraint_rescaling_out, cum_e, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, h->stream);
  if (variable_rescaling_out && n > 0)
    (void)hipMemcpyAsync(variable_rescaling_out, cum_d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipStreamSynchronize(h->stream);
  cleanup();
#undef RS
  if (e2 != hipSuccess) return fail((int)e2, hipGetErrorString(e2));
  return 0;
}

int pdhg_get_problem_vectors(pdhg_handle *h, double *c, double *b, double *lb, double *ub) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (c) HIP_TRY(hipMemcpyAsync(c, h->c, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (b) HIP_TRY(hipMemcpyAsync(b, h->b, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
  if (lb) HIP_TRY(hipMemcpyAsync(lb, h->lb, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (ub) HIP_TRY(hipMemcpyAsync(ub, h->ub, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int pdhg_matrix_max_abs(pdhg_handle *h, double *out) {
  int rc = check_handle(h);
  if (rc) return rc;
  if ((rc = ev_alloc(h))) return rc;
  hipLaunchKernelGGL(maxabs_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int64_t)h->At.nnz, h->At.val,
                     h->ev_partials, h->ev_grid);
  return ev_finish(h, 0, 1, out);
}

// ---- measurement ------------------------------------------------------------

int pdhg_profile_enable(pdhg_handle *h, int enable) {
  if (!h) return fail(-1, "null handle");
  h->profile = enable != 0;
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
  const int64_t m = h->m, n = h->n, nnz = h->nnz;
  switch (kernel_id) {
    case PDHG_K_PRIMAL: return 8 * 7 * n;                               // r: x,c,aty,lb,ub  w: x',xbar
    case PDHG_K_SPMV_DUAL: return nnz * 12 + (m + 1) * 4 + n * 8 + 3 * m * 8;  // + r: y,b  w: y'
    case PDHG_K_SPMV_ATY: return nnz * 12 + (n + 1) * 4 + m * 8 + 4 * n * 8;   // + r: x,x',aty  w: aty'
    case PDHG_K_FINAL: return 8 * (int64_t)(3 * h->At.slots() + h->A.slots());
    case PDHG_K_ACCEPT: return 8 * 3 * (n + m);
    default: return -1;
  }
}

namespace {
__global__ __launch_bounds__(TPB) void triad_kernel(int64_t len2, const double2 *__restrict__ b,
                                                    const double2 *__restrict__ c, double s,
                                                    double2 *__restrict__ a) {
  const int64_t stride = (int64_t)gridDim.x * TPB;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < len2; i += stride) {
    const double2 bv = b[i], cv = c[i];
    a[i] = make_double2(bv.x + s * cv.x, bv.y + s * cv.y);
  }
}
}  // namespace

int pdhg_measure_triad(pdhg_handle *h, int64_t len, int reps, double *gbps) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (len <= 0 || (len & 1) || reps <= 0 || !gbps) return fail(-1, "bad triad arguments (len must be even)");
  double *buf = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, sizeof(double) * 3 * (size_t)len));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float best = 1e30f;
  hipError_t err = hipMemsetAsync(buf, 0, sizeof(double) * 3 * (size_t)len, h->stream);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  const int64_t len2 = len / 2;
  const int grids[3] = {256 * 8, 256 * 16, 256 * 64};   // grid-stride; keep the best shape
  for (int g = 0; g < 3 && err == hipSuccess; ++g) {
    for (int r = 0; r <= reps && err == hipSuccess; ++r) {   // pass 0 warms up
      (void)hipEventRecord(e0, h->stream);
      hipLaunchKernelGGL(triad_kernel, dim3(grids[g]), dim3(TPB), 0, h->stream, len2,
                         reinterpret_cast<const double2 *>(buf + len), reinterpret_cast<const double2 *>(buf + 2 * len),
                         0.5, reinterpret_cast<double2 *>(buf));
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
  *gbps = 24.0 * (double)(2 * len2) / ((double)best * 1e-3) / 1e9;
  return 0;
}

int pdhg_layout_info(pdhg_handle *h, int64_t info[12]) {
  if (!h) return fail(-1, "null handle");
  info[0] = h->A.nblk; info[1] = h->A.nlong; info[2] = h->A.nchunks; info[3] = h->A.max_row_nnz;
  info[4] = h->At.nblk; info[5] = h->At.nlong; info[6] = h->At.nchunks; info[7] = h->At.max_row_nnz;
  info[8] = h->A.tiled ? h->A.nwaves : 0; info[9] = h->At.tiled ? h->At.nwaves : 0;
  info[10] = h->A.tiled ? h->A.tile_shift : 0; info[11] = h->At.tiled ? h->At.tile_shift : 0;
  return 0;
}

}  // extern "C"
