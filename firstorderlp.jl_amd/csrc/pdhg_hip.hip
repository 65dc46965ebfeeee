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
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off: elementwise updates must round like Julia's unfused
//  broadcasts; the product a*x and the sum are separate roundings).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pdhg_hip.h"

namespace {

constexpr int TPB = 256;             // 4 waves of 64
constexpr int WAVE = 64;
constexpr int BLOCK_NNZ = 2048;      // products staged in LDS per workgroup (16 KiB)
constexpr int UNROLL = BLOCK_NNZ / TPB;
constexpr int MAX_ROWS_PER_BLOCK = 4 * TPB;
constexpr int LONG_CHUNK = 8192;     // nnz per workgroup for rows longer than BLOCK_NNZ
constexpr int NUM_XCD = 8;
constexpr int EW_MAX_BLOCKS = 256 * 8;  // elementwise kernels: grid-stride above this
constexpr int FINAL_TPB = 1024;
// tiled-sweep layout (SpMV v2)
constexpr int TW_WPB = 8;              // waves per workgroup (512 threads), 2 workgroups per CU
constexpr int TW_MAX_ROWS = 1272;      // rows owned by one wave: 2 x 8 x 1272 x 8 B fits the 160 KiB LDS
constexpr unsigned TW_PAD = 0xFFFFFFFFu;
constexpr int TW_U = 3;               // 64-entry chunks prefetched per wave per tile

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);      \
      return (int)_e > 0 ? (int)_e : 999;                                    \
    }                                                                        \
  } while (0)

// ---------------------------------------------------------------- device utils

// Julia's max/min on Float64 for non-NaN inputs, including signed zeros
// (saddle_point.jl:88-91, :115 use min(ub, max(lb, v)) and max(y, 0.0)).
__device__ __forceinline__ double jl_max(double a, double b) {
  return (a > b) ? a : ((b > a) ? b : (signbit(a) ? b : a));
}
__device__ __forceinline__ double jl_min(double a, double b) {
  return (a < b) ? a : ((b < a) ? b : (signbit(a) ? a : b));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

// Deterministic block reduction of up to 3 per-thread accumulators; thread 0
// of the block returns the totals in acc[].  `red` is LDS [3][TPB/WAVE].
template <int NQ, int THREADS>
__device__ __forceinline__ void block_sum(double (&acc)[3],
                                          double (*red)[THREADS / WAVE]) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double w = wave_sum(acc[q]);
    if (lane == 0) red[q][wid] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < THREADS / WAVE; ++w) s += red[q][w];
      acc[q] = s;
    }
  }
}

// ---------------------------------------------------------------- CSR views

struct CsrView {
  int rows;
  const int *rowptr;   // [rows+1]
  const int *col;      // [nnz]
  const double *val;   // [nnz]
};

enum { MODE_PLAIN = 0, MODE_DUAL = 1, MODE_ATY = 2 };

// Everything a row epilogue may touch.  Passed by value to the kernels.
struct EpiArgs {
  // MODE_PLAIN
  double *out;
  // MODE_DUAL: y' = proj(y + sigma*(b - A xbar)); partial sum dy^2
  const double *y;
  const double *b;
  double *y_next;
  double sigma;
  int num_eq;
  // MODE_ATY: A'y' written; partial dx.(A'y'-A'y), dx^2, (A'y'-A'y)^2
  const double *x;
  const double *x_next;
  const double *aty;
  double *aty_next;
  // block partials: partials[q*stride + slot]
  double *partials;
  int stride;
};

template <int MODE>
__device__ __forceinline__ void row_epilogue(const EpiArgs &e, int r, double s,
                                             double (&acc)[3]) {
  if (MODE == MODE_PLAIN) {
    e.out[r] = s;
  } else if (MODE == MODE_DUAL) {
    // compute_dual_gradient: b .- A*x              saddle_point.jl:1102-1107
    const double yo = e.y[r];
    const double dg = e.b[r] - s;
    // next_dual = y .+ (pw*step) .* dual_gradient   pdhg.jl:489-490
    const double t = e.sigma * dg;
    double yn = yo + t;
    // project_dual!: only inequality rows           saddle_point.jl:110-117
    if (r >= e.num_eq) yn = jl_max(yn, 0.0);
    e.y_next[r] = yn;
    const double dy = yn - yo;                       // pdhg.jl:535
    acc[0] += dy * dy;
  } else {
    // next_dual_product = A' * next_dual            pdhg.jl:492
    e.aty_next[r] = s;
    const double dx = e.x_next[r] - e.x[r];          // pdhg.jl:534
    const double dd = s - e.aty[r];                  // pdhg.jl:543
    acc[0] += dx * dd;
    acc[1] += dx * dx;
    acc[2] += dd * dd;
  }
}

template <int MODE>
struct ModeNQ { static constexpr int value = (MODE == MODE_PLAIN) ? 0 : (MODE == MODE_DUAL ? 1 : 3); };

// CSR "stream" kernel: a workgroup owns a run of consecutive rows holding at
// most BLOCK_NNZ nonzeros.  Phase 1 streams val/col with fully coalesced
// loads (UNROLL independent load chains per lane for memory-level
// parallelism), gathers x and parks the products in LDS.  Phase 2: one lane
// per row adds that row's products in ascending column order -- the same
// order as Julia's SparseMatrixCSC A*x / A'*y loops, so short rows are
// bit-identical to the sequential CPU result -- and applies the fused
// epilogue.  Block->XCD: hardware places block b on XCD b%8; with `remap`
// each XCD walks a contiguous eighth of the row blocks so its private 4 MiB
// L2 sees a contiguous slice of the gathered vector for banded/local
// matrices.
template <int MODE>
__global__ __launch_bounds__(TPB) void spmv_stream_kernel(
    CsrView A, const double *__restrict__ xin, const int2 *__restrict__ blks,
    int nblk, int per_xcd, int remap, EpiArgs e) {
  __shared__ double prod[BLOCK_NNZ];
  __shared__ double red[3][TPB / WAVE];
  const int b = blockIdx.x;
  const int blk = remap ? ((b & (NUM_XCD - 1)) * per_xcd + (b >> 3)) : b;
  double acc[3] = {0.0, 0.0, 0.0};
  const bool active = remap ? ((b >> 3) < per_xcd && blk < nblk) : (blk < nblk);
  if (active) {
    const int2 rr = blks[blk];
    const int r0 = rr.x, r1 = rr.y;
    const int k0 = A.rowptr[r0];
    const int k1 = A.rowptr[r1];
    const int tid = threadIdx.x;
    int cidx[UNROLL];
    double v[UNROLL];
    double xv[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      const int k = k0 + tid + i * TPB;
      const bool ok = k < k1;
      cidx[i] = ok ? __builtin_nontemporal_load(A.col + k) : 0;
      v[i] = ok ? __builtin_nontemporal_load(A.val + k) : 0.0;
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      const int k = k0 + tid + i * TPB;
      xv[i] = (k < k1) ? xin[cidx[i]] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      const int k = tid + i * TPB;
      if (k0 + k < k1) prod[k] = v[i] * xv[i];
    }
    __syncthreads();
    for (int r = r0 + tid; r < r1; r += TPB) {
      const int ks = A.rowptr[r] - k0;
      const int ke = A.rowptr[r + 1] - k0;
      double s = 0.0;
      int k = ks;
      // 8 LDS reads in flight, adds still strictly left to right (bit-exact
      // order); a row of ~2K products is otherwise one LDS latency per add.
      for (; k + 8 <= ke; k += 8) {
        const double t0 = prod[k], t1 = prod[k + 1], t2 = prod[k + 2], t3 = prod[k + 3];
        const double t4 = prod[k + 4], t5 = prod[k + 5], t6 = prod[k + 6], t7 = prod[k + 7];
        s = s + t0; s = s + t1; s = s + t2; s = s + t3;
        s = s + t4; s = s + t5; s = s + t6; s = s + t7;
      }
      for (; k < ke; ++k) s = s + prod[k];
      row_epilogue<MODE>(e, r, s, acc);
    }
  }
  constexpr int NQ = ModeNQ<MODE>::value;
  if (NQ > 0) {
    block_sum<NQ, TPB>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) e.partials[q * e.stride + b] = acc[q];
    }
  }
}

// CSR "tiled sweep" kernel (SpMV v2) for matrices whose gathered vector is far
// larger than the 4 MiB per-XCD L2.  Measured on MI355X (tools/gather_probe):
// a uniformly random 8-byte gather tops out at ~56 G/s over an 80 MB vector but
// reaches ~190-245 G/s when the window fits L2.
// Layout: every wave owns up to TW_ROWS consecutive rows and streams ITS
// nonzeros, pre-sorted on the host by (column tile, row, column) and packed
// tile-locally as {row_local << tile_shift | col_local} + value, with one
// offset per (wave, tile).  A workgroup is 8 such waves; all of them process
// tile t, then meet at a barrier -- the barrier is pacing, not correctness: it
// keeps the 16 waves of a CU (and, statistically, the CUs of an XCD) inside
// the same ~1 MiB slice of the gathered vector, which therefore stays in L2.
// The next tile's entries are prefetched into registers before the barrier.
// Accumulators live in the wave's private LDS slice; one wave's DS operations
// execute in order, each row receives its products in ascending column order
// (tile-major order preserves it) => bit-identical to the sequential CPU
// loops.  Entries of one row inside a tile are adjacent; the run head adds
// them left to right via lane shuffles.
__device__ __forceinline__ void tiled_chunk(double *acc, unsigned p, double v, double xv,
                                            int tile_shift, int lane) {
  const bool valid = p != TW_PAD;
  const unsigned row = valid ? (p >> tile_shift) : 0xFFFFFFFFu;
  const double prod = v * xv;
  const unsigned rowp = __shfl_up(row, 1, WAVE);
  const bool head = valid && (lane == 0 || rowp != row);
  double s = head ? acc[row] : 0.0;
  for (int j = 0; j < WAVE; ++j) {
    const double pj = __shfl_down(prod, j, WAVE);
    const unsigned rj = __shfl_down(row, j, WAVE);
    const bool take = head && (lane + j < WAVE) && (rj == row);
    if (!__any(take)) break;
    if (take) s = s + pj;
  }
  if (head) acc[row] = s;
}

// Variant for matrices with long same-row runs inside a tile (rows with hundreds
// of entries): run lengths from two ballots, followers' products handed to the
// run head through a 64-double LDS scratch per wave and added left to right
// (same order as above; ~6x cheaper than lane shuffles for a 64-long run).
__device__ __forceinline__ void tiled_chunk_scratch(double *acc, double *scratch, unsigned p, double v,
                                                    double xv, int tile_shift, int lane) {
  const bool valid = p != TW_PAD;
  const unsigned row = valid ? (p >> tile_shift) : 0xFFFFFFFFu;
  const double prod = v * xv;
  const unsigned rowp = __shfl_up(row, 1, WAVE);
  const bool head = valid && (lane == 0 || rowp != row);
  const unsigned long long hmask = __ballot(head);
  const unsigned long long vmask = __ballot(valid);
  const unsigned long long above = (lane == WAVE - 1) ? 0ull : (hmask >> (lane + 1));
  const int nvalid = __popcll(vmask);                                   // valid lanes are a prefix
  const int len = above ? __ffsll((long long)above) : (nvalid - lane);  // run length (heads only)
  if (__any(head && len > 1)) scratch[lane] = prod;
  if (head) {
    double s = acc[row] + prod;
    int q = 1;
    for (; q + 4 <= len; q += 4) {
      const double t0 = scratch[lane + q], t1 = scratch[lane + q + 1];
      const double t2 = scratch[lane + q + 2], t3 = scratch[lane + q + 3];
      s = s + t0; s = s + t1; s = s + t2; s = s + t3;
    }
    for (; q < len; ++q) s = s + scratch[lane + q];
    acc[row] = s;
  }
}

template <int MODE, bool SCR>
__global__ __launch_bounds__(TW_WPB * WAVE) void spmv_tiled_kernel(
    const int2 *__restrict__ wave_rows, const int *__restrict__ step_ptr,
    const int *__restrict__ wave_step_off, const int *__restrict__ step_tile,
    const int *__restrict__ wg_step_off, int nwaves, int tile_shift, int TW_ROWS,
    const unsigned *__restrict__ pk, const double *__restrict__ tv,
    const double *__restrict__ xin, EpiArgs e) {
  constexpr int TW_THREADS = TW_WPB * WAVE;
  constexpr int U = TW_U;   // 64-entry chunks held in registers per (wave, tile)
  constexpr int D = 3;      // entry loads run D tiles ahead of the accumulate
  constexpr int R = D + 1;  // register ring (statically indexed: the tile loop is unrolled R times)
  extern __shared__ double tw_lds[];  // [TW_WPB][TW_ROWS] accumulators, then red[3][TW_WPB]
  double(*red)[TW_WPB] = reinterpret_cast<double(*)[TW_WPB]>(tw_lds + TW_WPB * TW_ROWS);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
  double *scratch = tw_lds + TW_WPB * TW_ROWS + 3 * TW_WPB + wid * WAVE;   // SCR only: 64 doubles per wave
  const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * TW_WPB + wid);
  const bool live = w < nwaves;
  double acc3[3] = {0.0, 0.0, 0.0};
  double *acc = tw_lds + wid * TW_ROWS;
  int2 rr = make_int2(0, 0);
  if (live) rr = wave_rows[w];
  for (int r = lane; r < TW_ROWS; r += WAVE) acc[r] = 0.0;
  // This workgroup's step list: one step = one column tile, or a slice of a
  // heavy tile (cells are cut on the host so that no wave has more than
  // TW_U*64 entries in a step); tiles in which none of the 8 waves has an entry
  // are skipped.  ntiles below is the number of STEPS of this workgroup.
  const int ntiles = wg_step_off[blockIdx.x + 1] - wg_step_off[blockIdx.x];
  const int *stile = step_tile + wg_step_off[blockIdx.x];
  const int *tp = step_ptr + (live ? wave_step_off[w] : 0);
  const unsigned cmask = (1u << tile_shift) - 1u;

  unsigned p[R][U];
  double v[R][U];
  double xv[U];
  int ks[R], ke[R], tl[R];

  auto load_set = [&](unsigned(&pp)[U], double(&vv)[U], int kbeg, int kend) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int k = kbeg + i * WAVE + lane;
      const bool ok = k < kend;
      pp[i] = ok ? __builtin_nontemporal_load(pk + k) : TW_PAD;
      vv[i] = ok ? __builtin_nontemporal_load(tv + k) : 0.0;
    }
  };

  // prologue: entries of tiles 0..D-1
  int kprev = live ? tp[0] : 0;
#pragma unroll
  for (int s = 0; s < D; ++s) {
    ks[s] = kprev;
    ke[s] = (live && s < ntiles) ? tp[s + 1] : kprev;
    tl[s] = (s < ntiles) ? stile[s] : 0;
    kprev = ke[s];
    load_set(p[s], v[s], ks[s], ke[s]);
  }
  ks[D] = ke[D] = kprev;
  tl[D] = 0;
#pragma unroll
  for (int i = 0; i < U; ++i) { p[D][i] = TW_PAD; v[D][i] = 0.0; }
  int ke_ahead = (live && D < ntiles) ? tp[D + 1] : kprev;  // end of tile D

  for (int t0 = 0; t0 < ntiles; t0 += R) {
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const int t = t0 + s;
      if (t < ntiles) {  // workgroup-uniform
        const int f = (s + D) % R;       // ring slot being refilled (held tile t-1)
#ifdef TWD_ONE_TILE   // timing diagnostic, wrong results (tools/variants.sh): every gather hits tile 0
        const double *xt = xin;
#else
        const double *xt = xin + ((size_t)tl[s] << tile_shift);
#endif
        // 1. gathers for tile t (entries requested D steps ago).  Issued BEFORE
        //    the prefetch: a wave's loads return in order, so the L2-latency
        //    gathers must not queue behind HBM-latency streaming loads.
        //    (Gathering one tile ahead was measured slower: it widens the L2
        //    working window of the sweep.)
#pragma unroll
#ifdef TWD_NO_GATHER  // timing diagnostic, wrong results: stream + accumulate only
        for (int i = 0; i < U; ++i) xv[i] = 1.0 + (double)(size_t)xt * 0.0;
#else
        for (int i = 0; i < U; ++i) xv[i] = (p[s][i] != TW_PAD) ? xt[p[s][i] & cmask] : 0.0;
#endif
        // 2. entry loads for tile t+D
        ks[f] = ke[(s + D - 1) % R];
        ke[f] = ke_ahead;
        tl[f] = (t + D < ntiles) ? stile[t + D] : 0;
        load_set(p[f], v[f], ks[f], ke[f]);
        ke_ahead = (live && t + D + 1 < ntiles) ? tp[t + D + 2] : ke_ahead;  // end of tile t+D+1
        // 3. accumulate tile t
#pragma unroll
        for (int i = 0; i < U; ++i) {
          if (ks[s] + i * WAVE < ke[s]) {  // wave-uniform
            if (SCR) tiled_chunk_scratch(acc, scratch, p[s][i], v[s][i], xv[i], tile_shift, lane);
            else tiled_chunk(acc, p[s][i], v[s][i], xv[i], tile_shift, lane);
          }
        }
        for (int kb = ks[s] + U * WAVE; kb < ke[s]; kb += WAVE) {  // cells beyond the register window
          const int k = kb + lane;
          const bool ok = k < ke[s];
          const unsigned pp = ok ? __builtin_nontemporal_load(pk + k) : TW_PAD;
          const double vv = ok ? __builtin_nontemporal_load(tv + k) : 0.0;
          const double xx = ok ? xt[pp & cmask] : 0.0;
          if (SCR) tiled_chunk_scratch(acc, scratch, pp, vv, xx, tile_shift, lane);
          else tiled_chunk(acc, pp, vv, xx, tile_shift, lane);
        }
        // 4. pacing barrier: keep the workgroup inside one column tile
        //    (without it the kernel is 1.7x slower: waves drift apart and the
        //    gathers stop hitting L2)
        __syncthreads();
      }
    }
  }
  const int nrows = rr.y - rr.x;
  for (int r = lane; r < nrows; r += WAVE) row_epilogue<MODE>(e, rr.x + r, acc[r], acc3);
  constexpr int NQ = ModeNQ<MODE>::value;
  if (NQ > 0) {
    block_sum<NQ, TW_THREADS>(acc3, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) e.partials[q * e.stride + blockIdx.x] = acc3[q];
    }
  }
}

// Rows longer than BLOCK_NNZ: split into LONG_CHUNK pieces, one workgroup
// each (tree sum inside the chunk), partial per chunk.
__global__ __launch_bounds__(TPB) void spmv_long_partial_kernel(
    CsrView A, const double *__restrict__ xin, const int *__restrict__ chunk_row,
    const int *__restrict__ chunk_off, double *__restrict__ chunk_partial) {
  __shared__ double red[3][TPB / WAVE];
  const int c = blockIdx.x;
  const int r = chunk_row[c];
  const int kbeg = A.rowptr[r] + chunk_off[c];
  const int kend = min(kbeg + LONG_CHUNK, A.rowptr[r + 1]);
  double acc[3] = {0.0, 0.0, 0.0};
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = kbeg + threadIdx.x;
  for (; k + 3 * TPB < kend; k += 4 * TPB) {
    const int c0 = __builtin_nontemporal_load(A.col + k);
    const int c1 = __builtin_nontemporal_load(A.col + k + TPB);
    const int c2 = __builtin_nontemporal_load(A.col + k + 2 * TPB);
    const int c3 = __builtin_nontemporal_load(A.col + k + 3 * TPB);
    const double v0 = __builtin_nontemporal_load(A.val + k);
    const double v1 = __builtin_nontemporal_load(A.val + k + TPB);
    const double v2 = __builtin_nontemporal_load(A.val + k + 2 * TPB);
    const double v3 = __builtin_nontemporal_load(A.val + k + 3 * TPB);
    s0 += v0 * xin[c0];
    s1 += v1 * xin[c1];
    s2 += v2 * xin[c2];
    s3 += v3 * xin[c3];
  }
  for (; k < kend; k += TPB) s0 += A.val[k] * xin[A.col[k]];
  acc[0] = (s0 + s1) + (s2 + s3);
  block_sum<1, TPB>(acc, red);
  if (threadIdx.x == 0) chunk_partial[c] = acc[0];
}

// One lane per long row: add the chunk partials in order, run the epilogue.
template <int MODE>
__global__ __launch_bounds__(TPB) void spmv_long_final_kernel(
    const int *__restrict__ long_row, const int *__restrict__ long_chunk_ptr,
    int nlong, const double *__restrict__ chunk_partial, EpiArgs e,
    int slot_base) {
  __shared__ double red[3][TPB / WAVE];
  double acc[3] = {0.0, 0.0, 0.0};
  const int l = blockIdx.x * TPB + threadIdx.x;
  if (l < nlong) {
    double s = 0.0;
    for (int c = long_chunk_ptr[l]; c < long_chunk_ptr[l + 1]; ++c)
      s = s + chunk_partial[c];
    row_epilogue<MODE>(e, long_row[l], s, acc);
  }
  constexpr int NQ = ModeNQ<MODE>::value;
  if (NQ > 0) {
    block_sum<NQ, TPB>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        e.partials[q * e.stride + slot_base + blockIdx.x] = acc[q];
    }
  }
}

// ---------------------------------------------------------------- elementwise

// K1+K2: x' = proj(x - tau*(Qx + c - A'y)), xbar = x' + theta*(x' - x).
//   compute_primal_gradient_from_dual_product  saddle_point.jl:1093-1100
//   next_primal = x .- (step/pw) .* g          pdhg.jl:466-467
//   projection!                                saddle_point.jl:87-92
//   xbar                                       pdhg.jl:486-487
template <bool HAS_Q, bool WRITE_XBAR>
__device__ __forceinline__ void primal_one(double x, double c, double aty,
                                           double qx, double lb, double ub,
                                           double tau, double theta, double &xn,
                                           double &xb) {
  const double q = HAS_Q ? qx : 0.0;
  const double t0 = q + c;
  const double g = t0 - aty;
  const double t1 = tau * g;
  double v = x - t1;
  v = jl_min(ub, jl_max(lb, v));
  xn = v;
  if (WRITE_XBAR) {
    const double d = v - x;
    const double t2 = theta * d;
    xb = v + t2;
  }
}

template <bool HAS_Q, bool WRITE_XBAR>
__global__ __launch_bounds__(TPB) void primal_kernel(
    int n, const double *__restrict__ x, const double *__restrict__ c,
    const double *__restrict__ aty, const double *__restrict__ qx,
    const double *__restrict__ lb, const double *__restrict__ ub, double tau,
    double theta, double *__restrict__ x_next, double *__restrict__ xbar) {
  const int npair = n >> 1;
  const int stride = gridDim.x * TPB;
  for (int p = blockIdx.x * TPB + threadIdx.x; p < npair; p += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[p];
    const double2 cv = reinterpret_cast<const double2 *>(c)[p];
    const double2 av = reinterpret_cast<const double2 *>(aty)[p];
    const double2 lv = reinterpret_cast<const double2 *>(lb)[p];
    const double2 uv = reinterpret_cast<const double2 *>(ub)[p];
    double2 qv = {0.0, 0.0};
    if (HAS_Q) qv = reinterpret_cast<const double2 *>(qx)[p];
    double2 xn, xb;
    primal_one<HAS_Q, WRITE_XBAR>(xv.x, cv.x, av.x, qv.x, lv.x, uv.x, tau, theta, xn.x, xb.x);
    primal_one<HAS_Q, WRITE_XBAR>(xv.y, cv.y, av.y, qv.y, lv.y, uv.y, tau, theta, xn.y, xb.y);
    reinterpret_cast<double2 *>(x_next)[p] = xn;
    if (WRITE_XBAR) reinterpret_cast<double2 *>(xbar)[p] = xb;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int j = n - 1;
    double xn, xb;
    primal_one<HAS_Q, WRITE_XBAR>(x[j], c[j], aty[j], HAS_Q ? qx[j] : 0.0, lb[j], ub[j], tau, theta, xn, xb);
    x_next[j] = xn;
    if (WRITE_XBAR) xbar[j] = xb;
  }
}

// xbar = x' + theta*(x' - x) on its own (Malitsky-Pock retries, pdhg.jl:590-601)
__global__ __launch_bounds__(TPB) void xbar_kernel(int n, const double *__restrict__ x,
                                                   const double *__restrict__ x_next,
                                                   double theta, double *__restrict__ xbar) {
  const int stride = gridDim.x * TPB;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += stride) {
    const double v = x_next[j];
    const double d = v - x[j];
    const double t = theta * d;
    xbar[j] = v + t;
  }
}

// dx = x' - x  (for the QP interaction term 0.5*dx'Q dx, pdhg.jl:536-541)
__global__ __launch_bounds__(TPB) void diff_kernel(int n, const double *__restrict__ a,
                                                   const double *__restrict__ b,
                                                   double *__restrict__ out) {
  const int stride = gridDim.x * TPB;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += stride) out[j] = a[j] - b[j];
}

// Reductions over the replicated n-vectors (row-partitioned form, after the
// all-reduce delivered A'y'):  dx.(A'y'-A'y), dx^2, (A'y'-A'y)^2.
__global__ __launch_bounds__(TPB) void interaction_kernel(
    int n, const double *__restrict__ x, const double *__restrict__ x_next,
    const double *__restrict__ aty, const double *__restrict__ aty_next,
    double *__restrict__ partials, int pstride) {
  __shared__ double red[3][TPB / WAVE];
  double acc[3] = {0.0, 0.0, 0.0};
  const int stride = gridDim.x * TPB;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += stride) {
    const double dx = x_next[j] - x[j];
    const double dd = aty_next[j] - aty[j];
    acc[0] += dx * dd;
    acc[1] += dx * dx;
    acc[2] += dd * dd;
  }
  block_sum<3, TPB>(acc, red);
  if (threadIdx.x == 0) {
    partials[0 * pstride + blockIdx.x] = acc[0];
    partials[1 * pstride + blockIdx.x] = acc[1];
    partials[2 * pstride + blockIdx.x] = acc[2];
  }
}

// dot(a, b) partials (QP term)
__global__ __launch_bounds__(TPB) void dot_kernel(int n, const double *__restrict__ a,
                                                  const double *__restrict__ b,
                                                  double *__restrict__ partials) {
  __shared__ double red[3][TPB / WAVE];
  double acc[3] = {0.0, 0.0, 0.0};
  const int stride = gridDim.x * TPB;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += stride) acc[0] += a[j] * b[j];
  block_sum<1, TPB>(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// K7: sum_x += w*x', sum_y += w*y'      saddle_point.jl:258-259, 271
__global__ __launch_bounds__(TPB) void accept_kernel(int n, int m, double w,
                                                     const double *__restrict__ xs,
                                                     double *__restrict__ sum_x,
                                                     const double *__restrict__ ys,
                                                     double *__restrict__ sum_y) {
  const int stride = gridDim.x * TPB;
  const int tid = blockIdx.x * TPB + threadIdx.x;
  for (int j = tid; j < n; j += stride) {
    const double t = xs[j] * w;
    sum_x[j] = sum_x[j] + t;
  }
  for (int i = tid; i < m; i += stride) {
    const double t = ys[i] * w;
    sum_y[i] = sum_y[i] + t;
  }
}

// compute_average: sum / weight (a division, saddle_point.jl:296-301)
__global__ __launch_bounds__(TPB) void div_kernel(int n, const double *__restrict__ s,
                                                  double w, double *__restrict__ out) {
  const int stride = gridDim.x * TPB;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += stride) out[j] = s[j] / w;
}

// Second-stage, fixed-order sum of the block partials.  One workgroup.
// spec[q] = {ptr, count}; out[q] = sum(ptr[0..count)).  count==0 -> 0.
struct FinalSpec {
  const double *ptr[5];
  int count[5];
  double *out;     // 5 doubles (host-mapped or device)
};
__global__ __launch_bounds__(FINAL_TPB) void final_reduce_kernel(FinalSpec sp) {
  __shared__ double red[3][FINAL_TPB / WAVE];
  for (int q = 0; q < 5; ++q) {
    double acc[3] = {0.0, 0.0, 0.0};
    const double *p = sp.ptr[q];
    const int cnt = sp.count[q];
    for (int i = threadIdx.x; i < cnt; i += FINAL_TPB) acc[0] += p[i];
    block_sum<1, FINAL_TPB>(acc, red);
    if (threadIdx.x == 0) sp.out[q] = acc[0];
    __syncthreads();
  }
}

// ============================================================ evaluation branch (N1)
// Evaluation-cadence kernels (every termination_evaluation_frequency
// iterations): plain SpMVs into temporaries followed by elementwise kernels
// with multi-quantity sum/max reductions.  Simplicity over fusion here: the
// extra vector passes are noise at this cadence.
constexpr int EV_MAXQ = 20;

template <int NS, int NM>
struct RedAcc {
  double s[NS > 0 ? NS : 1];
  double m[NM > 0 ? NM : 1];
  __device__ RedAcc() {
    for (int i = 0; i < (NS > 0 ? NS : 1); ++i) s[i] = 0.0;
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) m[i] = 0.0;
  }
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, WAVE));
  return v;
}

// partials[q*stride + blockIdx.x]: q < NS sums, then NM maxes (all maxes are of non-negative values)
template <int NS, int NM>
__device__ __forceinline__ void block_reduce_store(const RedAcc<NS, NM> &a, double *partials, int stride) {
  __shared__ double red[NS + NM][TPB / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < NS; ++q) { const double w = wave_sum(a.s[q]); if (lane == 0) red[q][wid] = w; }
#pragma unroll
  for (int q = 0; q < NM; ++q) { const double w = wave_max(a.m[q]); if (lane == 0) red[NS + q][wid] = w; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NS; ++q) { double t = 0.0; for (int w = 0; w < TPB / WAVE; ++w) t += red[q][w]; partials[q * stride + blockIdx.x] = t; }
#pragma unroll
    for (int q = 0; q < NM; ++q) { double t = 0.0; for (int w = 0; w < TPB / WAVE; ++w) t = fmax(t, red[NS + q][w]); partials[(NS + q) * stride + blockIdx.x] = t; }
  }
}

__global__ __launch_bounds__(FINAL_TPB) void multi_final_kernel(const double *__restrict__ partials, int stride,
                                                                int count, int ns, int nm, double *__restrict__ out) {
  __shared__ double red[3][FINAL_TPB / WAVE];
  for (int q = 0; q < ns + nm; ++q) {
    const double *p = partials + (size_t)q * stride;
    const bool is_max = q >= ns;
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += FINAL_TPB) v = is_max ? fmax(v, p[i]) : v + p[i];
    v = is_max ? wave_max(v) : wave_sum(v);
    if ((threadIdx.x & (WAVE - 1)) == 0) red[0][threadIdx.x / WAVE] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < FINAL_TPB / WAVE; ++w) t = is_max ? fmax(t, red[0][w]) : t + red[0][w];
      out[q] = t;
    }
    __syncthreads();
  }
}

// Row side of compute_convergence_information / compute_infeasibility_information
// (iteration_stats_utils.jl:30-63, 157-197, 228-349) on the UNSCALED point:
//   activities A_o x_o = E .* (A_s x_s),  y_o = y_s ./ E.
// sums: 0 sum viol^2, 1 sum y_o^2, 2 b_o.y_o, 3 sum max(-y_o,0)^2 (ineq rows)
// maxs: 0 max|viol|, 1 max|viol_homogeneous|, 2 max|y_o|, 3 max max(-y_o,0)
__global__ __launch_bounds__(TPB) void eval_rows_kernel(int m, int ne, const double *__restrict__ ax_s,
                                                        const double *__restrict__ py, const double *__restrict__ E,
                                                        const double *__restrict__ b_o, double *__restrict__ partials,
                                                        int stride) {
  RedAcc<4, 4> a;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < m; i += gridDim.x * TPB) {
    const double e = E[i];
    const double act = e * ax_s[i];
    const double r = b_o[i] - act;
    const double rh = 0.0 - act;
    const bool eq = i < ne;
    const double viol = eq ? r : fmax(r, 0.0);
    const double violh = eq ? rh : fmax(rh, 0.0);
    const double yo = py[i] / e;
    const double dres = eq ? 0.0 : fmax(-yo, 0.0);
    a.s[0] += viol * viol; a.s[1] += yo * yo; a.s[2] += b_o[i] * yo; a.s[3] += dres * dres;
    a.m[0] = fmax(a.m[0], fabs(viol)); a.m[1] = fmax(a.m[1], fabs(violh));
    a.m[2] = fmax(a.m[2], fabs(yo)); a.m[3] = fmax(a.m[3], dres);
  }
  block_reduce_store<4, 4>(a, partials, stride);
}

// Column side (LP): g = c_o - D .* (A_s' y_s), reduced costs, bound violations,
// and the homogeneous (c = 0) dual statistics for the infeasibility certificate.
// sums: 0 sum resid^2, 1 sum bound*rc, 2 sum x_o^2, 3 c_o.x_o, 4 sum bound-viol^2, 5 sum bound*rc_h
// maxs: 0 max|resid|, 1 max|x_o|, 2 max bound viol, 3 max|resid_h|, 4 max|rc_h|, 5 max ray bound viol
__global__ __launch_bounds__(TPB) void eval_cols_kernel(int n, const double *__restrict__ aty_s,
                                                        const double *__restrict__ px, const double *__restrict__ D,
                                                        const double *__restrict__ c_o, const double *__restrict__ lb_o,
                                                        const double *__restrict__ ub_o, double *__restrict__ partials,
                                                        int stride) {
  RedAcc<6, 6> a;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += gridDim.x * TPB) {
    const double d = D[j];
    const double aty = d * aty_s[j];
    const double xo = px[j] / d;
    const double lb = lb_o[j], ub = ub_o[j];
    const bool lbf = isfinite(lb), ubf = isfinite(ub);
    // compute_reduced_costs_from_primal_gradient        iteration_stats_utils.jl:128-148
    const double g = c_o[j] - aty;
    const double rc = ((g > 0.0) ? lbf : ubf) ? g : 0.0;
    const double resid = g - rc;
    const double contrib = (rc == 0.0) ? 0.0 : ((rc > 0.0 ? lb : ub) * rc);
    const double gh = 0.0 - aty;
    const double rch = ((gh > 0.0) ? lbf : ubf) ? gh : 0.0;
    const double residh = gh - rch;
    const double contribh = (rch == 0.0) ? 0.0 : ((rch > 0.0 ? lb : ub) * rch);
    const double lv = fmax(lb - xo, 0.0), uv = fmax(xo - ub, 0.0);
    const double rayv = fmax(lbf ? fmax(-xo, 0.0) : 0.0, ubf ? fmax(xo, 0.0) : 0.0);
    a.s[0] += resid * resid; a.s[1] += contrib; a.s[2] += xo * xo; a.s[3] += c_o[j] * xo;
    a.s[4] += lv * lv + uv * uv; a.s[5] += contribh;
    a.m[0] = fmax(a.m[0], fabs(resid)); a.m[1] = fmax(a.m[1], fabs(xo)); a.m[2] = fmax(a.m[2], fmax(lv, uv));
    a.m[3] = fmax(a.m[3], fabs(residh)); a.m[4] = fmax(a.m[4], fabs(rch)); a.m[5] = fmax(a.m[5], rayv);
  }
  block_reduce_store<6, 6>(a, partials, stride);
}

// sum (a-b)^2 over two vector pairs: distances to the last restart point
// (saddle_point.jl:445-477, 911-920; weights are uniform per block in PDHG).
__global__ __launch_bounds__(TPB) void dist2_kernel(int n, int m, const double *__restrict__ xa,
                                                    const double *__restrict__ xb, const double *__restrict__ ya,
                                                    const double *__restrict__ yb, double *__restrict__ partials,
                                                    int stride) {
  RedAcc<2, 0> a;
  const int tid = blockIdx.x * TPB + threadIdx.x, st = gridDim.x * TPB;
  for (int j = tid; j < n; j += st) { const double d = xb ? xa[j] - xb[j] : xa[j]; a.s[0] += d * d; }
  for (int i = tid; i < m; i += st) { const double d = yb ? ya[i] - yb[i] : ya[i]; a.s[1] += d * d; }
  block_reduce_store<2, 0>(a, partials, stride);
}

// bound_optimal_objective (trust_region_utils.jl:271-360) set-up on the SCALED
// problem at point z = (x, y): gradient g = [c - A'y ; -(b - A x)], direction
// d = -g/w (0 if the bound blocks it), breakpoint thr (trust_region_utils.jl:86-110).
// range: 0 both blocks (EUCLIDEAN_NORM), 1 primal only, 2 dual only (MAX_NORM halves).
// sums: 0 c.x, 1 x.(A'y), 2 y.b, 3 sum_{thr=inf} w d^2, 4 sum g^2 (in range),
//       5 sum w d^2 (in range), 6 sum g.d primal, 7 sum g.d dual, 8 sum x^2, 9 sum y^2
// maxs: 0 max finite thr (in range)
__global__ __launch_bounds__(TPB) void tr_setup_kernel(int n, int m, int ne, const double *__restrict__ px,
                                                       const double *__restrict__ py, const double *__restrict__ aty_s,
                                                       const double *__restrict__ ax_s, const double *__restrict__ c_s,
                                                       const double *__restrict__ b_s, const double *__restrict__ lb_s,
                                                       const double *__restrict__ ub_s, double wp, double wd, int range,
                                                       double *__restrict__ gvec, double *__restrict__ dir,
                                                       double *__restrict__ thr, double *__restrict__ partials,
                                                       int stride) {
  RedAcc<10, 1> a;
  const int tid = blockIdx.x * TPB + threadIdx.x, st = gridDim.x * TPB;
  for (int k = tid; k < n + m; k += st) {
    const bool primal = k < n;
    const int i = primal ? k : k - n;
    double z, g, lo, hi, w;
    if (primal) {
      z = px[i]; g = c_s[i] - aty_s[i]; lo = lb_s[i]; hi = ub_s[i]; w = wp;
      a.s[0] += c_s[i] * z; a.s[1] += z * aty_s[i]; a.s[8] += z * z;
    } else {
      z = py[i]; g = -(b_s[i] - ax_s[i]); lo = (i < ne) ? -INFINITY : 0.0; hi = INFINITY; w = wd;
      a.s[2] += z * b_s[i]; a.s[9] += z * z;
    }
    const bool in_range = (range == 0) || (range == 1 && primal) || (range == 2 && !primal);
    double d = 0.0, t = 0.0;
    if (in_range && !((z >= hi && g <= 0.0) || (z <= lo && g >= 0.0))) {
      d = -g / w;
      if (d > 0.0) t = (hi - z) / d;
      else if (d < 0.0) t = (lo - z) / d;
      else t = 0.0;
    }
    gvec[k] = g; dir[k] = d; thr[k] = t;
    if (in_range) {
      a.s[4] += g * g;
      a.s[5] += w * d * d;
      if (primal) a.s[6] += g * d; else a.s[7] += g * d;
      if (isinf(t)) a.s[3] += w * d * d; else a.m[0] = fmax(a.m[0], t);
    }
  }
  block_reduce_store<10, 1>(a, partials, stride);
}

// radius^2 as a function of the step t at K probe values:
//   low_k = sum_{thr <= t_k} w d^2 thr^2 ,  high_k = sum_{thr > t_k} w d^2
constexpr int TR_K = 7;
struct TrProbes { double t[TR_K]; };
__global__ __launch_bounds__(TPB) void tr_probe_kernel(int n, int total, const double *__restrict__ dir,
                                                       const double *__restrict__ thr, double wp, double wd,
                                                       TrProbes pr, double *__restrict__ partials, int stride) {
  RedAcc<2 * TR_K, 0> a;
  for (int k = blockIdx.x * TPB + threadIdx.x; k < total; k += gridDim.x * TPB) {
    const double d = dir[k];
    if (d == 0.0) continue;
    const double w = (k < n) ? wp : wd;
    const double t = thr[k];
    const double wd2 = w * d * d;
    const double lowc = wd2 * t * t;   // inf for thr = inf: never selected below
#pragma unroll
    for (int q = 0; q < TR_K; ++q) {
      if (t <= pr.t[q]) a.s[2 * q] += lowc; else a.s[2 * q + 1] += wd2;
    }
  }
  block_reduce_store<2 * TR_K, 0>(a, partials, stride);
}

// value parts sum g_i (clamp(z_i + t d_i) - z_i), primal block and dual block
__global__ __launch_bounds__(TPB) void tr_value_kernel(int n, int m, int ne, const double *__restrict__ px,
                                                       const double *__restrict__ py, const double *__restrict__ lb_s,
                                                       const double *__restrict__ ub_s, const double *__restrict__ gvec,
                                                       const double *__restrict__ dir, double t,
                                                       double *__restrict__ partials, int stride) {
  RedAcc<2, 0> a;
  const int tid = blockIdx.x * TPB + threadIdx.x, st = gridDim.x * TPB;
  for (int k = tid; k < n + m; k += st) {
    const bool primal = k < n;
    const int i = primal ? k : k - n;
    const double d = dir[k];
    if (d == 0.0) continue;
    const double z = primal ? px[i] : py[i];
    const double lo = primal ? lb_s[i] : ((i < ne) ? -INFINITY : 0.0);
    const double hi = primal ? ub_s[i] : INFINITY;
    const double cand = fmin(fmax(z + t * d, lo), hi);   // clamp.(center + t*direction, lb, ub)
    const double v = gvec[k] * (cand - z);
    if (primal) a.s[0] += v; else a.s[1] += v;
  }
  block_reduce_store<2, 0>(a, partials, stride);
}

// ============================================================ rescaling on the device (N2)
// rescale_problem (preprocess.jl:631-687) applied in place to every resident
// layout.  One wave per CSR row (one-time work, simplicity over speed).
enum { ROP_MAXABS = 0, ROP_SUMPOW = 1, ROP_SUMSQ_SCALED = 2 };

// out[r] = max |a| ; sum |a|^p (+ structural zeros when p == 0: Julia's
// mapreduce visits them and 0.0^0 == 1.0) ; sum (a * inv_scale[r])^2
template <int OP>
__global__ __launch_bounds__(TPB) void row_op_kernel(CsrView A, int cols, double pexp,
                                                     const double *__restrict__ inv_scale,
                                                     double *__restrict__ out) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int r = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
  if (r >= A.rows) return;
  const int k0 = A.rowptr[r], k1 = A.rowptr[r + 1];
  const double sc = (OP == ROP_SUMSQ_SCALED) ? inv_scale[r] : 1.0;
  double acc = 0.0;
  for (int k = k0 + lane; k < k1; k += WAVE) {
    const double a = A.val[k];
    if (OP == ROP_MAXABS) acc = fmax(acc, fabs(a));
    else if (OP == ROP_SUMPOW) acc += pow(fabs(a), pexp);
    else { const double t = a * sc; acc += t * t; }
  }
  acc = (OP == ROP_MAXABS) ? wave_max(acc) : wave_sum(acc);
  if (lane == 0) {
    if (OP == ROP_SUMPOW && pexp == 0.0) acc += (double)(cols - (k1 - k0));
    out[r] = acc;
  }
}

// val[k] = (val[k] * inv_a[ia]) * inv_b[ib] with ia/ib chosen so that the
// multiplication order is always (a * (1/e_row_of_A)) * (1/d_col_of_A), as in
// (Diagonal(1 ./ E) * A) * Diagonal(1 ./ D) (preprocess.jl:567-571).
// transposed == false: CSR(A) (row -> E, col -> D); true: CSR(A') (row -> D, col -> E).
__global__ __launch_bounds__(TPB) void scale_csr_kernel(int rows, const int *__restrict__ rowptr,
                                                        const int *__restrict__ col, double *__restrict__ val,
                                                        const double *__restrict__ inv_e,
                                                        const double *__restrict__ inv_d, int transposed) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int r = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
  if (r >= rows) return;
  const int k0 = rowptr[r], k1 = rowptr[r + 1];
  for (int k = k0 + lane; k < k1; k += WAVE) {
    const int c = col[k];
    const double ie = transposed ? inv_e[c] : inv_e[r];
    const double id = transposed ? inv_d[r] : inv_d[c];
    val[k] = (val[k] * ie) * id;
  }
}

// same for the tiled-sweep copy: one wave per wave-row-block, walking its steps
__global__ __launch_bounds__(TPB) void scale_tiled_kernel(const int2 *__restrict__ wave_rows,
                                                          const int *__restrict__ step_ptr,
                                                          const int *__restrict__ wave_step_off,
                                                          const int *__restrict__ step_tile,
                                                          const int *__restrict__ wg_step_off, int nwaves,
                                                          int tile_shift, const unsigned *__restrict__ pk,
                                                          double *__restrict__ tv,
                                                          const double *__restrict__ inv_e,
                                                          const double *__restrict__ inv_d, int transposed) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int w = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
  if (w >= nwaves) return;
  const int g = w / TW_WPB;
  const int nst = wg_step_off[g + 1] - wg_step_off[g];
  const int *stile = step_tile + wg_step_off[g];
  const int *sp = step_ptr + wave_step_off[w];
  const int r0 = wave_rows[w].x;
  const unsigned cmask = (1u << tile_shift) - 1u;
  for (int st = 0; st < nst; ++st) {
    const int tile = stile[st];
    for (int k = sp[st] + lane; k < sp[st + 1]; k += WAVE) {
      const unsigned p = pk[k];
      const int r = r0 + (int)(p >> tile_shift);
      const int c = (int)(((unsigned)tile << tile_shift) | (p & cmask));
      const double ie = transposed ? inv_e[c] : inv_e[r];
      const double id = transposed ? inv_d[r] : inv_d[c];
      tv[k] = (tv[k] * ie) * id;
    }
  }
}

// elementwise helpers on rescaling vectors
//  mode 0: v = sqrt(v), zeros -> 1          (Ruiz / Pock-Chambolle factors)
//  mode 1: v = sqrt(max(a, 0)) from a       (unused)  mode 2: inv = 1/v ; cum *= v
__global__ __launch_bounds__(TPB) void resc_sqrt_kernel(int n, double *__restrict__ v) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    double t = sqrt(v[i]);
    if (t == 0.0) t = 1.0;
    v[i] = t;
  }
}
__global__ __launch_bounds__(TPB) void resc_l2norm_kernel(int n, const double *__restrict__ scale,
                                                          double *__restrict__ sumsq_inout) {
  // l2_norm (preprocess.jl:99-113): scale .* sqrt(sum (a/scale)^2)
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB)
    sumsq_inout[i] = scale[i] * sqrt(sumsq_inout[i]);
}
__global__ __launch_bounds__(TPB) void resc_zero_to_one_inv_kernel(int n, double *__restrict__ v,
                                                                   double *__restrict__ inv, int do_zero_to_one) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    double t = v[i];
    if (do_zero_to_one && t == 0.0) { t = 1.0; v[i] = t; }
    inv[i] = 1.0 / t;
  }
}
__global__ __launch_bounds__(TPB) void resc_apply_vectors_kernel(int n, int m, const double *__restrict__ dv,
                                                                 const double *__restrict__ ev,
                                                                 double *__restrict__ c, double *__restrict__ lb,
                                                                 double *__restrict__ ub, double *__restrict__ b,
                                                                 double *__restrict__ cum_d,
                                                                 double *__restrict__ cum_e) {
  // scale_problem (preprocess.jl:555-573): c ./= D ; ub .*= D ; lb .*= D ; b ./= E
  const int tid = blockIdx.x * TPB + threadIdx.x, st = gridDim.x * TPB;
  for (int j = tid; j < n; j += st) {
    const double d = dv[j];
    c[j] = c[j] / d; ub[j] = ub[j] * d; lb[j] = lb[j] * d; cum_d[j] = cum_d[j] * d;
  }
  for (int i = tid; i < m; i += st) {
    const double e = ev[i];
    b[i] = b[i] / e; cum_e[i] = cum_e[i] * e;
  }
}
__global__ __launch_bounds__(TPB) void fill_kernel(int n, double v, double *__restrict__ out) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) out[i] = v;
}
__global__ __launch_bounds__(TPB) void maxabs_kernel(int64_t n, const double *__restrict__ v,
                                                     double *__restrict__ partials, int stride) {
  RedAcc<0, 1> a;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
    a.m[0] = fmax(a.m[0], fabs(v[i]));
  block_reduce_store<0, 1>(a, partials, stride);
}

// One-quantity variant writing to a device slot (row-partitioned form).
__global__ __launch_bounds__(FINAL_TPB) void final_to_slot_kernel(const double *p, int cnt, double *slot) {
  __shared__ double red[3][FINAL_TPB / WAVE];
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < cnt; i += FINAL_TPB) acc[0] += p[i];
  block_sum<1, FINAL_TPB>(acc, red);
  if (threadIdx.x == 0) *slot = acc[0];
}

// ---------------------------------------------------------------- host side

struct CsrDev {
  int rows = 0, cols = 0;
  int64_t nnz = 0;
  int *rowptr = nullptr, *col = nullptr;
  double *val = nullptr;
  int2 *blks = nullptr;
  int nblk = 0, per_xcd = 0, grid = 0;
  int nlong = 0, nchunks = 0, long_grid = 0;
  int *long_row = nullptr, *long_chunk_ptr = nullptr, *chunk_row = nullptr, *chunk_off = nullptr;
  double *chunk_partial = nullptr;
  int64_t max_row_nnz = 0;
  // tiled-sweep layout (optional)
  bool tiled = false;
  int tile_shift = 0, nwaves = 0, ntiles = 0, tw_rows = 0;
  int2 *wave_rows = nullptr;
  int *wave_ent = nullptr;        // per-wave entry offsets, one per step of its workgroup (+1)
  int *wave_step_off = nullptr;   // [nwaves] start of a wave's offsets inside wave_ent
  int *step_tile = nullptr;       // tile id of every step, workgroup after workgroup
  int *wg_step_off = nullptr;     // [grid+1] start of a workgroup's steps inside step_tile
  int64_t total_steps = 0;
  bool tw_scratch = false;        // long same-row runs: use the LDS-scratch chunk variant
  unsigned *pk = nullptr;
  double *tv = nullptr;
  int slots() const { return grid + long_grid; }
  CsrView view() const { return CsrView{rows, rowptr, col, val}; }
};

template <typename T>
int upload(T **dst, const std::vector<T> &src) {
  const size_t bytes = sizeof(T) * std::max<size_t>(src.size(), 1);
  HIP_TRY(hipMalloc((void **)dst, bytes));
  if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
  return 0;
}

int alloc_zero(double **dst, int64_t len) {
  const size_t bytes = sizeof(double) * (size_t)std::max<int64_t>(len, 1);
  HIP_TRY(hipMalloc((void **)dst, bytes));
  HIP_TRY(hipMemset(*dst, 0, bytes));
  return 0;
}

// Host-side construction of the tiled-sweep layout: wave row blocks (runs of
// <= TW_ROWS consecutive non-long rows) and their entries counting-sorted by
// column tile (stable, so (row, col) order is kept inside a tile).
int build_tiled(CsrDev &D, int rows, const std::vector<int> &rowptr, const std::vector<int> &col,
                const std::vector<double> &val, int tile_shift) {
  // Geometry.  A CU holds 2 workgroups of 8 waves; the grid runs in rounds of
  // 256 CUs x 16 waves.  Rows per wave is chosen so that the rounds are full
  // (no tail round), within the LDS budget (160 KiB / 16 waves).
  const int64_t slots = 256LL * 2 * TW_WPB;               // resident waves per round
  const int max_rows = std::min<int>(TW_MAX_ROWS, 1 << (32 - tile_shift));
  int TW_ROWS;
  {
    int64_t rounds = std::max<int64_t>(1, ((int64_t)rows + slots * max_rows - 1) / (slots * max_rows));
    int64_t rpw = ((int64_t)rows + slots * rounds - 1) / (slots * rounds);
    TW_ROWS = (int)std::max<int64_t>(64, std::min<int64_t>(max_rows, rpw));
  }
  if (const char *ev = getenv("PDHG_TW_ROWS")) TW_ROWS = std::max(1, std::min(atoi(ev), max_rows));
  D.tw_rows = TW_ROWS;
  const int ntiles = std::max<int>(1, (int)((((int64_t)D.cols) + (1LL << tile_shift) - 1) >> tile_shift));
  const unsigned cmask = (1u << tile_shift) - 1u;
  const int WIN = TW_U * WAVE;  // entries a wave holds in registers per step
  // pass 1: wave row blocks.  A wave owns <= TW_ROWS rows AND <= nnz_cap
  // nonzeros: hub regions (PageRank's oldest nodes) would otherwise give one
  // wave 100x the average work and the whole launch would wait for its workgroup.
  const int64_t est_waves = std::max<int64_t>(1, ((int64_t)rows + TW_ROWS - 1) / TW_ROWS);
  const int64_t nnz_cap = std::max<int64_t>(4096, 2 * (D.nnz / est_waves));   // 2x the average wave
  std::vector<int2> wave_rows;
  {
    int r = 0;
    while (r < rows) {
      if (rowptr[r + 1] - rowptr[r] > BLOCK_NNZ) { ++r; continue; }  // long row: separate path
      const int r0 = r;
      while (r < rows && (r - r0) < TW_ROWS && rowptr[r + 1] - rowptr[r] <= BLOCK_NNZ &&
             (r == r0 || (int64_t)rowptr[r + 1] - rowptr[r0] <= nnz_cap)) ++r;
      wave_rows.push_back(make_int2(r0, r));
    }
  }
  const int nwaves = (int)wave_rows.size();
  const int grid = (nwaves + TW_WPB - 1) / TW_WPB;
  std::vector<unsigned> pk;
  std::vector<double> tv;
  pk.reserve((size_t)D.nnz);
  tv.reserve((size_t)D.nnz);
  std::vector<int> step_ptr, wave_step_off((size_t)std::max(nwaves, 1), 0), step_tile, wg_step_off(1, 0);
  std::vector<std::vector<int>> cnt(TW_WPB, std::vector<int>((size_t)ntiles + 1));
  std::vector<int> nsub((size_t)ntiles);
  int max_run = 0;   // longest same-row run inside one tile
  for (int g = 0; g < grid; ++g) {
    const int w0 = g * TW_WPB, w1 = std::min(nwaves, w0 + TW_WPB);
    // cell sizes of the workgroup's waves
    std::fill(nsub.begin(), nsub.end(), 0);
    for (int w = w0; w < w1; ++w) {
      std::vector<int> &c = cnt[w - w0];
      std::fill(c.begin(), c.end(), 0);
      for (int k = rowptr[wave_rows[w].x]; k < rowptr[wave_rows[w].y]; ++k) c[(col[k] >> tile_shift) + 1] += 1;
      for (int t = 0; t < ntiles; ++t) nsub[t] = std::max(nsub[t], (c[t + 1] + WIN - 1) / WIN);
      for (int t = 0; t < ntiles; ++t) c[t + 1] += c[t];   // prefix: cell start offsets
    }
    // the workgroup's step list (heavy tiles repeated, empty tiles skipped)
    for (int t = 0; t < ntiles; ++t)
      for (int j = 0; j < nsub[t]; ++j) step_tile.push_back(t);
    wg_step_off.push_back((int)step_tile.size());
    // entries of each wave, tile-major (stable in (row, col)), and its step offsets
    for (int w = w0; w < w1; ++w) {
      std::vector<int> &c = cnt[w - w0];
      const int r0 = wave_rows[w].x, r1 = wave_rows[w].y;
      const size_t base = pk.size();
      const int total = rowptr[r1] - rowptr[r0];
      wave_step_off[w] = (int)step_ptr.size();
      for (int t = 0; t < ntiles; ++t) {
        const int cs = c[t], ce = c[t + 1], len = ce - cs;
        const int per = nsub[t] ? (len + nsub[t] - 1) / nsub[t] : 0;
        for (int j = 0; j < nsub[t]; ++j) step_ptr.push_back((int)base + std::min(ce, cs + j * per));
      }
      step_ptr.push_back((int)base + total);
      pk.resize(base + (size_t)total);
      tv.resize(base + (size_t)total);
      std::vector<int> next(c.begin(), c.end() - 1);
      for (int rr = r0; rr < r1; ++rr) {
        const unsigned rl = (unsigned)(rr - r0) << tile_shift;
        int run = 0, run_tile = -1;
        for (int k = rowptr[rr]; k < rowptr[rr + 1]; ++k) {
          const int tt = col[k] >> tile_shift;
          run = (tt == run_tile) ? run + 1 : 1;
          run_tile = tt;
          if (run > max_run) max_run = run;
          const int pos = next[tt]++;
          pk[base + pos] = rl | ((unsigned)col[k] & cmask);
          tv[base + pos] = val[k];
        }
      }
    }
  }
  // Rows with long same-row runs inside a tile (hub rows of the PageRank LP,
  // dense-ish blocks) are summed by one lane, sequentially, to keep the
  // ascending-column order; the stream layout does that from LDS with 8 reads
  // in flight and wins on such matrices (PageRank-1M: 0.106 ms vs 0.18 ms), and
  // hubs give it natural cache locality anyway.  PDHG_SPMV=tiled overrides.
  {
    const char *mode_env = getenv("PDHG_SPMV");
    const bool forced = mode_env && !strcmp(mode_env, "tiled");
    if (!forced && max_run > 32) return 0;
  }
  D.tiled = true;
  D.tile_shift = tile_shift;
  D.ntiles = ntiles;
  D.nwaves = nwaves;
  D.grid = grid;
  D.total_steps = (int64_t)step_tile.size();
  D.tw_scratch = max_run > 8;
  int rc;
  if ((rc = upload(&D.wave_rows, wave_rows))) return rc;
  if ((rc = upload(&D.wave_ent, step_ptr))) return rc;
  if ((rc = upload(&D.wave_step_off, wave_step_off))) return rc;
  if ((rc = upload(&D.step_tile, step_tile))) return rc;
  if ((rc = upload(&D.wg_step_off, wg_step_off))) return rc;
  if ((rc = upload(&D.pk, pk))) return rc;
  if ((rc = upload(&D.tv, tv))) return rc;
  return 0;
}

int build_csr_dev(CsrDev &D, int rows, int cols, const std::vector<int> &rowptr,
                  const std::vector<int> &col, const std::vector<double> &val,
                  bool remap, int tile_shift = 0) {
  D.rows = rows;
  D.cols = cols;
  D.nnz = rowptr[rows];
  std::vector<int2> blks;
  std::vector<int> long_row, long_chunk_ptr(1, 0), chunk_row, chunk_off;
  int r = 0;
  while (r < rows) {
    int len = rowptr[r + 1] - rowptr[r];
    D.max_row_nnz = std::max<int64_t>(D.max_row_nnz, len);
    if (len > BLOCK_NNZ) {
      const int l = (int)long_row.size();
      long_row.push_back(r);
      for (int off = 0; off < len; off += LONG_CHUNK) {
        chunk_row.push_back(r);
        chunk_off.push_back(off);
      }
      long_chunk_ptr.push_back((int)chunk_row.size());
      (void)l;
      ++r;
      continue;
    }
    const int r0 = r;
    int nn = 0;
    while (r < rows && (r - r0) < MAX_ROWS_PER_BLOCK) {
      len = rowptr[r + 1] - rowptr[r];
      if (len > BLOCK_NNZ - nn) break;
      D.max_row_nnz = std::max<int64_t>(D.max_row_nnz, len);
      nn += len;
      ++r;
    }
    blks.push_back(make_int2(r0, r));
  }
  D.nblk = (int)blks.size();
  D.per_xcd = (D.nblk + NUM_XCD - 1) / NUM_XCD;
  D.grid = remap ? D.per_xcd * NUM_XCD : D.nblk;
  D.nlong = (int)long_row.size();
  D.nchunks = (int)chunk_row.size();
  D.long_grid = (D.nlong + TPB - 1) / TPB;
  int rc;
  if ((rc = upload(&D.rowptr, rowptr))) return rc;
  if ((rc = upload(&D.col, col))) return rc;
  if ((rc = upload(&D.val, val))) return rc;
  if ((rc = upload(&D.blks, blks))) return rc;
  if ((rc = upload(&D.long_row, long_row))) return rc;
  if ((rc = upload(&D.long_chunk_ptr, long_chunk_ptr))) return rc;
  if ((rc = upload(&D.chunk_row, chunk_row))) return rc;
  if ((rc = upload(&D.chunk_off, chunk_off))) return rc;
  if ((rc = alloc_zero(&D.chunk_partial, D.nchunks))) return rc;
  if (tile_shift > 0) {
    if ((rc = build_tiled(D, rows, rowptr, col, val, tile_shift))) return rc;
  }
  return 0;
}

void free_csr_dev(CsrDev &D) {
  void *ptrs[] = {D.rowptr, D.col, D.val, D.blks, D.long_row, D.long_chunk_ptr,
                  D.chunk_row, D.chunk_off, D.chunk_partial, D.wave_rows, D.wave_ent, D.pk, D.tv,
                  D.wave_step_off, D.step_tile, D.wg_step_off};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  D = CsrDev();
}

}  // namespace

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
  double *tr_g = nullptr, *tr_dir = nullptr, *tr_thr = nullptr;  // n+m each
  double *ev_partials = nullptr, *ev_out = nullptr, *ev_host = nullptr;
  int ev_grid = 1;
  bool has_original = false;

  bool profile = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int64_t prof_count[PDHG_K_COUNT] = {0};
  double prof_ms[PDHG_K_COUNT] = {0};
  bool dist_pending = false;
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

template <int MODE>
int launch_spmv(pdhg_handle *h, const CsrDev &D, const double *xin, EpiArgs e) {
  if (D.tiled) {
    if (D.grid > 0) {
      const size_t lds = sizeof(double) * ((size_t)TW_WPB * D.tw_rows + 3 * TW_WPB + (D.tw_scratch ? TW_WPB * WAVE : 0));
      static size_t attr_set[3][2] = {{0, 0}, {0, 0}, {0, 0}};
      if (D.tw_scratch) {
        if (attr_set[MODE][1] < lds) {
          HIP_TRY(hipFuncSetAttribute((const void *)spmv_tiled_kernel<MODE, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          attr_set[MODE][1] = lds;
        }
        hipLaunchKernelGGL((spmv_tiled_kernel<MODE, true>), dim3(D.grid), dim3(TW_WPB * WAVE), lds, h->stream,
                           D.wave_rows, D.wave_ent, D.wave_step_off, D.step_tile, D.wg_step_off, D.nwaves,
                           D.tile_shift, D.tw_rows, D.pk, D.tv, xin, e);
      } else {
        if (attr_set[MODE][0] < lds) {
          HIP_TRY(hipFuncSetAttribute((const void *)spmv_tiled_kernel<MODE, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          attr_set[MODE][0] = lds;
        }
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
  if (rows >= INT32_MAX || cols >= INT32_MAX || nnz >= INT32_MAX)
    return fail(-2, "dimensions/nnz >= 2^31 need the 64-bit index path (not built)");
  if (colptr[0] != base) return fail(-1, "colptr[0] != index_base");
  if (colptr[cols] - base != nnz) return fail(-1, "colptr[n] - base != nnz");
  t_rowptr.resize(cols + 1);
  for (int64_t j = 0; j <= cols; ++j) {
    const int64_t v = colptr[j] - base;
    if (v < 0 || v > nnz || (j > 0 && v < t_rowptr[j - 1])) return fail(-1, "colptr not monotone");
    t_rowptr[j] = (int)v;
  }
  t_col.resize(nnz);
  t_val.assign(nzval, nzval + nnz);
  rowptr.assign(rows + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t r = rowval[k] - base;
    if (r < 0 || r >= rows) return fail(-1, "rowval out of range");
    t_col[k] = (int)r;
    rowptr[r + 1] += 1;
  }
  for (int64_t i = 0; i < rows; ++i) rowptr[i + 1] += rowptr[i];
  col.resize(nnz);
  val.resize(nnz);
  std::vector<int> next(rowptr.begin(), rowptr.end() - 1);
  for (int64_t j = 0; j < cols; ++j) {
    for (int k = t_rowptr[j]; k < t_rowptr[j + 1]; ++k) {
      const int p = next[t_col[k]]++;
      col[p] = (int)j;
      val[p] = t_val[k];
    }
  }
  return 0;
}

}  // namespace

// ================================================================== C ABI

extern "C" {

const char *pdhg_last_error(void) { return g_last_error.c_str(); }
int pdhg_abi_version(void) { return 3; }

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

  std::vector<int> t_rowptr, t_col, rowptr, col;
  std::vector<double> t_val, val;
  int rc = csc_to_both(m, n, nnz, colptr, rowval, nzval, index_base, t_rowptr, t_col, t_val, rowptr, col, val);
  if (rc) return rc;

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
  CK(build_csr_dev(h->At, (int)n, (int)m, t_rowptr, t_col, t_val, h->remap, choose_tile_shift(m, nnz, n)));
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
                    h->tr_dir, h->tr_thr, h->ev_partials, h->ev_out};
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
  HIP_TRY(hipMemsetAsync(h->sum_x, 0, sizeof(double) * (size_t)std::max<int64_t>(h->n, 1), h->stream));
  HIP_TRY(hipMemsetAsync(h->sum_y, 0, sizeof(double) * (size_t)std::max<int64_t>(h->m, 1), h->stream));
  h->sum_x_count = h->sum_y_count = 0;
  h->sum_x_weights = h->sum_y_weights = 0.0;
  return 0;
}

int pdhg_restart_to_average(pdhg_handle *h) {
  int rc = check_handle(h);
  if (rc) return rc;
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
  if (h->has_q) return fail(-2, "row-partitioned form supports LPs only");
  if ((rc = launch_primal(h, step_size / primal_weight, theta, true))) return rc;
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
  if (h->has_q) return fail(-2, "row-partitioned form supports LPs only");
  hipLaunchKernelGGL(xbar_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->x, h->x_next, theta, h->xbar);
  HIP_TRY(hipGetLastError());
  if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  if ((rc = launch_aty_plain(h, h->y_next, h->aty_next))) return rc;
  hipLaunchKernelGGL(final_to_slot_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->pA, h->A.slots(), h->aty_next + h->n);
  HIP_TRY(hipGetLastError());
  h->dist_pending = true;
  return 0;
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
  return finish_scalars(h, h->pAt, h->ew_grid_n, h->pAt_stride, h->aty_next + h->n, 1, 0, out);
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
    hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, h->sum_x, h->sum_x_weights, h->px_avg);
    hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->py_avg);
    HIP_TRY(hipGetLastError());
    *px = h->px_avg; *py = h->py_avg;
    return 0;
  }
  return fail(-1, "unknown point selector");
}

int pdhg_set_original_problem(pdhg_handle *h, const double *constraint_rescaling,
                              const double *variable_rescaling, const double *c_o, const double *b_o,
                              const double *lb_o, const double *ub_o) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (h->has_q) return fail(-2, "device evaluation supports LPs only");
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

int pdhg_eval_point(pdhg_handle *h, int point, double out[20]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!h->has_original) return fail(-1, "pdhg_set_original_problem has not been called");
  const double *px, *py;
  if ((rc = select_point(h, point, &px, &py))) return rc;
  EpiArgs e{};
  e.out = h->ev_ax;
  if ((rc = launch_spmv<MODE_PLAIN>(h, h->A, px, e))) return rc;
  e.out = h->ev_aty;
  if ((rc = launch_spmv<MODE_PLAIN>(h, h->At, py, e))) return rc;
  hipLaunchKernelGGL(eval_rows_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->m, (int)h->num_eq,
                     h->ev_ax, py, h->E, h->b_o, h->ev_partials, h->ev_grid);
  if ((rc = ev_finish(h, 4, 4, out))) return rc;
  hipLaunchKernelGGL(eval_cols_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, h->ev_aty, px,
                     h->Dv, h->c_o, h->lb_o, h->ub_o, h->ev_partials, h->ev_grid);
  return ev_finish(h, 6, 6, out + 8);
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
  if (h->has_q) return fail(-2, "device trust region supports LPs only");
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
  EpiArgs e{};
  e.out = h->ev_ax;
  if ((rc = launch_spmv<MODE_PLAIN>(h, h->A, px, e))) return rc;
  e.out = h->ev_aty;
  if ((rc = launch_spmv<MODE_PLAIN>(h, h->At, py, e))) return rc;
  hipLaunchKernelGGL(tr_setup_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->n, (int)h->m,
                     (int)h->num_eq, px, py, h->ev_aty, h->ev_ax, h->c, h->b, h->lb, h->ub, wp, wd, range,
                     h->tr_g, h->tr_dir, h->tr_thr, h->ev_partials, h->ev_grid);
  double r[EV_MAXQ];
  if ((rc = ev_finish(h, 10, 1, r))) return rc;
  // compute_lagrangian_value (saddle_point.jl:1109-1120) without objective_constant
  out[0] = r[0] - r[1] + r[2];
  out[1] = out[2] = 0.0;
  out[3] = r[8]; out[4] = r[9];
  out[5] = 0.0; out[6] = 0.0; out[7] = 0.0;
  const double hinf = r[3], g2 = r[4], wd2_all = r[5], tmax = r[10];
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
  if (h->has_q) return fail(-2, "device rescaling supports LPs only");
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
