// sj_kernels.hpp -- part of the single translation unit pdhg_hip.hip (included there, after spmv_kernels.hpp).
// The SLICED JAGGED layout of the stream class (round 5) and its product kernel.
//
// Why.  The CSR "stream" kernel (spmv_kernels.hpp) parks a row block's products in LDS and lets one lane per row add
// them: two phases, a workgroup barrier, row pointers, LDS reads with bank conflicts.  Round 5 measured what that costs
// on a matrix whose gathers all hit L2 (banded 10M +-50000): 0.81 ms = 123 G nonzeros/s, against 0.70 ms for
// rocSPARSE's adaptive CSR kernel and 0.58-0.60 ms for a skeleton that streams the same 12 bytes per nonzero from HBM
// and gathers the same doubles with every lane walking ITS OWN row in registers (tools/slot_probe.hip, variant 14;
// profiles/r05_slot_probe.txt).  This file is that skeleton as a product kernel.
//
// Layout.  Rows are taken in GROUPS of SJ_SIGMA = 256 consecutive rows (one workgroup trip; the wide form of round 6:
// windows of 2 048 rows, see the kernel) and, inside a group, ordered
// by DECREASING length (stable: equal lengths keep their row order -- the layout is a deterministic function of the row
// pointers).  64 consecutive slots of that order are a SLICE, one wave's work: slot -> (row, length) in one word of `meta`.
// A slice's entries are stored LEVEL-MAJOR and jagged: level j holds the j-th entry (ascending column) of every row of
// the slice with more than j entries -- because the lengths decrease along the slice these are the first cnt_j lanes, so
// level j is cnt_j consecutive (col, val) pairs and a wave reads it with one coalesced load pair; no padding is stored,
// and sorting a whole group (not one slice) makes the four slices nearly uniform inside.  Lane l adds its row's
// products level by level, i.e. strictly left to right in a register: EVERY row of this layout has the reference's
// order of additions (saddle_point.jl:1102-1107, pdhg.jl:492; bit-identical to the CPU loops), whatever its length --
// there is no "relaxed" mode for the rows a lane walks (hub rows, below, follow the CSR kernel's rule).  The row sums then cross the workgroup through 2 KB of LDS so that thread t runs the
// fused epilogue of row (group base + t): y, b, x, A'y are read and y', A'y' written in ROW order, fully coalesced
// (with the epilogue in slot order -- round 5's first cut, sorting windows of 2 048 rows -- every epilogue operand became a
// 64-line gather and the A' product of a banded matrix ran at HALF the CSR kernel's speed).  Rows beyond
// CsrDev::long_thr stay with the long-row kernels (slot row = -1, no epilogue here).
// Column-slab passes (INIT) carry the row sums through e.init exactly as the CSR stream kernel does.
// Rows of more than SjDev::max_len (128) entries that are not "long" are HUB rows (round 6): a lane walks its row in batches of
// SJ_U dependent load -> gather -> add trips, so a 2 000-entry hub row would hold its workgroup for 125 memory round trips
// (PageRank-1M in round 5's first cut: 1.19 ms against 0.10) -- the kernel runs them as whole-workgroup row blocks of the CSR
// arrays before its slices instead (see the kernel).  Which matrices get the layout, and in which of its two forms:
// layout.hpp, sj_plan().
//
// Launch.  Persistent: SJ_WGS_PER_CU workgroups per compute unit (the skeleton is fastest at 1-4: more waves only
// widen the window of the gathered vector in flight), one group of four slices per workgroup and trip, XCD x walking the
// contiguous eighth of the groups (remap) so that its L2 sees one moving window.  Block partials: one slot per
// workgroup; the remaining slots of the stream part (the CSR kernel's grid, which the reductions still walk) are zeroed.
#pragma once

namespace {

constexpr int SJ_SIGMA = TPB;        // rows per workgroup trip of the narrow form (4 slices); the wide form sorts SJ_SIGMA * SJ_WIDE_G rows
constexpr int SJ_WIDE_G = 8;         // wide form (round 6): windows of 2 048 rows, every wave walks 8 of the window's 32 slices
constexpr int SJ_MAX_LEN = 128;      // longest row a lane walks; longer ones (long rows apart) are HUB rows: whole-workgroup row blocks
#ifndef PDHG_SJ_U
#define PDHG_SJ_U 16
#endif
constexpr int SJ_U = PDHG_SJ_U;      // levels per batch (one batch in registers, the next in flight)
#ifndef PDHG_SJ_WGS
#define PDHG_SJ_WGS 2
#endif
constexpr int SJ_WGS_PER_CU = PDHG_SJ_WGS;

struct SjDev {
  int nslices = 0, grid = 0, rows = 0;
  int G = 1;                         // slices per wave and window: 1 (256-row windows) or SJ_WIDE_G (2 048-row windows)
  int max_len = SJ_MAX_LEN;          // rows beyond this many entries (in this matrix / slab) are hub rows
  int64_t nnz = 0;                   // entries in the slices (hub and long rows hold theirs in the CSR arrays only)
  unsigned *meta = nullptr;          // [nslices * 64] slot -> (row - window base) << 16 | entries; SJ_NONE in the high half: no row
  int *slice_off = nullptr;          // [nslices + 1] first entry of every slice
  int *col = nullptr;                // [nnz] level-major inside a slice
  double *val = nullptr;
  int2 *hub = nullptr;               // [nhub] (r, r + 1): hub rows, one stream-kernel row block each, on the CSR arrays below
  int nhub = 0;
  int64_t hub_nnz = 0;
  double fill_narrow = 0.0, fill_wide = 0.0, ragged = 0.0;      // what the builder's rule measured (layout.hpp: sj_plan)
  const int *csr_rowptr = nullptr, *csr_col = nullptr;     // the CSR arrays the copy was filled from (not owned)
  const double *csr_val = nullptr;
  bool on() const { return nslices > 0; }
};

constexpr unsigned SJ_NONE = 0xFFFFu;
struct SjView {
  int nslices, rows;
  const unsigned *meta;
  const int *slice_off;
  const int *col;
  const double *val;
  const int2 *hub;
  int nhub, relaxed;
  CsrView csr;
};
inline SjView sj_view(const SjDev &J, int relaxed = 1) {
  return SjView{J.nslices, J.rows, J.meta, J.slice_off, J.col, J.val, J.hub, J.nhub, relaxed, CsrView{J.rows, J.csr_rowptr, J.csr_col, J.csr_val}};
}

// one wave per slice: the CSR entries of its rows into the level-major order.  slices_per_window = 4 * G.
__global__ __launch_bounds__(TPB) void sj_fill_kernel(int nslices, int slices_per_window, const unsigned *__restrict__ meta, const int *__restrict__ slice_off,
                                                      const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      const double *__restrict__ val, int *__restrict__ sj_col, double *__restrict__ sj_val) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int slice = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
  if (slice >= nslices) return;
  const unsigned w = meta[slice * WAVE + lane];
  const int l = (int)(w & 0xFFFFu);
  const int r = (w >> 16) == SJ_NONE ? -1 : (slice / slices_per_window) * (slices_per_window * WAVE) + (int)(w >> 16);
  const int src = r >= 0 ? rowptr[r] : 0;
  int off = slice_off[slice];
  const int L = __builtin_amdgcn_readfirstlane(l);      // lane 0 holds the slice's longest row
  for (int j = 0; j < L; ++j) {
    const bool act = j < l;
    const int cnt = __popcll(__ballot(act));
    if (act) {
      sj_col[off + lane] = col[src + j];
      sj_val[off + lane] = val[src + j];
    }
    off += cnt;
  }
}

// SJ_U levels of a slice in registers
struct SjBatch {
  int c[SJ_U];
  double v[SJ_U];
};
// levels j0 .. j0 + SJ_U - 1 of the slice for this lane: (col, val) of level j at off + lane when the lane's row has more
// than j entries; `off` moves on by the number of such lanes.  L: levels of the slice (uniform): the branch skips levels
// no lane has.
__device__ __forceinline__ void sj_load_batch(const SjView &J, int &off, int l, int L, int j0, int lane, SjBatch &B) {
#pragma unroll
  for (int jj = 0; jj < SJ_U; ++jj) {
    B.c[jj] = 0;
    B.v[jj] = 0.0;
    if (j0 + jj < L) {                                    // wave-uniform
      const bool act = j0 + jj < l;
      const int k = off + lane;
      off += __popcll(__ballot(act));
      if (act) {
        B.c[jj] = __builtin_nontemporal_load(J.col + k);
        B.v[jj] = __builtin_nontemporal_load(J.val + k);
      }
    }
  }
}

// Software pipeline at BATCH granularity (SJ_U levels): while a batch's gathers are in flight the next batch -- of the same
// slice, or the first of the wave's slice in the workgroup's next group -- is already requested, together with that group's
// epilogue operands; the slot words run two groups ahead.  So a batch costs one gather round trip (L2) and a group one
// workgroup barrier, instead of the chain slot words -> entries -> gathers -> operands -> store (0.82 ms on banded 10M, the
// CSR kernel's time; 0.62 pipelined).  The barrier that hands the row sums to the row-order epilogue is also PACING: a
// variant without it (every wave its own pipeline, the epilogue in the lane that walked the row) ran at 0.74 -- the waves
// of an XCD drift apart and the window of the gathered vector in flight widens, exactly as in the sweep
// (profiles/r05_sj_layout.txt).
//
// Round 6, the WIDE form (G = SJ_WIDE_G): the sorting window is 2 048 rows -- 32 slices, wave w walks slices w, w + 4, ... of
// the window (long and short ones alike) -- and the row sums of the whole window cross the workgroup through LDS before the
// epilogue runs in ROW order, eight rows per thread.  What it is for: rows of power-law or Poisson length.  Sorted inside
// 256 rows, a group's four slices last as long as their longest rows (Poisson(10) columns of a banded matrix: 82 % of the
// lane-levels carry an entry; PageRank's rows: a fraction); sorted inside 2 048 rows the slices are nearly uniform.  The
// epilogue's operands are requested just before the window's barrier (24 coalesced loads per thread in flight) instead of a
// trip ahead.  HUB rows (round 6; more than SjDev::max_len entries, the long-row path's apart): no lane walks them --
// every workgroup first takes its share of them as row blocks of the CSR arrays (stream_block_load / _finish: the
// products through LDS, added left to right by one lane, or by a wave in relaxed order beyond 256 entries: the CSR
// kernel's bits), which is what lets PageRank's matrices (rows of 5 ... 4 000 entries) use the layout at all.
template <int MODE, bool INIT = false, int TAG = 0, int G = 1>
__global__ __launch_bounds__(TPB) void spmv_sj_kernel(SjView J, const double *__restrict__ xin, int remap, int stream_slots, EpiArgs e) {
  constexpr int SIGMA = SJ_SIGMA * G;                     // rows per window
  constexpr int SPW = (TPB / WAVE) * G;                   // slices per window
  constexpr size_t SUM_BYTES = sizeof(double) * 2 * SIGMA;
#ifdef PDHG_SJ_NOHUB
  constexpr size_t LDS_BYTES = SUM_BYTES;
#else
  constexpr size_t LDS_BYTES = SUM_BYTES > sizeof(double) * BLOCK_NNZ ? SUM_BYTES : sizeof(double) * BLOCK_NNZ;
#endif
  __shared__ double red[6][TPB / WAVE];
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];     // hub phase: prod[BLOCK_NNZ]; then row_sum[2][SIGMA]
#ifdef PDHG_SJ_OKINT
  __shared__ int row_ok[2][SIGMA];
#else
  __shared__ unsigned char row_ok[2][SIGMA];
#endif
  double (*row_sum)[SIGMA] = reinterpret_cast<double (*)[SIGMA]>(lds_raw);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  Acc3 acc3 = acc3_zero();
  // ---- hub rows first: one row block each, dealt round robin
#ifndef PDHG_SJ_NOHUB
  if (J.nhub > 0) {
    double *prod = reinterpret_cast<double *>(lds_raw);
    for (int hb = (int)blockIdx.x; hb < J.nhub; hb += (int)gridDim.x) {
      StreamRegs sg;
      stream_block_load(J.csr, J.hub[hb], sg);
      stream_block_finish<MODE, INIT>(J.csr, xin, sg, e, J.relaxed, acc3, prod);
      __syncthreads();
    }
  }
#endif
  const int ngroups = (J.nslices + SPW - 1) / SPW;
  const int per_xcd = (ngroups + NUM_XCD - 1) / NUM_XCD;
  // remap: workgroup b runs on XCD b % 8 (round-robin dispatch); it walks groups x * per_xcd + i, i = b / 8, b / 8 + gridDim / 8, ...
  const int first = remap ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int stride = remap ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int limit = remap ? per_xcd : ngroups;
  const int xbase = remap ? (int)(blockIdx.x & (NUM_XCD - 1)) * per_xcd : 0;
#pragma unroll
  for (int qq = 0; qq < G; ++qq) {
    row_ok[0][qq * TPB + tid] = 0;
    row_ok[1][qq * TPB + tid] = 0;
  }
  __syncthreads();
  auto group_of = [&](int i) { const int g = xbase + i; return (i < limit && g < ngroups) ? g : -1; };
  // the wave's walk: (trip i, slice q of the window) -> (i, q + 1) ... (i, G - 1) -> (i + stride, 0)
  struct Pos { int i, g, q; };
  auto next_pos = [&](Pos p) {
    if (G > 1 && p.q + 1 < G) return Pos{p.i, p.g, p.q + 1};
    const int i2 = p.i + stride;
    return Pos{i2, p.g >= 0 ? group_of(i2) : -1, 0};
  };
  // ---- in flight: the slot words of the next two slices (w_n, w_nn), the next batch (B_n), (G == 1) the next group's operands
  unsigned w_n = (SJ_NONE << 16), w_nn = (SJ_NONE << 16);
  int off_n = 0, off_nn = 0;
  EpiOps ops_n{0.0, 0.0, 0.0};
  double init_n = 0.0;
  SjBatch B_n;
  auto request_meta = [&](Pos p, unsigned &w, int &off) {
    w = (SJ_NONE << 16);
    off = 0;
    const int slice = p.g * SPW + p.q * (TPB / WAVE) + wave;
    if (p.g >= 0 && slice < J.nslices) {                  // wave-uniform
      w = J.meta[slice * WAVE + lane];
      off = J.slice_off[slice];
    }
  };
  // the first batch, (G == 1) the epilogue operands, and the carried sums of the slice at p, whose slot words (w_n, off_n) have arrived
  auto request_slice = [&](Pos p) {
    if (G == 1) {
      const int row = p.g * SIGMA + tid;
      if (p.g >= 0 && row < J.rows) ops_n = epi_load<MODE>(e, row);
    }
    const int l = (int)(w_n & 0xFFFFu);
    if (INIT) {
      const int r = (w_n >> 16) == SJ_NONE ? -1 : p.g * SIGMA + (int)(w_n >> 16);
      init_n = r >= 0 ? e.init[r] : 0.0;
    }
    sj_load_batch(J, off_n, l, __builtin_amdgcn_readfirstlane(l), 0, lane, B_n);
  };
  // wide form: the epilogue operands of the window's rows (row order, G per thread), requested behind the first batch of
  // the window's FIRST slice -- a whole window ahead of their use; requested just before the window's barrier they cost a
  // round trip to HBM per window with nothing to hide it (A x on banded 10M 0.61 -> 0.69 ms)
  EpiOps wo[G];
#pragma unroll
  for (int qq = 0; qq < G; ++qq) wo[qq] = EpiOps{0.0, 0.0, 0.0};
  Pos cur{first, group_of(first), 0};
  Pos nxt = next_pos(cur);
  request_meta(cur, w_n, off_n);
  request_meta(nxt, w_nn, off_nn);
  request_slice(cur);
  int buf = 0;
  while (cur.g >= 0) {
    const unsigned w = w_n;
    const EpiOps ops = ops_n;
    const int l = (int)(w & 0xFFFFu);
    const int L = __builtin_amdgcn_readfirstlane(l);      // lane 0 holds the slice's longest row
    const int rl = (w >> 16) == SJ_NONE ? -1 : (int)(w >> 16);
    const int base = cur.g * SIGMA;
    double s = INIT ? init_n : 0.0;
    const Pos nxt2 = next_pos(nxt);
    int j0 = 0;
    do {                                                  // the slice's batches (at least one trip: it requests what follows)
      double xv[SJ_U], vv[SJ_U];
#ifndef PDHG_SJ_NOWAIT
      // Every column index of the batch must have arrived before its gather can be issued, and loads return in order:
      // ONE wait for all of them here costs nothing (the batch was requested a gather round trip ago) and lets the 16
      // gathers go out back to back.  Without it the compiler, which cannot count the conditionally issued loads of
      // sj_load_batch across the loop, put `s_waitcnt vmcnt(1)` in front of EVERY gather -- two gathers in flight per
      // wave: banded 10M 0.61 -> 0.68 ms when round 6 restructured the loop (profiles/r06_sj_waitcnt.txt).
      __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0), expcnt / lgkmcnt untouched
#endif
#pragma unroll
      for (int jj = 0; jj < SJ_U; ++jj) {
        xv[jj] = 0.0;
        if (j0 + jj < L) xv[jj] = (j0 + jj < l) ? xin[(unsigned)B_n.c[jj]] : 0.0;      // (uniform branch: levels no lane of the slice has)
        vv[jj] = B_n.v[jj];
      }
      // behind these gathers: the batch that follows
      if (j0 + SJ_U < L) {                                // wave-uniform: the same slice goes on
        sj_load_batch(J, off_n, l, L, j0 + SJ_U, lane, B_n);
      } else {                                            // the wave's next slice; slot words for the one after
        w_n = w_nn;
        off_n = off_nn;
        request_meta(nxt2, w_nn, off_nn);
        request_slice(nxt);
      }
      if (G > 1 && cur.q == 0 && j0 == 0) {               // (wave-uniform)
#pragma unroll
        for (int qq = 0; qq < G; ++qq) {
          const int row = base + qq * TPB + tid;
          if (row < J.rows) wo[qq] = epi_load<MODE>(e, row);
        }
      }
#pragma unroll
      for (int jj = 0; jj < SJ_U; ++jj) {
        if (j0 + jj < l) {
          const double p = vv[jj] * xv[jj];
          s = s + p;
        }
      }
      j0 += SJ_U;
    } while (j0 < L);
    if (rl >= 0) {
      row_sum[buf][rl] = s;
      row_ok[buf][rl] = 1;
    }
    if (G == 1) {
      __syncthreads();
      // thread t: the epilogue of row base + t, operands in row order (this buffer is next written two trips from now,
      // behind the next trip's barrier)
      if (row_ok[buf][tid]) {
        row_ok[buf][tid] = 0;
        epi_apply<MODE>(e, base + tid, row_sum[buf][tid], ops, acc3);
      }
      buf ^= 1;
    } else if (cur.q != G - 1) {
#ifndef PDHG_SJ_NOPACE
      // pacing only (nothing is handed over): the four waves enter their next slices together, as the narrow form's
      // barrier per 256 rows makes them -- without it they drift apart inside the window's eight slices and the part of
      // the gathered vector in flight widens (A x on banded 10M: 0.70 ms against the narrow form's 0.61)
      __syncthreads();
#endif
    } else {                                              // the window's last slice of this wave (the same trip for all four waves)
      __syncthreads();
#pragma unroll
      for (int qq = 0; qq < G; ++qq) {
        const int rr = qq * TPB + tid;
        if (row_ok[buf][rr]) {
          row_ok[buf][rr] = 0;
          epi_apply<MODE>(e, base + rr, row_sum[buf][rr], wo[qq], acc3);
        }
      }
      buf ^= 1;
    }
    cur = nxt;
    nxt = nxt2;
  }
  constexpr int NQ = ModeNQ<MODE>::value;
  if (NQ > 0) {
    block_sum_dd<NQ, TPB>(acc3, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        e.partials[q * e.stride + blockIdx.x] = acc3.hi[q];
        e.partials[e.lo_offset + q * e.stride + blockIdx.x] = acc3.lo[q];
      }
    }
    // the slots of the CSR kernel's row blocks this launch does not use: the reductions add them
    for (int sl = (int)gridDim.x + (int)blockIdx.x * TPB + (int)threadIdx.x; sl < stream_slots; sl += (int)gridDim.x * TPB) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        e.partials[q * e.stride + sl] = 0.0;
        e.partials[e.lo_offset + q * e.stride + sl] = 0.0;
      }
    }
  }
}

// the form a copy was built in picks the kernel instance
template <int MODE, bool INIT, int TAG>
inline const void *sj_kernel_fn(const SjDev &J) {
  return J.G > 1 ? (const void *)spmv_sj_kernel<MODE, INIT, TAG, SJ_WIDE_G> : (const void *)spmv_sj_kernel<MODE, INIT, TAG, 1>;
}
template <int MODE, bool INIT, int TAG>
inline void launch_sj(hipStream_t stream, const SjDev &J, const double *xin, int remap, int relaxed, int stream_slots, const EpiArgs &e) {
  if (J.G > 1)
    hipLaunchKernelGGL((spmv_sj_kernel<MODE, INIT, TAG, SJ_WIDE_G>), dim3(J.grid), dim3(TPB), 0, stream, sj_view(J, relaxed), xin, remap, stream_slots, e);
  else
    hipLaunchKernelGGL((spmv_sj_kernel<MODE, INIT, TAG, 1>), dim3(J.grid), dim3(TPB), 0, stream, sj_view(J, relaxed), xin, remap, stream_slots, e);
}

}  // namespace
