// spmv_kernels.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// The two SpMV layouts (CSR-adaptive "stream", L2-tiled "sweep"), the long-row path and their fused epilogues.
#pragma once

namespace {

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
  // block partials, double-double: hi at partials[q*stride + slot], lo at partials[lo_offset + q*stride + slot]
  // (lo_offset = number of quantities * stride)
  double *partials;
  int stride, lo_offset;
  // column-slab passes of the stream layout: the row sums so far (read when INIT)
  const double *init;
  // MODE_DUAL: the deferred K7 of the previous accept, sum_y += avg_w * y (nullptr: none)
  double *sum_y;
  double avg_w;
};

template <bool COH>
__device__ __forceinline__ void stc(double *p, double v) {
  *p = v;
}
// The operands a row's epilogue reads, apart from the row sum: loaded by epi_load, consumed by epi_apply, so that a
// kernel can request them long before the sum exists (sj_kernels.hpp).  row_epilogue is the two back to back: one
// definition of the arithmetic for every product kernel.
struct EpiOps {
  double a, b, c;
};
template <int MODE, bool COH = false>
__device__ __forceinline__ EpiOps epi_load(const EpiArgs &e, int r) {
  EpiOps o{0.0, 0.0, 0.0};
  if (MODE == MODE_DUAL) {
    o.a = ldc<COH>(e.y + r);
    o.b = e.b[r];
    if (e.sum_y) o.c = ldc<COH>(e.sum_y + r);
  } else if (MODE == MODE_ATY) {
    o.a = ldc<COH>(e.x_next + r);
    o.b = ldc<COH>(e.x + r);
    o.c = ldc<COH>(e.aty + r);
  }
  return o;
}
template <int MODE, bool COH = false>
__device__ __forceinline__ void epi_apply(const EpiArgs &e, int r, double s, const EpiOps &o, Acc3 &acc) {
  if (MODE == MODE_PLAIN) {
    e.out[r] = s;
  } else if (MODE == MODE_DUAL) {
    // compute_dual_gradient: b .- A*x              saddle_point.jl:1102-1107
    const double yo = o.a;
    if (e.sum_y) {
      const double t = yo * e.avg_w;
      stc<COH>(e.sum_y + r, o.c + t);
    }
    const double dg = o.b - s;
    // next_dual = y .+ (pw*step) .* dual_gradient   pdhg.jl:489-490
    const double t = e.sigma * dg;
    double yn = yo + t;
    // project_dual!: only inequality rows           saddle_point.jl:110-117
    if (r >= e.num_eq) yn = jl_max(yn, 0.0);
    stc<COH>(e.y_next + r, yn);
    const double dy = yn - yo;                       // pdhg.jl:535
    dd_add(acc.hi[0], acc.lo[0], dy * dy);
  } else {
    // next_dual_product = A' * next_dual            pdhg.jl:492
    stc<COH>(e.aty_next + r, s);
    const double dx = o.a - o.b;                     // pdhg.jl:534
    const double dd = s - o.c;                       // pdhg.jl:543
    dd_add(acc.hi[0], acc.lo[0], dx * dd);
    dd_add(acc.hi[1], acc.lo[1], dx * dx);
    dd_add(acc.hi[2], acc.lo[2], dd * dd);
  }
}
template <int MODE, bool COH = false>
__device__ __forceinline__ void row_epilogue(const EpiArgs &e, int r, double s,
                                             Acc3 &acc) {
  epi_apply<MODE, COH>(e, r, s, epi_load<MODE, COH>(e, r), acc);
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
// INIT: the row sum starts from e.init[r] instead of 0 -- a later COLUMN-SLAB pass of
// the same product.  When the gathered vector is a few times an XCD's L2 (4 MiB) the
// matrix is also kept split by column slab, one launch per slab in ascending column
// order: every workgroup of a launch then gathers from the same <= 4 MiB window, which
// stays L2-resident on every XCD (the kernel boundary is the chip-wide synchronisation
// the tiled sweep cannot afford), and the rows still receive their products strictly
// left to right, so the result is bit-identical to the single pass.
// One row block of the stream layout, in two halves so that the one-launch trial kernel
// (trial_kernel.hpp) can issue the first half -- which reads only the static matrix -- BEFORE
// the grid barrier that delivers the gathered vector.
// (dev A/B switches of the round-6 wait changes: -DPDHG_STREAM_SIGNED_OFFSETS, -DPDHG_STREAM_NO_GATHER_WAIT)
#ifdef PDHG_STREAM_SIGNED_OFFSETS
#define PDHG_COLOFF(c) (c)
#else
#define PDHG_COLOFF(c) ((unsigned)(c))
#endif
struct StreamRegs {
  int r0, r1, k0, k1;
  int cidx[UNROLL];
  double v[UNROLL];
};

// first half: the block's extent and its (col, val) entries, UNROLL independent coalesced
// non-temporal loads per lane.
// The column indices are used as UNSIGNED offsets further down (xin[(unsigned)cidx]).  With `int` offsets the compiler hoisted
// the sign extension of every index into the conditional block right behind its load -- `s_waitcnt vmcnt(1)` after each (col,
// val) pair: the eight "independent" load chains were eight DEPENDENT round trips to HBM per row block (round 6, found
// in the ISA after the same accident in sj_kernels.hpp; profiles/r06_stream_waitcnt.txt).
// THROTTLE (the column-slab passes): a lane waits for each index before it requests the next pair -- what rounds 1-5 did
// everywhere by accident (see above).  The slab passes exist because the gathered window barely fits an XCD's L2; with
// sixteen entry loads in flight per lane instead of two the streams crowd the window out: PageRank-1M 4 640 -> 4 470
// it/s without the waits, while the latency-bound products (one resident wave of workgroups: the L1-SVM LP 17.8 / 27.8
// -> 16.3 / 24.1 us per product) want them gone (profiles/r06_stream_waitcnt.txt).
template <bool THROTTLE = false>
__device__ __forceinline__ void stream_block_entries(const CsrView &A, StreamRegs &g) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) {
    const int k = g.k0 + tid + i * TPB;
    const bool ok = k < g.k1;
    g.cidx[i] = ok ? __builtin_nontemporal_load(A.col + k) : 0;
    g.v[i] = ok ? __builtin_nontemporal_load(A.val + k) : 0.0;
#ifndef PDHG_THROTTLE_PAIRS
#define PDHG_THROTTLE_PAIRS 1
#endif
    // (PDHG_THROTTLE_PAIRS, dev: wait after every k-th pair -- 1 is what rounds 1-5 did)
    if (THROTTLE && (i + 1) % PDHG_THROTTLE_PAIRS == 0) __builtin_amdgcn_s_waitcnt(0x0F71);     // vmcnt(1): the index is here, its value may still be on the way
  }
}
template <bool THROTTLE = false>
__device__ __forceinline__ void stream_block_load(const CsrView &A, int2 rr, StreamRegs &g) {
  g.r0 = rr.x; g.r1 = rr.y;
  g.k0 = A.rowptr[g.r0];
  g.k1 = A.rowptr[g.r1];
  stream_block_entries<THROTTLE>(A, g);
}
// the same from the block's extent word (r0, r1, k0, k1)
template <bool THROTTLE = false>
__device__ __forceinline__ void stream_block_load_ext(const CsrView &A, int4 x, StreamRegs &g) {
  g.r0 = x.x; g.r1 = x.y; g.k0 = x.z; g.k1 = x.w;
  stream_block_entries<THROTTLE>(A, g);
}

// Rows of more than this many entries are summed by a whole wave in relaxed-order mode.  Measured on the
// L1-SVM LP (feature columns of 65-2000 entries): with 64, a block of ~30 rows of ~70 entries costs its wave 30
// wave-wide reductions one after the other and the iteration got SLOWER (14.8k it/s against 15.9k strict); a
// lane adds 256 products in ~0.85 us, so only rows beyond that are worth a wave.
constexpr int RELAXED_MIN_ROW = 256;

// What a pipelined caller has requested a trip ahead for the row (r0 + tid) of a block: its extent, its epilogue
// operands and (column-slab passes) its carried sum.
struct RowPre {
  int rs, re;
  EpiOps ops;
  double init;
};

// the row data of the trip that starts at row `base` (this thread: row base + tid), requested in one go
template <int MODE, bool INIT, bool COH>
__device__ __forceinline__ RowPre stream_row_request(const CsrView &A, const EpiArgs &e, int base, int r1) {
  RowPre p{0, 0, EpiOps{0.0, 0.0, 0.0}, 0.0};
  const int r = base + (int)threadIdx.x;
  if (r < r1) {
    p.rs = A.rowptr[r];
    p.re = A.rowptr[r + 1];
    p.ops = epi_load<MODE, COH>(e, r);
    if (INIT) p.init = e.init[r];
  }
  return p;
}

// The row phase of a stream block whose products lie in prod[]: one lane per row adds them left to right (rows beyond
// RELAXED_MIN_ROW entries by their wave in relaxed order) and runs the fused epilogue.  pre != nullptr: the data of the
// rows r0 .. r0 + TPB - 1 is in *pre (spmv_stream_pipe_kernel); further row trips of a block of very short rows read theirs here.
template <int MODE, bool INIT, bool PIPE, bool COH>
__device__ __forceinline__ void stream_rows_phase(const CsrView &A, int r0, int r1, int k0, const EpiArgs &e, int relaxed, Acc3 &acc,
                                                  const double *prod, const RowPre *pre) {
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  // A trip's row data -- extent, epilogue operands, carried sum -- is requested a trip AHEAD (round 6): the first trip's by the
  // caller (`pre`: behind the block's gathers, or a block ahead in the pipelined kernel), every further trip's before the
  // current trip is summed.  Trip by trip a block of very short rows (1 024 rows of 2-3 entries: four trips) paid two
  // dependent round trips per trip -- row pointers, then operands -- with nothing to hide them.
  // (Not in the persistent trial kernels -- COH: at their register limit the ten more live registers become spills, the
  //  multi-step kernel's scratch grows from 20 to 88 bytes per lane and the L1-SVM leg loses 2 %; there a trip requests its
  //  own data; even the operands alone requested beside the row pointers spill 60 bytes and cost 4 %.  PageRank-1M's slab
  //  passes: 0.099 -> 0.093 ms per product, profiles/r06_stream_waitcnt.txt.)
  RowPre cur{0, 0, EpiOps{0.0, 0.0, 0.0}, 0.0}, nxt = cur;
  if (!COH) cur = pre ? *pre : stream_row_request<MODE, INIT, COH>(A, e, r0, r1);
  for (int base = r0; base < r1; base += TPB) {     // workgroup-uniform trip count (the wave sums below need all lanes)
    const int r = base + tid;
    const bool have = r < r1;
    const bool first = COH && pre && base == r0;          // (COH) this trip's row data came in ahead of time (workgroup-uniform)
    int ks, ke;
    double s;
    if (COH) {
      ks = have ? (first ? pre->rs : A.rowptr[r]) - k0 : 0;
      ke = have ? (first ? pre->re : A.rowptr[r + 1]) - k0 : 0;
      s = (INIT && have) ? (first ? pre->init : e.init[r]) : 0.0;
    } else {
      if (base + TPB < r1) nxt = stream_row_request<MODE, INIT, COH>(A, e, base + TPB, r1);      // (workgroup-uniform)
      ks = have ? cur.rs - k0 : 0;
      ke = have ? cur.re - k0 : 0;
      s = (INIT && have) ? cur.init : 0.0;
    }
    const bool wide = relaxed && (ke - ks > RELAXED_MIN_ROW);
    if (have && !wide) {
      int k = ks;
      // Adds strictly left to right (bit-exact order).  A row of hundreds of
      // products (hub rows, dense feature columns) is one dependent chain on one
      // lane, so its LDS reads are software-pipelined: the next 8 products are
      // requested before the current 8 are added, which leaves the chain at the
      // latency of the adds alone.
      if (!PIPE) {
        for (; k + 8 <= ke; k += 8) {
          const double t0 = prod[k], t1 = prod[k + 1], t2 = prod[k + 2], t3 = prod[k + 3];
          const double t4 = prod[k + 4], t5 = prod[k + 5], t6 = prod[k + 6], t7 = prod[k + 7];
          s = s + t0; s = s + t1; s = s + t2; s = s + t3; s = s + t4; s = s + t5; s = s + t6; s = s + t7;
        }
      } else if (k + 8 <= ke) {
#define LD8(p, q) const double p##0 = prod[q], p##1 = prod[(q) + 1], p##2 = prod[(q) + 2], p##3 = prod[(q) + 3], \
                               p##4 = prod[(q) + 4], p##5 = prod[(q) + 5], p##6 = prod[(q) + 6], p##7 = prod[(q) + 7]
#define ADD8(p) s = s + p##0; s = s + p##1; s = s + p##2; s = s + p##3; s = s + p##4; s = s + p##5; s = s + p##6; s = s + p##7
        double c0 = prod[k], c1 = prod[k + 1], c2 = prod[k + 2], c3 = prod[k + 3];
        double c4 = prod[k + 4], c5 = prod[k + 5], c6 = prod[k + 6], c7 = prod[k + 7];
        k += 8;
        // two register sets alive at once (c: being added, b: in flight) and scheduling
        // barriers: otherwise the compiler sinks the loads back behind the adds
        for (; k + 16 <= ke; k += 16) {
          LD8(b, k);
          __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the adds they overlap
          ADD8(c);
          c0 = prod[k + 8]; c1 = prod[k + 9]; c2 = prod[k + 10]; c3 = prod[k + 11];
          c4 = prod[k + 12]; c5 = prod[k + 13]; c6 = prod[k + 14]; c7 = prod[k + 15];
          __builtin_amdgcn_sched_barrier(0);
          ADD8(b);
        }
        if (k + 8 <= ke) {
          LD8(b, k);
          ADD8(c);
          ADD8(b);
          k += 8;
        } else {
          ADD8(c);
        }
#undef LD8
#undef ADD8
      }
      for (; k < ke; ++k) s = s + prod[k];
    }
    if (relaxed) {
      unsigned long long todo = __ballot(wide);
      while (todo) {                                   // wave-uniform
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int lks = __shfl(ks, l, WAVE), lke = __shfl(ke, l, WAVE);
        double part = 0.0;
        for (int k = lks + lane; k < lke; k += WAVE) part = part + prod[k];
        part = wave_sum(part);                          // fixed shuffle tree; lane 0 holds the total
        const double total = __shfl(part, 0, WAVE);
        if (lane == l) s = s + total;
      }
    }
    if (have) {
      if (!COH) epi_apply<MODE, COH>(e, r, s, cur.ops, acc);
      else if (first) epi_apply<MODE, COH>(e, r, s, pre->ops, acc);
      else row_epilogue<MODE, COH>(e, r, s, acc);
    }
    if (!COH) cur = nxt;
  }
}

// second half: gathers, products into LDS, per-row sums, fused epilogue.  acc[] receives this
// thread's contributions to the block partials.
// relaxed == 0 (PDHG_ROW_ORDER=strict): every row is added strictly left to right by one lane.
// relaxed != 0 (default): rows of more than RELAXED_MIN_ROW (256) entries are summed by their whole wave --
// lane l adds products l, l + 64, ... in ascending order, then a shuffle tree -- a fixed order
// (bitwise reproducible), but not the sequential one: |result - sequential| <= 1e-13 * sum |a x|,
// the bar the rows beyond BLOCK_NNZ have always had.  The row stays OWNED by the lane that
// would have added it (epilogue, partial sums): nothing else changes.
// PIPE: the two-register-set software pipeline of the strict per-lane row sum (32 VGPRs).  The one-launch trial
// kernel, which keeps a prefetched item's 24 registers alive across its phases, runs the plain 8-at-a-time loop
// instead (same order of additions, hence the same bits) to stay within 96 VGPRs.
template <int MODE, bool INIT, bool PIPE = true, bool COH = false>
__device__ __forceinline__ void stream_block_finish(const CsrView &A, const double *xin, const StreamRegs &g,
                                                    const EpiArgs &e, int relaxed, Acc3 &acc, double *prod) {
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int k0 = g.k0, k1 = g.k1;
  double xv[UNROLL];
  // every column index must be here before its gather can go out, and loads return in order: one wait for all sixteen
  // entry loads (requested together a round trip ago), then the eight gathers back to back -- the compiler, which cannot
  // count conditionally issued loads, otherwise waits in front of the first three gathers one by one
#ifndef PDHG_STREAM_NO_GATHER_WAIT
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
#endif
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) {
    const int k = k0 + tid + i * TPB;
    if (COH) {
      // unconditional: a lane without an entry holds column 0 (stream_block_load), a valid address -- an ATOMIC load
      // under a condition becomes a branch per load, and the waits the compiler puts in front of each serialise the
      // gathers.  (For plain loads the conditional form measured 1-2 % faster: kept.)
      const double t = ldc<true>(xin + PDHG_COLOFF(g.cidx[i]));
      xv[i] = (k < k1) ? t : 0.0;
    } else {
      xv[i] = (k < k1) ? xin[PDHG_COLOFF(g.cidx[i])] : 0.0;
    }
  }
  // behind the gathers: the first row trip's extents, operands and carried sums (nothing of it depends on the products)
  RowPre pre0{0, 0, EpiOps{0.0, 0.0, 0.0}, 0.0};
  if (!COH) pre0 = stream_row_request<MODE, INIT, COH>(A, e, g.r0, g.r1);
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) {
    const int k = tid + i * TPB;
    if (k0 + k < k1) prod[k] = g.v[i] * xv[i];
  }
  __syncthreads();
  stream_rows_phase<MODE, INIT, PIPE, COH>(A, g.r0, g.r1, g.k0, e, relaxed, acc, prod, COH ? nullptr : &pre0);
}

// TAG names the product in profiler output (0: the constraint matrix A, 1: its transpose,
// 2: the objective matrix): the MODE_PLAIN passes of a slab layout and the evaluation
// products would otherwise share one kernel name for both products.
template <int MODE, bool INIT = false, int TAG = 0>
__global__ __launch_bounds__(TPB) void spmv_stream_kernel(
    CsrView A, const double *__restrict__ xin, const int2 *__restrict__ blks, const int4 *__restrict__ ext,
    int nblk, int per_xcd, int remap, int relaxed, EpiArgs e) {
  __shared__ double prod[BLOCK_NNZ];
  __shared__ double red[6][TPB / WAVE];
  const int b = blockIdx.x;
  const int blk = remap ? ((b & (NUM_XCD - 1)) * per_xcd + (b >> 3)) : b;
  Acc3 acc = acc3_zero();
  const bool active = remap ? ((b >> 3) < per_xcd && blk < nblk) : (blk < nblk);
  if (active) {
    StreamRegs g;
    // ext (round 6): (r0, r1, k0, k1) of the block in ONE word -- the entry loads no longer wait for the block table AND two
    // row pointers (layouts built before it / callers without the table pass nullptr)
    if (ext) {
      const int4 x = ext[blk];
      if (relaxed & 2) stream_block_load_ext<true>(A, x, g);        // bit 1 of `relaxed`: a column-slab pass (throttled entry loads)
      else stream_block_load_ext<false>(A, x, g);
    } else if (relaxed & 2) stream_block_load<true>(A, blks[blk], g);
    else stream_block_load<false>(A, blks[blk], g);
    stream_block_finish<MODE, INIT>(A, xin, g, e, relaxed & 1, acc, prod);
  }
  constexpr int NQ = ModeNQ<MODE>::value;
  if (NQ > 0) {
    block_sum_dd<NQ, TPB>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        e.partials[q * e.stride + b] = acc.hi[q];
        e.partials[e.lo_offset + q * e.stride + b] = acc.lo[q];
      }
    }
  }
}

// The same product as a PERSISTENT, software-pipelined kernel (round 5), for stream-class matrices whose product is
// bandwidth work (more row blocks than the persistent trial kernels take).  spmv_stream_kernel walks a chain of dependent
// memory round trips per row block -- block table -> row pointers -> (col, val) -> gathers -> [LDS] -> row extents ->
// epilogue operands -> store -- with eight short-lived workgroups per CU to hide them; counters on a banded 10M matrix
// showed waves alive for 31 us each, 47 % of it in s_waitcnt (profiles/r05_stream_kernel_pmc_banded50k.json).  Here a
// workgroup stays and walks its blocks: while block t's gathers are in flight, block t + 1's entries, row extents and
// epilogue operands and block t + 2's extent word are already requested, so a trip costs one L2 round trip and one
// barrier.  Same lane-per-row sums in the same order (stream_rows_phase): the same bits as spmv_stream_kernel.
// `ext`: (r0, r1, k0, k1) per row block, so that a block's entry addresses do not wait for two more loads.
// Block partials: one slot per workgroup; the CSR grid's remaining slots (which the reductions walk) are zeroed.
constexpr int STREAM_PIPE_WGS_PER_CU = 4;
template <int MODE, bool INIT = false, int TAG = 0>
__global__ __launch_bounds__(TPB) void spmv_stream_pipe_kernel(
    CsrView A, const double *__restrict__ xin, const int4 *__restrict__ ext, int nblk, int per_xcd, int remap, int relaxed,
    int stream_slots, EpiArgs e) {
  __shared__ double prod[2][BLOCK_NNZ];
  __shared__ double red[6][TPB / WAVE];
  const int tid = threadIdx.x;
  Acc3 acc = acc3_zero();
  // remap: workgroup b runs on XCD b % 8 (round-robin dispatch) and walks that XCD's contiguous eighth of the blocks
  const int first = remap ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int stride = remap ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int limit = remap ? per_xcd : nblk;
  const int xbase = remap ? (int)(blockIdx.x & (NUM_XCD - 1)) * per_xcd : 0;
  auto block_of = [&](int t) { const int i = first + t * stride, b = xbase + i; return (i < limit && b < nblk) ? b : -1; };
  int4 x_n = make_int4(0, 0, 0, 0), x_nn = make_int4(0, 0, 0, 0);
  int c_n[UNROLL];
  double v_n[UNROLL];
  RowPre pre_n{0, 0, EpiOps{0.0, 0.0, 0.0}, 0.0};
  auto request_rest = [&](const int4 &x) {     // entries, row extents, epilogue operands of the block whose extent word is x
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      const int k = x.z + tid + i * TPB;
      const bool ok = k < x.w;
      c_n[i] = ok ? __builtin_nontemporal_load(A.col + k) : 0;
      v_n[i] = ok ? __builtin_nontemporal_load(A.val + k) : 0.0;
    }
    const int r = x.x + tid;
    if (r < x.y) {
      pre_n.rs = A.rowptr[r];
      pre_n.re = A.rowptr[r + 1];
      pre_n.ops = epi_load<MODE>(e, r);
      if (INIT) pre_n.init = e.init[r];
    }
  };
  int b = block_of(0), b_next = block_of(1);
  if (b >= 0) x_n = ext[b];
  if (b_next >= 0) x_nn = ext[b_next];
  if (b >= 0) request_rest(x_n);
  int buf = 0;
  for (int t = 0; b >= 0; ++t) {
    const int4 cur = x_n;
    const RowPre pre = pre_n;
    double xv[UNROLL], vv[UNROLL];
    // one wait for the block's entries (requested a trip ago), then the eight gathers back to back: the compiler's own
    // waits sat in front of every gather (vmcnt(1): two gathers in flight per wave; round 6, as in sj_kernels.hpp)
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      const int k = cur.z + tid + i * TPB;
      xv[i] = (k < cur.w) ? xin[PDHG_COLOFF(c_n[i])] : 0.0;
      vv[i] = v_n[i];
    }
    // behind this block's gathers: the next block's entries and row data (its extent word came in a trip ago), and the
    // extent word of the block after it
    x_n = x_nn;
    const int b_next2 = block_of(t + 2);
    if (b_next2 >= 0) x_nn = ext[b_next2];
    if (b_next >= 0) request_rest(x_n);
    double *pr = prod[buf];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      const int k = tid + i * TPB;
      if (cur.z + k < cur.w) pr[k] = vv[i] * xv[i];
    }
    __syncthreads();
    // (the other buffer is written next trip; this one again two trips on, behind the next trip's barrier)
    stream_rows_phase<MODE, INIT, true, false>(A, cur.x, cur.y, cur.z, e, relaxed, acc, pr, &pre);
    buf ^= 1;
    b = b_next;
    b_next = b_next2;
  }
  constexpr int NQ = ModeNQ<MODE>::value;
  if (NQ > 0) {
    block_sum_dd<NQ, TPB>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        e.partials[q * e.stride + blockIdx.x] = acc.hi[q];
        e.partials[e.lo_offset + q * e.stride + blockIdx.x] = acc.lo[q];
      }
    }
    for (int sl = (int)gridDim.x + (int)blockIdx.x * TPB + tid; sl < stream_slots; sl += (int)gridDim.x * TPB) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        e.partials[q * e.stride + sl] = 0.0;
        e.partials[e.lo_offset + q * e.stride + sl] = 0.0;
      }
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

// Variant for runs of 9 ... 32 entries (rows whose entries cluster in a few tiles): the run sums travel lane to lane.
// Lane i takes lane i - 1's partial sum with one DPP move per register half (wave_shr:1 -- a VALU operation; CDNA keeps
// GFX9's whole-wave shifts) and adds its own product, position by position along the runs, so that a run's LAST lane
// ends up with ((acc + p0) + p1) + ... -- the sequential order -- and stores it.  The shuffle loop above costs three
// LDS-pipe operations per position; the LDS scratch of the next variant costs the second workgroup per CU its LDS
// (8 x 1 221 rows + scratch > 80 KiB: a 10M-row matrix with runs of 10 ran in two residency rounds at 1.47 ms; 1.03 so).
// On runs of 1 - 2 (config S) the shuffle loop is faster (0.76 ms against 0.92), hence a variant and not a replacement.
__device__ __forceinline__ unsigned wave_shr1(unsigned x, unsigned lane0) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138, 0xF, 0xF, false);   // lane i <- lane i - 1
}
__device__ __forceinline__ unsigned wave_shl1(unsigned x, unsigned lane63) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)lane63, (int)x, 0x130, 0xF, 0xF, false);  // lane i <- lane i + 1
}
__device__ __forceinline__ double wave_shr1(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void tiled_chunk_scan(double *acc, unsigned p, double v, double xv, int tile_shift, int lane) {
  const bool valid = p != TW_PAD;
  // padding lanes get pairwise different pseudo-rows (no real row_local reaches 0xFFFFFF00: it is < 2^31): runs of one, never stored
  const unsigned row = valid ? (p >> tile_shift) : (0xFFFFFF00u | (unsigned)lane);
  const double prod = v * xv;
  const bool head = wave_shr1(row, ~row) != row;
  const bool tail = wave_shl1(row, ~row) != row;
  const unsigned long long hmask = __ballot(head);                                       // (lane 0 is a head)
  const int pos = lane - (63 - __clzll((long long)(hmask & ((2ull << lane) - 1ull))));   // place in the run
  double s = (head && valid) ? acc[row] + prod : prod;
  for (int j = 1; __any(pos >= j); ++j) {
    const double left = wave_shr1(s);
    if (pos == j) s = left + prod;
  }
  if (tail && valid) acc[row] = s;
}

// Variant 4: variant 3 for chunks that hold a run of more than three entries, variant 0's shuffle loop for the others.
// Which of 3 and 4 is faster depends on the memory regime, not on the run lengths: rows of log-normal length over uniform
// columns (10M: most chunks hold short runs) 0.89 ms with 3, 0.76 with 4; the transpose of the clustered matrix (its
// gathers hit a handful of lines per cell) 0.49 with 3, 0.59 with 4 -- so pdhg_create TIMES both on the matrix at hand
// (tune_tiled_variant).  Same sequential sums either way: not a bit depends on the choice.
constexpr int TW_HYBRID_RUN = 3;
__device__ __forceinline__ void tiled_chunk_hybrid(double *acc, unsigned p, double v, double xv, int tile_shift, int lane) {
  const bool valid = p != TW_PAD;
  const unsigned row = valid ? (p >> tile_shift) : (0xFFFFFF00u | (unsigned)lane);
  const double prod = v * xv;
  const bool head = wave_shr1(row, ~row) != row;
  const bool tail = wave_shl1(row, ~row) != row;
  const unsigned long long hmask = __ballot(head);
  const int pos = lane - (63 - __clzll((long long)(hmask & ((2ull << lane) - 1ull))));
  if (!__any(pos >= TW_HYBRID_RUN)) {
    const bool h = head && valid;
    double s0 = h ? acc[row] : 0.0;
    for (int j = 0; j < TW_HYBRID_RUN; ++j) {
      const double pj = __shfl_down(prod, j, WAVE);
      const unsigned rj = __shfl_down(row, j, WAVE);
      const bool take = h && (lane + j < WAVE) && (rj == row);
      if (!__any(take)) break;
      if (take) s0 = s0 + pj;
    }
    if (h) acc[row] = s0;
    return;
  }
  double s = (head && valid) ? acc[row] + prod : prod;
  for (int j = 1; __any(pos >= j); ++j) {
    const double left = wave_shr1(s);
    if (pos == j) s = left + prod;
  }
  if (tail && valid) acc[row] = s;
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

// Relaxed-order variant (PDHG_ROW_ORDER=relaxed, the default): same-row runs of up to
// TW_STRICT_RUN entries inside a chunk are still added strictly left to right by the run
// head (so rows whose entries scatter over the tiles -- every row of a random LP -- stay
// bit-identical to the CPU loops); when a chunk holds a longer run (hub rows, dense blocks)
// the whole chunk is reduced by a segmented shuffle tree instead: log2(64) steps whatever
// the run lengths, a fixed order (reproducible), |result - sequential| <= 1e-13 * sum |a x|.
// This is what lets matrices with hub rows use the sweep at all: in strict order one lane adds
// a 1 500-entry row's ~90 products per tile one after the other.
constexpr int TW_STRICT_RUN = 8;
__device__ __forceinline__ void tiled_chunk_relaxed(double *acc, unsigned p, double v, double xv,
                                                    int tile_shift, int lane) {
  const bool valid = p != TW_PAD;
  // padding lanes get pairwise different pseudo-rows (no real row_local reaches 0xFFFFFF00: it is < 2^31)
  const unsigned row = valid ? (p >> tile_shift) : (0xFFFFFF00u | (unsigned)lane);
  const double prod = v * xv;
  const unsigned rowp = __shfl_up(row, 1, WAVE);
  const bool head = valid && (lane == 0 || rowp != row);
  const unsigned long long hmask = __ballot(head);
  const unsigned long long vmask = __ballot(valid);
  const unsigned long long above = (lane == WAVE - 1) ? 0ull : (hmask >> (lane + 1));
  const int nvalid = __popcll(vmask);                                   // valid lanes are a prefix
  const int len = head ? (above ? __ffsll((long long)above) : (nvalid - lane)) : 0;
  if (!__any(len > TW_STRICT_RUN)) {
    double s = head ? acc[row] : 0.0;
    for (int j = 0; j <= TW_STRICT_RUN; ++j) {
      const double pj = __shfl_down(prod, j, WAVE);
      const unsigned rj = __shfl_down(row, j, WAVE);
      const bool take = head && (lane + j < WAVE) && (rj == row);
      if (!__any(take)) break;
      if (take) s = s + pj;
    }
    if (head) acc[row] = s;
    return;
  }
  // a chunk with a long run: the short runs beside it are still added left to right ...
  double s = head ? acc[row] : 0.0;
  for (int j = 0; j <= TW_STRICT_RUN; ++j) {
    const double pj = __shfl_down(prod, j, WAVE);
    const unsigned rj = __shfl_down(row, j, WAVE);
    const bool take = head && len <= TW_STRICT_RUN && (lane + j < WAVE) && (rj == row);
    if (!__any(take)) break;
    if (take) s = s + pj;
  }
  // ... the long ones by a segmented shuffle tree
  double val = valid ? prod : 0.0;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const unsigned rd = __shfl_down(row, d, WAVE);
    const double vd = __shfl_down(val, d, WAVE);
    const bool same = (lane + d < WAVE) && (rd == row);
    if (!__any(same)) break;             // runs are contiguous: none at distance d, none beyond
    if (same) val = val + vd;
  }
  if (head) acc[row] = (len <= TW_STRICT_RUN) ? s : acc[row] + val;
}

// CH: how a 64-entry chunk is accumulated -- 0 lane shuffles (strict order), 1 LDS scratch
// (strict order, long runs), 2 relaxed (see tiled_chunk_relaxed), 3 lane-to-lane (strict order, runs of 9 ... 32), 4 = 3 with the shuffle loop for chunks of short runs
template <int MODE, int CH>
__global__ __launch_bounds__(TW_WPB * WAVE) void spmv_tiled_kernel(
    const int2 *__restrict__ wave_rows, const int *__restrict__ step_ptr,
    const int *__restrict__ wave_step_off, const int *__restrict__ step_tile,
    const int *__restrict__ wg_step_off, int nwaves, int tile_shift, int TW_ROWS,
    const unsigned *__restrict__ pk, const double *__restrict__ tv,
    const double *__restrict__ xin, EpiArgs e, int ngroups, int per_xcd) {
  constexpr int TW_THREADS = TW_WPB * WAVE;
  constexpr int U = TW_U;   // 64-entry chunks held in registers per (wave, tile)
  constexpr int D = 3;      // entry loads run D tiles ahead of the accumulate
  constexpr int R = D + 1;  // register ring (statically indexed: the tile loop is unrolled R times)
  extern __shared__ double tw_lds[];  // [TW_WPB][TW_ROWS] accumulators, then red[6][TW_WPB]
  double(*red)[TW_WPB] = reinterpret_cast<double(*)[TW_WPB]>(tw_lds + TW_WPB * TW_ROWS);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
  double *scratch = tw_lds + TW_WPB * TW_ROWS + 6 * TW_WPB + wid * WAVE;   // CH == 1 only: 64 doubles per wave
  // XCD remap (per_xcd > 0; the grid is 8 x per_xcd): workgroup b runs on XCD b % 8 and takes row group
  // (b % 8) * per_xcd + b / 8, so that an XCD's resident workgroups own CONSECUTIVE row groups.  All the same to a matrix
  // whose rows scatter over every tile; on a wide band (10M, +-3M columns) the row groups resident together start their
  // sweeps up to 64 tiles apart, and dealt round robin every XCD's L2 saw all of those tiles at once.
  const int g = per_xcd > 0 ? (int)(blockIdx.x & (NUM_XCD - 1)) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const bool wg_live = g < ngroups;
  const int w = __builtin_amdgcn_readfirstlane(g * TW_WPB + wid);
  const bool live = wg_live && w < nwaves;
  Acc3 acc3 = acc3_zero();
  double *acc = tw_lds + wid * TW_ROWS;
  int2 rr = make_int2(0, 0);
  if (live) rr = wave_rows[w];
  // e.init (column-chunk passes of a shard group's A_p xbar, dist.hpp): the row sums of the chunks before this one
  if (e.init) {
    const int nr = live ? rr.y - rr.x : 0;
    for (int r = lane; r < TW_ROWS; r += WAVE) acc[r] = r < nr ? e.init[rr.x + r] : 0.0;
  } else {
    for (int r = lane; r < TW_ROWS; r += WAVE) acc[r] = 0.0;
  }
  // This workgroup's step list: one step = one column tile, or a slice of a
  // heavy tile (cells are cut on the host so that no wave has more than
  // TW_U*64 entries in a step); tiles in which none of the 8 waves has an entry
  // are skipped.  ntiles below is the number of STEPS of this workgroup.
  const int ntiles = wg_live ? wg_step_off[g + 1] - wg_step_off[g] : 0;
  const int *stile = step_tile + (wg_live ? wg_step_off[g] : 0);
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
        const double *xt = xin + (size_t)tl[s];      // the step table holds the tile's first column
        // 1. gathers for tile t (entries requested D steps ago).  Issued BEFORE
        //    the prefetch: a wave's loads return in order, so the L2-latency
        //    gathers must not queue behind HBM-latency streaming loads.
        //    (Gathering one tile ahead was measured slower: it widens the L2
        //    working window of the sweep.)
#pragma unroll
        for (int i = 0; i < U; ++i) xv[i] = (p[s][i] != TW_PAD) ? xt[p[s][i] & cmask] : 0.0;
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
            if (CH == 1) tiled_chunk_scratch(acc, scratch, p[s][i], v[s][i], xv[i], tile_shift, lane);
            else if (CH == 3) tiled_chunk_scan(acc, p[s][i], v[s][i], xv[i], tile_shift, lane);
            else if (CH == 4) tiled_chunk_hybrid(acc, p[s][i], v[s][i], xv[i], tile_shift, lane);
            else if (CH == 2) tiled_chunk_relaxed(acc, p[s][i], v[s][i], xv[i], tile_shift, lane);
            else tiled_chunk(acc, p[s][i], v[s][i], xv[i], tile_shift, lane);
          }
        }
        for (int kb = ks[s] + U * WAVE; kb < ke[s]; kb += WAVE) {  // cells beyond the register window
          const int k = kb + lane;
          const bool ok = k < ke[s];
          const unsigned pp = ok ? __builtin_nontemporal_load(pk + k) : TW_PAD;
          const double vv = ok ? __builtin_nontemporal_load(tv + k) : 0.0;
          const double xx = ok ? xt[pp & cmask] : 0.0;
          if (CH == 1) tiled_chunk_scratch(acc, scratch, pp, vv, xx, tile_shift, lane);
          else if (CH == 3) tiled_chunk_scan(acc, pp, vv, xx, tile_shift, lane);
          else if (CH == 4) tiled_chunk_hybrid(acc, pp, vv, xx, tile_shift, lane);
          else if (CH == 2) tiled_chunk_relaxed(acc, pp, vv, xx, tile_shift, lane);
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
    block_sum_dd<NQ, TW_THREADS>(acc3, red);
    if (threadIdx.x == 0 && wg_live) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        e.partials[q * e.stride + g] = acc3.hi[q];
        e.partials[e.lo_offset + q * e.stride + g] = acc3.lo[q];
      }
    }
  }
}

// Rows longer than BLOCK_NNZ: split into LONG_CHUNK (= BLOCK_NNZ) pieces, one workgroup each.
// A chunk is read like a row block of the stream kernel -- UNROLL independent (col, val) load
// pairs per lane, then UNROLL gathers -- and summed by a fixed tree: lane-local pairs, wave
// shuffles, the waves left to right.  (Round 2 used 8192-entry chunks walked in 8 dependent
// steps of 4 loads: ~11 us of latency per chunk on the critical path of a small LP's
// iteration, and 123 workgroups for PageRank's 1M-entry row.)  Thread 0 returns the chunk's sum.
// (two halves, like a row block: the one-launch trial kernel requests the entries before its grid
// barrier.  A chunk uses the row block's register set: k0 / k1 are its extent, r0 / r1 unused.)
__device__ __forceinline__ void long_chunk_load(const CsrView &A, int r, int off, StreamRegs &g) {
  g.r0 = g.r1 = r;
  g.k0 = A.rowptr[r] + off;
  g.k1 = min(g.k0 + LONG_CHUNK, A.rowptr[r + 1]);
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) {
    const int k = g.k0 + threadIdx.x + i * TPB;
    const bool ok = k < g.k1;
    g.cidx[i] = ok ? __builtin_nontemporal_load(A.col + k) : 0;
    g.v[i] = ok ? __builtin_nontemporal_load(A.val + k) : 0.0;
  }
}
template <bool COH = false>
__device__ __forceinline__ double long_chunk_finish(const double *xin, const StreamRegs &g, double (*red)[TPB / WAVE]) {
  double acc[3] = {0.0, 0.0, 0.0};
  double p[UNROLL];
  // all eight gathers first, the products behind a scheduling barrier: left to itself the compiler, short of registers in
  // the persistent kernels, reused ONE register pair for the eight gathered values -- load, wait, multiply, eight times
  // over: eight dependent L2 round trips per chunk (round 6, seen in steps_kernel's ISA)
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) {
    const int k = g.k0 + threadIdx.x + i * TPB;
    if (COH) p[i] = ldc<true>(xin + PDHG_COLOFF(g.cidx[i]));     // unconditional (column 0 for a lane without an entry): see stream_block_finish
    else p[i] = (k < g.k1) ? xin[PDHG_COLOFF(g.cidx[i])] : 0.0;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) {
    const int k = g.k0 + threadIdx.x + i * TPB;
    p[i] = (k < g.k1) ? g.v[i] * p[i] : 0.0;
  }
  static_assert(UNROLL == 8 && LONG_CHUNK == UNROLL * TPB, "the chunk sum below is written for 8 products per lane");
  acc[0] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  block_sum<1, TPB>(acc, red);
  return acc[0];
}
__device__ __forceinline__ double long_chunk_body(const CsrView &A, const double *xin, int r, int off,
                                                  double (*red)[TPB / WAVE]) {
  StreamRegs g;
  long_chunk_load(A, r, off, g);
  return long_chunk_finish(xin, g, red);
}

template <int TAG = 0>
__global__ __launch_bounds__(TPB) void spmv_long_partial_kernel(
    CsrView A, const double *__restrict__ xin, const int *__restrict__ chunk_row,
    const int *__restrict__ chunk_off, double *__restrict__ chunk_partial) {
  __shared__ double red[3][TPB / WAVE];
  const int c = blockIdx.x;
  const double s = long_chunk_body(A, xin, chunk_row[c], chunk_off[c], red);
  if (threadIdx.x == 0) chunk_partial[c] = s;
}

// One WAVE per long row: lane l adds chunk partials l, l + 64, ... in ascending order, then
// the wave's shuffle tree (a fixed order: reproducible; a 1M-entry row has ~490 partials and
// one lane adding them one after the other was 10 us), lane 0 runs the epilogue and writes the
// row's contribution to the reductions into the row's OWN slot (slot_base + l): whichever
// workgroup ends up finishing a long row -- a block of the separate kernel below, or, in the
// one-launch trial, the workgroup that completed the row's last chunk -- the slots hold the
// same bits.  Called by all 64 lanes of one wave.  AGENT: write-through stores (trial kernel).
template <int MODE, bool AGENT, bool COH = false>
__device__ __forceinline__ void long_final_row(int l, const int *long_row, const int *long_chunk_ptr,
                                               const double *chunk_partial, const EpiArgs &e, int slot_base,
                                               const double *init = nullptr) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int c0 = long_chunk_ptr[l], c1 = long_chunk_ptr[l + 1];
  double s = 0.0;
  for (int c = c0 + lane; c < c1; c += WAVE) s = s + chunk_partial[c];
  s = wave_sum(s);
  if (lane == 0) {
    if (init) s = init[long_row[l]] + s;       // a later column-chunk pass: the row's sum so far
    Acc3 acc = acc3_zero();
    row_epilogue<MODE, COH>(e, long_row[l], s, acc);
    constexpr int NQ = ModeNQ<MODE>::value;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double *dst = e.partials + q * e.stride + slot_base + l;
      if (AGENT) {
        __hip_atomic_store(dst, acc.hi[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + e.lo_offset, acc.lo[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *dst = acc.hi[q];
        dst[e.lo_offset] = acc.lo[q];
      }
    }
  }
}

constexpr int LONG_ROWS_PER_WG = TPB / WAVE;
template <int MODE>
__global__ __launch_bounds__(TPB) void spmv_long_final_kernel(
    const int *__restrict__ long_row, const int *__restrict__ long_chunk_ptr,
    int nlong, const double *__restrict__ chunk_partial, EpiArgs e,
    int slot_base) {
  const int l = blockIdx.x * LONG_ROWS_PER_WG + threadIdx.x / WAVE;
  if (l < nlong) long_final_row<MODE, false>(l, long_row, long_chunk_ptr, chunk_partial, e, slot_base, e.init);
}

}  // namespace
