// vector_kernels.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// Elementwise PDHG kernels (primal step, xbar, interaction, accept/average) and the final reductions.
#pragma once

namespace {

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

// Workgroup `bid` of `nb` (grid-stride; elementwise, so the distribution does not matter for the bits).
// COH: A'y was written by other compute units since this one last read it (multi-step trial kernel): coherent loads
template <bool HAS_Q, bool WRITE_XBAR, bool COH = false>
__device__ __forceinline__ void primal_body(
    int n, const double *x, const double *c, const double *aty, const double *qx,
    const double *lb, const double *ub, double tau, double theta, double *x_next, double *xbar,
    double avg_w, double *sum_x, int bid, int nb) {
  // sum_x != nullptr: the accept of the previous iteration left its K7 to this kernel
  // (sum_x += avg_w * x, x being the iterate accepted then; same two roundings)
  const int npair = n >> 1;
  const int stride = nb * TPB;
  for (int p = bid * TPB + threadIdx.x; p < npair; p += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[p];
    if (sum_x) {
      double2 sv = reinterpret_cast<double2 *>(sum_x)[p];
      const double t0 = xv.x * avg_w, t1 = xv.y * avg_w;
      sv.x = sv.x + t0;
      sv.y = sv.y + t1;
      reinterpret_cast<double2 *>(sum_x)[p] = sv;
    }
    const double2 cv = reinterpret_cast<const double2 *>(c)[p];
    double2 av;
    if (COH) { av.x = ldc<true>(aty + 2 * p); av.y = ldc<true>(aty + 2 * p + 1); }
    else av = reinterpret_cast<const double2 *>(aty)[p];
    const double2 lv = reinterpret_cast<const double2 *>(lb)[p];
    const double2 uv = reinterpret_cast<const double2 *>(ub)[p];
    double2 qv = {0.0, 0.0};
    if (HAS_Q) qv = reinterpret_cast<const double2 *>(qx)[p];
    double2 xn, xb;
    primal_one<HAS_Q, WRITE_XBAR>(xv.x, cv.x, av.x, qv.x, lv.x, uv.x, tau, theta, xn.x, xb.x);
    primal_one<HAS_Q, WRITE_XBAR>(xv.y, cv.y, av.y, qv.y, lv.y, uv.y, tau, theta, xn.y, xb.y);
    {
      reinterpret_cast<double2 *>(x_next)[p] = xn;
      if (WRITE_XBAR) reinterpret_cast<double2 *>(xbar)[p] = xb;
    }
  }
  if ((n & 1) && bid == 0 && threadIdx.x == 0) {
    const int j = n - 1;
    double xn, xb;
    if (sum_x) {
      const double t = x[j] * avg_w;
      sum_x[j] = sum_x[j] + t;
    }
    primal_one<HAS_Q, WRITE_XBAR>(x[j], c[j], ldc<COH>(aty + j), HAS_Q ? qx[j] : 0.0, lb[j], ub[j], tau, theta, xn, xb);
    x_next[j] = xn;
    if (WRITE_XBAR) xbar[j] = xb;
  }
}

template <bool HAS_Q, bool WRITE_XBAR>
__global__ __launch_bounds__(TPB) void primal_kernel(
    int n, const double *__restrict__ x, const double *__restrict__ c,
    const double *__restrict__ aty, const double *__restrict__ qx,
    const double *__restrict__ lb, const double *__restrict__ ub, double tau,
    double theta, double *__restrict__ x_next, double *__restrict__ xbar,
    double avg_w, double *__restrict__ sum_x) {
  primal_body<HAS_Q, WRITE_XBAR>(n, x, c, aty, qx, lb, ub, tau, theta, x_next, xbar, avg_w, sum_x, blockIdx.x, gridDim.x);
}

// xbar = x' + theta*(x' - x) on its own (Malitsky-Pock retries, pdhg.jl:590-601)
__device__ __forceinline__ void xbar_body(int n, const double *x, const double *x_next, double theta, double *xbar,
                                          int bid, int nb) {
  const int stride = nb * TPB;
  for (int j = bid * TPB + threadIdx.x; j < n; j += stride) {
    const double v = x_next[j];
    const double d = v - x[j];
    const double t = theta * d;
    xbar[j] = v + t;
  }
}
__global__ __launch_bounds__(TPB) void xbar_kernel(int n, const double *__restrict__ x,
                                                   const double *__restrict__ x_next,
                                                   double theta, double *__restrict__ xbar) {
  xbar_body(n, x, x_next, theta, xbar, blockIdx.x, gridDim.x);
}

// dx = x' - x  (for the QP interaction term 0.5*dx'Q dx, pdhg.jl:536-541)
__device__ __forceinline__ void diff_body(int n, const double *a, const double *b, double *out, int bid, int nb) {
  const int stride = nb * TPB;
  for (int j = bid * TPB + threadIdx.x; j < n; j += stride) out[j] = a[j] - b[j];
}
// the same with primal_body's element-to-thread mapping (pairs 2p, 2p + 1; the odd tail on block 0, thread 0):
// inside the one-launch trial kernel every thread then reads back only the x' it has just written itself
__device__ __forceinline__ void diff_pairs_body(int n, const double *a, const double *b, double *out, int bid, int nb) {
  const int npair = n >> 1;
  const int stride = nb * TPB;
  for (int p = bid * TPB + threadIdx.x; p < npair; p += stride) {
    out[2 * p] = a[2 * p] - b[2 * p];
    out[2 * p + 1] = a[2 * p + 1] - b[2 * p + 1];
  }
  if ((n & 1) && bid == 0 && threadIdx.x == 0) out[n - 1] = a[n - 1] - b[n - 1];
}
__global__ __launch_bounds__(TPB) void diff_kernel(int n, const double *__restrict__ a,
                                                   const double *__restrict__ b,
                                                   double *__restrict__ out) {
  diff_body(n, a, b, out, blockIdx.x, gridDim.x);
}

// Reductions over the replicated n-vectors (row-partitioned form, after the
// all-reduce delivered A'y'):  dx.(A'y'-A'y), dx^2, (A'y'-A'y)^2.
__global__ __launch_bounds__(TPB) void interaction_kernel(
    int n, const double *__restrict__ x, const double *__restrict__ x_next,
    const double *__restrict__ aty, const double *__restrict__ aty_next,
    double *__restrict__ partials, int pstride) {
  __shared__ double red[6][TPB / WAVE];
  Acc3 acc = acc3_zero();
  const int stride = gridDim.x * TPB;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += stride) {
    const double dx = x_next[j] - x[j];
    const double dd = aty_next[j] - aty[j];
    dd_add(acc.hi[0], acc.lo[0], dx * dd);
    dd_add(acc.hi[1], acc.lo[1], dx * dx);
    dd_add(acc.hi[2], acc.lo[2], dd * dd);
  }
  block_sum_dd<3, TPB>(acc, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      partials[q * pstride + blockIdx.x] = acc.hi[q];
      partials[(3 + q) * pstride + blockIdx.x] = acc.lo[q];
    }
  }
}

// dot(a, b) partials (QP term), double-double: block `bid` of `nb` writes partials[bid] (hi) and partials[nb + bid] (lo)
__device__ __forceinline__ void dot_body(int n, const double *a, const double *b, double *partials, int bid, int nb,
                                         double (*red)[TPB / WAVE], bool agent_store) {
  Acc3 acc = acc3_zero();
  const int stride = nb * TPB;
  for (int j = bid * TPB + threadIdx.x; j < n; j += stride) dd_add(acc.hi[0], acc.lo[0], a[j] * b[j]);
  block_sum_dd<1, TPB>(acc, red);
  if (threadIdx.x == 0) {
    if (agent_store) {
      __hip_atomic_store(partials + bid, acc.hi[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(partials + nb + bid, acc.lo[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      partials[bid] = acc.hi[0];
      partials[nb + bid] = acc.lo[0];
    }
  }
}
__global__ __launch_bounds__(TPB) void dot_kernel(int n, const double *__restrict__ a,
                                                  const double *__restrict__ b,
                                                  double *__restrict__ partials) {
  __shared__ double red[6][TPB / WAVE];
  dot_body(n, a, b, partials, blockIdx.x, gridDim.x, red, false);
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
  const double *ptr[5];     // hi parts of the block partials
  const double *ptr_lo[5];  // lo parts
  int count[5];
  double *out;     // 5 doubles (host-mapped or device)
};
// The five sums, each a double-double sum of its block partials: quantity q is summed by three
// "virtual waves" 3q..3q+2 (strided per-lane sums, wave shuffle tree, then the three wave totals
// left to right -- a fixed order, although the double-double result does not depend on it except
// with negligible probability).  A workgroup of NW waves runs virtual wave v on wave v mod NW: the
// separate final kernels have 16 waves (one virtual wave each), the one-launch trial kernel 4 --
// the arithmetic, hence the bits, are the same.  Thread 0 ends with all five, rounded to one
// double each, in res[].
#ifndef PDHG_FINAL_BATCH
#define PDHG_FINAL_BATCH 4
#endif
template <int NW>
__device__ __forceinline__ void final_reduce_body(const FinalSpec &sp, double (&res)[5]) {
  __shared__ double wsum[16], wsum_lo[16];
  const int wid = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  constexpr int VPW = (15 + NW - 1) / NW;        // virtual waves per physical wave
  // load pairs in flight per lane and virtual wave (registers: 2 * VPW * B doubles).  The 4-wave form (trial kernel) takes 5:
  // a one-launch LP has up to ~1000 block partials per quantity, i.e. 5 per lane and virtual wave -- one trip to memory
  constexpr int B = NW >= 16 ? PDHG_FINAL_BATCH : PDHG_FINAL_BATCH + 1;
  // The partials come from other CUs' stores: every load is a trip to memory.  All of a
  // physical wave's first B loads per virtual wave are requested before anything is added
  // (one round trip instead of VPW).
  double th[VPW][B], tl[VPW][B];
#pragma unroll
  for (int j = 0; j < VPW; ++j) {
    const int v = wid + j * NW;
    const int q = v < 15 ? v / 3 : 0, sub = v % 3;
    const double *p = sp.ptr[q], *pl = sp.ptr_lo[q];
    const int cnt = v < 15 ? sp.count[q] : 0;
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int i = sub * WAVE + lane + u * 3 * WAVE;
      th[j][u] = i < cnt ? p[i] : 0.0;
      tl[j][u] = i < cnt ? pl[i] : 0.0;
    }
  }
#pragma unroll
  for (int j = 0; j < VPW; ++j) {
    const int v = wid + j * NW;
    if (v < 15) {                                 // wave-uniform
      const int q = v / 3, sub = v % 3;
      const double *p = sp.ptr[q], *pl = sp.ptr_lo[q];
      const int cnt = sp.count[q];
      double hi = 0.0, lo = 0.0;
#pragma unroll
      for (int u = 0; u < B; ++u)
        if (sub * WAVE + lane + u * 3 * WAVE < cnt) dd_add_dd(hi, lo, th[j][u], tl[j][u]);
      for (int i0 = sub * WAVE + lane + B * 3 * WAVE; i0 < cnt; i0 += B * 3 * WAVE) {
        double rh[B], rl[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const int i = i0 + u * 3 * WAVE;
          rh[u] = i < cnt ? p[i] : 0.0;
          rl[u] = i < cnt ? pl[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < B; ++u)
          if (i0 + u * 3 * WAVE < cnt) dd_add_dd(hi, lo, rh[u], rl[u]);
      }
      wave_sum_dd(hi, lo);
      if (lane == WAVE - 1) { wsum[v] = hi; wsum_lo[v] = lo; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      double hi = wsum[3 * k], lo = wsum_lo[3 * k];
      dd_add_dd(hi, lo, wsum[3 * k + 1], wsum_lo[3 * k + 1]);
      dd_add_dd(hi, lo, wsum[3 * k + 2], wsum_lo[3 * k + 2]);
      res[k] = hi + lo;
    }
  }
}

// The same second stage for an LP (quantity 4, dx'Q dx, is absent) with ONE WAVE PER QUANTITY: wave q adds quantity q,
// B load pairs per lane in flight at a time -- a handful of registers (the general form above keeps 2 * 4 * 5 doubles
// in flight per lane and spills inside the multi-step kernel: 11.4 us from "barrier complete" to "decision known" on a
// 40-workgroup grid, 2.9 us with this one).  Exactly rounded double-double sums: the same bits whatever the grouping.
// Called by a 4-wave workgroup; lds8: 8 doubles of LDS.
__device__ __forceinline__ bool final_reduce_lp_ok(const FinalSpec &sp) { return sp.count[4] == 0; }
template <int B>
__device__ __forceinline__ void final_reduce_lp(const FinalSpec &sp, double (&res)[5], double *lds8) {
  const int wid = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const double *p = sp.ptr[wid], *pl = sp.ptr_lo[wid];
  const int cnt = sp.count[wid];
  double hi = 0.0, lo = 0.0;
  for (int base = 0; base < cnt; base += B * WAVE) {         // wave-uniform trip count
    double h[B], l[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int i = base + lane + u * WAVE;
      h[u] = i < cnt ? p[i] : 0.0;
      l[u] = i < cnt ? pl[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < B; ++u)
      if (base + lane + u * WAVE < cnt) dd_add_dd(hi, lo, h[u], l[u]);
  }
  wave_sum_dd(hi, lo);
  if (lane == WAVE - 1) { lds8[wid] = hi; lds8[4 + wid] = lo; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) res[k] = lds8[k] + lds8[4 + k];
    res[4] = 0.0;
  }
}

__global__ __launch_bounds__(FINAL_TPB) void final_reduce_kernel(FinalSpec sp) {
  double res[5];
  final_reduce_body<FINAL_TPB / WAVE>(sp, res);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) sp.out[k] = res[k];
  }
}

}  // namespace
