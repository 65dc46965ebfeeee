// eval_kernels.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// Evaluation branch on the device (N1): fused residual/objective reductions, distances, trust-region probes.
#pragma once

namespace {

// ============================================================ evaluation branch (N1)
// Evaluation-cadence kernels (every termination_evaluation_frequency
// iterations): plain SpMVs into temporaries followed by elementwise kernels
// with multi-quantity sum/max reductions.  Simplicity over fusion here: the
// extra vector passes are noise at this cadence.
constexpr int EV_MAXQ = 32;            // quantities one evaluation reduction may carry (partials: EV_MAXQ x grid)

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

// Wave sums of N quantities at once (N a power of two, 8 ... 64), round 6.  One tree per quantity (wave_sum_dpp) is six
// dependent steps of two DPP moves and an add EACH: 30 quantities x 3 problems were 4.8 us of a 21 us probe pass of the
// batched trust-region search (NOTEBOOK 10.4).  Here the lanes SHARE the work: at the level that pairs lanes l and l ^ m,
// a lane keeps half of the quantities it still holds (bit m of l clear: the lower half) and hands the other half to its
// partner, so the six levels cost N/2 + N/4 + ... exchanges instead of 6 N.  Every quantity is still summed over the same
// balanced tree of adjacent lanes as wave_sum_dpp builds (pairs, fours, ..., rows, row pairs, halves), each node the sum of
// the same two operands -- floating-point addition commutes, so THE BITS are those of the one-tree-per-quantity form.
// Afterwards lane l holds the wave's total of quantity wave_split_index<N>(l) in v[0].
template <int N>
__device__ __forceinline__ int wave_split_index(int lane) {
  int q = 0, h = N >> 1;
#pragma unroll
  for (int m = 1; h >= 1; m <<= 1, h >>= 1) q += (lane & m) ? h : 0;
  return q;
}
template <int N, int H, int M>
struct WaveSplit {
  static __device__ __forceinline__ void run(double (&v)[N], int lane) {
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const double give = up ? v[j] : v[j + H];
      const double keep = up ? v[j + H] : v[j];
      v[j] = keep + __shfl_xor(give, M, WAVE);
    }
    WaveSplit<N, H / 2, M * 2>::run(v, lane);
  }
};
template <int N, int M>
struct WaveSplit<N, 0, M> {       // one quantity left per lane: plain butterflies over the remaining levels
  static __device__ __forceinline__ void run(double (&v)[N], int) {
#pragma unroll
    for (int m = M; m < WAVE; m <<= 1) v[0] = v[0] + __shfl_xor(v[0], m, WAVE);
  }
};
constexpr int pow2_at_least(int n) { return n <= 8 ? 8 : (n <= 16 ? 16 : (n <= 32 ? 32 : 64)); }

// partials[q*stride + blockIdx.x]: q < NS sums, then NM maxes (all maxes are of non-negative values)
template <int NS, int NM>
__device__ __forceinline__ void block_reduce_store(const RedAcc<NS, NM> &a, double *partials, int stride) {
  __shared__ double red[NS + NM][TPB / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#ifndef PDHG_NO_WAVE_SPLIT
  if constexpr (NS >= 8 && NS <= 64) {
    constexpr int N = pow2_at_least(NS);
    double v[N];
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = q < NS ? a.s[q] : 0.0;
    WaveSplit<N, N / 2, 1>::run(v, lane);
    const int q = wave_split_index<N>(lane);
    if (lane < N && q < NS) red[q][wid] = v[0];
  } else
#endif
  {
  // DPP trees (common.hpp): the wave's total ends in lane 63
#pragma unroll
  for (int q = 0; q < NS; ++q) { const double w = wave_sum_dpp(a.s[q]); if (lane == WAVE - 1) red[q][wid] = w; }
  }
#pragma unroll
  for (int q = 0; q < NM; ++q) { const double w = wave_max_nonneg_dpp(a.m[q]); if (lane == WAVE - 1) red[NS + q][wid] = w; }
  __syncthreads();
  static_assert(NS + NM <= TPB, "one thread per quantity combines the waves");
  if (threadIdx.x < NS + NM) {
    const int q = threadIdx.x;
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < TPB / WAVE; ++w) t = (q >= NS) ? fmax(t, red[q][w]) : t + red[q][w];
    partials[q * stride + blockIdx.x] = t;
  }
}

// Self-test of the check kernels' block reduction (eval_kernels.hpp: WaveSplit): NS quantities per lane, pseudo-random
// magnitudes over 80 binades, both signs, signed zeros -- every wave total through the shared form and through one DPP tree
// per quantity (wave_sum_dpp, the form of rounds 1-5) must have the SAME BITS.
template <int NS>
__global__ __launch_bounds__(TPB) void wave_sums_selftest_kernel(unsigned long long seed, unsigned long long *mismatches) {
  __shared__ unsigned long long ref[NS][TPB / WAVE], got[NS][TPB / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  constexpr int N = pow2_at_least(NS);
  double v[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(((size_t)blockIdx.x * TPB + threadIdx.x) * 64 + q + 1);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;      // splitmix64
    const double mant = 1.0 + (double)(x >> 12) * (1.0 / 4503599627370496.0);
    const int ex = (int)((x >> 3) % 80u) - 40;
    double val = ldexp(mant, ex);
    if (x & 1ull) val = -val;
    if ((x & 0xF0ull) == 0) val = (x & 2ull) ? -0.0 : 0.0;
    v[q] = q < NS ? val : 0.0;
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) { const double w = wave_sum_dpp(v[q]); if (lane == WAVE - 1) ref[q][wid] = (unsigned long long)__double_as_longlong(w); }
  WaveSplit<N, N / 2, 1>::run(v, lane);
  const int q = wave_split_index<N>(lane);
  if (lane < N && q < NS) got[q][wid] = (unsigned long long)__double_as_longlong(v[0]);
  __syncthreads();
  if (threadIdx.x < NS) {
    unsigned long long bad = 0;
    for (int w = 0; w < TPB / WAVE; ++w) bad += ref[threadIdx.x][w] != got[threadIdx.x][w];
    if (bad) atomicAdd(mismatches, bad);
  }
}

// Second stage of the evaluation kernels' block partials: quantity q (ns sums, then nm maxes) belongs to
// wave q mod 16, which adds / maxes its `count` partials in a fixed order (lane i takes i, i + 64, ..., eight
// loads in flight at a time; then the shuffle tree) -- no workgroup barrier between quantities, so 16 of them
// cost what one does.
// host_out != nullptr: the results also go straight into pinned host memory, followed by a checksum and the
// call's sequence number ([EV_HOST_CK], [EV_HOST_SEQ]); the host polls those instead of a device-to-host
// copy + stream synchronisation (20-30 us per round trip on this runtime, and the trust-region search makes
// five to eight round trips per call).  No system-scope fence: the words may land in any order, a read counts
// only when sequence number AND checksum match (as for the trial kernel's result word).
constexpr int EV_HOST_SLOTS = 32;             // >= SCAL_MAX (dist.hpp)
constexpr int EV_HOST_CK = EV_HOST_SLOTS, EV_HOST_SEQ = EV_HOST_SLOTS + 1;
constexpr unsigned long long EV_CHECK_SALT = 0xD1B54A32D192ED03ull;
__global__ __launch_bounds__(FINAL_TPB) void multi_final_kernel(const double *__restrict__ partials, int stride,
                                                                int count, int ns, int nm, double *__restrict__ out,
                                                                double *host_out, unsigned long long seq,
                                                                unsigned max_mask) {
  __shared__ double res[EV_HOST_SLOTS];
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const int k = ns + nm;
  for (int q = wave; q < k; q += FINAL_TPB / WAVE) {
    const double *p = partials + (size_t)q * stride;
    const bool is_max = max_mask ? ((max_mask >> q) & 1u) != 0 : q >= ns;   // max_mask: several kernels' partials side by side (pdhg_eval_point)
    double v = 0.0;
    for (int base = lane; base < count; base += 8 * WAVE) {
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (base + j * WAVE < count) ? p[base + j * WAVE] : 0.0;   // partials of maxes are >= 0
#pragma unroll
      for (int j = 0; j < 8; ++j) v = is_max ? fmax(v, t[j]) : v + t[j];
    }
    v = is_max ? wave_max_nonneg_dpp(v) : wave_sum_dpp(v);
    if (lane == WAVE - 1) { out[q] = v; res[q] = v; }
  }
  if (host_out) {
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long ck = EV_CHECK_SALT ^ seq ^ ((unsigned long long)k << 56);
      for (int q = 0; q < k; ++q) {
        host_out[q] = res[q];
        ck ^= (unsigned long long)__double_as_longlong(res[q]) * (2ull * (unsigned long long)q + 1ull);
      }
      host_out[EV_HOST_CK] = __longlong_as_double((long long)ck);
      host_out[EV_HOST_SEQ] = __longlong_as_double((long long)seq);
    }
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
// sums: 0 sum resid^2, 1 sum bound*rc, 2 sum x_o^2, 3 c_o.x_o, 4 sum bound-viol^2, 5 sum bound*rc_h,
//       6 x_o.(Q_o x_o)
// maxs: 0 max|resid|, 1 max|x_o|, 2 max bound viol, 3 max|resid_h|, 4 max|rc_h|, 5 max ray bound viol,
//       6 max|Q_o x_o|
// qx_s = Q_s x_s on the scaled point (NULL for an LP); Q_o x_o = D .* (Q_s x_s) because Q_s = D^-1 Q_o D^-1.
__global__ __launch_bounds__(TPB) void eval_cols_kernel(int n, const double *__restrict__ aty_s,
                                                        const double *__restrict__ qx_s,
                                                        const double *__restrict__ px, const double *__restrict__ D,
                                                        const double *__restrict__ c_o, const double *__restrict__ lb_o,
                                                        const double *__restrict__ ub_o, double *__restrict__ partials,
                                                        int stride) {
  RedAcc<7, 7> a;
  for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += gridDim.x * TPB) {
    const double d = D[j];
    const double aty = d * aty_s[j];
    const double xo = px[j] / d;
    const double qxo = qx_s ? d * qx_s[j] : 0.0;
    const double lb = lb_o[j], ub = ub_o[j];
    const bool lbf = isfinite(lb), ubf = isfinite(ub);
    // compute_reduced_costs_from_primal_gradient        iteration_stats_utils.jl:128-148
    // primal gradient Q x + c - A'y                        saddle_point.jl:1093-1100
    const double g = qx_s ? (qxo + c_o[j]) - aty : c_o[j] - aty;
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
    a.s[4] += lv * lv + uv * uv; a.s[5] += contribh; a.s[6] += xo * qxo;
    a.m[6] = fmax(a.m[6], fabs(qxo));
    a.m[0] = fmax(a.m[0], fabs(resid)); a.m[1] = fmax(a.m[1], fabs(xo)); a.m[2] = fmax(a.m[2], fmax(lv, uv));
    a.m[3] = fmax(a.m[3], fabs(residh)); a.m[4] = fmax(a.m[4], fabs(rch)); a.m[5] = fmax(a.m[5], rayv);
  }
  block_reduce_store<7, 7>(a, partials, stride);
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
// Per element the search needs three numbers, written here: thr, wd2 = w d^2 and gd = g d --
//   radius^2(t) = sum_{thr <= t} wd2 thr^2 + t^2 sum_{thr > t} wd2
//   value(t)    = sum g (clamp(z + t d) - z) = sum_{thr <= t} gd thr + t sum_{thr > t} gd
// (clamp(z + t d) - z = d min(t, thr) for a direction that moves towards its bound), so every
// probe of the search yields the value at its t as well and no pass over x, y, the bounds and g
// is needed once t* is known.
// sums: 0 c.x, 1 x.(A'y), 2 y.b, 3 sum_{thr=inf} wd2, 4 sum g^2 (in range),
//       5 sum wd2 (in range), 6 sum gd primal, 7 sum gd dual, 8 sum x^2, 9 sum y^2, 10 x.(Q x),
//       11 sum_{thr finite} wd2 thr^2, 12 / 13 sum_{thr finite} gd thr (primal / dual),
//       14 / 15 sum_{thr=inf} gd (primal / dual)      [11-15: the probe at t = max finite thr]
// maxs: 0 max finite thr (in range)
// qx_s = Q x at the point (NULL for an LP): the primal gradient is Q x + c - A'y.
constexpr int TR_SETUP_NS = 16;
__global__ __launch_bounds__(TPB) void tr_setup_kernel(int n, int m, int ne, const double *__restrict__ px,
                                                       const double *__restrict__ py, const double *__restrict__ aty_s,
                                                       const double *__restrict__ qx_s,
                                                       const double *__restrict__ ax_s, const double *__restrict__ c_s,
                                                       const double *__restrict__ b_s, const double *__restrict__ lb_s,
                                                       const double *__restrict__ ub_s, double wp, double wd, int range,
                                                       double *__restrict__ gdv, double *__restrict__ wd2v,
                                                       double *__restrict__ thr, double *__restrict__ partials,
                                                       int stride) {
  RedAcc<TR_SETUP_NS, 1> a;
  const int tid = blockIdx.x * TPB + threadIdx.x, st = gridDim.x * TPB;
  for (int k = tid; k < n + m; k += st) {
    const bool primal = k < n;
    const int i = primal ? k : k - n;
    double z, g, lo, hi, w;
    if (primal) {
      z = px[i]; lo = lb_s[i]; hi = ub_s[i]; w = wp;
      if (qx_s) { g = (qx_s[i] + c_s[i]) - aty_s[i]; a.s[10] += z * qx_s[i]; }
      else g = c_s[i] - aty_s[i];
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
    const double wd2 = w * d * d, gd = g * d;
    gdv[k] = gd; wd2v[k] = wd2; thr[k] = t;
    if (in_range) {
      a.s[4] += g * g;
      a.s[5] += wd2;
      if (primal) a.s[6] += gd; else a.s[7] += gd;
      if (isinf(t)) {
        a.s[3] += wd2;
        if (primal) a.s[14] += gd; else a.s[15] += gd;
      } else {
        a.m[0] = fmax(a.m[0], t);
        if (wd2 != 0.0) {
          a.s[11] += wd2 * t * t;
          if (primal) a.s[12] += gd * t; else a.s[13] += gd * t;
        }
      }
    }
  }
  block_reduce_store<TR_SETUP_NS, 1>(a, partials, stride);
}

// radius^2 and the value as functions of the step t at K probe values, per probe q (6 sums):
//   low = sum_{thr <= t_q} wd2 thr^2, high = sum_{thr > t_q} wd2,
//   vlow / vhigh = sum_{thr <= t_q} gd thr / sum_{thr > t_q} gd, primal block then dual block
constexpr int TR_K = 5, TR_Q = 6;
static_assert(TR_K * TR_Q <= EV_MAXQ && TR_SETUP_NS + 1 <= EV_MAXQ, "evaluation partials are sized for EV_MAXQ quantities");
struct TrProbes { double t[TR_K]; };
template <int BASE>
__device__ __forceinline__ void tr_probe_range(int k0, int k1, const double *__restrict__ thr,
                                               const double *__restrict__ wd2v, const double *__restrict__ gdv,
                                               const TrProbes &pr, RedAcc<TR_Q * TR_K, 0> &a) {
  for (int k = k0 + blockIdx.x * TPB + threadIdx.x; k < k1; k += gridDim.x * TPB) {
    const double wd2 = wd2v[k];
    if (wd2 == 0.0) continue;          // d == 0: blocked by its bound, or outside the range
    const double t = thr[k], gd = gdv[k];
    const double lowc = wd2 * t * t;   // inf for thr = inf: never selected below
    const double vlow = gd * t;
#pragma unroll
    for (int q = 0; q < TR_K; ++q) {
      if (t <= pr.t[q]) { a.s[TR_Q * q] += lowc; a.s[TR_Q * q + BASE] += vlow; }
      else { a.s[TR_Q * q + 1] += wd2; a.s[TR_Q * q + BASE + 1] += gd; }
    }
  }
}
__global__ __launch_bounds__(TPB) void tr_probe_kernel(int n, int total, const double *__restrict__ thr,
                                                       const double *__restrict__ wd2v, const double *__restrict__ gdv,
                                                       TrProbes pr, double *__restrict__ partials, int stride) {
  RedAcc<TR_Q * TR_K, 0> a;
  tr_probe_range<2>(0, n, thr, wd2v, gdv, pr, a);
  tr_probe_range<4>(n, total, thr, wd2v, gdv, pr, a);
  block_reduce_store<TR_Q * TR_K, 0>(a, partials, stride);
}


// ---- the breakpoint search as a state machine, ONE definition for the host loop (pdhg_trust_region_bound: a probe
// pass is a kernel + a reduction) and for the one-workgroup kernel of small problems (tr_small_kernel: a probe pass is
// a loop between two __syncthreads).
//   tr_search_begin(set-up sums)  ->  while (tr_search_next(probes)) { sums of the probes; tr_search_feed(sums); }
//   then tstar and `at` (the sums t* is computed from) are final:  value(t*) = at.v[0] + t* at.v[1] (primal), v[2], v[3] (dual)
struct TrEnd { double low, high, v[4]; };     // v: vlow primal, vhigh primal, vlow dual, vhigh dual
struct TrSearch {
  double r2, tstar;
  unsigned long long lo, hi, pb[TR_K];
  TrEnd lo_end, at;
  int have_lo, q0, zero_probe, done, passes;
};
__host__ __device__ inline unsigned long long tr_d2bits(double v) { unsigned long long b; memcpy(&b, &v, 8); return b; }
__host__ __device__ inline double tr_bits2d(unsigned long long b) { double v; memcpy(&v, &b, 8); return v; }
__host__ __device__ inline TrEnd tr_end_of(const double *p6) { return TrEnd{p6[0], p6[1], {p6[2], p6[3], p6[4], p6[5]}}; }
// floor(span * num / den) for 1 <= num < den <= 8 without 128-bit division (the device has none)
__host__ __device__ inline unsigned long long tr_mul_div(unsigned long long span, unsigned num, unsigned den) {
  const unsigned long long lo32 = span & 0xFFFFFFFFull, hi32 = span >> 32;
  const unsigned long long p0 = lo32 * num, p1 = hi32 * num + (p0 >> 32);      // span * num = p1 * 2^32 + (p0 & mask): < 2^67
  const unsigned long long q1 = p1 / den, r1 = p1 % den;
  const unsigned long long t0 = (r1 << 32) | (p0 & 0xFFFFFFFFull);
  return (q1 << 32) + t0 / den;                                                 // < span: fits
}
__host__ __device__ inline void tr_search_finish(TrSearch &S) {
  S.tstar = S.lo_end.high > 0.0 ? sqrt(fmax(S.r2 - S.lo_end.low, 0.0) / S.lo_end.high) : tr_bits2d(S.lo);
  S.at = S.lo_end;
  S.done = 1;
}
// at_tmax: the sums of the probe at t = tmax (every finite breakpoint passed), which the set-up pass produces itself
__host__ __device__ inline void tr_search_begin(TrSearch &S, double r2, double tmax, double hinf, const TrEnd &at_tmax) {
  S.r2 = r2; S.passes = 0; S.done = 0; S.have_lo = 0; S.zero_probe = 0; S.q0 = 0; S.tstar = 0.0;
  S.lo_end = TrEnd{0.0, 0.0, {0.0, 0.0, 0.0, 0.0}};
  S.at = S.lo_end;
  if (at_tmax.low + tmax * tmax * at_tmax.high <= r2) {
    // every finite breakpoint is reached before the radius
    if (hinf <= 0.0) S.tstar = tmax;                     // "all bounds hit" special case
    else S.tstar = sqrt((r2 - at_tmax.low) / hinf);
    S.at = at_tmax;
    S.done = 1;
    return;
  }
  S.lo = 0; S.hi = tr_d2bits(tmax);
}
// false: the search is over.  true: pr holds the next pass's probes.
__host__ __device__ inline bool tr_search_next(TrSearch &S, TrProbes &pr) {
  if (S.done) return false;
  if (S.hi - S.lo > 1) {
    const unsigned long long span = S.hi - S.lo;
    S.q0 = 0;
    // Probe 0: the closed-form candidate from the current lower end, t' = sqrt((r2 - low)/high).  If no breakpoint
    // lies in (lo, t'] the probe returns the same (low, high) and t' is the exact answer -- this fixed-point step
    // usually lands within a few passes; the remaining probes keep a guaranteed bracket in IEEE bit space.
    if (S.have_lo && S.lo_end.high > 0.0) {
      const double cand = sqrt(fmax(S.r2 - S.lo_end.low, 0.0) / S.lo_end.high);
      const unsigned long long cb = tr_d2bits(cand);
      if (cb > S.lo && cb < S.hi) { S.pb[0] = cb; pr.t[0] = cand; S.q0 = 1; }
    }
    for (int q = S.q0; q < TR_K; ++q) {
      unsigned long long off = tr_mul_div(span, (unsigned)(q - S.q0 + 1), (unsigned)(TR_K - S.q0 + 1));
      if (off == 0) off = 1;
      if (off >= span) off = span - 1;
      S.pb[q] = S.lo + off;
      pr.t[q] = tr_bits2d(S.pb[q]);
    }
    S.zero_probe = 0;
    return true;
  }
  if (!S.have_lo) {            // bracket collapsed at t = 0: evaluate the sums there
    for (int q = 0; q < TR_K; ++q) pr.t[q] = 0.0;
    S.zero_probe = 1;
    return true;
  }
  tr_search_finish(S);
  return false;
}
__host__ __device__ inline void tr_search_feed(TrSearch &S, const TrProbes &pr, const double *lh) {
  S.passes += 1;
  if (S.zero_probe) {
    S.lo_end = tr_end_of(lh);
    S.have_lo = 1;
    tr_search_finish(S);
    return;
  }
  if (S.q0 == 1 && lh[0] == S.lo_end.low && lh[1] == S.lo_end.high) {
    S.tstar = pr.t[0];
    S.at = S.lo_end;
    S.done = 1;
    return;
  }
  unsigned long long nlo = S.lo, nhi = S.hi;
  for (int q = 0; q < TR_K; ++q) {
    const double f = lh[TR_Q * q] + pr.t[q] * pr.t[q] * lh[TR_Q * q + 1];
    if (f <= S.r2) { if (S.pb[q] > nlo) { nlo = S.pb[q]; S.lo_end = tr_end_of(lh + TR_Q * q); S.have_lo = 1; } }
    else { if (S.pb[q] < nhi) nhi = S.pb[q]; }
  }
  S.lo = nlo; S.hi = nhi;
}


// ---- bound_optimal_objective for SMALL problems (n + m <= TRS_MAX) in ONE workgroup, one launch: the set-up pass,
// the whole breakpoint search (tr_search_*: thread 0 decides, every thread sums its elements for the five probes of a
// pass, a pass is a loop between two __syncthreads) and the eight results, published like an evaluation reduction.
// The multi-launch form costs a kernel + a second-stage kernel + a trip to the host per pass: 50-70 us per call for an
// LP whose vectors fit a few KB, five calls per termination / restart check.
constexpr int TRS_TPB = 256, TRS_MAX = 4096;      // 4 waves, one per SIMD: the 30 DPP trees of a pass run once per SIMD (1024 threads: 4 waves per SIMD, 68 us per call against 4x fewer trees here)
struct TrSmallArgs {
  int n, m, ne, range, approximate;
  const double *px, *py, *aty, *qx, *ax, *c, *b, *lb, *ub;
  double wp, wd, radius;
  double *host_out;
  unsigned long long seq;
};
// sums (then maxes) of a per-thread accumulator over the workgroup, into res[] (LDS); every thread calls it
template <int NS, int NM>
__device__ __forceinline__ void block_reduce_lds(const RedAcc<NS, NM> &a, double (*red)[TRS_TPB / WAVE], double *res) {
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < NS; ++q) { const double w = wave_sum_dpp(a.s[q]); if (lane == WAVE - 1) red[q][wid] = w; }
#pragma unroll
  for (int q = 0; q < NM; ++q) { const double w = wave_max_nonneg_dpp(a.m[q]); if (lane == WAVE - 1) red[NS + q][wid] = w; }
  __syncthreads();
  if (threadIdx.x < NS + NM) {
    const int q = threadIdx.x;
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < TRS_TPB / WAVE; ++w) t = (q >= NS) ? fmax(t, red[q][w]) : t + red[q][w];
    res[q] = t;
  }
  __syncthreads();
}
__global__ __launch_bounds__(TRS_TPB) void tr_small_kernel(TrSmallArgs a) {
  extern __shared__ double trs_dyn[];          // thr, w d^2, g d: n + m each
  double *s_thr = trs_dyn, *s_wd2 = trs_dyn + (a.n + a.m), *s_gd = trs_dyn + 2 * (a.n + a.m);
  __shared__ double red[TR_Q * TR_K + 2][TRS_TPB / WAVE];
  __shared__ double res[TR_Q * TR_K + 2];
  __shared__ TrProbes s_pr;
  __shared__ int s_go;
  __shared__ double s_out[8];
  const int n = a.n, total = a.n + a.m, tid = threadIdx.x;
  {
    // the set-up pass: tr_setup_kernel's arithmetic, element by element (see there for the sums' meaning)
    RedAcc<TR_SETUP_NS, 1> acc;
    for (int k = tid; k < total; k += TRS_TPB) {
      const bool primal = k < n;
      const int i = primal ? k : k - n;
      double z, g, lo, hi, w;
      if (primal) {
        z = a.px[i]; lo = a.lb[i]; hi = a.ub[i]; w = a.wp;
        if (a.qx) { g = (a.qx[i] + a.c[i]) - a.aty[i]; acc.s[10] += z * a.qx[i]; }
        else g = a.c[i] - a.aty[i];
        acc.s[0] += a.c[i] * z; acc.s[1] += z * a.aty[i]; acc.s[8] += z * z;
      } else {
        z = a.py[i]; g = -(a.b[i] - a.ax[i]); lo = (i < a.ne) ? -INFINITY : 0.0; hi = INFINITY; w = a.wd;
        acc.s[2] += z * a.b[i]; acc.s[9] += z * z;
      }
      const bool in_range = (a.range == 0) || (a.range == 1 && primal) || (a.range == 2 && !primal);
      double d = 0.0, t = 0.0;
      if (in_range && !((z >= hi && g <= 0.0) || (z <= lo && g >= 0.0))) {
        d = -g / w;
        if (d > 0.0) t = (hi - z) / d;
        else if (d < 0.0) t = (lo - z) / d;
        else t = 0.0;
      }
      const double wd2 = w * d * d, gd = g * d;
      s_gd[k] = gd; s_wd2[k] = wd2; s_thr[k] = t;
      if (in_range) {
        acc.s[4] += g * g;
        acc.s[5] += wd2;
        if (primal) acc.s[6] += gd; else acc.s[7] += gd;
        if (isinf(t)) {
          acc.s[3] += wd2;
          if (primal) acc.s[14] += gd; else acc.s[15] += gd;
        } else {
          acc.m[0] = fmax(acc.m[0], t);
          if (wd2 != 0.0) {
            acc.s[11] += wd2 * t * t;
            if (primal) acc.s[12] += gd * t; else acc.s[13] += gd * t;
          }
        }
      }
    }
    block_reduce_lds<TR_SETUP_NS, 1>(acc, red, res);
  }
  // thread 0: the host function's statements (pdhg_trust_region_bound) on the set-up sums
  __shared__ TrSearch S;
  if (tid == 0) {
    const double *r = res;
    s_out[0] = 0.5 * r[10] + r[0] - r[1] + r[2];
    s_out[1] = s_out[2] = 0.0;
    s_out[3] = r[8]; s_out[4] = r[9];
    s_out[5] = 0.0; s_out[6] = 0.0; s_out[7] = 0.0;
    const double hinf = r[3], g2 = r[4], wd2_all = r[5], tmax = r[TR_SETUP_NS];
    const double r2 = a.radius * a.radius;
    s_go = 0;
    if (a.approximate) {
      const double dn = sqrt(wd2_all);
      const double sc = dn > 0.0 ? a.radius / dn : 1.0;
      s_out[1] = sc * r[6]; s_out[2] = sc * r[7];
    } else if (!(a.radius == 0.0 || g2 == 0.0)) {
      tr_search_begin(S, r2, tmax, hinf, TrEnd{r[11], hinf, {r[12], r[14], r[13], r[15]}});
      s_go = tr_search_next(S, s_pr) ? 1 : 2;      // 1: a pass to run, 2: the search is over already
    }
  }
  __syncthreads();
  while (s_go == 1) {
    RedAcc<TR_Q * TR_K, 0> acc;
    const TrProbes pr = s_pr;
    for (int k = tid; k < total; k += TRS_TPB) {
      const double wd2 = s_wd2[k];
      if (wd2 == 0.0) continue;
      const double t = s_thr[k], gd = s_gd[k];
      const double lowc = wd2 * t * t, vlow = gd * t;
      const int base = (k < n) ? 2 : 4;
#pragma unroll
      for (int q = 0; q < TR_K; ++q) {
        if (t <= pr.t[q]) {
          acc.s[TR_Q * q] += lowc;
          if (base == 2) acc.s[TR_Q * q + 2] += vlow; else acc.s[TR_Q * q + 4] += vlow;
        } else {
          acc.s[TR_Q * q + 1] += wd2;
          if (base == 2) acc.s[TR_Q * q + 3] += gd; else acc.s[TR_Q * q + 5] += gd;
        }
      }
    }
    block_reduce_lds<TR_Q * TR_K, 0>(acc, red, res);
    if (tid == 0) {
      tr_search_feed(S, s_pr, res);
      s_go = tr_search_next(S, s_pr) ? 1 : 2;
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (s_go == 2) {
      s_out[1] = S.at.v[0] + S.tstar * S.at.v[1];
      s_out[2] = S.at.v[2] + S.tstar * S.at.v[3];
      s_out[5] = S.tstar; s_out[6] = (double)S.passes;
    }
    unsigned long long ck = EV_CHECK_SALT ^ a.seq ^ (8ull << 56);
    for (int q = 0; q < 8; ++q) {
      a.host_out[q] = s_out[q];
      ck ^= (unsigned long long)__double_as_longlong(s_out[q]) * (2ull * (unsigned long long)q + 1ull);
    }
    a.host_out[EV_HOST_CK] = __longlong_as_double((long long)ck);
    a.host_out[EV_HOST_SEQ] = __longlong_as_double((long long)a.seq);
  }
}

}  // namespace
