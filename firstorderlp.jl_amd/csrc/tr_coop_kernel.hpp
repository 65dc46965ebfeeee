// tr_coop_kernel.hpp -- part of the single translation unit pdhg_hip.hip (included there, after trial_kernel.hpp).
// bound_optimal_objective's trust-region problem (trust_region_utils.jl:57-224, 271-360) for MEDIUM problems as ONE
// persistent launch: the set-up pass, every probe pass of the breakpoint search and the eight results, with a grid
// barrier where the multi-launch form (pdhg_trust_region_bound: tr_setup_kernel / tr_probe_kernel + multi_final_kernel
// + a trip to the host PER PASS) has a launch pair and a host round trip.
//
// Why.  A termination / restart check makes five such calls of four to six passes each; on the L1-SVM LP a pass is
// 9 us of probe kernel + 4.4 us of second stage + 10-11 us of host round trip = 24 us, 30 passes = 0.72 of the check's
// 1.07 ms under the profiler (tools/archive/r4_eval_timeline.sh), and the checks are 30 % of a whole solve.  Here a pass is the
// probe arithmetic on elements the thread already owns + one XCD-scoped grid barrier (trial_kernel.hpp: ~3.3 us) + a
// reduction of <= 256 block partials per quantity that every workgroup repeats for itself (so the search state -- the
// tr_search_* machine of eval_kernels.hpp, thread 0 of every workgroup -- needs no broadcast: same sums, same
// decisions, everywhere).
//
// Arithmetic: tr_setup_kernel's and tr_probe_range's statements element by element; the sums are grouped by THIS
// kernel's grid (<= 256 workgroups, a multiple of 8), so results agree with the multi-launch form to rounding, not to
// the bit -- as the one-workgroup kernel of small problems (tr_small_kernel) already does.  Deterministic: fixed grid,
// fixed order.
//
// The block partials alternate between two buffers: a workgroup can run at most one barrier ahead of the slowest, so
// pass k + 1's partials never overwrite what somebody still reads from pass k.  Partials are read with agent-scope
// loads (past the CU's L1, which may hold the lines of two passes ago); across XCDs the barrier's write-back /
// invalidate makes them visible.
#pragma once

namespace {

constexpr int TRC_MAX_WGS = 256;                   // <= 4 partials per lane and quantity in the repeated second stage
constexpr int TRC_Q = TR_Q * TR_K;                 // 30 sums per probe pass
static_assert(TRC_Q <= EV_MAXQ && TR_SETUP_NS + 1 <= EV_MAXQ, "partials are EV_MAXQ quantities wide");

struct TrCoopArgs {
  int n, m, ne, range, approximate;
  const double *px, *py, *aty, *qx, *ax, *c, *b, *lb, *ub;
  double wp, wd, radius;
  double *gdv, *wd2v, *thr;                        // n + m each (the handle's tr_g / tr_dir / tr_thr)
  double *partials;                                // 2 x EV_MAXQ x gridDim.x
  GridSync *sync;
  unsigned long long epoch;                        // barriers this GridSync has passed so far
  unsigned nxcd;
  unsigned xcd_cnt[8];
  double *host_out;                                // pinned: out[0..7], error word, epoch after the launch
  unsigned long long seq;
};

// every workgroup: quantity q (NS sums, then NM maxes) over the `count` block partials of one pass, into res[q] (LDS).
// Wave w takes q = w, w + 4, ...; all its loads are issued before the first tree.
template <int NS, int NM>
__device__ __forceinline__ void trc_reduce(const double *partials, int stride, int count, double *res) {
  constexpr int K = NS + NM, WAVES = TPB / WAVE, PER_WAVE = (K + WAVES - 1) / WAVES, PER_LANE = TRC_MAX_WGS / WAVE;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  double t[PER_WAVE][PER_LANE];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int q = wave + i * WAVES;
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
      const int b = lane + j * WAVE;
      t[i][j] = (q < K && b < count) ? __hip_atomic_load(partials + (size_t)q * stride + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int q = wave + i * WAVES;
    const bool is_max = q >= NS;
    double v = 0.0;                                 // partials of maxes are >= 0
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) v = is_max ? fmax(v, t[i][j]) : v + t[i][j];
    v = is_max ? wave_max_nonneg_dpp(v) : wave_sum_dpp(v);
    if (q < K && lane == WAVE - 1) res[q] = v;
  }
  __syncthreads();
}

// The same for up to NP problems whose partials lie pstride apart, ALL their loads issued before the first tree and one
// workgroup barrier at the end (round 6: problem by problem this stage was three times one problem's latency -- 6.3 us of a
// 21 us pass on the L1-SVM LP, in-kernel trace).  Per problem and quantity the same lane grouping and the same tree as
// trc_reduce: the same bits.  go[p] != 0: problem p takes part (workgroup-uniform).
template <int NS, int NM, int NP>
__device__ __forceinline__ void trc_reduce_multi(const double *partials, size_t pstride, int stride, int count, double (*res)[EV_MAXQ],
                                                 const int *go, int P) {
  constexpr int K = NS + NM, WAVES = TPB / WAVE, PER_WAVE = (K + WAVES - 1) / WAVES, PER_LANE = TRC_MAX_WGS / WAVE;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  double t[NP][PER_WAVE][PER_LANE];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool on = p < P && go[p];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = wave + i * WAVES;
#pragma unroll
      for (int j = 0; j < PER_LANE; ++j) {
        const int b = lane + j * WAVE;
        t[p][i][j] = (on && q < K && b < count)
                         ? __hip_atomic_load(partials + p * pstride + (size_t)q * stride + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (!(p < P && go[p])) continue;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = wave + i * WAVES;
      const bool is_max = q >= NS;
      double v = 0.0;
#pragma unroll
      for (int j = 0; j < PER_LANE; ++j) v = is_max ? fmax(v, t[p][i][j]) : v + t[p][i][j];
      v = is_max ? wave_max_nonneg_dpp(v) : wave_sum_dpp(v);
      if (q < K && lane == WAVE - 1) res[p][q] = v;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(TPB) void tr_coop_kernel(TrCoopArgs a) {
  __shared__ double res[EV_MAXQ];
  __shared__ TrProbes s_pr;
  __shared__ int s_go;
  __shared__ double s_out[8];
  __shared__ TrSearch S;
  const int n = a.n, total = a.n + a.m;
  const int gtid = blockIdx.x * TPB + threadIdx.x, gstride = gridDim.x * TPB, stride = gridDim.x;
  unsigned long long epoch = a.epoch;
  int buf = 0;
  {
    // the set-up pass: tr_setup_kernel's statements (see there for the meaning of the sums)
    RedAcc<TR_SETUP_NS, 1> acc;
    for (int k = gtid; k < total; k += gstride) {
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
      a.gdv[k] = gd; a.wd2v[k] = wd2; a.thr[k] = t;           // read back by this very thread in the probe passes
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
    block_reduce_store<TR_SETUP_NS, 1>(acc, a.partials, stride);
    grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
    trc_reduce<TR_SETUP_NS, 1>(a.partials, stride, (int)gridDim.x, res);
    buf ^= 1;
  }
  if (threadIdx.x == 0) {
    // the host function's statements (pdhg_trust_region_bound) on the set-up sums, in every workgroup
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
      s_go = tr_search_next(S, s_pr) ? 1 : 2;
    }
  }
  __syncthreads();
  while (s_go == 1) {
    RedAcc<TRC_Q, 0> acc;
    const TrProbes pr = s_pr;
    for (int k = gtid; k < total; k += gstride) {
      const double wd2 = a.wd2v[k];
      if (wd2 == 0.0) continue;                                // d == 0: blocked by its bound, or outside the range
      const double t = a.thr[k], gd = a.gdv[k];
      const double lowc = wd2 * t * t, vlow = gd * t;
      const bool primal = k < n;
#pragma unroll
      for (int q = 0; q < TR_K; ++q) {
        if (t <= pr.t[q]) {
          acc.s[TR_Q * q] += lowc;
          if (primal) acc.s[TR_Q * q + 2] += vlow; else acc.s[TR_Q * q + 4] += vlow;
        } else {
          acc.s[TR_Q * q + 1] += wd2;
          if (primal) acc.s[TR_Q * q + 3] += gd; else acc.s[TR_Q * q + 5] += gd;
        }
      }
    }
    double *part = a.partials + (size_t)buf * EV_MAXQ * stride;
    block_reduce_store<TRC_Q, 0>(acc, part, stride);
    grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
    trc_reduce<TRC_Q, 0>(part, stride, (int)gridDim.x, res);
    buf ^= 1;
    if (threadIdx.x == 0) {
      tr_search_feed(S, s_pr, res);
      s_go = tr_search_next(S, s_pr) ? 1 : 2;
      // a barrier that timed out leaves the error word set and every later barrier falls through: stop searching on
      // sums that may be incomplete (the host sees the error word and repeats the call launch by launch)
      if (__hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) s_go = 3;
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (s_go == 2) {
      s_out[1] = S.at.v[0] + S.tstar * S.at.v[1];
      s_out[2] = S.at.v[2] + S.tstar * S.at.v[3];
      s_out[5] = S.tstar; s_out[6] = (double)S.passes;
    }
    double w[10];
    for (int q = 0; q < 8; ++q) w[q] = s_out[q];
    w[8] = (double)__hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w[9] = (double)epoch;
    unsigned long long ck = EV_CHECK_SALT ^ a.seq ^ (10ull << 56);
    for (int q = 0; q < 10; ++q) {
      a.host_out[q] = w[q];
      ck ^= (unsigned long long)__double_as_longlong(w[q]) * (2ull * (unsigned long long)q + 1ull);
    }
    a.host_out[EV_HOST_CK] = __longlong_as_double((long long)ck);
    a.host_out[EV_HOST_SEQ] = __longlong_as_double((long long)a.seq);
  }
}

// ---- several trust-region problems in ONE persistent launch (round 5) -------------------------------------------------
// A restart check asks for three bounds -- at the average, at the current iterate, at the last restart point
// (saddle_point.jl:432-496, 551-596) -- and a recorded iteration for two more; each was a launch of its own with a grid
// barrier per pass, 4-6 passes: on the L1-SVM LP five calls of ~80 us were 70 % of a 0.55 ms check.  The problems are
// independent, so their passes can share the barriers: here every pass works through the elements of EVERY problem still
// searching, then ONE barrier, then every workgroup reduces each problem's partials and advances each search -- the
// launch lasts as long as the problem with the most passes, not the sum.  Per problem the statements, the grouping of
// the sums (this grid) and therefore the BITS are those of tr_coop_kernel.
constexpr int TRB_MAX = 3;
struct TrBatchProblem {
  int range;
  const double *px, *py, *aty, *qx, *ax;
  double radius;
  double *gdv, *wd2v, *thr;                        // n + m each, this problem's own
};
struct TrBatchArgs {
  int n, m, ne, approximate, count;
  const double *c, *b, *lb, *ub;
  double wp, wd;
  TrBatchProblem pb[TRB_MAX];
  double *partials;                                // 2 x TRB_MAX x EV_MAXQ x gridDim.x
  GridSync *sync;
  unsigned long long epoch;
  unsigned nxcd;
  unsigned xcd_cnt[8];
  double *host_out;                                // pinned: 8 x count results, error word, epoch after the launch
  unsigned long long seq;
};

#ifdef PDHG_TRB_TRACE
// dev: where a probe pass of the batched search spends its time (workgroup 0's wall clock, 100 MHz): sums over the passes
// of [0] the element walk, [1] the block reductions, [2] the grid barrier, [3] the second stage, [4] the searches' step;
// [5] passes, [6] the set-up pass up to its barrier's end, [7] launches
__device__ unsigned long long g_trb_trace[8];
#define TRB_STAMP(v) do { if (blockIdx.x == 0 && threadIdx.x == 0) v = wall_clock64(); } while (0)
#else
#define TRB_STAMP(v) do { } while (0)
#endif
__global__ __launch_bounds__(TPB) void tr_coop_batch_kernel(TrBatchArgs a) {
  __shared__ double res[TRB_MAX][EV_MAXQ];
  __shared__ TrProbes s_pr[TRB_MAX];
  __shared__ int s_go[TRB_MAX];
  __shared__ int s_any;
  __shared__ double s_out[TRB_MAX][8];
  __shared__ TrSearch S[TRB_MAX];
  const int n = a.n, total = a.n + a.m, P = a.count;
  const int gtid = blockIdx.x * TPB + threadIdx.x, gstride = gridDim.x * TPB, stride = gridDim.x;
  const size_t pstride = (size_t)EV_MAXQ * stride;          // one problem's partials of one pass
  unsigned long long epoch = a.epoch;
  int buf = 0;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
  (void)ts0; (void)ts1; (void)ts2; (void)ts3; (void)ts4; (void)ts5;
  TRB_STAMP(ts0);
  for (int p = 0; p < P; ++p) {
    // the set-up pass of problem p: tr_coop_kernel's statements
    const TrBatchProblem &q = a.pb[p];
    RedAcc<TR_SETUP_NS, 1> acc;
    for (int k = gtid; k < total; k += gstride) {
      const bool primal = k < n;
      const int i = primal ? k : k - n;
      double z, g, lo, hi, w;
      if (primal) {
        z = q.px[i]; lo = a.lb[i]; hi = a.ub[i]; w = a.wp;
        if (q.qx) { g = (q.qx[i] + a.c[i]) - q.aty[i]; acc.s[10] += z * q.qx[i]; }
        else g = a.c[i] - q.aty[i];
        acc.s[0] += a.c[i] * z; acc.s[1] += z * q.aty[i]; acc.s[8] += z * z;
      } else {
        z = q.py[i]; g = -(a.b[i] - q.ax[i]); lo = (i < a.ne) ? -INFINITY : 0.0; hi = INFINITY; w = a.wd;
        acc.s[2] += z * a.b[i]; acc.s[9] += z * z;
      }
      const bool in_range = (q.range == 0) || (q.range == 1 && primal) || (q.range == 2 && !primal);
      double d = 0.0, t = 0.0;
      if (in_range && !((z >= hi && g <= 0.0) || (z <= lo && g >= 0.0))) {
        d = -g / w;
        if (d > 0.0) t = (hi - z) / d;
        else if (d < 0.0) t = (lo - z) / d;
        else t = 0.0;
      }
      const double wd2 = w * d * d, gd = g * d;
      q.gdv[k] = gd; q.wd2v[k] = wd2; q.thr[k] = t;           // read back by this very thread in the probe passes
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
    block_reduce_store<TR_SETUP_NS, 1>(acc, a.partials + p * pstride, stride);
  }
  grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
  TRB_STAMP(ts1);
#ifdef PDHG_TRB_TRACE
  if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&g_trb_trace[6], ts1 - ts0); atomicAdd(&g_trb_trace[7], 1ull); }
#endif
  {
    __shared__ int s_all[TRB_MAX];
    if (threadIdx.x < TRB_MAX) s_all[threadIdx.x] = 1;
    __syncthreads();
    trc_reduce_multi<TR_SETUP_NS, 1, TRB_MAX>(a.partials, pstride, stride, (int)gridDim.x, res, s_all, P);
  }
  buf ^= 1;
  // one thread PER PROBLEM (each in a wave of its own) starts / advances its search: the problems are independent, and
  // thread 0 doing them one after the other was 4.5 us of a 21 us pass
  if ((threadIdx.x & (WAVE - 1)) == 0 && (int)(threadIdx.x / WAVE) < P) {
    const int p = (int)(threadIdx.x / WAVE);
    {
      const double *r = res[p];
      double *o = s_out[p];
      o[0] = 0.5 * r[10] + r[0] - r[1] + r[2];
      o[1] = o[2] = 0.0;
      o[3] = r[8]; o[4] = r[9];
      o[5] = 0.0; o[6] = 0.0; o[7] = 0.0;
      const double hinf = r[3], g2 = r[4], wd2_all = r[5], tmax = r[TR_SETUP_NS];
      const double radius = a.pb[p].radius, r2 = radius * radius;
      s_go[p] = 0;
      if (a.approximate) {
        const double dn = sqrt(wd2_all);
        const double sc = dn > 0.0 ? radius / dn : 1.0;
        o[1] = sc * r[6]; o[2] = sc * r[7];
      } else if (!(radius == 0.0 || g2 == 0.0)) {
        tr_search_begin(S[p], r2, tmax, hinf, TrEnd{r[11], hinf, {r[12], r[14], r[13], r[15]}});
        s_go[p] = tr_search_next(S[p], s_pr[p]) ? 1 : 2;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int any = 0;
    for (int p = 0; p < P; ++p) any |= s_go[p] == 1;
    s_any = any;
  }
  __syncthreads();
  while (s_any) {
    double *part = a.partials + (size_t)buf * TRB_MAX * pstride;
    // one walk over the elements for ALL the problems still searching: their loads are independent, so a thread has three
    // times the requests in flight for the same latency (problem by problem a pass is ~9 us of mostly waiting)
    RedAcc<TRC_Q, 0> acc[TRB_MAX];
    bool act[TRB_MAX];
#pragma unroll
    for (int p = 0; p < TRB_MAX; ++p) act[p] = p < P && s_go[p] == 1;       // workgroup-uniform, and the same in every workgroup
    TRB_STAMP(ts0);
    for (int k = gtid; k < total; k += gstride) {
      const bool primal = k < n;
      double wd2[TRB_MAX], t[TRB_MAX], gd[TRB_MAX];
#pragma unroll
      for (int p = 0; p < TRB_MAX; ++p) {
        wd2[p] = 0.0; t[p] = 0.0; gd[p] = 0.0;
        if (act[p]) { wd2[p] = a.pb[p].wd2v[k]; t[p] = a.pb[p].thr[k]; gd[p] = a.pb[p].gdv[k]; }
      }
#pragma unroll
      for (int p = 0; p < TRB_MAX; ++p) {
        if (!act[p] || wd2[p] == 0.0) continue;                // d == 0: blocked by its bound, or outside the range
        const double lowc = wd2[p] * t[p] * t[p], vlow = gd[p] * t[p];
#pragma unroll
        for (int j = 0; j < TR_K; ++j) {
          if (t[p] <= s_pr[p].t[j]) {
            acc[p].s[TR_Q * j] += lowc;
            if (primal) acc[p].s[TR_Q * j + 2] += vlow; else acc[p].s[TR_Q * j + 4] += vlow;
          } else {
            acc[p].s[TR_Q * j + 1] += wd2[p];
            if (primal) acc[p].s[TR_Q * j + 3] += gd[p]; else acc[p].s[TR_Q * j + 5] += gd[p];
          }
        }
      }
    }
    TRB_STAMP(ts1);
#pragma unroll
    for (int p = 0; p < TRB_MAX; ++p)
      if (act[p]) block_reduce_store<TRC_Q, 0>(acc[p], part + p * pstride, stride);
    TRB_STAMP(ts2);
    grid_barrier(a.sync, ++epoch, a.nxcd, a.xcd_cnt);
    TRB_STAMP(ts3);
    {
      __shared__ int s_act[TRB_MAX];
      if (threadIdx.x < TRB_MAX) s_act[threadIdx.x] = (int)threadIdx.x < P && s_go[threadIdx.x] == 1;
      __syncthreads();
      trc_reduce_multi<TRC_Q, 0, TRB_MAX>(part, pstride, stride, (int)gridDim.x, res, s_act, P);
    }
    buf ^= 1;
    TRB_STAMP(ts4);
    if ((threadIdx.x & (WAVE - 1)) == 0 && (int)(threadIdx.x / WAVE) < P) {
      const int p = (int)(threadIdx.x / WAVE);
      if (s_go[p] == 1) {
        const bool broken = __hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        // (state, probes and sums stay in LDS: copied into registers for the step the kernel spills 200 bytes per lane and the
        //  step takes 5.8 us instead of 2.9)
        tr_search_feed(S[p], s_pr[p], res[p]);
        s_go[p] = tr_search_next(S[p], s_pr[p]) ? 1 : 2;
        if (broken) s_go[p] = 3;                               // a barrier timed out: the host repeats the calls one by one
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int any = 0;
      for (int p = 0; p < P; ++p) any |= s_go[p] == 1;
      s_any = any;
    }
    __syncthreads();
#ifdef PDHG_TRB_TRACE
    TRB_STAMP(ts5);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      atomicAdd(&g_trb_trace[0], ts1 - ts0); atomicAdd(&g_trb_trace[1], ts2 - ts1); atomicAdd(&g_trb_trace[2], ts3 - ts2);
      atomicAdd(&g_trb_trace[3], ts4 - ts3); atomicAdd(&g_trb_trace[4], ts5 - ts4); atomicAdd(&g_trb_trace[5], 1ull);
    }
#endif
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int k = 8 * P + 2;
    unsigned long long ck = EV_CHECK_SALT ^ a.seq ^ ((unsigned long long)k << 56);
    int slot = 0;
    auto put = [&](double v) {
      a.host_out[slot] = v;
      ck ^= (unsigned long long)__double_as_longlong(v) * (2ull * (unsigned long long)slot + 1ull);
      ++slot;
    };
    for (int p = 0; p < P; ++p) {
      double *o = s_out[p];
      if (s_go[p] == 2) {
        o[1] = S[p].at.v[0] + S[p].tstar * S[p].at.v[1];
        o[2] = S[p].at.v[2] + S[p].tstar * S[p].at.v[3];
        o[5] = S[p].tstar; o[6] = (double)S[p].passes;
      }
      for (int j = 0; j < 8; ++j) put(o[j]);
    }
    put((double)__hip_atomic_load(&a.sync->error[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    put((double)epoch);
    a.host_out[EV_HOST_CK] = __longlong_as_double((long long)ck);
    a.host_out[EV_HOST_SEQ] = __longlong_as_double((long long)a.seq);
  }
}

}  // namespace
