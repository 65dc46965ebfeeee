// small_lp_kernel.hpp -- part of the single translation unit pdhg_hip.hip (included there, after trial_kernel.hpp).
// SMALL LPs (n, m up to ~1400: the class of the reference's Netlib runs, BASELINE configs[1]): a batch of adaptive
// take_steps (pdhg.jl:653-731; pdhg_take_steps_adaptive) in ONE workgroup with every vector in LDS.
//
// Why a kernel of its own.  For an LP of a few thousand nonzeros a trial is nothing but latency: one launch per trial
// costs ~28 us whatever the size (launch, two grid barriers, the completion ticket, the result's trip to the host), and
// walking the row blocks of the stream layout in one workgroup does not help either (a phase is a chain of ~5 dependent
// trips to memory, ~7 us per row block: profiles/r03_trial_kernel.txt).  Here nothing leaves the compute unit between
// two trials: x, x', xbar, A'y, A'y', c, the bounds, sum_x (n each) and y, y', b, sum_y (m each) live in LDS, the
// matrix (static, a few hundred KB at most) streams from the L1 / L2, one thread owns one row, and the step rule
// (adaptive_step_rule, the host loop's own function) runs on thread 0.  A trial is three phases between
// __syncthreads: ~3 us instead of ~28.
//
// Same bits as every other path: the phases restate the element arithmetic of primal_one / row_epilogue verbatim
// (separate multiply and add, -ffp-contract=off), a row's products are added left to right as the stream kernel adds
// rows of up to 256 entries in either row order (longer rows make the LP ineligible), and the three sums are
// double-double (exactly rounded whatever the grouping).  tests/test_gpu_small_lp.py: bitwise against one launch per
// trial and against the oracle.
#pragma once

namespace {

constexpr int SMALL_TPB = 1024;          // threads of the workgroup for n or m beyond SMALL_FEW_ROWS; 256 below (fewer waves per barrier)
constexpr int SMALL_FEW_ROWS = 256;      // (30 x 30: 173k it/s with 256 threads against 126k with 1024; 300 x 300: 116k against 120k)
constexpr int SMALL_MAX_ROW = 256;       // rows of more entries are summed wave-parallel in relaxed order elsewhere

struct SmallLpArgs {
  int n, m, num_eq;
  CsrView A, T;                           // CSR(A) (m rows), CSR(A') (n rows)
  double *x, *y, *aty, *sum_x, *sum_y;    // read at the start, written back at the end
  const double *c, *lb, *ub, *b;
  double primal_weight, step_size;
  int n_steps, max_trials, table_len;
  int pend;                               // an accept before this launch left its average update pending
  double pend_w;
  double wsum_x, wsum_y;
  const double *pow_red, *pow_growth;
  volatile double *res_host;
  unsigned long long seq;
};

// one row's sum: products added strictly left to right, eight entries requested at a time
__device__ __forceinline__ double small_row_sum(const CsrView &M, int r, const double *xs) {
  int k = M.rowptr[r];
  const int ke = M.rowptr[r + 1];
  double s = 0.0;
  for (; k + 8 <= ke; k += 8) {
    int ci[8];
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { ci[u] = M.col[k + u]; v[u] = M.val[k + u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const double p = v[u] * xs[ci[u]]; s = s + p; }
  }
  for (; k < ke; ++k) { const double p = M.val[k] * xs[M.col[k]]; s = s + p; }
  return s;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void small_lp_steps_kernel(SmallLpArgs a) {
  extern __shared__ double lds[];
  __shared__ double red[6][THREADS / WAVE];
  __shared__ double s_dec[3];
  __shared__ double s_st[5];              // step size of the trial, step size on entry, (unused), weight sums x / y
  const int n = a.n, m = a.m, tid = threadIdx.x;
  double *xs = lds, *xn = xs + n, *xb = xn + n, *at = xb + n, *atn = at + n;
  double *cs = atn + n, *lbs = cs + n, *ubs = lbs + n, *sx = ubs + n;
  double *ys = sx + n, *yn = ys + m, *bs = yn + m, *sy = bs + m;
  for (int j = tid; j < n; j += THREADS) {
    xs[j] = a.x[j]; at[j] = a.aty[j]; cs[j] = a.c[j]; lbs[j] = a.lb[j]; ubs[j] = a.ub[j]; sx[j] = a.sum_x[j];
  }
  for (int r = tid; r < m; r += THREADS) { ys[r] = a.y[r]; bs[r] = a.b[r]; sy[r] = a.sum_y[r]; }
  if (tid == 0) { s_st[0] = a.step_size; s_st[1] = a.step_size; s_st[3] = a.wsum_x; s_st[4] = a.wsum_y; }
  __syncthreads();
  if (a.pend) {                           // the deferred K7 of the accept before this launch (saddle_point.jl:252-301)
    for (int j = tid; j < n; j += THREADS) { const double t = xs[j] * a.pend_w; sx[j] = sx[j] + t; }
    for (int r = tid; r < m; r += THREADS) { const double t = ys[r] * a.pend_w; sy[r] = sy[r] + t; }
  }
  int steps = 0, trials = 0, num_err = 0, mid = 0;
  while (steps < a.n_steps && (trials < a.max_trials || mid) && trials < a.table_len) {
    const double step = s_st[0];
    const double tau = step / a.primal_weight, sigma = a.primal_weight * step;
    double pw_r = 0.0, pw_g = 0.0;
    if (tid == 0) { pw_r = a.pow_red[trials]; pw_g = a.pow_growth[trials]; }
    // ---- K1 + K2: x' = proj(x - tau (c - A'y)), xbar = x' + (x' - x)        (primal_one, vector_kernels.hpp)
    for (int j = tid; j < n; j += THREADS) {
      double v, b2;
      primal_one<false, true>(xs[j], cs[j], at[j], 0.0, lbs[j], ubs[j], tau, 1.0, v, b2);
      xn[j] = v; xb[j] = b2;
    }
    __syncthreads();
    // ---- K3 + K4: y' = proj(y + sigma (b - A xbar)), sum dy^2                 (row_epilogue<MODE_DUAL>)
    Acc3 acc = acc3_zero();
    for (int r = tid; r < m; r += THREADS) {
      const double s = small_row_sum(a.A, r, xb);
      const double yo = ys[r];
      const double dg = bs[r] - s;
      const double t = sigma * dg;
      double v = yo + t;
      if (r >= a.num_eq) v = jl_max(v, 0.0);
      yn[r] = v;
      const double dy = v - yo;
      dd_add(acc.hi[0], acc.lo[0], dy * dy);
    }
    block_sum_dd<1, THREADS>(acc, red);          // (ends with the totals on thread 0; a barrier inside)
    const double dy2 = acc.hi[0] + acc.lo[0];
    __syncthreads();
    // ---- K5 + K6: A'y' and the interaction sums                                  (row_epilogue<MODE_ATY>)
    Acc3 acc3 = acc3_zero();
    for (int j = tid; j < n; j += THREADS) {
      const double s = small_row_sum(a.T, j, yn);
      atn[j] = s;
      const double dx = xn[j] - xs[j];
      const double dd = s - at[j];
      dd_add(acc3.hi[0], acc3.lo[0], dx * dd);
      dd_add(acc3.hi[1], acc3.lo[1], dx * dx);
      dd_add(acc3.hi[2], acc3.lo[2], dd * dd);
    }
    block_sum_dd<3, THREADS>(acc3, red);
    if (tid == 0) {
      double raw[5];
      raw[0] = acc3.hi[0] + acc3.lo[0]; raw[1] = acc3.hi[1] + acc3.lo[1]; raw[2] = dy2; raw[3] = acc3.hi[2] + acc3.lo[2];
      raw[4] = 0.0;
      const StepRule rule = adaptive_step_rule(raw, a.primal_weight, step, pw_r, pw_g);
      s_dec[0] = (double)rule.accept; s_dec[1] = (double)rule.numerical_error; s_dec[2] = rule.next_step;
    }
    __syncthreads();
    const int nerr = __builtin_amdgcn_readfirstlane((int)(s_dec[1] != 0.0));
    const int acc_ok = __builtin_amdgcn_readfirstlane((int)(s_dec[0] != 0.0));
    trials += 1;
    if (nerr) { num_err = 1; mid = 0; break; }
    mid = !acc_ok;
    if (acc_ok) {
      // update_solution_in_solver_state (pdhg.jl:496-525): the trial point becomes the iterate; the averages take it
      // with the step size on entry as weight
      double *t0 = xs; xs = xn; xn = t0;
      double *t1 = ys; ys = yn; yn = t1;
      double *t2 = at; at = atn; atn = t2;
      const double wgt = s_st[1];
      for (int j = tid; j < n; j += THREADS) { const double t = xs[j] * wgt; sx[j] = sx[j] + t; }
      for (int r = tid; r < m; r += THREADS) { const double t = ys[r] * wgt; sy[r] = sy[r] + t; }
      steps += 1;
    }
    __syncthreads();
    if (tid == 0) {
      const double next = s_dec[2];
      if (acc_ok) {
        const double entry = s_st[1];
        s_st[3] = s_st[3] + entry; s_st[4] = s_st[4] + entry;
        s_st[1] = next;
      }
      s_st[0] = next;
    }
    __syncthreads();
  }
  for (int j = tid; j < n; j += THREADS) { a.x[j] = xs[j]; a.aty[j] = at[j]; a.sum_x[j] = sx[j]; }
  for (int r = tid; r < m; r += THREADS) { a.y[r] = ys[r]; a.sum_y[r] = sy[r]; }
  __syncthreads();
  if (tid == 0) {
    __threadfence_system();
    unsigned long long ck = RESULT_CHECK_SALT;
    int k = 0;
#define PDHG_PUB(v) do { const double pv = (v); a.res_host[k] = pv; ck ^= (unsigned long long)__double_as_longlong(pv) * (2ull * k + 1ull); ++k; } while (0)
    PDHG_PUB(s_st[0]); PDHG_PUB((double)steps); PDHG_PUB((double)trials); PDHG_PUB(0.0); PDHG_PUB(0.0);
    PDHG_PUB(0.0); PDHG_PUB(s_st[3]); PDHG_PUB(s_st[4]); PDHG_PUB((double)num_err); PDHG_PUB(0.0);
    PDHG_PUB(0.0); PDHG_PUB(0.0);
    PDHG_PUB((double)a.seq);
#undef PDHG_PUB
    const double e14 = mid ? s_st[1] : 0.0;   // ended inside a take_step (table exhausted): its step size on entry, for the host to finish it
    ck ^= (unsigned long long)__double_as_longlong(e14) * 29ull;     // (word 14 is under the checksum too: steps_wait)
    a.res_host[14] = e14;
    a.res_host[13] = __longlong_as_double((long long)ck);
    a.res_host[15] = (double)a.seq;
  }
}

}  // namespace
