/*
 * pdhg_cpu_omp.c -- multi-threaded CPU COMPARATOR for bench.py's cpu_baseline leg.
 *
 * TEST/MEASUREMENT INFRASTRUCTURE, NOT THE PRODUCT and NOT the parity oracle
 * (that is pdhg_oracle.c, single-threaded and sequential like the reference).
 * The reference (Julia stdlib SparseArrays + broadcasts) is single-threaded;
 * BASELINE.json's north star nevertheless asks for a "single-socket CPU
 * reference", so this file runs the same adaptive PDHG step
 * (src/primal_dual_hybrid_gradient.jl:442-549, 653-731) with OpenMP across the
 * cores of one socket: A*x row-parallel over CSR(A), A'*y row-parallel over
 * CSR(A') (= the CSC arrays), vector updates and reductions parallel.  Same
 * per-element arithmetic (no FMA contraction); only reduction order differs.
 */
#define _GNU_SOURCE
#include <math.h>
#include <omp.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t m, n, ne, nnz;
  int64_t *rp, *ci; double *va;      /* CSR(A)  */
  int64_t *cp, *ri; double *vt;      /* CSR(A') */
  /* round 5: what a tuned socket code would do (the GPU side has had both since round 1) --
   * 32-bit column / row indices when m, n, nnz < 2^31 (12 instead of 16 bytes streamed per nonzero; the 64-bit index
   * arrays are then freed), and one contiguous row range per thread cut at equal NONZEROS (not equal row counts), with
   * every array first-touched by the thread that will stream it; gathers are software-prefetched PF entries ahead. */
  int idx32, nthreads, pf;
  int32_t *ci32, *ri32;
  int64_t *cutA, *cutT;              /* [nthreads+1] row cuts of CSR(A) / CSR(A') */
  double *c, *b, *lb, *ub;
  double *x, *y, *aty, *xn, *yn, *atyn, *xbar, *sx, *sy;
  double step_size, primal_weight;
  int64_t total_iterations;
} omp_state;

/* Thread count and pinning are set explicitly (the process may already host an
 * OpenMP runtime initialised by another library, so environment variables are
 * too late): thread i is pinned to cpus[i]. */
static int g_threads = 0;
void omp_configure(int nthreads, const int *cpus) {
  g_threads = nthreads > 0 ? nthreads : 1;
  omp_set_dynamic(0);
  omp_set_num_threads(g_threads);
#pragma omp parallel num_threads(g_threads)
  {
    const int t = omp_get_thread_num();
    if (cpus) {
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET(cpus[t], &set);
      sched_setaffinity(0, sizeof(set), &set);
    }
  }
}

/* first-touch by the (pinned) worker threads so pages land on their socket */
static double *zd(int64_t k) {
  double *p = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < k; ++i) p[i] = 0.0;
  return p;
}
static double *cpd(const double *s, int64_t k) {
  double *p = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < k; ++i) p[i] = s[i];
  return p;
}
static int64_t *cpi(const int64_t *s, int64_t k) {
  int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < k; ++i) p[i] = s[i];
  return p;
}

/* gathers can be software-prefetched `pf` entries ahead (a random 8-byte read per nonzero is what bounds the CPU too);
 * default off (PDHG_CPU_PF, omp_set_prefetch): bench.py times both and reports the faster */
static int pf_distance(void) { const char *e = getenv("PDHG_CPU_PF"); return e ? atoi(e) : 0; }

/* row cuts at equal nonzeros: cut[t] = first row r with ptr[r] >= nnz*t/T */
static int64_t *nnz_cuts(const int64_t *ptr, int64_t rows, int T) {
  int64_t *cut = (int64_t *)malloc(sizeof(int64_t) * (size_t)(T + 1));
  const int64_t nnz = ptr[rows];
  cut[0] = 0;
  for (int t = 1; t < T; ++t) {
    const int64_t target = (int64_t)((__int128)nnz * t / T);
    int64_t lo = cut[t - 1], hi = rows;
    while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (ptr[mid] < target) lo = mid + 1; else hi = mid; }
    cut[t] = lo;
  }
  cut[T] = rows;
  return cut;
}

/* the arrays of one CSR re-allocated so that thread t first-touches (and later streams) the entries of ITS row range */
static void place_by_cuts(const int64_t *ptr, const int64_t *cut, int T, int64_t nnz, const int64_t *idx, const double *val,
                          int32_t **idx32_out, double **val_out) {
  int32_t *i32 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  double *v = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
#pragma omp parallel num_threads(T)
  {
    const int t = omp_get_thread_num();
    for (int64_t k = ptr[cut[t]]; k < ptr[cut[t + 1]]; ++k) { i32[k] = (int32_t)idx[k]; v[k] = val[k]; }
  }
  *idx32_out = i32;
  *val_out = v;
}

omp_state *omp_create(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, const double *c, const double *b, const double *lb,
                      const double *ub, int64_t ne) {
  omp_state *s = (omp_state *)calloc(1, sizeof(omp_state));
  const int64_t nnz = colptr[n];
  s->m = m; s->n = n; s->ne = ne; s->nnz = nnz;
  s->cp = cpi(colptr, n + 1);
  s->ri = cpi(rowval, nnz);
  s->vt = cpd(nzval, nnz);
  s->rp = (int64_t *)calloc((size_t)(m + 1), sizeof(int64_t));
  s->ci = cpi(rowval, nnz);   /* placeholder contents; first-touched in parallel, filled below */
  s->va = zd(nnz);
  for (int64_t k = 0; k < nnz; ++k) s->rp[rowval[k] + 1] += 1;
  for (int64_t i = 0; i < m; ++i) s->rp[i + 1] += s->rp[i];
  int64_t *next = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
  memcpy(next, s->rp, sizeof(int64_t) * (size_t)m);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) { int64_t p = next[rowval[k]]++; s->ci[p] = j; s->va[p] = nzval[k]; }
  free(next);
  s->c = cpd(c, n); s->b = cpd(b, m); s->lb = cpd(lb, n); s->ub = cpd(ub, n);
  s->x = zd(n); s->y = zd(m); s->aty = zd(n); s->xn = zd(n); s->yn = zd(m); s->atyn = zd(n);
  s->xbar = zd(n); s->sx = zd(n); s->sy = zd(m);
  s->primal_weight = 1.0;
  s->nthreads = g_threads > 0 ? g_threads : omp_get_max_threads();
  s->pf = pf_distance();
  s->cutA = nnz_cuts(s->rp, m, s->nthreads);
  s->cutT = nnz_cuts(s->cp, n, s->nthreads);
  const char *force64 = getenv("PDHG_CPU_IDX64");
  s->idx32 = m < INT32_MAX && n < INT32_MAX && nnz < INT32_MAX && !(force64 && force64[0] == '1');
  if (s->idx32) {
    double *v;
    place_by_cuts(s->rp, s->cutA, s->nthreads, nnz, s->ci, s->va, &s->ci32, &v);
    free(s->ci); free(s->va); s->ci = NULL; s->va = v;
    place_by_cuts(s->cp, s->cutT, s->nthreads, nnz, s->ri, s->vt, &s->ri32, &v);
    free(s->ri); free(s->vt); s->ri = NULL; s->vt = v;
  }
  return s;
}

void omp_destroy(omp_state *s) {
  if (!s) return;
  free(s->rp); free(s->ci); free(s->va); free(s->cp); free(s->ri); free(s->vt);
  free(s->ci32); free(s->ri32); free(s->cutA); free(s->cutT);
  free(s->c); free(s->b); free(s->lb); free(s->ub); free(s->x); free(s->y); free(s->aty);
  free(s->xn); free(s->yn); free(s->atyn); free(s->xbar); free(s->sx); free(s->sy); free(s);
}

void omp_set_scalars(omp_state *s, double step, double pw) { s->step_size = step; s->primal_weight = pw; }
double omp_get_step_size(const omp_state *s) { return s->step_size; }
int64_t omp_get_total_iterations(const omp_state *s) { return s->total_iterations; }
int omp_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }
void omp_set_prefetch(omp_state *s, int entries_ahead) { s->pf = entries_ahead > 0 ? entries_ahead : 0; }
int omp_index_bytes(const omp_state *s) { return s->idx32 ? 4 : 8; }
/* bytes one trial streams at the least: both matrix copies once, the vectors of SURVEY 8d's B_vec */
double omp_bytes_per_trial(const omp_state *s) {
  const double ib = s->idx32 ? 4.0 : 8.0;
  return 2.0 * (double)s->nnz * (8.0 + ib) + 8.0 * (double)(s->m + s->n + 2) + 8.0 * (13.0 * (double)s->n + 6.0 * (double)s->m);
}


/* rows [i0, i1) of a CSR with 32-bit indices: out-of-line so that the two tight loops are compiled on their own */
static inline double row_dot32(const double *restrict val, const int32_t *restrict idx, const double *restrict x, int64_t k0, int64_t k1) {
  double acc = 0.0;
  for (int64_t k = k0; k < k1; ++k) { const double p = val[k] * x[idx[k]]; acc = acc + p; }
  return acc;
}
static inline double row_dot32_pf(const double *restrict val, const int32_t *restrict idx, const double *restrict x, int64_t k0, int64_t k1,
                                  int pf) {
  double acc = 0.0;
  for (int64_t k = k0; k < k1; ++k) {
    __builtin_prefetch(&x[idx[k + pf]], 0, 0);
    const double p = val[k] * x[idx[k]]; acc = acc + p;
  }
  return acc;
}

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* take_step(::AdaptiveStepsizeParams) -- pdhg.jl:653-731; returns trials used */
int omp_take_step_adaptive(omp_state *s, double red_exp, double grow_exp) {
  double step = s->step_size;
  const double pw = s->primal_weight;
  int done = 0, iter = 0;
  const int64_t n = s->n, m = s->m, ne = s->ne;
  const int pf = s->pf;
  while (!done) {
    ++iter; s->total_iterations += 1;
    const double tau = step / pw, sigma = pw * step;
    double dx2 = 0.0, dy2 = 0.0, inter = 0.0;
#pragma omp parallel num_threads(s->nthreads)
    {
#pragma omp for schedule(static) reduction(+ : dx2)
      for (int64_t j = 0; j < n; ++j) {                       /* pdhg.jl:442-470, 486-487 */
        const double g = (0.0 + s->c[j]) - s->aty[j];
        const double t = tau * g;
        double v = s->x[j] - t;
        v = dmin(s->ub[j], dmax(s->lb[j], v));
        s->xn[j] = v;
        const double d = v - s->x[j];
        s->xbar[j] = v + 1.0 * d;
        dx2 += d * d;
      }
#pragma omp barrier
      const int tid_ = omp_get_thread_num();
      const int64_t iA0 = s->idx32 ? s->cutA[tid_] : (m * tid_) / s->nthreads, iA1 = s->idx32 ? s->cutA[tid_ + 1] : (m * (tid_ + 1)) / s->nthreads;
      double dy2_t = 0.0;
      for (int64_t i = iA0; i < iA1; ++i) {                   /* pdhg.jl:472-494 */
        double acc = 0.0;
        if (s->idx32) {
          const int64_t ke = s->rp[i + 1];
          acc = (pf > 0 && ke + pf <= s->nnz) ? row_dot32_pf(s->va, s->ci32, s->xbar, s->rp[i], ke, pf)
                                              : row_dot32(s->va, s->ci32, s->xbar, s->rp[i], ke);
        } else {
          for (int64_t k = s->rp[i]; k < s->rp[i + 1]; ++k) { const double p = s->va[k] * s->xbar[s->ci[k]]; acc = acc + p; }
        }
        const double dg = s->b[i] - acc;
        const double t = sigma * dg;
        double yn = s->y[i] + t;
        if (i >= ne) yn = dmax(yn, 0.0);
        s->yn[i] = yn;
        const double d = yn - s->y[i];
        dy2_t += d * d;
      }
#pragma omp atomic
      dy2 += dy2_t;
#pragma omp barrier
      const int64_t jT0 = s->idx32 ? s->cutT[tid_] : (n * tid_) / s->nthreads, jT1 = s->idx32 ? s->cutT[tid_ + 1] : (n * (tid_ + 1)) / s->nthreads;
      double inter_t = 0.0;
      for (int64_t j = jT0; j < jT1; ++j) {                   /* pdhg.jl:492, 527-549 */
        double acc = 0.0;
        if (s->idx32) {
          const int64_t ke = s->cp[j + 1];
          acc = (pf > 0 && ke + pf <= s->nnz) ? row_dot32_pf(s->vt, s->ri32, s->yn, s->cp[j], ke, pf)
                                              : row_dot32(s->vt, s->ri32, s->yn, s->cp[j], ke);
        } else {
          for (int64_t k = s->cp[j]; k < s->cp[j + 1]; ++k) { const double p = s->vt[k] * s->yn[s->ri[k]]; acc = acc + p; }
        }
        s->atyn[j] = acc;
        inter_t += (s->xn[j] - s->x[j]) * (acc - s->aty[j]);
      }
#pragma omp atomic
      inter += inter_t;
    }
    const double interaction = fabs(inter);
    const double nx = sqrt(dx2), ny = sqrt(dy2);
    const double movement = 0.5 * pw * (nx * nx) + (0.5 / pw) * (ny * ny);
    if (movement == 0.0) break;
    const double limit = interaction > 0 ? movement / interaction : INFINITY;
    if (step <= limit) {
      const double w = s->step_size;                          /* pdhg.jl:500-519 */
#pragma omp parallel
      {
#pragma omp for schedule(static)
        for (int64_t j = 0; j < n; ++j) { const double t = s->xn[j] * w; s->sx[j] = s->sx[j] + t; }
#pragma omp for schedule(static)
        for (int64_t i = 0; i < m; ++i) { const double t = s->yn[i] * w; s->sy[i] = s->sy[i] + t; }
      }
      double *t;
      t = s->x; s->x = s->xn; s->xn = t;
      t = s->y; s->y = s->yn; s->yn = t;
      t = s->aty; s->aty = s->atyn; s->atyn = t;
      done = 1;
    }
    const double k1 = (double)(s->total_iterations + 1);
    const double first = (1 - pow(k1, -red_exp)) * limit, second = (1 + pow(k1, -grow_exp)) * step;
    step = first < second ? first : second;
  }
  s->step_size = step;
  return iter;
}

void omp_get_xy(const omp_state *s, double *x, double *y) {
  memcpy(x, s->x, sizeof(double) * (size_t)s->n);
  memcpy(y, s->y, sizeof(double) * (size_t)s->m);
}
