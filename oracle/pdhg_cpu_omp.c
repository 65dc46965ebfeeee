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
  int64_t m, n, ne;
  int64_t *rp, *ci; double *va;      /* CSR(A)  */
  int64_t *cp, *ri; double *vt;      /* CSR(A') */
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

omp_state *omp_create(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, const double *c, const double *b, const double *lb,
                      const double *ub, int64_t ne) {
  omp_state *s = (omp_state *)calloc(1, sizeof(omp_state));
  const int64_t nnz = colptr[n];
  s->m = m; s->n = n; s->ne = ne;
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
  return s;
}

void omp_destroy(omp_state *s) {
  if (!s) return;
  free(s->rp); free(s->ci); free(s->va); free(s->cp); free(s->ri); free(s->vt);
  free(s->c); free(s->b); free(s->lb); free(s->ub); free(s->x); free(s->y); free(s->aty);
  free(s->xn); free(s->yn); free(s->atyn); free(s->xbar); free(s->sx); free(s->sy); free(s);
}

void omp_set_scalars(omp_state *s, double step, double pw) { s->step_size = step; s->primal_weight = pw; }
double omp_get_step_size(const omp_state *s) { return s->step_size; }
int64_t omp_get_total_iterations(const omp_state *s) { return s->total_iterations; }
int omp_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* take_step(::AdaptiveStepsizeParams) -- pdhg.jl:653-731; returns trials used */
int omp_take_step_adaptive(omp_state *s, double red_exp, double grow_exp) {
  double step = s->step_size;
  const double pw = s->primal_weight;
  int done = 0, iter = 0;
  const int64_t n = s->n, m = s->m, ne = s->ne;
  while (!done) {
    ++iter; s->total_iterations += 1;
    const double tau = step / pw, sigma = pw * step;
    double dx2 = 0.0, dy2 = 0.0, inter = 0.0;
#pragma omp parallel
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
#pragma omp for schedule(static) reduction(+ : dy2)
      for (int64_t i = 0; i < m; ++i) {                       /* pdhg.jl:472-494 */
        double acc = 0.0;
        for (int64_t k = s->rp[i]; k < s->rp[i + 1]; ++k) { const double p = s->va[k] * s->xbar[s->ci[k]]; acc = acc + p; }
        const double dg = s->b[i] - acc;
        const double t = sigma * dg;
        double yn = s->y[i] + t;
        if (i >= ne) yn = dmax(yn, 0.0);
        s->yn[i] = yn;
        const double d = yn - s->y[i];
        dy2 += d * d;
      }
#pragma omp for schedule(static) reduction(+ : inter)
      for (int64_t j = 0; j < n; ++j) {                       /* pdhg.jl:492, 527-549 */
        double acc = 0.0;
        for (int64_t k = s->cp[j]; k < s->cp[j + 1]; ++k) { const double p = s->vt[k] * s->yn[s->ri[k]]; acc = acc + p; }
        s->atyn[j] = acc;
        inter += (s->xn[j] - s->x[j]) * (acc - s->aty[j]);
      }
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
