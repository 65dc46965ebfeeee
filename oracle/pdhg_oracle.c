/*
 * pdhg_oracle.c -- CPU ORACLE for the PDHG inner step of FirstOrderLp.jl.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (firstorderlp.jl_amd/) never imports, links or calls anything
 * in oracle/.
 *
 * It is a literal, single-threaded, sequential-sum restatement of the Julia
 * reference's per-iteration arithmetic, one C statement per Julia broadcast,
 * compiled with -ffp-contract=off so that no multiply-add is fused (Julia does
 * not contract a*b+c either).  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/src).
 *
 * Parity pin: the reference is Julia and cannot run here; this restatement is
 * pinned through the reference's own known-answer tests
 * (test/test_primal_dual_hybrid_gradient.jl, test/test_saddle_point.jl) -- see
 * tests/test_kat_*.py -- and golden vectors generated from it afterwards are
 * committed under tests/golden/.
 *
 * Arithmetic that lives outside /root/reference (Julia stdlib SparseArrays /
 * LinearAlgebra, shipped with Julia 1.6/1.7, no pinned version in
 * Manifest.toml) is restated from its published algorithm:
 *   A*x   on SparseMatrixCSC : for col j ascending, for k in nzrange(A,j):
 *                              y[rowval[k]] += nzval[k]*x[j]     (y zeroed first)
 *   A'*y  on Adjoint{CSC}    : per column j, tmp=0; tmp += nzval[k]*y[rowval[k]]
 *                              for k ascending; out[j] = tmp
 *   norm(v)  : sqrt(sum v_i^2) (generic_norm2 for length<32, BLAS dnrm2 above;
 *              we use the sequential form for all lengths)
 *   dot(a,b) : sequential sum a_i*b_i (BLAS ddot order is implementation
 *              defined; sequential is the stand-in)
 * Test aid beside the literal restatement: oracle_set_exact_sums(1) takes the three
 * step-acceptance sums (pdhg.jl:540-547) in double-double instead -- the correctly rounded
 * exact sum of the same terms, independent of summation order.  Off by default; every KAT and
 * golden vector is produced with it off.
 *
 * Index convention here: 0-based int64 CSC (the Python wrapper converts).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t m, n, num_equalities;
  /* constraint_matrix, CSC (quadratic_programming.jl:64) */
  int64_t *colptr, *rowval;
  double *nzval;
  /* objective_matrix, CSC n x n (quadratic_programming.jl:49); q_nnz may be 0 */
  int64_t q_nnz;
  int64_t *q_colptr, *q_rowval;
  double *q_nzval;
  int q_is_zero; /* iszero(objective_matrix), pdhg.jl:536 */
  double *c, *b, *lb, *ub;

  /* PdhgSolverState (primal_dual_hybrid_gradient.jl:205-258) */
  double *x, *y, *delta_x, *delta_y, *aty;
  /* SolutionWeightedAverage (saddle_point.jl:215-222) */
  double *sum_x, *sum_y;
  int64_t sum_x_count, sum_y_count;
  double sum_x_weights, sum_y_weights;

  double step_size, primal_weight;
  int numerical_error;
  int exact_sums; /* test aid, NOT the reference's arithmetic: see oracle_set_exact_sums */
  double cumulative_kkt_passes;
  int64_t total_number_iterations;
  double ratio_step_sizes;

  /* scratch for trial iterates */
  double *x_next, *y_next, *aty_next, *tmp_n, *tmp_m, *tmp_n2;
} oracle_state;

static double *dup_d(const double *src, int64_t len) {
  double *p = (double *)malloc(sizeof(double) * (size_t)(len > 0 ? len : 1));
  if (src && len > 0) memcpy(p, src, sizeof(double) * (size_t)len);
  return p;
}
static int64_t *dup_i(const int64_t *src, int64_t len) {
  int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)(len > 0 ? len : 1));
  if (src && len > 0) memcpy(p, src, sizeof(int64_t) * (size_t)len);
  return p;
}
static double *zeros_d(int64_t len) {
  return (double *)calloc((size_t)(len > 0 ? len : 1), sizeof(double));
}

/* ---- Julia stdlib SparseArrays restatements ------------------------------ */

/* A*x for SparseMatrixCSC (call sites saddle_point.jl:1106, 1098). */
void oracle_spmv(int64_t m, int64_t n, const int64_t *colptr,
                 const int64_t *rowval, const double *nzval, const double *x,
                 double *out) {
  for (int64_t i = 0; i < m; ++i) out[i] = 0.0;
  for (int64_t j = 0; j < n; ++j) {
    const double xj = x[j];
    for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
      const double p = nzval[k] * xj;
      out[rowval[k]] = out[rowval[k]] + p;
    }
  }
}

/* A'*y for Adjoint{SparseMatrixCSC} (call site pdhg.jl:492, 1021). */
void oracle_spmv_t(int64_t m, int64_t n, const int64_t *colptr,
                   const int64_t *rowval, const double *nzval, const double *y,
                   double *out) {
  (void)m;
  for (int64_t j = 0; j < n; ++j) {
    double tmp = 0.0;
    for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
      const double p = nzval[k] * y[rowval[k]];
      tmp = tmp + p;
    }
    out[j] = tmp;
  }
}

/* sum v_i^2, sequential (LinearAlgebra.norm restated as sqrt of this). */
static double sumsq(const double *v, int64_t len) {
  double s = 0.0;
  for (int64_t i = 0; i < len; ++i) {
    const double p = v[i] * v[i];
    s = s + p;
  }
  return s;
}
/* (hi, lo) += t, TwoSum: the rounding error of the addition goes to lo */
static inline void dd_add(double *hi, double *lo, double t) {
  const double s = *hi + t;
  const double bb = s - *hi;
  const double e = (*hi - (s - bb)) + (t - bb);
  *hi = s;
  *lo = *lo + e;
}
/* sum of a[i] * b[i] (each product rounded to a double), accumulated in double-double */
static double dd_dot(const double *a, const double *b, int64_t len) {
  double hi = 0.0, lo = 0.0;
  for (int64_t i = 0; i < len; ++i) {
    const double p = a[i] * b[i];
    dd_add(&hi, &lo, p);
  }
  return hi + lo;
}

static double dot_seq(const double *a, const double *b, int64_t len) {
  double s = 0.0;
  for (int64_t i = 0; i < len; ++i) {
    const double p = a[i] * b[i];
    s = s + p;
  }
  return s;
}

/* ---- construction --------------------------------------------------------- */

oracle_state *oracle_create(int64_t m, int64_t n, const int64_t *colptr,
                            const int64_t *rowval, const double *nzval,
                            int64_t q_nnz, const int64_t *q_colptr,
                            const int64_t *q_rowval, const double *q_nzval,
                            const double *c, const double *b, const double *lb,
                            const double *ub, int64_t num_equalities) {
  oracle_state *s = (oracle_state *)calloc(1, sizeof(oracle_state));
  s->m = m;
  s->n = n;
  s->num_equalities = num_equalities;
  const int64_t nnz = colptr[n];
  s->colptr = dup_i(colptr, n + 1);
  s->rowval = dup_i(rowval, nnz);
  s->nzval = dup_d(nzval, nnz);
  s->q_nnz = q_nnz;
  if (q_colptr) {
    s->q_colptr = dup_i(q_colptr, n + 1);
  } else {
    s->q_colptr = (int64_t *)calloc((size_t)(n + 1), sizeof(int64_t));
  }
  s->q_rowval = dup_i(q_rowval, q_nnz);
  s->q_nzval = dup_d(q_nzval, q_nnz);
  s->q_is_zero = 1;
  for (int64_t k = 0; k < q_nnz; ++k)
    if (q_nzval[k] != 0.0) s->q_is_zero = 0;
  s->c = dup_d(c, n);
  s->b = dup_d(b, m);
  s->lb = dup_d(lb, n);
  s->ub = dup_d(ub, n);
  /* zeros(primal_size) etc., pdhg.jl:805-819 */
  s->x = zeros_d(n);
  s->y = zeros_d(m);
  s->delta_x = zeros_d(n);
  s->delta_y = zeros_d(m);
  s->aty = zeros_d(n);
  s->sum_x = zeros_d(n);
  s->sum_y = zeros_d(m);
  s->step_size = 0.0;
  s->primal_weight = 1.0;
  s->ratio_step_sizes = 1.0;
  s->x_next = zeros_d(n);
  s->y_next = zeros_d(m);
  s->aty_next = zeros_d(n);
  s->tmp_n = zeros_d(n);
  s->tmp_n2 = zeros_d(n);
  s->tmp_m = zeros_d(m);
  return s;
}

void oracle_destroy(oracle_state *s) {
  if (!s) return;
  free(s->colptr); free(s->rowval); free(s->nzval);
  free(s->q_colptr); free(s->q_rowval); free(s->q_nzval);
  free(s->c); free(s->b); free(s->lb); free(s->ub);
  free(s->x); free(s->y); free(s->delta_x); free(s->delta_y); free(s->aty);
  free(s->sum_x); free(s->sum_y);
  free(s->x_next); free(s->y_next); free(s->aty_next);
  free(s->tmp_n); free(s->tmp_n2); free(s->tmp_m);
  free(s);
}

/* ---- accessors (the Python wrapper reads/writes the state through these) -- */
#define GETTER(name, field, len)                                   \
  void oracle_get_##name(const oracle_state *s, double *out) {     \
    memcpy(out, s->field, sizeof(double) * (size_t)(s->len));      \
  }                                                                \
  void oracle_set_##name(oracle_state *s, const double *in) {      \
    memcpy(s->field, in, sizeof(double) * (size_t)(s->len));       \
  }
GETTER(x, x, n)
GETTER(y, y, m)
GETTER(aty, aty, n)
GETTER(delta_x, delta_x, n)
GETTER(delta_y, delta_y, m)
GETTER(sum_x, sum_x, n)
GETTER(sum_y, sum_y, m)
GETTER(x_next, x_next, n)
GETTER(y_next, y_next, m)
GETTER(aty_next, aty_next, n)

double oracle_get_step_size(const oracle_state *s) { return s->step_size; }
void oracle_set_step_size(oracle_state *s, double v) { s->step_size = v; }
double oracle_get_primal_weight(const oracle_state *s) { return s->primal_weight; }
void oracle_set_primal_weight(oracle_state *s, double v) { s->primal_weight = v; }
double oracle_get_ratio_step_sizes(const oracle_state *s) { return s->ratio_step_sizes; }
void oracle_set_ratio_step_sizes(oracle_state *s, double v) { s->ratio_step_sizes = v; }
int oracle_get_numerical_error(const oracle_state *s) { return s->numerical_error; }
void oracle_set_numerical_error(oracle_state *s, int v) { s->numerical_error = v; }
/* Exact-sums mode (test aid).  The reference takes the three step-acceptance sums with BLAS
 * dot / nrm2 (pdhg.jl:540-547), whose summation order is implementation-defined; the default here is
 * the plain sequential loop.  With exact_sums != 0 the same TERMS (each product rounded to a double
 * first) are added in double-double arithmetic and rounded once at the end: the correctly rounded
 * exact sum (up to a ~1e-14 chance per sum), which no longer depends on the order of the additions.
 * The HIP library accumulates these sums the same way, so in this mode the two produce bitwise
 * identical scalars, hence -- everything else being bit-exact -- bitwise identical free-running
 * trajectories (tests/test_gpu_exact_sums.py). */
void oracle_set_exact_sums(oracle_state *s, int v) { s->exact_sums = v; }
int oracle_get_exact_sums(const oracle_state *s) { return s->exact_sums; }
double oracle_get_cumulative_kkt_passes(const oracle_state *s) { return s->cumulative_kkt_passes; }
void oracle_set_cumulative_kkt_passes(oracle_state *s, double v) { s->cumulative_kkt_passes = v; }
int64_t oracle_get_total_number_iterations(const oracle_state *s) { return s->total_number_iterations; }
void oracle_get_average_counts(const oracle_state *s, int64_t *counts, double *weights) {
  counts[0] = s->sum_x_count; counts[1] = s->sum_y_count;
  weights[0] = s->sum_x_weights; weights[1] = s->sum_y_weights;
}

/* ---- saddle_point.jl restatements ----------------------------------------- */

/* projection!/project_primal!  (saddle_point.jl:82-106):
 *   primal[idx] = min(ub[idx], max(lb[idx], primal[idx]))
 * Julia's min/max on Float64 propagate NaN; inputs here are never NaN. */
static inline double jl_max(double a, double b) { return (a > b) ? a : ((b > a) ? b : ((a != a) ? a : ((b != b) ? b : (signbit(a) ? b : a)))); }
static inline double jl_min(double a, double b) { return (a < b) ? a : ((b < a) ? b : ((a != a) ? a : ((b != b) ? b : (signbit(a) ? a : b)))); }

static void project_primal(const oracle_state *s, double *primal) {
  for (int64_t j = 0; j < s->n; ++j)
    primal[j] = jl_min(s->ub[j], jl_max(s->lb[j], primal[j]));
}

/* project_dual! (saddle_point.jl:110-117): only inequality_range
 * (quadratic_programming.jl:302-304) = num_equalities+1 : m. */
static void project_dual(const oracle_state *s, double *dual) {
  for (int64_t i = s->num_equalities; i < s->m; ++i)
    dual[i] = jl_max(dual[i], 0.0);
}

/* compute_primal_gradient_from_dual_product (saddle_point.jl:1093-1100):
 *   objective_matrix * x .+ objective_vector .- dual_product */
static void primal_gradient_from_dual_product(const oracle_state *s,
                                              const double *x,
                                              const double *dual_product,
                                              double *g) {
  oracle_spmv(s->n, s->n, s->q_colptr, s->q_rowval, s->q_nzval, x, s->tmp_n2);
  for (int64_t j = 0; j < s->n; ++j) {
    const double t = s->tmp_n2[j] + s->c[j];
    g[j] = t - dual_product[j];
  }
}

/* add_to_{primal,dual}_solution_weighted_average (saddle_point.jl:252-276):
 *   sum .+= current * weight ; count += 1 ; weights += weight */
void oracle_add_to_primal_average(oracle_state *s, const double *x, double weight) {
  for (int64_t j = 0; j < s->n; ++j) {
    const double t = x[j] * weight;
    s->sum_x[j] = s->sum_x[j] + t;
  }
  s->sum_x_count += 1;
  s->sum_x_weights += weight;
}
void oracle_add_to_dual_average(oracle_state *s, const double *y, double weight) {
  for (int64_t i = 0; i < s->m; ++i) {
    const double t = y[i] * weight;
    s->sum_y[i] = s->sum_y[i] + t;
  }
  s->sum_y_count += 1;
  s->sum_y_weights += weight;
}

/* reset_solution_weighted_average (saddle_point.jl:238-250) */
void oracle_reset_average(oracle_state *s) {
  memset(s->sum_x, 0, sizeof(double) * (size_t)s->n);
  memset(s->sum_y, 0, sizeof(double) * (size_t)s->m);
  s->sum_x_count = 0;
  s->sum_y_count = 0;
  s->sum_x_weights = 0.0;
  s->sum_y_weights = 0.0;
}

/* compute_average (saddle_point.jl:296-301): sum / weights (division) */
void oracle_compute_average(const oracle_state *s, double *x_avg, double *y_avg) {
  for (int64_t j = 0; j < s->n; ++j) x_avg[j] = s->sum_x[j] / s->sum_x_weights;
  for (int64_t i = 0; i < s->m; ++i) y_avg[i] = s->sum_y[i] / s->sum_y_weights;
}

/* Recompute the cached A'y after x,y were overwritten (pdhg.jl:1018-1022). */
void oracle_recompute_dual_product(oracle_state *s) {
  oracle_spmv_t(s->m, s->n, s->colptr, s->rowval, s->nzval, s->y, s->aty);
}

/* ---- primal_dual_hybrid_gradient.jl restatements --------------------------- */

/* compute_next_primal_solution (pdhg.jl:442-470):
 *   g = primal gradient; next = x .- (step/pw) .* g ; project_primal! */
void oracle_compute_next_primal(oracle_state *s, double step_size,
                                double primal_weight, double *x_next) {
  primal_gradient_from_dual_product(s, s->x, s->aty, s->tmp_n);
  const double tau = step_size / primal_weight;
  for (int64_t j = 0; j < s->n; ++j) {
    const double t = tau * s->tmp_n[j];
    x_next[j] = s->x[j] - t;
  }
  project_primal(s, x_next);
}

/* compute_next_dual_solution (pdhg.jl:472-494):
 *   xbar = next .+ theta .* (next - x)
 *   dual_gradient = b .- A*xbar                      (saddle_point.jl:1102-1107)
 *   next_dual = y .+ (pw*step) .* dual_gradient ; project_dual!
 *   next_dual_product = A' * next_dual */
void oracle_compute_next_dual(oracle_state *s, const double *x_next,
                              double step_size, double primal_weight,
                              double theta, double *y_next, double *aty_next) {
  for (int64_t j = 0; j < s->n; ++j) {
    const double d = x_next[j] - s->x[j];
    const double t = theta * d;
    s->tmp_n[j] = x_next[j] + t;
  }
  oracle_spmv(s->m, s->n, s->colptr, s->rowval, s->nzval, s->tmp_n, s->tmp_m);
  const double sigma = primal_weight * step_size;
  for (int64_t i = 0; i < s->m; ++i) {
    const double dg = s->b[i] - s->tmp_m[i];
    const double t = sigma * dg;
    y_next[i] = s->y[i] + t;
  }
  project_dual(s, y_next);
  oracle_spmv_t(s->m, s->n, s->colptr, s->rowval, s->nzval, y_next, aty_next);
}

/* compute_interaction_and_movement (pdhg.jl:527-549).
 * raw[0] = delta_primal' * (next_dual_product - current_dual_product)
 * raw[1] = sum delta_primal^2, raw[2] = sum delta_dual^2 (before sqrt/square)
 * raw[3] = sum (next_dual_product - current_dual_product)^2
 * raw[4] = 0.5 * delta_primal' * Q * delta_primal  -- the C-ABI's out[5] */
void oracle_interaction_and_movement(oracle_state *s, const double *x_next,
                                     const double *y_next,
                                     const double *aty_next,
                                     double *interaction, double *movement,
                                     double *raw) {
  for (int64_t j = 0; j < s->n; ++j) s->tmp_n[j] = x_next[j] - s->x[j];
  for (int64_t i = 0; i < s->m; ++i) s->tmp_m[i] = y_next[i] - s->y[i];
  double primal_objective_interaction = 0.0;
  if (!s->q_is_zero) {
    /* 0.5 * (dx' * Q * dx): Julia evaluates (dx' * Q) * dx left to right;
     * dx'*Q is the adjoint-vector x CSC product = per-column dots. */
    oracle_spmv_t(s->n, s->n, s->q_colptr, s->q_rowval, s->q_nzval, s->tmp_n,
                  s->tmp_n2);
    primal_objective_interaction = 0.5 * (s->exact_sums ? dd_dot(s->tmp_n2, s->tmp_n, s->n)
                                                        : dot_seq(s->tmp_n2, s->tmp_n, s->n));
  }
  double pdi = 0.0, pdi_lo = 0.0;
  for (int64_t j = 0; j < s->n; ++j) {
    const double dd = aty_next[j] - s->aty[j];
    const double p = s->tmp_n[j] * dd;
    if (s->exact_sums) dd_add(&pdi, &pdi_lo, p);
    else pdi = pdi + p;
  }
  if (s->exact_sums) pdi = pdi + pdi_lo;
  const double ssx = s->exact_sums ? dd_dot(s->tmp_n, s->tmp_n, s->n) : sumsq(s->tmp_n, s->n);
  const double ssy = s->exact_sums ? dd_dot(s->tmp_m, s->tmp_m, s->m) : sumsq(s->tmp_m, s->m);
  const double nx = sqrt(ssx), ny = sqrt(ssy);
  *interaction = fabs(pdi) + fabs(primal_objective_interaction);
  *movement = 0.5 * s->primal_weight * (nx * nx) +
              (0.5 / s->primal_weight) * (ny * ny);
  if (raw) {
    /* raw[3] = sum (A'y' - A'y)^2 (Malitsky-Pock test, pdhg.jl:615) */
    double ssd = 0.0, ssd_lo = 0.0;
    for (int64_t j = 0; j < s->n; ++j) {
      const double dd = aty_next[j] - s->aty[j];
      const double p = dd * dd;
      if (s->exact_sums) dd_add(&ssd, &ssd_lo, p);
      else ssd = ssd + p;
    }
    if (s->exact_sums) ssd = ssd + ssd_lo;
    raw[0] = pdi; raw[1] = ssx; raw[2] = ssy; raw[3] = ssd;
    raw[4] = primal_objective_interaction;
  }
}

/* update_solution_in_solver_state (pdhg.jl:500-519).  Note quirk Q1: the
 * average weight is solver_state.step_size, i.e. the value on ENTRY to
 * take_step, not the accepted trial's step. */
void oracle_update_solution(oracle_state *s, const double *x_next,
                            const double *y_next, const double *aty_next) {
  for (int64_t j = 0; j < s->n; ++j) s->delta_x[j] = x_next[j] - s->x[j];
  for (int64_t i = 0; i < s->m; ++i) s->delta_y[i] = y_next[i] - s->y[i];
  memcpy(s->x, x_next, sizeof(double) * (size_t)s->n);
  memcpy(s->y, y_next, sizeof(double) * (size_t)s->m);
  memcpy(s->aty, aty_next, sizeof(double) * (size_t)s->n);
  const double weight = s->step_size;
  oracle_add_to_primal_average(s, s->x, weight);
  oracle_add_to_dual_average(s, s->y, weight);
}

/* take_step(::AdaptiveStepsizeParams, ...) (pdhg.jl:653-731).
 * Returns the number of trials (inner iterations) this call made. */
int oracle_take_step_adaptive(oracle_state *s, double reduction_exponent,
                              double growth_exponent) {
  double step_size = s->step_size;
  int done = 0, iter = 0;
  while (!done) {
    iter += 1;
    s->total_number_iterations += 1;
    oracle_compute_next_primal(s, step_size, s->primal_weight, s->x_next);
    oracle_compute_next_dual(s, s->x_next, step_size, s->primal_weight, 1.0,
                             s->y_next, s->aty_next);
    double interaction, movement;
    oracle_interaction_and_movement(s, s->x_next, s->y_next, s->aty_next,
                                    &interaction, &movement, NULL);
    s->cumulative_kkt_passes += 1;
    if (movement == 0.0) {
      s->numerical_error = 1;
      break;
    }
    double step_size_limit;
    if (interaction > 0)
      step_size_limit = movement / interaction;
    else
      step_size_limit = INFINITY;
    if (step_size <= step_size_limit) {
      oracle_update_solution(s, s->x_next, s->y_next, s->aty_next);
      done = 1;
    }
    const double k1 = (double)(s->total_number_iterations + 1);
    const double first_term =
        (1 - pow(k1, -reduction_exponent)) * step_size_limit;
    const double second_term = (1 + pow(k1, -growth_exponent)) * step_size;
    /* Julia's min: NaN if either operand is NaN */
    step_size = (first_term != first_term || second_term != second_term)
                    ? NAN : (first_term < second_term ? first_term : second_term);
  }
  s->step_size = step_size;
  return iter;
}

/* take_step(::ConstantStepsizeParams, ...) (pdhg.jl:737-767). */
int oracle_take_step_constant(oracle_state *s) {
  oracle_compute_next_primal(s, s->step_size, s->primal_weight, s->x_next);
  oracle_compute_next_dual(s, s->x_next, s->step_size, s->primal_weight, 1.0,
                           s->y_next, s->aty_next);
  s->cumulative_kkt_passes += 1;
  oracle_update_solution(s, s->x_next, s->y_next, s->aty_next);
  return 1;
}

/* take_step(::MalitskyPockStepsizeParameters, ...) (pdhg.jl:555-647).
 * LP only (pdhg.jl:560-565): returns -1 for a QP. */
int oracle_take_step_malitsky_pock(oracle_state *s, double downscaling_factor,
                                   double breaking_factor,
                                   double interpolation_coefficient) {
  if (!s->q_is_zero) return -1;
  double step_size = s->step_size;
  double ratio_step_sizes = s->ratio_step_sizes;
  int done = 0, iter = 0;
  oracle_compute_next_primal(s, step_size, s->primal_weight, s->x_next);
  s->cumulative_kkt_passes += 0.5;
  step_size = step_size + interpolation_coefficient *
                              (sqrt(1 + ratio_step_sizes) - 1) * step_size;
  const int max_iter = 60;
  while (!done && iter < max_iter) {
    iter += 1;
    s->total_number_iterations += 1;
    ratio_step_sizes = step_size / s->step_size;
    oracle_compute_next_dual(s, s->x_next, step_size, s->primal_weight,
                             ratio_step_sizes, s->y_next, s->aty_next);
    for (int64_t i = 0; i < s->m; ++i) s->tmp_m[i] = s->y_next[i] - s->y[i];
    for (int64_t j = 0; j < s->n; ++j) s->tmp_n[j] = s->aty_next[j] - s->aty[j];
    s->cumulative_kkt_passes += 0.5;
    const double n_dp = sqrt(s->exact_sums ? dd_dot(s->tmp_n, s->tmp_n, s->n) : sumsq(s->tmp_n, s->n));
    const double n_dd = sqrt(s->exact_sums ? dd_dot(s->tmp_m, s->tmp_m, s->m) : sumsq(s->tmp_m, s->m));
    if (step_size * n_dp <= breaking_factor * n_dd) {
      if (s->sum_x_count == 0) {
        oracle_add_to_primal_average(s, s->x, step_size * ratio_step_sizes);
      }
      oracle_update_solution(s, s->x_next, s->y_next, s->aty_next);
      done = 1;
    } else {
      step_size *= downscaling_factor;
    }
  }
  if (iter == max_iter && !done) {
    s->numerical_error = 1;
    return iter;
  }
  s->step_size = step_size;
  s->ratio_step_sizes = ratio_step_sizes;
  return iter;
}
