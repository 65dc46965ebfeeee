// rescale_kernels.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// Diagonal rescaling on the device (N2): Ruiz, L2 and Pock-Chambolle passes over both CSR copies.
#pragma once

namespace {

// ============================================================ rescaling on the device (N2)
// rescale_problem (preprocess.jl:631-687) applied in place to every resident
// layout.  One wave per CSR row (one-time work, simplicity over speed).
enum { ROP_MAXABS = 0, ROP_SUMPOW = 1, ROP_SUMSQ_SCALED = 2 };

// out[r] = max |a| ; sum |a|^p (+ structural zeros when p == 0: Julia's
// mapreduce visits them and 0.0^0 == 1.0) ; sum (a * inv_scale[r])^2
template <int OP>
__global__ __launch_bounds__(TPB) void row_op_kernel(CsrView A, int cols, double pexp,
                                                     const double *__restrict__ inv_scale,
                                                     double *__restrict__ out, int long_thr) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int r = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
  if (r >= A.rows) return;
  const int k0 = A.rowptr[r], k1 = A.rowptr[r + 1];
  if (k1 - k0 > long_thr) return;        // long rows: row_op_long_* below (one workgroup per 8192-entry chunk)
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
                                                        const double *__restrict__ inv_d, int transposed,
                                                        int skip_long) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int r = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
  if (r >= rows) return;
  const int k0 = rowptr[r], k1 = rowptr[r + 1];
  if (skip_long && k1 - k0 > skip_long) return;      // skip_long = the long-row threshold: scale_long_kernel does those rows chunk by chunk
  for (int k = k0 + lane; k < k1; k += WAVE) {
    const int c = col[k];
    const double ie = transposed ? inv_e[c] : inv_e[r];
    const double id = transposed ? inv_d[r] : inv_d[c];
    val[k] = (val[k] * ie) * id;
  }
}

// Rows longer than BLOCK_NNZ (a dense equality row, an intercept column: up to n entries)
// would keep one wave busy for milliseconds per pass: one workgroup per LONG_CHUNK entries.
__global__ __launch_bounds__(TPB) void scale_long_kernel(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                         double *__restrict__ val, const int *__restrict__ chunk_row,
                                                         const int *__restrict__ chunk_off,
                                                         const double *__restrict__ inv_e,
                                                         const double *__restrict__ inv_d, int transposed) {
  const int r = chunk_row[blockIdx.x];
  const int kb = rowptr[r] + chunk_off[blockIdx.x];
  const int ke = min(kb + LONG_CHUNK, rowptr[r + 1]);
  for (int k = kb + threadIdx.x; k < ke; k += TPB) {
    const int c = col[k];
    const double ie = transposed ? inv_e[c] : inv_e[r];
    const double id = transposed ? inv_d[r] : inv_d[c];
    val[k] = (val[k] * ie) * id;
  }
}

// chunk partial of a long row's statistic (same three operations as row_op_kernel)
template <int OP>
__global__ __launch_bounds__(TPB) void row_op_long_partial_kernel(CsrView A, const int *__restrict__ chunk_row,
                                                                  const int *__restrict__ chunk_off, double pexp,
                                                                  const double *__restrict__ inv_scale,
                                                                  double *__restrict__ chunk_partial) {
  __shared__ double red[TPB / WAVE];
  const int r = chunk_row[blockIdx.x];
  const int kb = A.rowptr[r] + chunk_off[blockIdx.x];
  const int ke = min(kb + LONG_CHUNK, A.rowptr[r + 1]);
  const double sc = (OP == ROP_SUMSQ_SCALED) ? inv_scale[r] : 1.0;
  double acc = 0.0;
  for (int k = kb + threadIdx.x; k < ke; k += TPB) {
    const double a = A.val[k];
    if (OP == ROP_MAXABS) acc = fmax(acc, fabs(a));
    else if (OP == ROP_SUMPOW) acc += pow(fabs(a), pexp);
    else { const double t = a * sc; acc += t * t; }
  }
  acc = (OP == ROP_MAXABS) ? wave_max(acc) : wave_sum(acc);
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  if (lane == 0) red[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = red[0];
    for (int w = 1; w < TPB / WAVE; ++w) t = (OP == ROP_MAXABS) ? fmax(t, red[w]) : t + red[w];
    chunk_partial[blockIdx.x] = t;
  }
}
// one lane per long row: combine its chunk partials in order (+ the structural zeros of 0^0)
template <int OP>
__global__ __launch_bounds__(TPB) void row_op_long_final_kernel(CsrView A, const int *__restrict__ long_row,
                                                                const int *__restrict__ long_chunk_ptr, int nlong,
                                                                const double *__restrict__ chunk_partial, int cols,
                                                                double pexp, double *__restrict__ out) {
  const int l = blockIdx.x * TPB + threadIdx.x;
  if (l >= nlong) return;
  const int r = long_row[l];
  double t = 0.0;
  for (int c = long_chunk_ptr[l]; c < long_chunk_ptr[l + 1]; ++c)
    t = (OP == ROP_MAXABS) ? fmax(t, chunk_partial[c]) : t + chunk_partial[c];
  if (OP == ROP_SUMPOW && pexp == 0.0) t += (double)(cols - (A.rowptr[r + 1] - A.rowptr[r]));
  out[r] = t;
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
      const int c = tile + (int)(p & cmask);      // step_tile holds the tile's first column
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
// a = max(a, b): column max over [A; Q] for the QP Ruiz step (preprocess.jl:425-433)
__global__ __launch_bounds__(TPB) void resc_max_kernel(int n, double *__restrict__ a, const double *__restrict__ b) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) a[i] = fmax(a[i], b[i]);
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

}  // namespace
