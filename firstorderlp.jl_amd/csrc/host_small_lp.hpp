// host_small_lp.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// small LPs: a batch of take_steps in one workgroup (small_lp_kernel.hpp): eligibility, staging, launch (host side).

// ---- small LPs: a batch of take_steps in one workgroup, vectors in LDS (small_lp_kernel.hpp) ---------------------
int flush_pending(const Shards &L);
bool small_lp_eligible(pdhg_handle *h) {
  if (h->small_lp_mode < 0) {
    const char *ev = getenv("PDHG_SMALL_LP");
    const size_t lds = sizeof(double) * (9 * (size_t)h->n + 4 * (size_t)h->m);
    bool on = !h->grp && !h->has_q && h->n > 0 && h->m > 0 && h->A.segs.empty() && h->At.segs.empty() && !h->A.tiled && !h->At.tiled && h->A.slabs.empty() &&
              h->At.slabs.empty() && h->A.max_row_nnz <= SMALL_MAX_ROW && h->At.max_row_nnz <= SMALL_MAX_ROW &&
              lds <= (size_t)144 * 1024;
    if (ev) on = on && ev[0] != '0';
    h->small_lp_mode = on ? 1 : 0;
  }
  return h->small_lp_mode == 1 && !h->profile;
}

// returns 1 when not eligible (nothing launched)
int small_lp_steps(pdhg_handle *h, int64_t n_steps, double reduction_exponent, double growth_exponent, double *step_size_io,
                   double primal_weight, int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
                   int *numerical_error_out, int64_t *steps_done, double *unfinished_entry) {
  *steps_done = 0;
  *unfinished_entry = 0.0;
  if (!small_lp_eligible(h)) return 1;
  HIP_TRY(hipSetDevice(h->device));
  int rc;
  if (h->pend_x != h->pend_y) { Shards L = shards_of(h); if ((rc = flush_pending(L))) return rc; }
  const int n = (int)std::min<int64_t>(n_steps, 1 << 20);
  int max_trials = 0, table_len = 0;
  if ((rc = steps_prepare(h, n, *total_number_iterations_io, reduction_exponent, growth_exponent, &max_trials, &table_len))) return rc;
  const size_t lds = sizeof(double) * (9 * (size_t)h->n + 4 * (size_t)h->m);
  {
    static size_t limit[64] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    size_t &cur = limit[h->device & 63];
    if (cur < lds) {
      HIP_TRY(hipFuncSetAttribute((const void *)small_lp_steps_kernel<SMALL_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(hipFuncSetAttribute((const void *)small_lp_steps_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      cur = lds;
    }
  }

  SmallLpArgs a{};
  a.n = (int)h->n; a.m = (int)h->m; a.num_eq = (int)h->num_eq;
  a.A = h->A.view(); a.T = h->At.view();
  a.x = h->x; a.y = h->y; a.aty = h->aty; a.sum_x = h->sum_x; a.sum_y = h->sum_y;
  a.c = h->c; a.lb = h->lb; a.ub = h->ub; a.b = h->b;
  a.primal_weight = primal_weight; a.step_size = *step_size_io;
  a.n_steps = n; a.max_trials = max_trials; a.table_len = table_len;
  a.pend = h->pend_x ? 1 : 0; a.pend_w = h->pend_w;
  a.wsum_x = h->sum_x_weights; a.wsum_y = h->sum_y_weights;
  a.pow_red = h->steps_pow_dev; a.pow_growth = h->steps_pow_dev + table_len;
  a.res_host = h->steps_res;
  a.seq = ++h->steps_seq;
  h->pend_x = h->pend_y = false;             // the launch applies it
  const auto c1 = std::chrono::steady_clock::now();
  static const int few_env = dev_env("PDHG_SMALL_FEW_ROWS") ? atoi(dev_env("PDHG_SMALL_FEW_ROWS")) : SMALL_FEW_ROWS;   // dev knob
  if (std::max(h->n, h->m) <= few_env) hipLaunchKernelGGL(small_lp_steps_kernel<256>, dim3(1), dim3(256), lds, h->stream, a);
  else hipLaunchKernelGGL(small_lp_steps_kernel<SMALL_TPB>, dim3(1), dim3(SMALL_TPB), lds, h->stream, a);
  HIP_TRY(hipGetLastError());
  const auto c2 = std::chrono::steady_clock::now();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  double r[13], r14 = 0.0;
  if ((rc = steps_wait(h, a.seq, r, &r14))) return rc;
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  const int64_t steps = (int64_t)r[1], trials = (int64_t)r[2];
  h->small_lp_launches += 1; h->n_graph_trials += trials;
  h->sum_x_count += steps; h->sum_y_count += steps;
  h->sum_x_weights = r[6]; h->sum_y_weights = r[7];
  h->state_version += 1;
  *step_size_io = r[0];
  *total_number_iterations_io += trials;
  *cumulative_kkt_passes_io += (double)trials;
  *steps_done = steps;
  *unfinished_entry = r14;
  if (r[8] != 0.0) { *numerical_error_out = 1; *steps_done = steps + 1; }
  return 0;
}
