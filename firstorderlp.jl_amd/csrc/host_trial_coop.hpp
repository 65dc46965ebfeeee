// host_trial_coop.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// the trial step / whole batches of take_steps as ONE persistent kernel (trial_kernel.hpp): grid, census, launch, fallback (host side).

// ---- the trial step as ONE persistent kernel (trial_kernel.hpp) -----------------------------

int coop_prepare(pdhg_handle *h, int cap_limit = 0, bool several_items = false);
bool graph_eligible(pdhg_handle *h);

bool coop_eligible(pdhg_handle *h) {
  if (h->coop_mode < 0) {
    const char *ev = getenv("PDHG_COOP");
    bool on = !h->grp && !h->A.tiled && !h->At.tiled && h->A.slabs.empty() && h->At.slabs.empty() &&
              h->A.segs.empty() && h->At.segs.empty() && h->n > 0 && h->m > 0;
    if (h->has_q) on = on && !h->Q.tiled && !h->Qt.tiled && h->Q.slabs.empty() && h->Qt.slabs.empty();
    if (ev) on = on && ev[0] != '0';
    const char *gv = getenv("PDHG_GRAPH");             // PDHG_GRAPH=0: separate launches, no one-launch path of either kind
    if (gv) on = on && gv[0] != '0';
    h->coop_mode = on ? 1 : 0;
    if (on && coop_prepare(h) != 0) h->coop_mode = 0;  // too many items for one co-resident grid, or no census: graph / plain path
  }
  return h->coop_mode == 1 && !h->profile;
}

// grid of the persistent launch + the census of workgroups per XCD (once per handle)
// cap_limit: at most this many workgroups (several shards share a device); several_items: accept more items than
// workgroups (the phases then walk several row blocks per workgroup)
int coop_prepare(pdhg_handle *h, int cap_limit, bool several_items) {
  if (h->gsync) return 0;
  HIP_TRY(hipSetDevice(h->device));
  int rc = ensure_result_word(h);
  if (rc) return rc;
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trial_kernel<false>, TPB, 0));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, h->device));
  int cap = std::max(8, per_cu * prop.multiProcessorCount / 8 * 8);
  if (const char *ev = dev_env("PDHG_COOP_WGS")) cap = std::max(8, std::min(cap, atoi(ev) / 8 * 8));
  if (cap_limit > 0) cap = std::max(8, std::min(cap, cap_limit / 8 * 8));
  // test knob: pretend the device holds this many workgroups (more than it does: the barriers cannot complete)
  const char *pretend = dev_env("PDHG_COOP_TEST_PRETEND_WGS");
  if (pretend) cap = std::max(8, atoi(pretend) / 8 * 8);
  // one item per workgroup and phase where the device can hold that many: row blocks from the front, long-row chunks from the end
  int items = std::max(h->A.grid + h->A.nchunks, h->At.grid + h->At.nchunks);
  if (h->has_q) items = std::max(items, std::max(h->Q.grid + h->Q.nchunks, h->A.grid + h->A.nchunks + h->Qt.grid + h->Qt.nchunks));
  h->coop_grid = pretend ? cap : std::min(cap, std::max(8, (items + 7) / 8 * 8));
  // More items than co-resident workgroups: the persistent kernel would walk several row blocks per workgroup at
  // 5 workgroups per CU, where the separate stream kernels keep 8 per CU in flight -- measured slower (PageRank-1M,
  // 4 552 items on 1 280 workgroups: 4 380 it/s against 4 620 as a graph of slab passes).  Leave those to the graph.
  if (items > cap && !several_items && !dev_env("PDHG_COOP_FORCE")) {
    h->coop_mode = 0;
    return 1;       // not an error: the caller falls through to the graph / plain path
  }
  HIP_TRY(hipMalloc((void **)&h->gsync, sizeof(GridSync)));
  HIP_TRY(hipMemsetAsync(h->gsync, 0, sizeof(GridSync), h->stream));
  if (getenv("PDHG_COOP_TRACE")) {
    HIP_TRY(hipMalloc((void **)&h->coop_trace, sizeof(unsigned long long) * 8 * (size_t)h->coop_grid));
    HIP_TRY(hipMemsetAsync(h->coop_trace, 0, sizeof(unsigned long long) * 8 * (size_t)h->coop_grid, h->stream));
  }
  hipLaunchKernelGGL(xcd_register_kernel, dim3(h->coop_grid), dim3(TPB), 0, h->stream, h->gsync);
  HIP_TRY(hipGetLastError());
  GridSync host;
  HIP_TRY(hipMemcpyAsync(&host, h->gsync, sizeof(GridSync), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned long long total = 0;
  h->coop_nxcd = 0;
  for (int x = 0; x < 8; ++x) { total += host.xcd_count[x][0]; h->coop_nxcd += host.xcd_count[x][0] > 0; h->coop_xcd_cnt[x] = (unsigned)host.xcd_count[x][0]; }
  if (total != (unsigned long long)h->coop_grid || h->coop_nxcd == 0) {
    h->coop_mode = 0;
    return fail(996, "one-launch trial: workgroup census does not add up");
  }
  if (getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] one-launch trial: %d workgroups (%d per CU possible) on %u XCDs, %d + %d / %d + %d row blocks + long chunks\n",
            h->coop_grid, per_cu, h->coop_nxcd, h->A.grid, h->A.nchunks, h->At.grid, h->At.nchunks);
  return 0;
}

TrialProduct trial_product(pdhg_handle *h, CsrDev &D, const double *xin, const EpiArgs &e) {
  TrialProduct P{};
  P.M = D.view();
  P.blks = D.blks; P.nblk = D.nblk; P.per_xcd = D.per_xcd; P.grid = D.grid; P.remap = h->remap ? 1 : 0;
  P.nchunks = D.nchunks; P.nlong = D.nlong; P.long_grid = D.long_grid;
  P.chunk_row = D.chunk_row; P.chunk_off = D.chunk_off; P.chunk_lidx = D.chunk_lidx; P.chunk_partial = D.chunk_partial;
  P.long_ticket = D.long_ticket;
  P.long_row = D.long_row; P.long_chunk_ptr = D.long_chunk_ptr;
  P.xin = xin; P.e = e;
  P.uses = D.coop_uses;
  D.coop_uses += 1;
  return P;
}

// returns 1 when the handle turned out not to suit the one-launch kernel (nothing was launched)
// Two persistent launches that are both only PARTLY resident would wait for each other's workgroups for ever
// (until the spin limit): one such kernel at a time per device, from launch until its results are back.
std::mutex &coop_device_mutex(int device) {
  static std::mutex mu[64];
  return mu[device & 63];
}

int coop_trial(pdhg_handle *h, double step_size, double primal_weight, double theta, bool xbar_only, double out[5]) {
  int rc = coop_prepare(h);
  if (rc) return rc;
  std::lock_guard<std::mutex> one_at_a_time(coop_device_mutex(h->device));
  const auto c0 = std::chrono::steady_clock::now();
  TrialKernelArgs a{};
  a.n = (int)h->n; a.xbar_only = xbar_only ? 1 : 0;
  a.x = h->x; a.c = h->c; a.aty = h->aty; a.lb = h->lb; a.ub = h->ub;
  a.tau = step_size / primal_weight; a.theta = theta;
  a.x_next = h->x_next; a.xbar = h->xbar;
  a.avg_w = h->pend_w; a.sum_x = (h->pend_x && !xbar_only) ? h->sum_x : nullptr;
  EpiArgs de{};
  de.y = h->y; de.b = h->b; de.y_next = h->y_next; de.sigma = primal_weight * step_size; de.num_eq = (int)h->num_eq;
  de.partials = h->pA; de.stride = h->A.slots(); de.lo_offset = h->A.slots();
  if (h->pend_y) { de.sum_y = h->sum_y; de.avg_w = h->pend_w; }
  a.A = trial_product(h, h->A, h->xbar, de);
  EpiArgs te{};
  te.x = h->x; te.x_next = h->x_next; te.aty = h->aty; te.aty_next = h->aty_next;
  te.partials = h->pAt; te.stride = h->pAt_stride; te.lo_offset = 3 * h->pAt_stride;
  a.T = trial_product(h, h->At, h->y_next, te);
  a.sp.ptr[0] = h->pAt;                       a.sp.count[0] = h->At.slots();
  a.sp.ptr[1] = h->pAt + h->pAt_stride;       a.sp.count[1] = h->At.slots();
  a.sp.ptr[2] = h->pA;                        a.sp.count[2] = h->A.slots();
  a.sp.ptr[3] = h->pAt + 2 * h->pAt_stride;   a.sp.count[3] = h->At.slots();
  a.sp.ptr[4] = h->pQ;                        a.sp.count[4] = 0;
  for (int q : {0, 1, 3}) a.sp.ptr_lo[q] = a.sp.ptr[q] + 3 * h->pAt_stride;
  a.sp.ptr_lo[2] = h->pA + h->A.slots();
  a.sp.ptr_lo[4] = h->pQ + h->ew_grid_n;
  a.sp.out = nullptr;
  a.has_q = h->has_q ? 1 : 0;
  a.epoch = h->coop_epoch;
  h->coop_epoch += 2;
  if (h->has_q) {
    a.q_blocks = h->ew_grid_n;
    a.sp.count[4] = h->ew_grid_n;
    a.qx = h->qx; a.dx = h->tmp_n; a.qtdx = h->tmp_n2; a.pq = h->pQ;
    EpiArgs qe{};
    qe.out = h->tmp_n2;
    a.Qtdx = trial_product(h, h->Qt, h->tmp_n, qe);
    if (!xbar_only) {
      qe.out = h->qx;
      a.Qx = trial_product(h, h->Q, h->x, qe);
      h->coop_epoch += 1;
    }
  }
  a.seq_dev = h->seq_dev; a.res_host = h->res_host; a.sync = h->gsync;
  h->seq_expected += 1;
  a.launch = h->coop_launches; a.seq = h->seq_expected; a.nxcd = h->coop_nxcd; a.relaxed = h->relaxed ? 1 : 0;
  a.trace = h->coop_trace;
  for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->coop_xcd_cnt[x];
  // test knob: raise the barriers' error word in front of launch number k, as a time-out in it would (the launch
  // then runs without synchronisation and reports the error; the recovery below is what is being tested)
  static const long break_at = dev_env("PDHG_COOP_TEST_BREAK_AT") ? atol(dev_env("PDHG_COOP_TEST_BREAK_AT")) : -1;
  if (break_at >= 0 && (long)h->coop_launches == break_at) {
    static const unsigned long long nine = 9ull;
    HIP_TRY(hipMemcpyAsync(&h->gsync->error[0], &nine, sizeof nine, hipMemcpyHostToDevice, h->stream));
  }
  h->coop_launches += 1;
  const auto c1 = std::chrono::steady_clock::now();
  static const bool coh_single = dev_env("PDHG_COOP_COH") && dev_env("PDHG_COOP_COH")[0] == '1';   // dev: L1-bypassing loads in the single-trial kernel too
  if (coh_single) hipLaunchKernelGGL(trial_kernel<true>, dim3(h->coop_grid), dim3(TPB), 0, h->stream, a);
  else hipLaunchKernelGGL(trial_kernel<false>, dim3(h->coop_grid), dim3(TPB), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  const auto c2 = std::chrono::steady_clock::now();
  h->t_set += std::chrono::duration<double>(c1 - c0).count();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  h->n_graph_trials += 1;
  if (!xbar_only) h->pend_x = false;
  h->pend_y = false;                  // the launch carries the deferred average update
  rc = wait_result_word(h, out, true);
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  if (rc) return rc;
  if (h->res_error != 0.0) {
    // A grid barrier ran into its spin limit: the workgroups were not all co-resident (another process runs a
    // persistent kernel on this device, or a debugger / profiler serialises dispatch).  Once the error word is up no
    // workgroup waits any more, every workgroup still runs every phase, so the launch has ended and the elementwise
    // work that does not depend on the barriers -- the deferred average update it carried -- is applied exactly once.
    // x', y', A'y' and the sums are not trustworthy: the caller repeats the trial on the graph / plain path (its
    // inputs x, y, A'y are untouched), and this handle stays there (the barrier counters are out of step now).
    h->coop_mode = 0;
    h->coop_fallbacks += 1;
    fprintf(stderr, "[pdhg_hip] one-launch trial: a grid barrier timed out (code %g; is the device shared with another "
                    "persistent kernel?) -- this handle uses the %s path from here on\n", h->res_error,
            graph_eligible(h) ? "graph" : "separate-launch");
    return 1;
  }
  return 0;
}

// ---- what the two multi-step launchers (coop_steps, small_lp_steps) share ---------------------------------------
// Trial budget of a launch of n steps, the tables of (total_number_iterations + 1)^-exponent for its trials (host pow,
// uploaded), the pinned result words.  Budget: the steps asked for plus room for rejections (a launch that runs out
// returns at a take_step boundary and the caller launches again); 64 more table entries for finishing the take_step
// the budget ends in.  The tables cost two pow() per entry on the host: sized to the batch, not to the worst case.
static int steps_prepare(pdhg_handle *h, int n, int64_t total_number_iterations, double reduction_exponent,
                         double growth_exponent, int *max_trials_out, int *table_len_out) {
  int max_trials = n + n / 8 + 16, table_len = max_trials + 64;
  if (const char *tv = dev_env("PDHG_STEPS_TEST_TABLE")) max_trials = table_len = std::max(1, atoi(tv));   // test knob: launches end inside take_steps
  if (!h->steps_res) {
    HIP_TRY(hipHostMalloc((void **)&h->steps_res, STEPS_RES_WORDS * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
    memset(h->steps_res, 0, STEPS_RES_WORDS * sizeof(double));
  }
  if (h->steps_pow_cap < table_len) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->steps_pow_dev) (void)hipFree(h->steps_pow_dev);
    if (h->steps_pow_host) (void)hipHostFree(h->steps_pow_host);
    h->steps_pow_dev = h->steps_pow_host = nullptr;
    h->steps_pow_cap = std::max(table_len, 512);
    HIP_TRY(hipMalloc((void **)&h->steps_pow_dev, sizeof(double) * 2 * (size_t)h->steps_pow_cap));
    // (room behind the tables: coop_steps stages its FinalSpec there)
    HIP_TRY(hipHostMalloc((void **)&h->steps_pow_host, sizeof(double) * 2 * (size_t)h->steps_pow_cap + sizeof(FinalSpec) + 64, hipHostMallocDefault));
  }
  // the t-th trial of the launch runs with total_number_iterations = total + t + 1 and uses k1 = that + 1 (pdhg.jl:713-714)
  for (int t = 0; t < table_len; ++t) {
    const double k1 = (double)(total_number_iterations + t + 2);
    h->steps_pow_host[t] = pow(k1, -reduction_exponent);
    h->steps_pow_host[table_len + t] = pow(k1, -growth_exponent);
  }
  HIP_TRY(hipMemcpyAsync(h->steps_pow_dev, h->steps_pow_host, sizeof(double) * 2 * (size_t)table_len, hipMemcpyHostToDevice, h->stream));
  *max_trials_out = max_trials;
  *table_len_out = table_len;
  return 0;
}
// wait for a multi-step launch's result words: r[0..12] once sequence number and checksum match (bounded spin, then the stream)
// Every word is read through the volatile pointer (a plain read in the spin loop may be hoisted).  r14: the step size
// on entry of a take_step the launch ended inside (0: none); it is under the checksum like the other words.
static int steps_wait(pdhg_handle *h, unsigned long long seq, double r[13], double *r14) {
  const volatile unsigned long long *bits = reinterpret_cast<const volatile unsigned long long *>(h->steps_res);
  const double seq_d = (double)seq;
  unsigned long long seq_bits;
  memcpy(&seq_bits, &seq_d, 8);
  auto ready = [&]() -> bool {
    if (bits[15] != seq_bits) return false;
    unsigned long long w[13], ck = RESULT_CHECK_SALT;
    for (int k = 0; k < 13; ++k) { w[k] = bits[k]; ck ^= w[k] * (2ull * (unsigned long long)k + 1ull); }
    const unsigned long long w14 = bits[14];
    ck ^= w14 * 29ull;
    if (ck != bits[13]) return false;
    for (int k = 0; k < 13; ++k) memcpy(&r[k], &w[k], 8);
    memcpy(r14, &w14, 8);
    return r[12] == seq_d;
  };
  for (long spin = 0; spin < 400000000L; ++spin) {
    if (ready()) return 0;
    if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(h->stream) != hipErrorNotReady) break;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (ready()) return 0;
  return fail(998, "multi-step kernel finished without publishing its results");
}

// Up to n_steps adaptive take_steps in ONE launch (steps_kernel, trial_kernel.hpp).  On return *steps_done take_steps
// have been taken (fewer when the launch ran out of its trial budget, met numerical_error, or a barrier timed out: the
// caller goes on from the state left).  Returns 1 when nothing could be launched (not eligible).
// The multi-step kernel's XCD-local mode: LPs whose products are at most PDHG_COOP_LOCAL_MAX (default 32: one per compute
// unit of an XCD) items.  8 x coop_grid workgroups are launched, the dispatcher deals them round the XCDs, those on XCD 0
// work -- the census must find exactly coop_grid of them there.  Own barrier words and epoch (the single-trial kernel
// keeps the handle's all-XCD census).  PDHG_COOP_LOCAL=0 turns it off.  Returns 0 when the mode is on.
static int steps_local_prepare(pdhg_handle *h) {
  if (h->local_mode >= 0) return h->local_mode ? 0 : 1;
  h->local_mode = 0;
  const char *ev = dev_env("PDHG_COOP_LOCAL");
  if (ev && ev[0] == '0') return 1;
  const int cap = dev_env("PDHG_COOP_LOCAL_MAX") ? atoi(dev_env("PDHG_COOP_LOCAL_MAX")) : 32;
  if (h->coop_grid <= 0 || h->coop_grid > cap || dev_env("PDHG_COOP_TEST_PRETEND_WGS")) return 1;
  HIP_TRY(hipMalloc((void **)&h->lsync, sizeof(GridSync)));
  HIP_TRY(hipMemsetAsync(h->lsync, 0, sizeof(GridSync), h->stream));
  hipLaunchKernelGGL(xcd_register_kernel, dim3(8 * h->coop_grid), dim3(TPB), 0, h->stream, h->lsync);
  HIP_TRY(hipGetLastError());
  GridSync host;
  HIP_TRY(hipMemcpyAsync(&host, h->lsync, sizeof(GridSync), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (host.xcd_count[0][0] != (unsigned long long)h->coop_grid) return 1;      // the dispatcher dealt them otherwise: all-XCD mode
  h->local_epoch = 0;
  h->local_mode = 1;
  if (getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] multi-step kernel: XCD-local mode, %d workgroups on XCD 0 (of %d launched)\n", h->coop_grid, 8 * h->coop_grid);
  return 0;
}

int coop_steps(pdhg_handle *h, int64_t n_steps, double reduction_exponent, double growth_exponent, double *step_size_io,
               double primal_weight, int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
               int *numerical_error_out, int64_t *steps_done, double *unfinished_entry) {
  *steps_done = 0;
  *unfinished_entry = 0.0;
  if (!coop_eligible(h) || h->has_q || !h->lazy_accept || h->pend_x != h->pend_y) return 1;
  int rc = coop_prepare(h);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  const int n = (int)std::min<int64_t>(n_steps, 1 << 20);
  int max_trials = 0, table_len = 0;
  if (!h->steps_ctl) {
    HIP_TRY(hipMalloc((void **)&h->steps_ctl, sizeof(StepsCtl)));
    HIP_TRY(hipMemsetAsync(h->steps_ctl, 0, sizeof(StepsCtl), h->stream));
  }
  std::lock_guard<std::mutex> one_at_a_time(coop_device_mutex(h->device));
  if ((rc = steps_prepare(h, n, *total_number_iterations_io, reduction_exponent, growth_exponent, &max_trials, &table_len))) return rc;
  StepsKernelArgs a{};
  a.n = (int)h->n; a.num_eq = (int)h->num_eq;
  a.xa = h->x; a.xb = h->x_next; a.ya = h->y; a.yb = h->y_next; a.atya = h->aty; a.atyb = h->aty_next;
  a.c = h->c; a.lb = h->lb; a.ub = h->ub; a.b = h->b;
  a.xbar = h->xbar; a.sum_x = h->sum_x; a.sum_y = h->sum_y;
  EpiArgs none{};
  a.A = trial_product(h, h->A, nullptr, none);
  a.T = trial_product(h, h->At, nullptr, none);
  h->A.coop_uses -= 1; h->At.coop_uses -= 1;       // (trial_product counted one use: the launch's own count comes back with the results)
  a.uses_a = a.A.uses; a.uses_t = a.T.uses;
  a.pA = h->pA; a.pAt = h->pAt; a.pA_slots = h->A.slots(); a.pAt_stride = h->pAt_stride;
  FinalSpec sp{};
  sp.ptr[0] = h->pAt;                       sp.count[0] = h->At.slots();
  sp.ptr[1] = h->pAt + h->pAt_stride;       sp.count[1] = h->At.slots();
  sp.ptr[2] = h->pA;                        sp.count[2] = h->A.slots();
  sp.ptr[3] = h->pAt + 2 * h->pAt_stride;   sp.count[3] = h->At.slots();
  sp.ptr[4] = h->pQ;                        sp.count[4] = 0;
  for (int q : {0, 1, 3}) sp.ptr_lo[q] = sp.ptr[q] + 3 * h->pAt_stride;
  sp.ptr_lo[2] = h->pA + h->A.slots();
  sp.ptr_lo[4] = h->pQ + h->ew_grid_n;
  sp.out = nullptr;
  memcpy(h->steps_pow_host + 2 * (size_t)table_len, &sp, sizeof sp);        // staged behind the pow tables (pinned)
  HIP_TRY(hipMemcpyAsync(&h->steps_ctl->sp, h->steps_pow_host + 2 * (size_t)table_len, sizeof sp, hipMemcpyHostToDevice, h->stream));
  a.primal_weight = primal_weight; a.step_size = *step_size_io;
  a.n_steps = n; a.max_trials = max_trials; a.table_len = table_len;
  a.pend = h->pend_x ? 1 : 0; a.pend_w = h->pend_w;
  a.wsum_x = h->sum_x_weights; a.wsum_y = h->sum_y_weights;
  a.pow_red = h->steps_pow_dev; a.pow_growth = h->steps_pow_dev + table_len;
  const bool local = steps_local_prepare(h) == 0;
  a.epoch = local ? h->local_epoch : h->coop_epoch;
  a.sync = local ? h->lsync : h->gsync; a.ctl = h->steps_ctl; a.res_host = h->steps_res;
  a.local_g = local ? h->coop_grid : 0; a.local_home = 0;
  // test knob: the kernel expects eight workgroups more than are launched on the home XCD -- its first barrier times out
  if (local && dev_env("PDHG_COOP_LOCAL_TEST_BAD")) a.local_g += 8;
  // The stayers number themselves with a ticket; the word is zeroed before every launch (one 8-byte fill in stream order,
  // once per batch of steps) instead of the host adding coop_grid per launch to a running base: a launch that placed
  // more than the census' workgroups on the home XCD drew more tickets than the host assumed, and every later launch
  // numbered its workers wrongly until the barriers' time-out dropped the mode.
  if (local) {
    HIP_TRY(hipMemsetAsync(&h->lsync->ticket[0][0], 0, sizeof(unsigned long long), h->stream));
    a.local_ticket_base = 0;
  }
  a.seq = ++h->steps_seq;
  a.nxcd = h->coop_nxcd; a.relaxed = h->relaxed ? 1 : 0;
  a.trace = h->coop_trace;
  for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->coop_xcd_cnt[x];
  const auto c1 = std::chrono::steady_clock::now();
  if (local) hipLaunchKernelGGL(steps_kernel<true>, dim3(8 * h->coop_grid), dim3(TPB), 0, h->stream, a);
  else hipLaunchKernelGGL(steps_kernel<false>, dim3(h->coop_grid), dim3(TPB), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  const auto c2 = std::chrono::steady_clock::now();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  double r[13], r14 = 0.0;
  if ((rc = steps_wait(h, a.seq, r, &r14))) return rc;
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  const int64_t steps = (int64_t)r[1], trials = (int64_t)r[2];
  const bool flip = r[3] != 0.0, aborted = r[9] != 0.0 || r[11] != 0.0;
  h->steps_launches += 1; h->steps_trials += trials; h->n_graph_trials += trials;
  if (local) { h->local_epoch = (unsigned long long)r[10]; h->local_launches += 1; }
  else h->coop_epoch = (unsigned long long)r[10];
  h->A.coop_uses += (unsigned long long)trials + (aborted ? 1ull : 0ull);
  h->At.coop_uses += (unsigned long long)trials + (aborted ? 1ull : 0ull);
  if (flip) { std::swap(h->x, h->x_next); std::swap(h->y, h->y_next); std::swap(h->aty, h->aty_next); }
  h->pend_x = h->pend_y = r[4] != 0.0;
  h->pend_w = r[5];
  h->sum_x_count += steps; h->sum_y_count += steps;
  h->sum_x_weights = r[6]; h->sum_y_weights = r[7];
  if (trials > 0 || aborted) h->state_version += 1;     // (bump_version of a single handle)
  *step_size_io = r[0];
  *total_number_iterations_io += trials;
  *cumulative_kkt_passes_io += (double)trials;
  *steps_done = steps;
  // (also after a barrier time-out: the launch may have aborted inside a take_step whose earlier trials were rejected,
  //  and the word is written by the same thread as the other result words)
  *unfinished_entry = r14;
  if (r[8] != 0.0) { *numerical_error_out = 1; *steps_done = steps + 1; }   // the failing take_step counts as taken (it is not repeated)
  if (aborted && local) {
    // the XCD-local form failed (a workgroup of the launch was not where the census saw it): the all-XCD form from here on
    h->local_mode = 0;
    fprintf(stderr, "[pdhg_hip] multi-step trial kernel, XCD-local mode: a barrier timed out (code %g) -- all-XCD mode from here on\n", r[11]);
  } else if (aborted) {
    h->coop_mode = 0;
    h->coop_fallbacks += 1;
    fprintf(stderr, "[pdhg_hip] multi-step trial kernel: a grid barrier timed out (code %g; is the device shared with another "
                    "persistent kernel?) -- this handle uses the %s path from here on\n", r[11],
            graph_eligible(h) ? "graph" : "separate-launch");
  }
  return 0;
}

