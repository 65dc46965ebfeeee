// abi_eval.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// C ABI: the evaluation branch on the device (N1): point products, eval_point, distances, the trust-region searches.

// ---- evaluation branch on the device (N1) -----------------------------------

static int ev_alloc(pdhg_handle *h) {
  if (h->ev_partials) return 0;
  int rc;
  {
    // the evaluation kernels reduce up to 30 quantities per workgroup and a second stage reads every workgroup's
    // partials: two elements per thread and at most 1024 workgroups measured best (L1-SVM 229K elements: 448
    // workgroups 67 us per trust-region call against 82 with 895; 2M elements: 977 workgroups 123 us against 147 with 2048)
    static const int per_thread = dev_env("PDHG_EV_ELEMS") ? std::max(1, atoi(dev_env("PDHG_EV_ELEMS"))) : 2;
    h->ev_grid = std::min(1024, ew_grid((h->n + h->m + per_thread) / per_thread));
  }
  if ((rc = alloc_zero(&h->ev_partials, (int64_t)EV_MAXQ * h->ev_grid))) return rc;
  if ((rc = alloc_zero(&h->ev_ax, h->m))) return rc;
  if ((rc = alloc_zero(&h->ev_aty, h->n_alloc))) return rc;
  for (int k = 0; k < 3; ++k) {
    if ((rc = alloc_zero(&h->ev_cax[k], h->m))) return rc;
    if ((rc = alloc_zero(&h->ev_caty[k], h->n_alloc))) return rc;
  }
  if (h->grp && (rc = alloc_zero(&h->ev_xg, h->n_alloc))) return rc;
  if ((rc = alloc_zero(&h->px_avg, h->n))) return rc;
  if ((rc = alloc_zero(&h->py_avg, h->m))) return rc;
  if ((rc = alloc_zero(&h->x_r, h->n))) return rc;   // zeros == the initial restart point (pdhg.jl:869)
  if ((rc = alloc_zero(&h->y_r, h->m))) return rc;
  return 0;
}

// second stage of every shard's block partials (ns sums then nm maxes), then the
// combination over ranks in rank order
// the pinned result words of the evaluation reductions (multi_final_kernel, tr_small_kernel): k values, checksum, sequence number
static int ev_ensure_host(pdhg_handle *h) {
  if (!h->ev_host) {
    HIP_TRY(hipHostMalloc((void **)&h->ev_host, (EV_HOST_SLOTS + 2) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
    memset(h->ev_host, 0, (EV_HOST_SLOTS + 2) * sizeof(double));
  }
  return 0;
}
static int ev_wait_host(pdhg_handle *h, int k, unsigned long long seq, double *out) {
  const volatile unsigned long long *bits = reinterpret_cast<const volatile unsigned long long *>(h->ev_host);
  auto ready = [&]() -> bool {
    if (bits[EV_HOST_SEQ] != seq) return false;
    unsigned long long w[EV_HOST_SLOTS];
    unsigned long long ck = EV_CHECK_SALT ^ seq ^ ((unsigned long long)k << 56);
    for (int q = 0; q < k; ++q) { w[q] = bits[q]; ck ^= w[q] * (2ull * (unsigned long long)q + 1ull); }
    if (ck != bits[EV_HOST_CK]) return false;
    for (int q = 0; q < k; ++q) memcpy(&out[q], &w[q], 8);
    return true;
  };
  for (long spin = 0; spin < 40000000L; ++spin) {
    if (ready()) return 0;
    if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(h->stream) != hipErrorNotReady) break;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (ready()) return 0;
  return fail(998, "evaluation reduction finished without publishing its results");
}

// The evaluation reductions publish into pinned host memory (one handle) unless PDHG_EVAL_HOST_WORD=0 (dev): decided in
// ONE place and per call -- pdhg_eval_point's combined 22-quantity reduction (which needs the max mask) and ev_finish
// must never disagree (a cached copy here once could: sums where maxima belong).
static bool eval_host_word() {
  const char *hw = dev_env("PDHG_EVAL_HOST_WORD");
  return !(hw && hw[0] == '0');
}

static int ev_finish(const Shards &L, int ns, int nm, double *out, unsigned max_mask = 0) {
  if (ns + nm > EV_MAXQ) return fail(-1, "too many scalars in one reduction");
  const bool host_word = eval_host_word();
  if (max_mask != 0 && (L.g || !host_word)) return fail(-1, "a mixed sum / max reduction needs the host-word form");
  if (!L.g && host_word) {
    // one handle: the second stage publishes into pinned memory and the host polls (see multi_final_kernel)
    pdhg_handle *h = L.p[0];
    const int k = ns + nm;
    if (k > EV_HOST_SLOTS) return fail(-1, "too many scalars in one reduction");
    HIP_TRY(hipSetDevice(h->device));
    int rc0 = ev_ensure_host(h);
    if (rc0) return rc0;
    const unsigned long long seq = ++h->ev_seq;
    hipLaunchKernelGGL(multi_final_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->ev_partials, h->ev_grid,
                       h->ev_grid, ns, nm, h->scal_dev, h->ev_host, seq, max_mask);
    HIP_TRY(hipGetLastError());
    return ev_wait_host(h, k, seq, out);
  }
  FOR_SHARDS(L, h) {
    hipLaunchKernelGGL(multi_final_kernel, dim3(1), dim3(FINAL_TPB), 0, h->stream, h->ev_partials, h->ev_grid,
                       h->ev_grid, ns, nm, h->scal_dev, (double *)nullptr, 0ull, 0u);
    HIP_TRY(hipGetLastError());
  }
  return combine_scalars(L, ns + nm, ns, out);
}

// px: column vector (valid on the owned slice), py: this shard's rows
static int select_point(pdhg_handle *h, int point, const double **px, const double **py) {
  int rc = ev_alloc(h);
  if (rc) return rc;
  if (point == PDHG_POINT_CURRENT) { *px = h->x; *py = h->y; return 0; }
  if (point == PDHG_POINT_RESTART) { *px = h->x_r; *py = h->y_r; return 0; }
  if (point == PDHG_POINT_AVERAGE) {
    if (h->sum_x_count == 0 || h->sum_y_count == 0) return fail(-1, "average is empty");
    if (h->avg_version != h->state_version) {
      const int64_t o = h->clo;
      hipLaunchKernelGGL(div_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, h->sum_x + o, h->sum_x_weights, h->px_avg + o);
      hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->py_avg);
      HIP_TRY(hipGetLastError());
      h->avg_version = h->state_version;
    }
    *px = h->px_avg; *py = h->py_avg;
    return 0;
  }
  return fail(-1, "unknown point selector");
}

// A*x (this shard's rows), A'*y (owned slice) and, for a QP, Q*x (full) at `point`
// on every shard; cached for CURRENT / AVERAGE until the state changes.  Results in
// h->pt_ax / pt_aty / pt_qx together with the point itself in h->pt_x / pt_y.
static int point_products(const Shards &L, int point) {
  int rc;
  static const bool cache_off = dev_env("PDHG_NO_EVAL_CACHE") != nullptr;   // debugging aid
  const bool cached = !cache_off && (point == PDHG_POINT_CURRENT || point == PDHG_POINT_AVERAGE ||
                                     point == PDHG_POINT_RESTART);
  bool fresh = true;
  FOR_SHARDS(L, h) {
    if ((rc = select_point(h, point, &h->pt_x, &h->pt_y))) return rc;
    h->pt_ax = h->ev_ax; h->pt_aty = h->ev_aty;
    double **dqx = &h->ev_qx;
    bool f = true;
    if (cached) {
      const int k = point == PDHG_POINT_CURRENT ? 0 : (point == PDHG_POINT_AVERAGE ? 1 : 2);
      h->pt_ax = h->ev_cax[k]; h->pt_aty = h->ev_caty[k]; dqx = &h->ev_cqx[k];
      if (k < 2) {
        f = h->ev_cversion[k] != h->state_version;
        h->ev_cversion[k] = h->state_version;
      } else {
        const uint64_t key = (h->matrix_version << 32) + h->restart_version;
        f = h->ev_rkey != key;
        h->ev_rkey = key;
      }
    }
    if (h->has_q && !*dqx) {
      if ((rc = alloc_zero(dqx, h->n))) return rc;
      f = true;
    }
    h->pt_qx = h->has_q ? *dqx : nullptr;
    fresh = f;              // shards move in lock step: the same answer on all of them
  }
  if (!fresh) return 0;
  // full x at the point on every shard (a plain handle's vectors are full already)
  if (L.g) {
    if ((rc = gather_cols_device(L, [](pdhg_handle *s) { return s->pt_x; }, [](pdhg_handle *s) { return s->ev_xg; }))) return rc;
  }
  FOR_SHARDS(L, h) {
    const double *xfull = L.g ? h->ev_xg : h->pt_x;
    EpiArgs e{};
    e.out = h->pt_ax;
    if ((rc = launch_spmv<MODE_PLAIN, 0>(h, h->A, xfull, e))) return rc;
    if (h->has_q) {
      e.out = h->pt_qx;
      if ((rc = launch_spmv<MODE_PLAIN, 2>(h, h->Q, xfull, e))) return rc;
    }
  }
  return dual_product(L, [](pdhg_handle *s) { return s->pt_y; }, [](pdhg_handle *s) { return s->pt_aty; });
}

int pdhg_set_original_problem(pdhg_handle *h0, const double *constraint_rescaling,
                              const double *variable_rescaling, const double *c_o, const double *b_o,
                              const double *lb_o, const double *ub_o) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!constraint_rescaling || !variable_rescaling || !c_o || !lb_o || !ub_o || (h0->m_global > 0 && !b_o))
    return fail(-1, "null input array");
  const Shards L = shards_of(h0);
  FOR_SHARDS(L, h) {
    auto up = [&](double **dst, const double *src, int64_t len) -> int {
      if (!*dst) { int r2 = alloc_zero(dst, len); if (r2) return r2; }
      if (len > 0) { HIP_TRY(hipMemcpy(*dst, src, sizeof(double) * (size_t)len, hipMemcpyHostToDevice)); HIP_TRY(hipStreamSynchronize(nullptr)); }
      return 0;
    };
    // row vectors arrive with their GLOBAL length: a shard keeps its rows
    if ((rc = up(&h->E, constraint_rescaling + h->row_lo, h->m))) return rc;
    if ((rc = up(&h->b_o, b_o ? b_o + h->row_lo : nullptr, h->m))) return rc;
    if ((rc = up(&h->Dv, variable_rescaling, h->n))) return rc;
    if ((rc = up(&h->c_o, c_o, h->n))) return rc;
    if ((rc = up(&h->lb_o, lb_o, h->n))) return rc;
    if ((rc = up(&h->ub_o, ub_o, h->n))) return rc;
    h->has_original = true;
    if ((rc = ev_alloc(h))) return rc;
  }
  return 0;
}

int pdhg_eval_point(pdhg_handle *h0, int point, double out[24]) {
  RoctxRange roctx_range("pdhg_eval_point");
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!h0->has_original) return fail(-1, "pdhg_set_original_problem has not been called");
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  if ((rc = point_products(L, point))) return rc;
  if (!L.g && eval_host_word()) {                              // (read per call: tests compare the two forms in one process)
    // one handle: the row and the column kernels leave their block partials side by side (8 + 14 quantities), ONE second
    // stage reduces all 22 and the host makes one round trip instead of two.  Same partials, same order per quantity:
    // the same bits as the two-round form below.
    pdhg_handle *h = L.p[0];
    hipLaunchKernelGGL(eval_rows_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->m, (int)h->num_eq,
                       h->pt_ax, h->pt_y, h->E, h->b_o, h->ev_partials, h->ev_grid);
    hipLaunchKernelGGL(eval_cols_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, h->pt_aty,
                       h->pt_qx, h->pt_x, h->Dv, h->c_o, h->lb_o, h->ub_o, h->ev_partials + (size_t)8 * h->ev_grid, h->ev_grid);
    HIP_TRY(hipGetLastError());
    // What the rest of a termination / restart check asks for next (saddle_point.jl:432-477: the distances of the
    // average and the current iterate to the last restart point; :1015-1047: the sum of squares of the evaluated point)
    // rides in the same reduction: three more launches of dist2_kernel, partials behind the 22, ONE second stage, one
    // round trip.  pdhg_distance_to_restart / pdhg_point_sumsq answer from these until the state moves -- the same
    // kernel, grid and per-quantity second stage as their own launches: the same bits.  (three round trips of a check's six)
    const char *pf = dev_env("PDHG_EVAL_PREFETCH");                   // dev knob, read per call: the test compares both forms in one process
    const bool pre = !(pf && pf[0] == '0');
    bool have_avg = false;
    if (pre) {
      have_avg = h->sum_x_count != 0 && h->sum_y_count != 0;
      const double *ax = nullptr, *ay = nullptr;
      if (have_avg && (rc = select_point(h, PDHG_POINT_AVERAGE, &ax, &ay))) return rc;
      double *extra = h->ev_partials + (size_t)22 * h->ev_grid;
      if (have_avg)
        hipLaunchKernelGGL(dist2_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)h->m, ax, (const double *)h->x_r, ay,
                           (const double *)h->y_r, extra, h->ev_grid);
      else
        HIP_TRY(hipMemsetAsync(extra, 0, sizeof(double) * 2 * (size_t)h->ev_grid, h->stream));
      hipLaunchKernelGGL(dist2_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)h->m, (const double *)h->x,
                         (const double *)h->x_r, (const double *)h->y, (const double *)h->y_r, extra + (size_t)2 * h->ev_grid, h->ev_grid);
      hipLaunchKernelGGL(dist2_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)h->m, h->pt_x, (const double *)nullptr,
                         h->pt_y, (const double *)nullptr, extra + (size_t)4 * h->ev_grid, h->ev_grid);
      HIP_TRY(hipGetLastError());
    }
    double r[28];
    // quantities 0-3 sums, 4-7 maxes (rows); 8-14 sums, 15-21 maxes (columns); 22-27 sums (the prefetched scalars)
    if ((rc = ev_finish(L, pre ? 28 : 22, 0, r, 0xF0u | (0x7Fu << 15)))) return rc;
    if (pre) {
      for (int q = 0; q < 6; ++q) h->chk_vals[q] = r[22 + q];
      h->chk_state = h->state_version;
      h->chk_restart = h->restart_version;
      h->chk_point = point;
      h->chk_have_avg = have_avg;
    }
    for (int q = 0; q < 8; ++q) out[q] = r[q];
    for (int q = 0; q < 6; ++q) { out[8 + q] = r[8 + q]; out[14 + q] = r[8 + 7 + q]; }
    out[20] = r[8 + 6]; out[21] = r[8 + 13]; out[22] = out[23] = 0.0;
    return 0;
  }
  FOR_SHARDS(L, h) {
    hipLaunchKernelGGL(eval_rows_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->m, (int)h->num_eq,
                       h->pt_ax, h->pt_y, h->E, h->b_o, h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  if ((rc = ev_finish(L, 4, 4, out))) return rc;
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(eval_cols_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, h->pt_aty + o,
                       h->pt_qx ? h->pt_qx + o : nullptr, h->pt_x + o, h->Dv + o, h->c_o + o, h->lb_o + o,
                       h->ub_o + o, h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  double r[14];
  if ((rc = ev_finish(L, 7, 7, r))) return rc;
  for (int q = 0; q < 6; ++q) { out[8 + q] = r[q]; out[14 + q] = r[7 + q]; }
  out[20] = r[6]; out[21] = r[13]; out[22] = out[23] = 0.0;
  return 0;
}

int pdhg_save_restart_point(pdhg_handle *h0) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  FOR_SHARDS(L, h) {
    if ((rc = ev_alloc(h))) return rc;
    h->restart_version += 1;
    if (h->cn > 0)
      HIP_TRY(hipMemcpyAsync(h->x_r + h->clo, h->x + h->clo, sizeof(double) * (size_t)h->cn, hipMemcpyDeviceToDevice, h->stream));
    if (h->m > 0)
      HIP_TRY(hipMemcpyAsync(h->y_r, h->y, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToDevice, h->stream));
  }
  return 0;
}

static int dist2_common(pdhg_handle *h0, int point, bool to_restart, double out[2]) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  if (!L.g && h0->chk_state == h0->state_version) {          // reduced with the last pdhg_eval_point (see there)?
    const pdhg_handle *h = h0;
    if (to_restart && h->chk_restart == h->restart_version && (point == PDHG_POINT_CURRENT || (point == PDHG_POINT_AVERAGE && h->chk_have_avg))) {
      const int k = point == PDHG_POINT_AVERAGE ? 0 : 2;
      out[0] = h->chk_vals[k]; out[1] = h->chk_vals[k + 1];
      return 0;
    }
    if (!to_restart && point == h->chk_point && point != PDHG_POINT_RESTART) {
      out[0] = h->chk_vals[4]; out[1] = h->chk_vals[5];
      return 0;
    }
  }
  FOR_SHARDS(L, h) {
    const double *px, *py;
    if ((rc = select_point(h, point, &px, &py))) return rc;
    const int64_t o = h->clo;
    hipLaunchKernelGGL(dist2_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)h->m, px + o,
                       to_restart ? (const double *)(h->x_r + o) : (const double *)nullptr, py,
                       to_restart ? (const double *)h->y_r : (const double *)nullptr, h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  return ev_finish(L, 2, 0, out);
}

int pdhg_distance_to_restart(pdhg_handle *h, int point, double out[2]) { return dist2_common(h, point, true, out); }
int pdhg_point_sumsq(pdhg_handle *h, int point, double out[2]) { return dist2_common(h, point, false, out); }

int pdhg_get_point(pdhg_handle *h0, int point, double *x, double *y) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  FOR_SHARDS(L, h) { if ((rc = select_point(h, point, &h->pt_x, &h->pt_y))) return rc; }
  if (x && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->pt_x; }, x))) return rc;
  if (y && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->pt_y; }, y))) return rc;
  return sync_all(L);
}

static inline uint64_t d2bits(double v) { uint64_t b; memcpy(&b, &v, 8); return b; }
static inline double bits2d(uint64_t b) { double v; memcpy(&v, &b, 8); return v; }

// ---- the trust-region problem as ONE persistent launch (tr_coop_kernel.hpp) ----
// Decided once per handle: a single handle (no shard group) whose n + m elements fit PDHG_TR_COOP_MAX (default 1M;
// measured per call, 5 passes: n + m = 40K 94 -> 73 us, 229K (L1-SVM) 104 -> 80, 500K 138 -> 90, 1M 143 -> 110, 2M 167 -> 160:
// beyond that a pass is bandwidth, not latency, and the multi-launch kernels' 1 024 workgroups stream it as fast as 256 do).
// PDHG_TR_COOP=0 turns it off.  Returns 0 (prepared), 1 (does not apply) or an error code.
static int tr_coop_prepare(pdhg_handle *h) {
  if (h->tr_coop >= 0) return h->tr_coop ? 0 : 1;
  h->tr_coop = 0;
  const char *ev = getenv("PDHG_TR_COOP");
  if (ev && ev[0] == '0') return 1;
  const int64_t total = h->n + h->m;
  const int64_t cap = dev_env("PDHG_TR_COOP_MAX") ? atoll(dev_env("PDHG_TR_COOP_MAX")) : 1000000;
  if (total > cap || total < 1) return 1;
  HIP_TRY(hipSetDevice(h->device));
  int grid = (int)std::min<int64_t>(TRC_MAX_WGS, (total + TPB * 4 - 1) / (TPB * 4));
  grid = std::max(8, (grid + 7) / 8 * 8);
  if (const char *g = dev_env("PDHG_TR_COOP_WGS")) grid = std::max(8, std::min(TRC_MAX_WGS, atoi(g) / 8 * 8));
  HIP_TRY(hipMalloc((void **)&h->tr_sync, sizeof(GridSync)));
  HIP_TRY(hipMemsetAsync(h->tr_sync, 0, sizeof(GridSync), h->stream));
  HIP_TRY(hipMalloc((void **)&h->tr_partials, sizeof(double) * 2 * EV_MAXQ * (size_t)grid));
  HIP_TRY(hipMemsetAsync(h->tr_partials, 0, sizeof(double) * 2 * EV_MAXQ * (size_t)grid, h->stream));
  hipLaunchKernelGGL(xcd_register_kernel, dim3(grid), dim3(TPB), 0, h->stream, h->tr_sync);
  HIP_TRY(hipGetLastError());
  GridSync host;
  HIP_TRY(hipMemcpyAsync(&host, h->tr_sync, sizeof(GridSync), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned long long seen = 0;
  h->tr_nxcd = 0;
  for (int x = 0; x < 8; ++x) { seen += host.xcd_count[x][0]; h->tr_nxcd += host.xcd_count[x][0] > 0; h->tr_xcd_cnt[x] = (unsigned)host.xcd_count[x][0]; }
  if (seen != (unsigned long long)grid || h->tr_nxcd == 0) return 1;     // no census: the multi-launch form
  // test knob: a census that expects one workgroup too many -- the first barrier cannot complete (spin limit, error word)
  if (dev_env("PDHG_TR_COOP_TEST_BAD_CENSUS")) h->tr_xcd_cnt[0] += 1;
  h->tr_grid = grid;
  h->tr_epoch = 0;
  h->tr_coop = 1;
  if (getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] trust-region search: one persistent launch of %d workgroups on %u XCDs per call\n", grid, h->tr_nxcd);
  return 0;
}

// one call; returns 0 with out[] filled, 1 when a barrier could not complete (the caller repeats the call launch by
// launch, and this handle stays with that form), or an error code
static int tr_coop_call(pdhg_handle *h, double wp, double wd, double radius, int range, int approximate, double out[8]) {
  int rc = ev_ensure_host(h);
  if (rc) return rc;
  TrCoopArgs a{};
  a.n = (int)h->n; a.m = (int)h->m; a.ne = (int)h->num_eq; a.range = range; a.approximate = approximate ? 1 : 0;
  a.px = h->pt_x; a.py = h->pt_y; a.aty = h->pt_aty; a.qx = h->pt_qx; a.ax = h->pt_ax;
  a.c = h->c; a.b = h->b; a.lb = h->lb; a.ub = h->ub;
  a.wp = wp; a.wd = wd; a.radius = radius;
  a.gdv = h->tr_g; a.wd2v = h->tr_dir; a.thr = h->tr_thr;
  a.partials = h->tr_partials;
  a.sync = h->tr_sync;
  a.epoch = h->tr_epoch;
  a.nxcd = h->tr_nxcd;
  for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->tr_xcd_cnt[x];
  a.host_out = h->ev_host;
  a.seq = ++h->ev_seq;
  double r[10];
  {
    // one partly resident persistent kernel at a time per device (as the trial kernels): from launch to results
    std::lock_guard<std::mutex> lock(coop_device_mutex(h->device));
    hipLaunchKernelGGL(tr_coop_kernel, dim3(h->tr_grid), dim3(TPB), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    if ((rc = ev_wait_host(h, 10, a.seq, r))) return rc;
  }
  h->tr_epoch = (unsigned long long)r[9];
  if (r[8] != 0.0) {
    h->tr_coop = 0;
    if (getenv("PDHG_VERBOSE")) fprintf(stderr, "[pdhg_hip] trust-region search: a grid barrier timed out (code %g); back to one launch per pass\n", r[8]);
    return 1;
  }
  for (int q = 0; q < 8; ++q) out[q] = r[q];
  h->tr_coop_calls += 1;
  return 0;
}

int pdhg_trust_region_bound(pdhg_handle *h0, int point, double primal_weight_norm, double dual_weight_norm,
                            double radius, int range, int approximate, double out[8]) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (range < 0 || range > 2) return fail(-1, "range must be 0, 1 or 2");
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  const double wp = primal_weight_norm, wd = dual_weight_norm;
  if ((rc = point_products(L, point))) return rc;
  {
    // small problems on one handle: set-up, search and results in ONE workgroup and one launch (tr_small_kernel)
    const char *se = dev_env("PDHG_SMALL_EVAL");
    pdhg_handle *h = L.p[0];
    if (!L.g && h->n + h->m <= TRS_MAX && !(se && se[0] == '0') && !h->profile) {
      HIP_TRY(hipSetDevice(h->device));
      if ((rc = ev_ensure_host(h))) return rc;
      const size_t lds = sizeof(double) * 3 * (size_t)(h->n + h->m);
      {
        static size_t limit[64] = {};
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        size_t &cur = limit[h->device & 63];
        if (cur < lds) {
          HIP_TRY(hipFuncSetAttribute((const void *)tr_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          cur = lds;
        }
      }
      TrSmallArgs a{};
      a.n = (int)h->n; a.m = (int)h->m; a.ne = (int)h->num_eq; a.range = range; a.approximate = approximate ? 1 : 0;
      a.px = h->pt_x; a.py = h->pt_y; a.aty = h->pt_aty; a.qx = h->pt_qx; a.ax = h->pt_ax;
      a.c = h->c; a.b = h->b; a.lb = h->lb; a.ub = h->ub;
      a.wp = wp; a.wd = wd; a.radius = radius;
      a.host_out = h->ev_host;
      a.seq = ++h->ev_seq;
      hipLaunchKernelGGL(tr_small_kernel, dim3(1), dim3(TRS_TPB), lds, h->stream, a);
      HIP_TRY(hipGetLastError());
      return ev_wait_host(h, 8, a.seq, out);
    }
  }
  // every shard works on the concatenation [its column slice ; its rows]
  FOR_SHARDS(L, h) {
    if (!h->tr_g) {
      const int64_t total = h->n + h->m;
      if ((rc = alloc_zero(&h->tr_g, total))) return rc;
      if ((rc = alloc_zero(&h->tr_dir, total))) return rc;
      if ((rc = alloc_zero(&h->tr_thr, total))) return rc;
    }
  }
  if (!L.g && !L.p[0]->profile) {
    // medium problems on one handle: set-up, every probe pass and the results in ONE persistent launch (tr_coop_kernel.hpp)
    pdhg_handle *h = L.p[0];
    rc = tr_coop_prepare(h);
    if (rc > 1 || rc < 0) return rc;
    if (rc == 0) {
      rc = tr_coop_call(h, wp, wd, radius, range, approximate, out);
      if (rc != 1) return rc;
    }
  }
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(tr_setup_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)h->m,
                       (int)h->num_eq, h->pt_x + o, h->pt_y, h->pt_aty + o, h->pt_qx ? h->pt_qx + o : nullptr, h->pt_ax,
                       h->c + o, h->b, h->lb + o, h->ub + o, wp, wd, range,
                       h->tr_g, h->tr_dir, h->tr_thr, h->ev_partials, h->ev_grid);   // tr_g: g d, tr_dir: w d^2
    HIP_TRY(hipGetLastError());
  }
  double r[EV_MAXQ];
  if ((rc = ev_finish(L, TR_SETUP_NS, 1, r))) return rc;
  // compute_lagrangian_value (saddle_point.jl:1109-1120) without objective_constant
  out[0] = 0.5 * r[10] + r[0] - r[1] + r[2];
  out[1] = out[2] = 0.0;
  out[3] = r[8]; out[4] = r[9];
  out[5] = 0.0; out[6] = 0.0; out[7] = 0.0;
  const double hinf = r[3], g2 = r[4], wd2_all = r[5], tmax = r[TR_SETUP_NS];
  const double r2 = radius * radius;
  if (approximate) {
    // approximately_solve_bound_constrained_trust_region (trust_region_utils.jl:194-224)
    const double dn = sqrt(wd2_all);
    const double sc = dn > 0.0 ? radius / dn : 1.0;
    out[1] = sc * r[6]; out[2] = sc * r[7];
    return 0;
  }
  if (radius == 0.0 || g2 == 0.0) return 0;   // trust_region_utils.jl:81-83
  // Find t* with radius^2(t*) = r2, radius^2(t) = low(t) + t^2 high(t).  The
  // reference eliminates breakpoints by repeated medians (trust_region_utils.jl:112-165);
  // here: TR_K-ary search over the IEEE bit patterns of t in [0, max finite
  // breakpoint] until no breakpoint lies strictly inside the bracket, then the
  // same closed form (trust_region_utils.jl:167-175).  Every probe carries the value sums of its t
  // (tr_probe_kernel), and the set-up pass those of t = tmax, so t* needs no pass of its own:
  //   value(t*) = vlow + t* vhigh  at the bracket's lower end (no breakpoint lies in between).
  auto probe = [&](const TrProbes &pr, double *sums) -> int {
    FOR_SHARDS(L, h) {
      hipLaunchKernelGGL(tr_probe_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int)h->cn, (int)(h->cn + h->m),
                         h->tr_thr, h->tr_dir, h->tr_g, pr, h->ev_partials, h->ev_grid);
      HIP_TRY(hipGetLastError());
    }
    return ev_finish(L, TR_Q * TR_K, 0, sums);
  };
  double lh[TR_Q * TR_K];
  TrProbes pr;
  TrSearch S;                                // the search itself: eval_kernels.hpp (shared with the one-workgroup kernel)
  tr_search_begin(S, r2, tmax, hinf, TrEnd{r[11], hinf, {r[12], r[14], r[13], r[15]}});
  while (tr_search_next(S, pr)) {
    if ((rc = probe(pr, lh))) return rc;
    tr_search_feed(S, pr, lh);
  }
  out[1] = S.at.v[0] + S.tstar * S.at.v[1];
  out[2] = S.at.v[2] + S.tstar * S.at.v[3];
  out[5] = S.tstar; out[6] = (double)S.passes;     // probe passes (the set-up pass evaluates t = tmax itself)
  return 0;
}

int pdhg_trust_region_bounds(pdhg_handle *h0, int count, const int *points, double primal_weight_norm, double dual_weight_norm,
                             const double *radii, const int *ranges, int approximate, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (count < 1 || count > TRB_MAX || !points || !radii || !ranges || !out) return fail(-1, "count must be 1..3 and the arrays non-null");
  for (int p = 0; p < count; ++p) if (ranges[p] < 0 || ranges[p] > 2) return fail(-1, "range must be 0, 1 or 2");
  const Shards L = shards_of(h0);
  pdhg_handle *h = L.p[0];
  const char *se = dev_env("PDHG_SMALL_EVAL");
  const bool small = !L.g && h->n + h->m <= TRS_MAX && !(se && se[0] == '0');
  const char *be = dev_env("PDHG_TR_BATCH");
  // (the batch snapshots every point's products, which needs the per-point buffers: with the evaluation cache switched off
  //  -- PDHG_NO_EVAL_CACHE, dev -- all points share one set and only the last point's products would survive)
  const char *nc = dev_env("PDHG_NO_EVAL_CACHE");
  bool batch = count > 1 && !L.g && !h->profile && !small && !(be && be[0] == '0') && nc == nullptr;
  if (batch) {
    if ((rc = flush_pending(L))) return rc;
    rc = tr_coop_prepare(h);
    if (rc > 1 || rc < 0) return rc;
    batch = rc == 0;
  }
  if (batch) {
    if ((rc = ev_ensure_host(h))) return rc;
    const int64_t total = h->n + h->m;
    if (!h->trb_scratch) {
      if ((rc = alloc_zero(&h->trb_scratch, 3 * (int64_t)TRB_MAX * total))) return rc;
      if ((rc = alloc_zero(&h->trb_partials, 2 * (int64_t)TRB_MAX * EV_MAXQ * h->tr_grid))) return rc;
    }
    TrBatchArgs a{};
    a.n = (int)h->n; a.m = (int)h->m; a.ne = (int)h->num_eq; a.approximate = approximate ? 1 : 0; a.count = count;
    a.c = h->c; a.b = h->b; a.lb = h->lb; a.ub = h->ub;
    a.wp = primal_weight_norm; a.wd = dual_weight_norm;
    for (int p = 0; p < count; ++p) {
      if ((rc = point_products(L, points[p]))) return rc;       // (cached per point: nothing is recomputed for a point seen before)
      TrBatchProblem &q = a.pb[p];
      q.range = ranges[p]; q.radius = radii[p];
      q.px = h->pt_x; q.py = h->pt_y; q.aty = h->pt_aty; q.qx = h->pt_qx; q.ax = h->pt_ax;
      q.gdv = h->trb_scratch + (3 * (int64_t)p + 0) * total;
      q.wd2v = h->trb_scratch + (3 * (int64_t)p + 1) * total;
      q.thr = h->trb_scratch + (3 * (int64_t)p + 2) * total;
    }
    a.partials = h->trb_partials;
    a.sync = h->tr_sync;
    a.epoch = h->tr_epoch;
    a.nxcd = h->tr_nxcd;
    for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = h->tr_xcd_cnt[x];
    a.host_out = h->ev_host;
    a.seq = ++h->ev_seq;
    double r[8 * TRB_MAX + 2];
    {
      std::lock_guard<std::mutex> lock(coop_device_mutex(h->device));
      hipLaunchKernelGGL(tr_coop_batch_kernel, dim3(h->tr_grid), dim3(TPB), 0, h->stream, a);
      HIP_TRY(hipGetLastError());
      if ((rc = ev_wait_host(h, 8 * count + 2, a.seq, r))) {
        h->tr_coop = 0;         // the launch may have passed barriers without reporting its epoch: never reuse this GridSync's counters
        return rc;
      }
    }
    h->tr_epoch = (unsigned long long)r[8 * count + 1];
#ifdef PDHG_TRB_TRACE
    if ((h->trb_calls % 100) == 99) {
      unsigned long long t[8];
      if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_trb_trace), sizeof(t)) == hipSuccess && t[5] > 0 && t[7] > 0)
        fprintf(stderr, "[pdhg_hip] batched searches: %llu launches, %.1f probe passes each; per pass (us): walk %.2f, block reductions %.2f, "
                        "barrier %.2f, second stage %.2f, search step %.2f; set-up pass incl. barrier %.2f\n", t[7], (double)t[5] / t[7],
                0.01 * t[0] / t[5], 0.01 * t[1] / t[5], 0.01 * t[2] / t[5], 0.01 * t[3] / t[5], 0.01 * t[4] / t[5], 0.01 * t[6] / t[7]);
    }
#endif
    if (r[8 * count] == 0.0) {
      for (int q = 0; q < 8 * count; ++q) out[q] = r[q];
      h->trb_calls += 1;
      h->tr_coop_calls += count;
      return 0;
    }
    h->tr_coop = 0;             // a barrier timed out: this handle goes back to one launch per pass, starting with these problems
    if (getenv("PDHG_VERBOSE")) fprintf(stderr, "[pdhg_hip] trust-region batch: a grid barrier timed out (code %g); back to one launch per pass\n", r[8 * count]);
  }
  for (int p = 0; p < count; ++p)
    if ((rc = pdhg_trust_region_bound(h0, points[p], primal_weight_norm, dual_weight_norm, radii[p], ranges[p], approximate, out + 8 * p))) return rc;
  return 0;
}

