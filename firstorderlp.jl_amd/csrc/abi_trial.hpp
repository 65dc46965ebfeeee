// abi_trial.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// C ABI: the trial step, accept, take_step(s), the persistent group trial (group_kernel.hpp) host side, state getters / setters.

// ---- the trial step -----------------------------------------------------------------

// Single GPU: K1+K2, K3+K4, K5+K6 (fused epilogues), second-stage reduction.
static int trial_dual_single(pdhg_handle *h, double step_size, double primal_weight, double out[5]) {
  int rc;
  if ((rc = launch_dual(h, primal_weight * step_size))) return rc;
  if ((rc = launch_aty_fused(h))) return rc;
  int qcount = 0;
  if ((rc = launch_q_interaction(h, &qcount))) return rc;
  // The five sums go straight into pinned host memory and the host polls the launch's sequence number there
  // (as on the graph path) instead of a device-to-host copy + stream synchronisation: ~10 us per trial, which
  // is 5 % of a 1M x 1M LP's iteration.  While profiling: the copy, so that the event brackets stay simple.
  static const bool host_word = !(dev_env("PDHG_TRIAL_HOST_WORD") && dev_env("PDHG_TRIAL_HOST_WORD")[0] == '0');
  if (host_word && !h->profile) {
    if ((rc = ensure_result_word(h))) return rc;
    if ((rc = launch_final(h, h->pAt, h->At.slots(), h->pAt_stride, h->pA, h->A.slots(), qcount, true))) return rc;
    return wait_result_word(h, out);
  }
  if ((rc = launch_final(h, h->pAt, h->At.slots(), h->pAt_stride, h->pA, h->A.slots(), qcount))) return rc;
  HIP_TRY(hipMemcpyAsync(h->scal_host, h->scal_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int q = 0; q < 5; ++q) out[q] = h->scal_host[q];
  out[4] *= 0.5;
  return 0;
}

// One shard's whole trial, issued by that shard's own host thread (ShardPool): the same
// launches, in the same order, as trial_dual_group issues for it from the calling thread.
struct TrialArgs {
  double step_size, primal_weight, theta;
  bool primal;      // K1+K2 first (pdhg_trial_step); false: xbar only (pdhg_trial_dual)
};
static int trial_shard_mt(DistGroup &g, pdhg_handle *s, int i, const TrialArgs &a, double *t_issued) {
  HIP_TRY(hipSetDevice(s->device));
  int rc;
  if (a.primal) { if ((rc = launch_primal(s, a.step_size / a.primal_weight, a.theta, true))) return rc; }
  else if ((rc = launch_xbar(s, a.theta))) return rc;
  const double sigma = a.primal_weight * a.step_size;
  if (g.ag_chunks > 1 && !s->has_q) {
    // xbar chunk by chunk, A_p xbar as one pass per chunk (see trial_dual_group)
    if (g.backend == COMM_RCCL) {
      // the owned slice into the chunk layout, then one all-gather per chunk on the comm stream; ag_mode 2: the passes wait
      // for the LAST chunk (one all-gather's worth of waiting: nothing overlapped, the same bits)
      if ((rc = launch_chunk_pack(g, s, s->xbar, s->xchunk, s->rank, s->rank + 1, s->n_alloc, s->stream))) return rc;
      HIP_TRY(hipEventRecord(s->ev_xbar, s->stream));
      for (int c = 0; c < g.ag_chunks; ++c)
        if ((rc = mt_all_gather_chunk(g, s, i, c))) return rc;
      if (g.ag_mode != 1) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_ag[(size_t)g.ag_chunks - 1], 0));
      if ((rc = launch_dual_chunked(s, sigma, g.ag_mode == 1))) return rc;
    } else {
      // peer back end: the ordinary all-gather, then the whole vector into the chunk layout
      if ((rc = mt_all_gather(g, s, i, [](pdhg_handle *q) { return q->xbar; }, g.S))) return rc;
      if ((rc = launch_chunk_pack(g, s, s->xbar, s->xchunk, 0, g.world, s->n_alloc, s->stream))) return rc;
      if ((rc = launch_dual_chunked(s, sigma, false))) return rc;
    }
  } else {
    if ((rc = mt_all_gather(g, s, i, [](pdhg_handle *q) { return q->xbar; }, g.S))) return rc;
    if (s->has_q && (rc = mt_all_gather(g, s, i, [](pdhg_handle *q) { return q->x_next; }, g.S))) return rc;
    if ((rc = launch_dual(s, sigma))) return rc;
  }
  if (!g.overlap) {
    if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc;
    if ((rc = mt_reduce_scatter(g, s, i, [](pdhg_handle *q) { return q->aty_next; }, g.S))) return rc;
  } else {
    // see trial_dual_group: the product in residency rounds, slice k reduced as soon as its rows are complete
    const char *rw_env = dev_env("PDHG_DIST_ROUND_WGS");
    const int round_wgs = rw_env ? std::max(1, atoi(rw_env)) : 256 * 2;
    const CsrDev &T = s->At;
    int issued = 0, next_wg = 0;
    for (int k = 0; k < g.world; ++k) {
      const int64_t need = std::min<int64_t>(s->n, (int64_t)(k + 1) * g.S);
      if (!T.tiled) {
        if (!issued) { if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc; issued = 1; }
      } else {
        while (next_wg < T.grid || !issued) {
          const int g0 = next_wg;
          const bool covered = g0 >= T.grid || (int64_t)T.wg_first_row[(size_t)g0] >= need;
          if (covered && issued) break;
          int g1 = std::min(T.grid, g0 + round_wgs);
          if (T.grid - g1 < round_wgs / 2) g1 = T.grid;
          if ((rc = launch_spmv_plain_part(s, T, s->y_next, s->aty_next, g0, g1, !issued))) return rc;
          issued = 1;
          next_wg = g1;
        }
      }
      HIP_TRY(hipEventRecord(s->ev_part[(size_t)k], s->stream));
      if ((rc = mt_reduce_slice_async(g, s, i, [](pdhg_handle *q) { return q->aty_next; }, g.S, k))) return rc;
    }
    if ((rc = mt_join_comm(g, s, i))) return rc;
  }
  {
    const int64_t o = s->clo;
    hipLaunchKernelGGL(interaction_kernel, dim3(ew_grid(s->cn)), dim3(TPB), 0, s->stream, (int)s->cn, s->x + o,
                       s->x_next + o, s->aty + o, s->aty_next + o, s->pAt, s->pAt_stride);
    HIP_TRY(hipGetLastError());
  }
  int qcount = 0;
  if ((rc = launch_q_interaction(s, &qcount))) return rc;
  {
    const bool chunked = g.ag_chunks > 1 && !s->has_q;
    if ((rc = launch_final(s, s->pAt, ew_grid(s->cn), s->pAt_stride, s->pA, chunked ? dual_chunk_slots(s) : s->A.slots(), qcount, false,
                           chunked ? dual_chunk_slots(s) : -1))) return rc;
  }
  HIP_TRY(hipMemcpyAsync(s->scal_host, s->scal_dev, sizeof(double) * 5, hipMemcpyDeviceToHost, s->stream));
  *t_issued = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

static int trial_group_mt(const Shards &L, const TrialArgs &a, double out[5]) {
  DistGroup &g = *L.g;
  const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  std::vector<double> issued((size_t)L.count, t0);
  int rc = g.pool->run([&](int i) { return trial_shard_mt(g, L.p[i], i, a, &issued[(size_t)i]); });
  if (rc) return rc;
  const double t2 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  double t1 = t0;
  for (double v : issued) t1 = std::max(t1, v);
  g.t_issue += t1 - t0; g.t_wait += t2 - t1; g.n_trials += 1;
  // the shards' scalars, added in rank order ([4], dx'Q dx, is replicated: maxed) -- as combine_scalars does
  for (int q = 0; q < 5; ++q) {
    double v = L.p[0]->scal_host[q];
    for (int i = 1; i < L.count; ++i) {
      const double t = L.p[i]->scal_host[q];
      v = (q < 4) ? v + t : std::fmax(v, t);
    }
    out[q] = v;
  }
  out[4] *= 0.5;
  return 0;
}

// ---- a group's trial as ONE persistent kernel per device (group_kernel.hpp) -----------------------------------------
// Eligible: every shard of the group lives in this process on the peer back end, LP, stream layouts without slabs, and
// the shards' grids fit their device side by side.  Default: on when all shards share ONE device (the configuration this
// environment can test -- bitwise the ordinary group path); for shards on distinct devices the protocol has never run,
// so it waits for PDHG_GROUP_COOP=1.  PDHG_GROUP_COOP=0: off.
static int group_coop_prepare(const Shards &L) {
  DistGroup &g = *L.g;
  std::vector<int> devs;
  for (int i = 0; i < L.count; ++i) devs.push_back(L.p[i]->device);
  std::sort(devs.begin(), devs.end());
  devs.erase(std::unique(devs.begin(), devs.end()), devs.end());
  const char *pretend = dev_env("PDHG_COOP_TEST_PRETEND_WGS");       // test knob: a grid the device cannot hold
  for (int dev : devs) {
    HIP_TRY(hipSetDevice(dev));
    // (the record enters g.coop_dev FIRST: whatever fails below, the caller's group_coop_release frees what it holds by then)
    g.coop_dev.emplace_back();
    GroupDevLaunch &D = g.coop_dev.back();
    D.device = dev;
    for (int i = 0; i < L.count; ++i) if (L.p[i]->device == dev) D.members.push_back(i);
    D.stream = L.p[D.members[0]]->stream;
    int per_cu = 0, per_cu_inline = 0;
    hipDeviceProp_t prop;
    // co-residency of the kernel that WILL be launched: up to GROUP_INLINE_SHARDS members per device take the inline-argument
    // variant (group_coop_trial), whose registers and kernel arguments differ -- size for the smaller of the two
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, group_trial_kernel, TPB, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_inline, group_trial_inline_kernel, TPB, 0));
    per_cu = std::min(per_cu, per_cu_inline);
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const int cap = std::max(8, per_cu * prop.multiProcessorCount / 8 * 8);
    // every shard one workgroup per item where the device holds that many side by side, else in proportion
    std::vector<int> items, grid;
    int64_t total_items = 0;
    for (int i : D.members) {
      const pdhg_handle *s = L.p[i];
      items.push_back(std::max(8, (std::max(s->A.grid + s->A.nchunks, s->At.grid + s->At.nchunks) + 7) / 8 * 8));
      total_items += items.back();
    }
    int total = 0;
    for (size_t k = 0; k < items.size(); ++k) {
      int gk = total_items <= cap ? items[k] : std::max(8, (int)((int64_t)cap * items[k] / total_items) / 8 * 8);
      if (pretend) gk = std::max(8, atoi(pretend) / 8 * 8);
      if (items[k] > 2 * gk) { g_last_error = "too many row blocks for one persistent launch per device"; return 1; }
      grid.push_back(gk);
      total += gk;
    }
    if (total > cap && !pretend) return 1;
    D.base.assign(1, 0);
    for (int gk : grid) D.base.push_back(D.base.back() + gk);
    D.grid = total;
    const size_t k_n = D.members.size();
    HIP_TRY(hipMalloc((void **)&D.args_dev, sizeof(GroupTrialArgs) * k_n));
    HIP_TRY(hipHostMalloc((void **)&D.args_host, sizeof(GroupTrialArgs) * k_n, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&D.sync_dev, sizeof(GridSync *) * k_n));
    HIP_TRY(hipEventCreateWithFlags(&D.ev_done, hipEventDisableTiming));
    D.ev.assign(k_n, nullptr);
    std::vector<GridSync *> syncs;
    for (size_t k = 0; k < k_n; ++k) {
      pdhg_handle *s = L.p[D.members[k]];
      HIP_TRY(hipEventCreateWithFlags(&D.ev[k], hipEventDisableTiming));
      int rc = ensure_result_word(s);
      if (rc) return rc;
      if (!s->gsync) HIP_TRY(hipMalloc((void **)&s->gsync, sizeof(GridSync)));
      HIP_TRY(hipMemset(s->gsync, 0, sizeof(GridSync)));
      s->coop_grid = grid[k];
      s->coop_epoch = 0; s->coop_launches = 0;
      if (s->coop_grid > s->pAt_stride) {          // the interaction partials take one slot per workgroup
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (s->pAt) (void)hipFree(s->pAt);
        s->pAt = nullptr;
        s->pAt_stride = s->coop_grid;
        if ((rc = alloc_zero(&s->pAt, 6 * (int64_t)s->pAt_stride))) return rc;
      }
      syncs.push_back(s->gsync);
    }
    HIP_TRY(hipMemcpy(D.sync_dev, syncs.data(), sizeof(GridSync *) * k_n, hipMemcpyHostToDevice));
    // census of the merged launch shape: workgroups of every shard per XCD
    GroupDeviceArgs da{};
    da.shard = D.args_dev; da.nshards = (int)k_n;
    for (size_t k = 0; k <= k_n; ++k) da.base[k] = D.base[k];
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(group_register_kernel, dim3(D.grid), dim3(TPB), 0, D.stream, da, (GridSync *const *)D.sync_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(D.stream));
    for (size_t k = 0; k < k_n; ++k) {
      pdhg_handle *s = L.p[D.members[k]];
      GridSync host;
      HIP_TRY(hipMemcpy(&host, s->gsync, sizeof(GridSync), hipMemcpyDeviceToHost));
      unsigned long long seen = 0;
      s->coop_nxcd = 0;
      for (int x = 0; x < 8; ++x) { seen += host.xcd_count[x][0]; s->coop_nxcd += host.xcd_count[x][0] > 0; s->coop_xcd_cnt[x] = (unsigned)host.xcd_count[x][0]; }
      if (seen != (unsigned long long)grid[k] || s->coop_nxcd == 0) return fail(996, "group trial kernel: workgroup census does not add up");
    }
  }
  if (!g.gsync) {
    HIP_TRY(hipSetDevice(L.p[0]->device));
    void *p = nullptr;
    // fine-grained device memory when the runtime offers it: the devices poll these words with system-scope atomics
    if (hipExtMallocWithFlags(&p, sizeof(GroupSync), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      HIP_TRY(hipMalloc(&p, sizeof(GroupSync)));
    }
    HIP_TRY(hipMemset(p, 0, sizeof(GroupSync)));
    HIP_TRY(hipDeviceSynchronize());
    g.gsync = reinterpret_cast<GroupSync *>(p);
  }
  return 0;
}

static bool group_coop_eligible(const Shards &L) {
  DistGroup &g = *L.g;
  if (g.coop_mode < 0) {
    const char *ev = getenv("PDHG_GROUP_COOP");
    bool on = g.all_local() && g.backend == COMM_P2P && L.count == g.world && g.world >= 2 && g.world <= P2P_MAX_WORLD &&
              !(ev && ev[0] == '0') && !(getenv("PDHG_GRAPH") && getenv("PDHG_GRAPH")[0] == '0') &&
              g.ag_chunks <= 1;      // (column-chunk passes fix another order of additions: the per-launch path runs them)
    bool one_device = true;
    for (int i = 0; i < L.count && on; ++i) {
      const pdhg_handle *s = L.p[i];
      one_device = one_device && s->device == L.p[0]->device;
      on = !s->has_q && s->lazy_accept && s->n > 0 && s->cn > 0 && !s->A.tiled && !s->At.tiled && s->A.slabs.empty() &&
           s->At.slabs.empty() && s->A.segs.empty() && s->At.segs.empty() && s->coop_mode != 1 && !s->gsync;
    }
    if (on && !one_device && !(ev && ev[0] == '1')) on = false;
    g.coop_mode = 0;
    if (on) {
      const int rc = group_coop_prepare(L);
      if (rc == 0) g.coop_mode = 1;
      else { (void)hipGetLastError(); group_coop_release(g); }
      if (getenv("PDHG_VERBOSE")) {
        fprintf(stderr, "[pdhg_hip] group of %d shards: one persistent kernel per device and trial %s", g.world, rc == 0 ? "ON" : "not possible");
        for (const GroupDevLaunch &D : g.coop_dev) fprintf(stderr, " [device %d: %zu shards, %d workgroups]", D.device, D.members.size(), D.grid);
        fprintf(stderr, "\n");
      }
    }
  }
  return g.coop_mode == 1 && !L.p[0]->profile;
}

// returns 1 when the trial was not taken here (the caller runs the ordinary group path)
static int group_coop_trial(const Shards &L, const TrialArgs &ta, double out[5]) {
  DistGroup &g = *L.g;
  // one persistent launch set at a time per device (two half-resident sets would wait for each other)
  std::vector<std::unique_lock<std::mutex>> locks;
  for (GroupDevLaunch &D : g.coop_dev) locks.emplace_back(coop_device_mutex(D.device));      // (ascending device ids)
  const auto t_begin = std::chrono::steady_clock::now();
  const double sigma = ta.primal_weight * ta.step_size;
  for (GroupDevLaunch &D : g.coop_dev) {
    HIP_TRY(hipSetDevice(D.device));
    for (size_t k = 0; k < D.members.size(); ++k) {
      pdhg_handle *s = L.p[D.members[k]];
      GroupTrialArgs a{};
      a.rank = s->rank; a.world = g.world;
      const int64_t o = s->clo;
      a.cn = (int)s->cn; a.clo = o; a.xbar_only = ta.primal ? 0 : 1;
      a.x = s->x + o; a.c = s->c + o; a.aty = s->aty + o; a.lb = s->lb + o; a.ub = s->ub + o;
      a.tau = ta.step_size / ta.primal_weight; a.theta = ta.theta;
      a.x_next = s->x_next + o;
      a.avg_w = s->pend_w; a.sum_x = (s->pend_x && ta.primal) ? s->sum_x + o : nullptr;
      for (int q = 0; q < L.count; ++q) {
        a.xbar_peer[L.p[q]->rank] = L.p[q]->xbar;
        a.part_peer[L.p[q]->rank] = L.p[q]->aty_next;
      }
      EpiArgs de{};
      de.y = s->y; de.b = s->b; de.y_next = s->y_next; de.sigma = sigma; de.num_eq = (int)s->num_eq;
      de.partials = s->pA; de.stride = s->A.slots(); de.lo_offset = s->A.slots();
      if (s->pend_y) { de.sum_y = s->sum_y; de.avg_w = s->pend_w; }
      a.A = trial_product(s, s->A, s->xbar, de);
      EpiArgs te{};
      te.out = s->aty_next;
      a.T = trial_product(s, s->At, s->y_next, te);
      a.off = o; a.aty_next = s->aty_next;
      a.pAt = s->pAt; a.pAt_stride = s->pAt_stride;
      a.sp.ptr[0] = s->pAt;                         a.sp.count[0] = s->coop_grid;
      a.sp.ptr[1] = s->pAt + s->pAt_stride;         a.sp.count[1] = s->coop_grid;
      a.sp.ptr[2] = s->pA;                          a.sp.count[2] = s->A.slots();
      a.sp.ptr[3] = s->pAt + 2 * s->pAt_stride;     a.sp.count[3] = s->coop_grid;
      a.sp.ptr[4] = s->pQ;                          a.sp.count[4] = 0;
      for (int q : {0, 1, 3}) a.sp.ptr_lo[q] = a.sp.ptr[q] + 3 * s->pAt_stride;
      a.sp.ptr_lo[2] = s->pA + s->A.slots();
      a.sp.ptr_lo[4] = s->pQ + s->ew_grid_n;
      a.sp.out = nullptr;
      a.sync = s->gsync; a.gsync = g.gsync;
      a.epoch = s->coop_epoch; s->coop_epoch += 3;
      a.xepoch = g.xepoch;
      a.launch = s->coop_launches; s->coop_launches += 1;
      s->seq_expected += 1;
      a.seq = s->seq_expected;
      a.nxcd = s->coop_nxcd;
      for (int x = 0; x < 8; ++x) a.xcd_cnt[x] = s->coop_xcd_cnt[x];
      a.seq_dev = s->seq_dev; a.res_host = s->res_host; a.relaxed = s->relaxed ? 1 : 0;
      D.args_host[k] = a;                       // (the previous launch has returned its results: the staging copy is free)
      if (ta.primal) s->pend_x = false;
      s->pend_y = false;                        // the launch carries the deferred average update
      // whatever the shard's own stream has queued since the last trial (a flush, a set_current, an evaluation) comes first
      if (g.members_dirty && s->stream != D.stream) {
        HIP_TRY(hipEventRecord(D.ev[k], s->stream));
        HIP_TRY(hipStreamWaitEvent(D.stream, D.ev[k], 0));
      }
    }
    if ((int)D.members.size() <= GROUP_INLINE_SHARDS) {            // the argument blocks by value: nothing to upload
      const int k_n = (int)D.members.size();
      hipLaunchKernelGGL(group_trial_inline_kernel, dim3(D.grid), dim3(TPB), 0, D.stream, D.args_host[0], D.args_host[k_n > 1 ? 1 : 0],
                         k_n, k_n > 1 ? D.base[1] : D.grid, D.grid);
    } else {
      HIP_TRY(hipMemcpyAsync(D.args_dev, D.args_host, sizeof(GroupTrialArgs) * D.members.size(), hipMemcpyHostToDevice, D.stream));
      GroupDeviceArgs da{};
      da.shard = D.args_dev; da.nshards = (int)D.members.size();
      for (size_t k = 0; k <= D.members.size(); ++k) da.base[k] = D.base[k];
      hipLaunchKernelGGL(group_trial_kernel, dim3(D.grid), dim3(TPB), 0, D.stream, da);
    }
    HIP_TRY(hipGetLastError());
    // ... and whatever is queued on the members' streams next comes after this launch (check_handle: lazily)
    HIP_TRY(hipEventRecord(D.ev_done, D.stream));
  }
  g.members_dirty = false;
  g.join_pending = true;
  g.xepoch += 2;
  const auto t_issued = std::chrono::steady_clock::now();
  bool failed = false;
  double sums[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < L.count; ++i) {
    pdhg_handle *s = L.p[i];
    HIP_TRY(hipSetDevice(s->device));
    double r[5];
    const int rc = wait_result_word(s, r, true);
    if (rc) return rc;
    failed = failed || s->res_error != 0.0;
    for (int q = 0; q < 4; ++q) sums[q] = (i == 0) ? r[q] : sums[q] + r[q];      // rank order (L.p is ascending in rank)
  }
  g.t_issue += std::chrono::duration<double>(t_issued - t_begin).count();
  g.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_issued).count();
  g.n_trials += 1;
  if (failed) {
    // a barrier ran into its spin limit (the launches were not all co-resident): every workgroup still ran every phase, so
    // the deferred average updates are applied exactly once; x', y', A'y' and the sums are not trustworthy -- the caller
    // repeats the trial on the ordinary group path (its inputs are untouched) and the group stays there
    g.coop_mode = 0;
    g.coop_fallbacks += 1;
    fprintf(stderr, "[pdhg_hip] group trial kernel: a barrier timed out -- this group uses the per-launch path from here on\n");
    return check_handle(L.p[0]) ? -1 : 1;        // (the members' streams wait for the failed launches before the repeat)
  }
  for (int q = 0; q < 4; ++q) out[q] = sums[q];
  out[4] = 0.0;
  g.coop_trials += 1;
  return 0;
}

// Row-partitioned group: the dual half of a trial.  xbar's owned slices are ready.
static int trial_dual_group(const Shards &L, double step_size, double primal_weight, double out[5]) {
  DistGroup &g = *L.g;
  pdhg_handle *lead = L.p[0];
  int rc;
  const auto t_begin = std::chrono::steady_clock::now();
  // The all-gather of xbar beside A_p xbar (DistGroup::ag_chunks; SURVEY 8e(ii), pdhg.jl:472-494): xbar travels in column
  // chunks on the comm streams -- chunk c = sub-range c of every rank's slice, so that every link carries a part of every
  // chunk -- and A_p xbar is one pass per chunk, pass c waiting for chunk c alone while chunk c + 1 is on the links.
  // ag_mode 2 (and the peer back end): the same passes behind one all-gather -- the same bits, nothing overlapped.
  const bool chunked = g.ag_chunks > 1 && !lead->has_q;
  const bool chunks_on_comm = chunked && g.ag_mode == 1 && g.backend == COMM_RCCL;
  if (chunked && g.backend == COMM_RCCL) {
    // the owned slices into the chunk layout (dist.hpp), then ONE ncclAllGather per chunk on the comm streams
    ProfScope ps(lead, PDHG_K_ALLGATHER);
    FOR_SHARDS(L, s) {
      if ((rc = launch_chunk_pack(g, s, s->xbar, s->xchunk, s->rank, s->rank + 1, s->n_alloc, s->stream))) return rc;
      HIP_TRY(hipEventRecord(s->ev_xbar, s->stream));
    }
    for (int c = 0; c < g.ag_chunks; ++c)
      if ((rc = dist_all_gather_chunk(g, c))) return rc;
    if (!chunks_on_comm)       // ag_mode 2: every pass behind the whole all-gather
      FOR_SHARDS(L, s) { HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_ag[(size_t)g.ag_chunks - 1], 0)); }
  } else if (chunked) {
    // peer back end: the ordinary all-gather, then the whole vector into the chunk layout
    ProfScope ps(lead, PDHG_K_ALLGATHER);
    if ((rc = dist_all_gather(g, [](pdhg_handle *s) { return s->xbar; }, g.S))) return rc;
    FOR_SHARDS(L, s) { if ((rc = launch_chunk_pack(g, s, s->xbar, s->xchunk, 0, g.world, s->n_alloc, s->stream))) return rc; }
  } else {
    ProfScope ps(lead, PDHG_K_ALLGATHER);
    if ((rc = dist_all_gather(g, [](pdhg_handle *s) { return s->xbar; }, g.S))) return rc;
    // QP: Q acts on full vectors, so x' is kept full as well (x becomes x' at accept)
    if (lead->has_q && (rc = dist_all_gather(g, [](pdhg_handle *s) { return s->x_next; }, g.S))) return rc;
  }
  auto dual_of = [&](pdhg_handle *s) { return chunked ? launch_dual_chunked(s, primal_weight * step_size, chunks_on_comm)
                                                      : launch_dual(s, primal_weight * step_size); };
  if (!g.overlap) {
    FOR_SHARDS(L, s) {
      if ((rc = dual_of(s))) return rc;
      if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc;      // t_p = A_p' y'_p, all n columns
    }
    ProfScope ps(lead, PDHG_K_REDUCE_SCATTER);
    if ((rc = dist_reduce_scatter(g, [](pdhg_handle *s) { return s->aty_next; }, g.S))) return rc;
  } else {
    // t_p in parts: a shard whose A_p' uses the tiled layout launches it one residency
    // round at a time (256 CUs x 2 workgroups: a smaller launch would idle CUs for the whole
    // sweep); as soon as the rows of slice k are complete, slice k is reduced to rank k on
    // the comm stream while the next round computes.  The sequence of collectives (slice
    // 0, 1, ..., P-1) is the same on every rank however the local product is cut.
    FOR_SHARDS(L, s) { if ((rc = dual_of(s))) return rc; }
    const char *rw_env = dev_env("PDHG_DIST_ROUND_WGS");            // tests use a finer granule on small problems
    const int round_wgs = rw_env ? std::max(1, atoi(rw_env)) : 256 * 2;
    std::vector<int> issued((size_t)L.count, 0), next_wg((size_t)L.count, 0);
    int k_issued = 0;
    while (k_issued < g.world) {
      // every local shard advances until slice k_issued is complete on it
      for (int i = 0; i < L.count; ++i) {
        pdhg_handle *s = L.p[i];
        HIP_TRY(hipSetDevice(s->device));
        const CsrDev &T = s->At;
        const int64_t need = std::min<int64_t>(s->n, (int64_t)(k_issued + 1) * g.S);   // rows [0, need) must be done
        if (!T.tiled) {
          if (issued[(size_t)i] == 0) {
            if ((rc = launch_aty_plain(s, s->y_next, s->aty_next))) return rc;
            issued[(size_t)i] = 1;
          }
        } else {
          while (next_wg[(size_t)i] < T.grid || issued[(size_t)i] == 0) {
            const int g0 = next_wg[(size_t)i];
            const bool covered = g0 >= T.grid || (int64_t)T.wg_first_row[(size_t)g0] >= need;
            if (covered && issued[(size_t)i] != 0) break;
            int g1 = std::min(T.grid, g0 + round_wgs);
            if (T.grid - g1 < round_wgs / 2) g1 = T.grid;       // no runt round at the end
            ProfScope ps(s, PDHG_K_SPMV_ATY);
            if ((rc = launch_spmv_plain_part(s, T, s->y_next, s->aty_next, g0, g1, issued[(size_t)i] == 0))) return rc;
            issued[(size_t)i] = 1;
            next_wg[(size_t)i] = g1;
          }
        }
        HIP_TRY(hipEventRecord(s->ev_part[(size_t)k_issued], s->stream));
      }
      if ((rc = dist_reduce_slice_async(g, [](pdhg_handle *s) { return s->aty_next; }, g.S, k_issued))) return rc;
      ++k_issued;
    }
    ProfScope ps(lead, PDHG_K_REDUCE_SCATTER);     // what is left of the exchange after the product
    if ((rc = dist_join_comm(g))) return rc;
  }
  FOR_SHARDS(L, s) {
    {
      ProfScope ps(s, PDHG_K_INTERACTION);
      const int64_t o = s->clo;
      hipLaunchKernelGGL(interaction_kernel, dim3(ew_grid(s->cn)), dim3(TPB), 0, s->stream, (int)s->cn, s->x + o,
                         s->x_next + o, s->aty + o, s->aty_next + o, s->pAt, s->pAt_stride);
      HIP_TRY(hipGetLastError());
    }
    int qcount = 0;
    if ((rc = launch_q_interaction(s, &qcount))) return rc;   // replicated: identical on every shard
    if ((rc = launch_final(s, s->pAt, ew_grid(s->cn), s->pAt_stride, s->pA, chunked ? dual_chunk_slots(s) : s->A.slots(), qcount, false,
                           chunked ? dual_chunk_slots(s) : -1))) return rc;
  }
  double r[5];
  const auto t_issued = std::chrono::steady_clock::now();
  // [0..4) are added in rank order; [4] (dx'Q dx, replicated: the same value on every rank) is "maxed"
  if ((rc = combine_scalars(L, 5, 4, r))) return rc;
  g.t_issue += std::chrono::duration<double>(t_issued - t_begin).count();
  g.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_issued).count();
  g.n_trials += 1;
  for (int q = 0; q < 4; ++q) out[q] = r[q];
  out[4] = 0.5 * r[4];
  return 0;
}

int pdhg_trial_primal(pdhg_handle *h, double step_size, double primal_weight) {
  int rc = check_handle(h);
  if (rc) return rc;
  const Shards L = shards_of(h);
  FOR_SHARDS(L, s) { if ((rc = launch_primal(s, step_size / primal_weight, 0.0, false))) return rc; }
  return 0;
}

// true when the trial will be taken as persistent group launches (group_kernel.hpp), which queue nothing on the members' own
// streams -- the same predicate group_coop_eligible() ends on (a profiled group takes the per-launch path, which DOES)
static bool trial_stays_off_member_streams(const pdhg_handle *h) {
  return h && h->grp && h->grp->coop_mode == 1 && !h->grp->sh.empty() && !h->grp->sh[0]->profile;
}

int pdhg_trial_dual(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  int rc = check_handle(h, !trial_stays_off_member_streams(h));
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  const Shards L = shards_of(h);
  if (!L.g && coop_eligible(h)) {         // Malitsky-Pock retries: xbar + the dual half
    if ((rc = coop_trial(h, step_size, primal_weight, theta, true, out)) != 1) return rc;    // 1: not run / timed out, repeat below
  }
  if (L.g && group_coop_eligible(L)) {
    if ((rc = group_coop_trial(L, TrialArgs{step_size, primal_weight, theta, false}, out)) != 1) return rc;
  }
  if (L.g && L.g->pool && !L.p[0]->profile) return trial_group_mt(L, TrialArgs{step_size, primal_weight, theta, false}, out);
  FOR_SHARDS(L, s) { if ((rc = launch_xbar(s, theta))) return rc; }
  if (L.g) return trial_dual_group(L, step_size, primal_weight, out);
  return trial_dual_single(h, step_size, primal_weight, out);
}

int pdhg_trial_step(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  RoctxRange roctx_range("pdhg_trial_step");
  int rc = check_handle(h, !trial_stays_off_member_streams(h));
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  const Shards L = shards_of(h);
  if (!L.g && coop_eligible(h)) {
    if ((rc = coop_trial(h, step_size, primal_weight, theta, false, out)) != 1) return rc;   // 1: not run / timed out, repeat below
  }
  if (!L.g && graph_eligible(h)) return graph_trial(h, step_size, primal_weight, theta, out);
  if (L.g && group_coop_eligible(L)) {
    if ((rc = group_coop_trial(L, TrialArgs{step_size, primal_weight, theta, true}, out)) != 1) return rc;
  }
  if (L.g && L.g->pool && !L.p[0]->profile) return trial_group_mt(L, TrialArgs{step_size, primal_weight, theta, true}, out);
  FOR_SHARDS(L, s) { if ((rc = launch_primal(s, step_size / primal_weight, theta, true))) return rc; }
  if (L.g) return trial_dual_group(L, step_size, primal_weight, out);
  return trial_dual_single(h, step_size, primal_weight, out);
}

int pdhg_accept(pdhg_handle *h0, double avg_weight) {
  RoctxRange roctx_range("pdhg_accept");
  // (a lazy accept with nothing pending queues no work: the iterates are swapped on the host)
  int rc = check_handle(h0, !(trial_stays_off_member_streams(h0) && h0->lazy_accept && !h0->pend_x && !h0->pend_y));
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;   // two accepts without a trial in between
  bump_version(L);
  FOR_SHARDS(L, h) {
    if (h->lazy_accept) {
      h->pend_x = h->pend_y = true;         // K7 rides on the next trial's kernels
      h->pend_w = avg_weight;
    } else {
      ProfScope ps(h, PDHG_K_ACCEPT);
      const int64_t o = h->clo;
      hipLaunchKernelGGL(accept_kernel, dim3(ew_grid(std::max(h->cn, h->m))), dim3(TPB), 0, h->stream, (int)h->cn,
                         (int)h->m, avg_weight, h->x_next + o, h->sum_x + o, h->y_next, h->sum_y);
      HIP_TRY(hipGetLastError());
    }
    std::swap(h->x, h->x_next);
    std::swap(h->y, h->y_next);
    std::swap(h->aty, h->aty_next);
    h->sum_x_count += 1; h->sum_y_count += 1;
    h->sum_x_weights += avg_weight; h->sum_y_weights += avg_weight;
  }
  return 0;
}

/* take_step(::AdaptiveStepsizeParams, ...) -- src/primal_dual_hybrid_gradient.jl:653-731 --
 * with its host part in C: the retry loop, compute_interaction_and_movement's scalar
 * arithmetic (:527-549), the step-size rule (:713-729) and the accept.  The same
 * statements as primal_dual_hybrid_gradient.py::take_step_adaptive (bitwise equal
 * results; tests/test_gpu_native_take_step.py); what it removes is the host
 * language's per-call overhead between the trial and the accept. */
// step_on_entry: the step size the take_step was entered with (the average's weight, pdhg.jl:512) -- equal to
// *step_size_io except when a multi-step kernel handed back a take_step it had begun (some trials already rejected)
static int take_step_adaptive_from(pdhg_handle *h, double reduction_exponent, double growth_exponent,
                                   double *step_size_io, double step_on_entry, double primal_weight,
                                   int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
                                   int *numerical_error_out);
int pdhg_take_step_adaptive(pdhg_handle *h, double reduction_exponent, double growth_exponent,
                            double *step_size_io, double primal_weight, int64_t *total_number_iterations_io,
                            double *cumulative_kkt_passes_io, int *numerical_error_out) {
  if (!h || !step_size_io || !total_number_iterations_io || !cumulative_kkt_passes_io || !numerical_error_out)
    return fail(-1, "null argument");
  return take_step_adaptive_from(h, reduction_exponent, growth_exponent, step_size_io, *step_size_io, primal_weight,
                                 total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out);
}
static int take_step_adaptive_from(pdhg_handle *h, double reduction_exponent, double growth_exponent,
                                   double *step_size_io, double step_on_entry, double primal_weight,
                                   int64_t *total_number_iterations_io, double *cumulative_kkt_passes_io,
                                   int *numerical_error_out) {
  double step_size = *step_size_io;
  *numerical_error_out = 0;
  bool done = false;
  while (!done) {
    *total_number_iterations_io += 1;
    double raw[5];
    int rc = pdhg_trial_step(h, step_size, primal_weight, 1.0, raw);
    if (rc) return rc;
    *cumulative_kkt_passes_io += 1;
    const double k1 = (double)(*total_number_iterations_io + 1);
    const StepRule rule = adaptive_step_rule(raw, primal_weight, step_size, pow(k1, -reduction_exponent), pow(k1, -growth_exponent));
    if (rule.numerical_error) {
      *numerical_error_out = 1;
      break;
    }
    if (rule.accept) {
      if ((rc = pdhg_accept(h, step_on_entry))) return rc;   // weight = step size on entry (pdhg.jl:512)
      done = true;
    }
    step_size = rule.next_step;
  }
  *step_size_io = step_size;
  return 0;
}

// Does pdhg_take_steps_adaptive take this handle's batches with the multi-step kernel (steps_kernel)?
static bool device_loop_for(pdhg_handle *h) {
  // Several take_steps per launch (steps_kernel: the rule on the device; stream-layout LPs on one handle).  Bitwise the
  // per-trial launches (tests/test_gpu_device_loop.py) and faster on every grid measured but one tie: L1-SVM 19.7k ->
  // 23.4k it/s, random 100K 22.5k -> 28.6k, 3000 x 2500 32.7k -> 52.7k (trial_kernel.hpp, profiles/r03_trial_kernel.txt).
  // PDHG_DEVICE_LOOP=0 / 1: never / whenever eligible; PDHG_DEVICE_LOOP_MAX_WGS: largest grid it is the default for.
  const char *dl_env = getenv("PDHG_DEVICE_LOOP");
  bool device_loop = dl_env && dl_env[0] == '1';
  if (!dl_env && !h->grp && !h->profile && !h->has_q && check_handle(h) == 0 && coop_eligible(h)) {
    static const int max_wgs = dev_env("PDHG_DEVICE_LOOP_MAX_WGS") ? atoi(dev_env("PDHG_DEVICE_LOOP_MAX_WGS")) : (1 << 30);
    device_loop = h->coop_grid <= max_wgs;
  }
  return device_loop;
}

/* `n_steps` consecutive take_steps (the iterations optimize() runs between two termination
 * evaluations, pdhg.jl:862-1046: nothing but take_step happens there).  Stops after the step that
 * raised numerical_error, like the reference's loop does at the top of the next iteration. */
int pdhg_take_steps_adaptive(pdhg_handle *h, int64_t n_steps, double reduction_exponent, double growth_exponent,
                             double *step_size_io, double primal_weight, int64_t *total_number_iterations_io,
                             double *cumulative_kkt_passes_io, int *numerical_error_out, int64_t *steps_done_out) {
  RoctxRange roctx_range("pdhg_take_steps_adaptive");
  if (!steps_done_out) return fail(-1, "null argument");
  if (n_steps < 0) return fail(-2, "pdhg_take_steps_adaptive: n_steps < 0");
  *steps_done_out = 0;
  if (!h || !step_size_io || !total_number_iterations_io || !cumulative_kkt_passes_io || !numerical_error_out)
    return fail(-1, "null argument");
  *numerical_error_out = 0;
  const bool device_loop = device_loop_for(h);
  int64_t s = 0;
  while (s < n_steps) {
    double entry = 0.0;         // nonzero: a multi-step kernel ended inside a take_step (its table of powers ran out)
    if (n_steps - s >= 2 && !h->grp && check_handle(h) == 0 && small_lp_eligible(h)) {
      // a small LP: the batch in one workgroup with the vectors in LDS (small_lp_kernel.hpp)
      int64_t k = 0;
      const int rc = small_lp_steps(h, n_steps - s, reduction_exponent, growth_exponent, step_size_io, primal_weight,
                                    total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out, &k, &entry);
      if (rc != 0 && rc != 1) return rc;
      if (rc == 0) {
        s += k;
        *steps_done_out = s;
        if (*numerical_error_out) break;
        if (k > 0 && entry == 0.0) continue;
      }
    }
    // (entry != 0: the small-LP launch above ended inside a take_step -- its step size on entry must reach the accept
    //  of THAT take_step, so it is finished launch by launch below, never handed to a fresh multi-step launch)
    if (entry == 0.0 && device_loop && n_steps - s >= 2 && !h->grp && !h->profile && check_handle(h) == 0) {
      int64_t k = 0;
      const int rc = coop_steps(h, n_steps - s, reduction_exponent, growth_exponent, step_size_io, primal_weight,
                                total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out, &k, &entry);
      if (rc != 0 && rc != 1) return rc;
      if (rc == 0) {
        s += k;
        *steps_done_out = s;
        if (*numerical_error_out) break;
        if (k > 0 && entry == 0.0) continue;   // (k == 0: trial budget spent on rejections, or a time-out: take the next step singly)
      }
    }
    if (s >= n_steps) break;
    // one take_step, launch by launch -- or the rest of one that a multi-step kernel began (entry: its step size on entry)
    const int rc = take_step_adaptive_from(h, reduction_exponent, growth_exponent, step_size_io,
                                           entry != 0.0 ? entry : *step_size_io, primal_weight,
                                           total_number_iterations_io, cumulative_kkt_passes_io, numerical_error_out);
    if (rc) return rc;
    *steps_done_out = ++s;
    if (*numerical_error_out) break;
  }
  return 0;
}

int pdhg_add_current_primal_to_average(pdhg_handle *h0, double weight) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(accept_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, 0, weight,
                       h->x + o, h->sum_x + o, h->y, h->sum_y);
    HIP_TRY(hipGetLastError());
    h->sum_x_count += 1;
    h->sum_x_weights += weight;
  }
  return 0;
}

int pdhg_get_average_info(pdhg_handle *h, int64_t counts[2], double weights[2]) {
  if (!h) return fail(-1, "null handle");
  counts[0] = h->sum_x_count; counts[1] = h->sum_y_count;
  weights[0] = h->sum_x_weights; weights[1] = h->sum_y_weights;
  return 0;
}

int pdhg_get_average(pdhg_handle *h0, double *x_avg, double *y_avg) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  if (x_avg) {
    FOR_SHARDS(L, h) {
      const int64_t o = h->clo;
      hipLaunchKernelGGL(div_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, h->sum_x + o,
                         h->sum_x_weights, h->tmp_n + o);
      HIP_TRY(hipGetLastError());
    }
    if ((rc = cols_to_host(L, [](pdhg_handle *s) { return s->tmp_n; }, x_avg))) return rc;
  }
  if (y_avg) {
    FOR_SHARDS(L, h) {
      hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->tmp_m);
      HIP_TRY(hipGetLastError());
    }
    if ((rc = rows_to_host(L, [](pdhg_handle *s) { return s->tmp_m; }, y_avg))) return rc;
  }
  return sync_all(L);
}

int pdhg_reset_average(pdhg_handle *h0) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  bump_version(L);
  FOR_SHARDS(L, h) {
    HIP_TRY(hipMemsetAsync(h->sum_x, 0, sizeof(double) * (size_t)std::max<int64_t>(h->n, 1), h->stream));
    HIP_TRY(hipMemsetAsync(h->sum_y, 0, sizeof(double) * (size_t)std::max<int64_t>(h->m, 1), h->stream));
    h->pend_x = h->pend_y = false;          // a deferred update belongs to the sums being discarded
    h->sum_x_count = h->sum_y_count = 0;
    h->sum_x_weights = h->sum_y_weights = 0.0;
  }
  return 0;
}

// after x (owned slices) changed outside a trial: QP groups keep x full on every shard
static int refresh_full_x(const Shards &L) {
  if (!L.g || !L.p[0]->has_q) return 0;
  return dist_all_gather(*L.g, [](pdhg_handle *s) { return s->x; }, L.g->S);
}

int pdhg_restart_to_average(pdhg_handle *h0) {
  RoctxRange roctx_range("pdhg_restart_to_average");
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  if (h0->sum_x_count == 0 || h0->sum_y_count == 0) return fail(-1, "average is empty");
  FOR_SHARDS(L, h) {
    const int64_t o = h->clo;
    hipLaunchKernelGGL(div_kernel, dim3(ew_grid(h->cn)), dim3(TPB), 0, h->stream, (int)h->cn, h->sum_x + o, h->sum_x_weights, h->x + o);
    hipLaunchKernelGGL(div_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, h->sum_y, h->sum_y_weights, h->y);
    HIP_TRY(hipGetLastError());
  }
  if ((rc = refresh_full_x(L))) return rc;
  return dual_product(L, [](pdhg_handle *s) { return (const double *)s->y; }, [](pdhg_handle *s) { return s->aty; });
}

int pdhg_get_current(pdhg_handle *h0, double *x, double *y, double *aty) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if (x && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->x; }, x))) return rc;
  if (y && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->y; }, y))) return rc;
  if (aty && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->aty; }, aty))) return rc;
  return sync_all(L);
}

int pdhg_get_trial(pdhg_handle *h0, double *x_next, double *y_next, double *aty_next) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if (x_next && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->x_next; }, x_next))) return rc;
  if (y_next && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->y_next; }, y_next))) return rc;
  if (aty_next && (rc = cols_to_host(L, [](pdhg_handle *s) { return s->aty_next; }, aty_next))) return rc;
  return sync_all(L);
}

int pdhg_set_current(pdhg_handle *h0, const double *x, const double *y) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  if (x && (rc = cols_from_host(L, x, [](pdhg_handle *s) { return s->x; }))) return rc;
  if (y && (rc = rows_from_host(L, y, [](pdhg_handle *s) { return s->y; }))) return rc;
  if ((rc = dual_product(L, [](pdhg_handle *s) { return (const double *)s->y; }, [](pdhg_handle *s) { return s->aty; }))) return rc;
  return sync_all(L);
}

int pdhg_spmv(pdhg_handle *h0, const double *x, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!x || !out) return fail(-1, "null vector");
  const Shards L = shards_of(h0);
  if ((rc = cols_from_host(L, x, [](pdhg_handle *s) { return s->tmp_n; }))) return rc;
  FOR_SHARDS(L, h) {
    EpiArgs e{};
    e.out = h->tmp_m;
    if ((rc = launch_spmv<MODE_PLAIN, 0>(h, h->A, h->tmp_n, e))) return rc;
  }
  if ((rc = rows_to_host(L, [](pdhg_handle *s) { return s->tmp_m; }, out))) return rc;
  return sync_all(L);
}

int pdhg_spmv_t(pdhg_handle *h0, const double *y, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (!y || !out) return fail(-1, "null vector");
  const Shards L = shards_of(h0);
  if ((rc = rows_from_host(L, y, [](pdhg_handle *s) { return s->tmp_m; }))) return rc;
  // the partial products go through tmp_n (n_alloc long); a group's gather-to-host then uses dn_buf
  if ((rc = dual_product(L, [](pdhg_handle *s) { return (const double *)s->tmp_m; }, [](pdhg_handle *s) { return s->tmp_n; }))) return rc;
  if ((rc = cols_to_host(L, [](pdhg_handle *s) { return s->tmp_n; }, out))) return rc;
  return sync_all(L);
}

