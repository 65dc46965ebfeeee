// host_trial_graph_build.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// The trial step as a HIP graph, second part: eligibility, the nodes of a fused product (stream / sliced jagged / pipelined / slab
// passes / sweep + the long-row pair), building, parameter updates and launch (host side; the node helpers are in host_trial_graph.hpp).

bool graph_eligible(pdhg_handle *h) {
  if (h->graph_mode < 0) {
    const char *ev = getenv("PDHG_GRAPH");
    // stream layouts only.  The sweep can run as graph nodes too (PDHG_GRAPH_TILED=1) but gains nothing: with the
    // take_step loop in C the separate launches already overlap the kernels -- random 1M x 1M 5 709 it/s as a graph
    // against 5 637, 4M x 4M 1 667 / 1 672, config S 611 / 613 (profiles/r03_trial_kernel.txt).
    const bool tiled_ok = dev_env("PDHG_GRAPH_TILED") != nullptr;
    bool on = !h->grp && !h->has_q && h->n > 0 && h->A.segs.empty() && h->At.segs.empty() && (tiled_ok || (!h->A.tiled && !h->At.tiled));
    if (ev) on = on && ev[0] != '0';
    h->graph_mode = on ? 1 : 0;
  }
  return h->graph_mode == 1 && !h->has_q && !h->profile;
}

void graph_destroy(pdhg_handle::TrialGraph &G) {
  if (G.exec) (void)hipGraphExecDestroy(G.exec);
  if (G.graph) (void)hipGraphDestroy(G.graph);
  G = pdhg_handle::TrialGraph();
}

// the argument packs of the three nodes whose scalars change from trial to trial
struct GraphArgs {
  pdhg_handle *h;
  int n;
  dim3 primal_grid;
  EpiArgs dual_epi;
  GraphArgs(pdhg_handle *h_, double sigma) : h(h_), n((int)h_->n), primal_grid(ew_grid((h_->n + 1) / 2)) {
    dual_epi = EpiArgs{};
    dual_epi.y = h->y; dual_epi.b = h->b; dual_epi.y_next = h->y_next; dual_epi.sigma = sigma;
    dual_epi.num_eq = (int)h->num_eq; dual_epi.partials = h->pA; dual_epi.stride = h->A.slots(); dual_epi.lo_offset = h->A.slots();
    if (h->pend_y) { dual_epi.sum_y = h->sum_y; dual_epi.avg_w = h->pend_w; }
  }
};

// the sweep kernel of a layout for graph nodes: function pointer (with the dynamic-LDS opt-in done)
template <int MODE>
int tiled_node_func(pdhg_handle *h, const CsrDev &D, const void **fn, size_t *lds) {
  *lds = tiled_lds_bytes(D);
  *fn = D.tw_mode == 1   ? (const void *)spmv_tiled_kernel<MODE, 1>
        : D.tw_mode == 2 ? (const void *)spmv_tiled_kernel<MODE, 2>
        : D.tw_mode == 3 ? (const void *)spmv_tiled_kernel<MODE, 3>
        : D.tw_mode == 4 ? (const void *)spmv_tiled_kernel<MODE, 4>
                         : (const void *)spmv_tiled_kernel<MODE, 0>;
  return ensure_lds_limit(h, MODE, D.tw_mode, *lds, *fn);
}

// nodes of one fused SpMV: stream kernel (one node) or its column-slab passes (a chain),
// beside the long-row pair.  `done` receives the nodes the next stage must wait for;
// main_node / long_node (optional) receive the nodes that carry the epilogue's scalars.
template <int MODE, int TAG>
int graph_add_spmv(pdhg_handle *h, hipGraph_t graph, const CsrDev &D, const double *xin, const EpiArgs &e,
                   const std::vector<hipGraphNode_t> &deps, std::vector<hipGraphNode_t> &done,
                   hipGraphNode_t *main_node, hipGraphNode_t *long_node) {
  const int rm = h->remap ? 1 : 0, rx = h->relaxed ? 1 : 0;
  if (D.tiled) {
    if (D.grid > 0) {
      const void *fn;
      size_t lds;
      int rc = tiled_node_func<MODE>(h, D, &fn, &lds);
      if (rc) return rc;
      hipGraphNode_t nd = nullptr;
      const int per_xcd = tiled_per_xcd(h, D, D.grid);
      HIP_TRY(graph_add_kernel_lds(graph, &nd, deps, fn, dim3(per_xcd > 0 ? per_xcd * NUM_XCD : D.grid), dim3(TW_WPB * WAVE), lds,
                                   (const int2 *)D.wave_rows, (const int *)D.wave_ent, (const int *)D.wave_step_off,
                                   (const int *)D.step_tile, (const int *)D.wg_step_off, D.nwaves, D.tile_shift, D.tw_rows,
                                   (const unsigned *)D.pk, (const double *)D.tv, xin, e, D.grid, per_xcd));
      if (main_node) *main_node = nd;
      done.push_back(nd);
    }
  } else if (!D.slabs.empty()) {
    const int P = (int)D.slabs.size();
    std::vector<hipGraphNode_t> prev = deps;
    for (int p = 0; p < P; ++p) {
      const SlabDev &S = D.slabs[(size_t)p];
      hipGraphNode_t nd = nullptr;
      if (S.sj.on()) {
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        EpiArgs le = e;
        le.init = D.slab_partial;
        if (p + 1 < P) {
          const void *fn = p == 0 ? sj_kernel_fn<MODE_PLAIN, false, TAG>(S.sj) : sj_kernel_fn<MODE_PLAIN, true, TAG>(S.sj);
          HIP_TRY(graph_add_kernel(graph, &nd, prev, fn, dim3(S.sj.grid), dim3(TPB), sj_view(S.sj, rx), xin, rm, 0, pe));
        } else {
          HIP_TRY(graph_add_kernel(graph, &nd, prev, sj_kernel_fn<MODE, true, TAG>(S.sj), dim3(S.sj.grid), dim3(TPB),
                                   sj_view(S.sj, rx), xin, rm, S.grid, le));
          if (main_node) *main_node = nd;
        }
        prev.assign(1, nd);
        continue;
      }
      if (S.pipe_grid > 0) {
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        EpiArgs le = e;
        le.init = D.slab_partial;
        if (p + 1 < P) {
          const void *fn = p == 0 ? (const void *)spmv_stream_pipe_kernel<MODE_PLAIN, false, TAG> : (const void *)spmv_stream_pipe_kernel<MODE_PLAIN, true, TAG>;
          HIP_TRY(graph_add_kernel(graph, &nd, prev, fn, dim3(S.pipe_grid), dim3(TPB), S.view(D.rows), xin, (const int4 *)S.ext, S.nblk,
                                   S.per_xcd, rm, rx, 0, pe));
        } else {
          HIP_TRY(graph_add_kernel(graph, &nd, prev, (const void *)spmv_stream_pipe_kernel<MODE, true, TAG>, dim3(S.pipe_grid), dim3(TPB),
                                   S.view(D.rows), xin, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx, S.grid, le));
          if (main_node) *main_node = nd;
        }
        prev.assign(1, nd);
        continue;
      }
      if (p + 1 < P) {
        EpiArgs pe{};
        pe.out = D.slab_partial;
        pe.init = D.slab_partial;
        const void *fn = p == 0 ? (const void *)spmv_stream_kernel<MODE_PLAIN, false, TAG> : (const void *)spmv_stream_kernel<MODE_PLAIN, true, TAG>;
        HIP_TRY(graph_add_kernel(graph, &nd, prev, fn, dim3(S.grid), dim3(TPB), S.view(D.rows), xin, (const int2 *)S.blks, (const int4 *)S.ext,
                                 S.nblk, S.per_xcd, rm, rx | 2, pe));
      } else {
        EpiArgs le = e;
        le.init = D.slab_partial;
        HIP_TRY(graph_add_kernel(graph, &nd, prev, (const void *)spmv_stream_kernel<MODE, true, TAG>, dim3(S.grid), dim3(TPB),
                                 S.view(D.rows), xin, (const int2 *)S.blks, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx | 2, le));
        if (main_node) *main_node = nd;
      }
      prev.assign(1, nd);
    }
    done.push_back(prev[0]);
  } else if (D.sj.on()) {
    hipGraphNode_t nd = nullptr;
    HIP_TRY(graph_add_kernel(graph, &nd, deps, sj_kernel_fn<MODE, false, TAG>(D.sj), dim3(D.sj.grid), dim3(TPB),
                             sj_view(D.sj, rx), xin, rm, D.grid, e));
    if (main_node) *main_node = nd;
    done.push_back(nd);
  } else if (D.pipe_grid > 0) {
    hipGraphNode_t nd = nullptr;
    HIP_TRY(graph_add_kernel(graph, &nd, deps, (const void *)spmv_stream_pipe_kernel<MODE, false, TAG>, dim3(D.pipe_grid), dim3(TPB),
                             D.view(), xin, (const int4 *)D.ext, D.nblk, D.per_xcd, rm, rx, D.grid, e));
    if (main_node) *main_node = nd;
    done.push_back(nd);
  } else if (D.grid > 0) {
    hipGraphNode_t nd = nullptr;
    HIP_TRY(graph_add_kernel(graph, &nd, deps, (const void *)spmv_stream_kernel<MODE, false, TAG>, dim3(D.grid), dim3(TPB),
                             D.view(), xin, (const int2 *)D.blks, (const int4 *)D.ext, D.nblk, D.per_xcd, rm, rx, e));
    if (main_node) *main_node = nd;
    done.push_back(nd);
  }
  if (D.nlong > 0) {
    hipGraphNode_t part = nullptr, fin = nullptr;
    HIP_TRY(graph_add_kernel(graph, &part, deps, (const void *)spmv_long_partial_kernel<TAG>, dim3(D.nchunks), dim3(TPB),
                             D.view(), xin, (const int *)D.chunk_row, (const int *)D.chunk_off, D.chunk_partial));
    HIP_TRY(graph_add_kernel(graph, &fin, {part}, (const void *)spmv_long_final_kernel<MODE>, dim3(D.long_grid), dim3(TPB),
                             (const int *)D.long_row, (const int *)D.long_chunk_ptr, D.nlong,
                             (const double *)D.chunk_partial, e, D.grid));
    if (long_node) *long_node = fin;
    done.push_back(fin);
  }
  return 0;
}

// the dual stream node's parameters again, with a new sigma
int graph_set_dual(pdhg_handle *h, pdhg_handle::TrialGraph &G, const EpiArgs &dual_epi) {
  const CsrDev &A = h->A;
  const int rm = h->remap ? 1 : 0, rx = h->relaxed ? 1 : 0;
  if (G.n_dual) {
    if (A.tiled) {
      const void *fn;
      size_t lds;
      int rc = tiled_node_func<MODE_DUAL>(h, A, &fn, &lds);
      if (rc) return rc;
      const int per_xcd = tiled_per_xcd(h, A, A.grid);
      HIP_TRY(graph_set_kernel_lds(G.exec, G.n_dual, fn, dim3(per_xcd > 0 ? per_xcd * NUM_XCD : A.grid), dim3(TW_WPB * WAVE), lds,
                                   (const int2 *)A.wave_rows, (const int *)A.wave_ent, (const int *)A.wave_step_off,
                                   (const int *)A.step_tile, (const int *)A.wg_step_off, A.nwaves, A.tile_shift, A.tw_rows,
                                   (const unsigned *)A.pk, (const double *)A.tv, (const double *)h->xbar, dual_epi, A.grid, per_xcd));
    } else if (!A.slabs.empty()) {
      const SlabDev &S = A.slabs.back();
      EpiArgs le = dual_epi;
      le.init = A.slab_partial;
      if (S.sj.on())
        HIP_TRY(graph_set_kernel(G.exec, G.n_dual, sj_kernel_fn<MODE_DUAL, true, 0>(S.sj), dim3(S.sj.grid), dim3(TPB),
                                 sj_view(S.sj, rx), (const double *)h->xbar, rm, S.grid, le));
      else if (S.pipe_grid > 0)
        HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_pipe_kernel<MODE_DUAL, true, 0>, dim3(S.pipe_grid), dim3(TPB),
                                 S.view(A.rows), (const double *)h->xbar, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx, S.grid, le));
      else
        HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_kernel<MODE_DUAL, true, 0>, dim3(S.grid), dim3(TPB),
                                 S.view(A.rows), (const double *)h->xbar, (const int2 *)S.blks, (const int4 *)S.ext, S.nblk, S.per_xcd, rm, rx | 2, le));
    } else if (A.sj.on()) {
      HIP_TRY(graph_set_kernel(G.exec, G.n_dual, sj_kernel_fn<MODE_DUAL, false, 0>(A.sj), dim3(A.sj.grid), dim3(TPB),
                               sj_view(A.sj, rx), (const double *)h->xbar, rm, A.grid, dual_epi));
    } else if (A.pipe_grid > 0) {
      HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_pipe_kernel<MODE_DUAL, false, 0>, dim3(A.pipe_grid), dim3(TPB),
                               A.view(), (const double *)h->xbar, (const int4 *)A.ext, A.nblk, A.per_xcd, rm, rx, A.grid, dual_epi));
    } else {
      HIP_TRY(graph_set_kernel(G.exec, G.n_dual, (const void *)spmv_stream_kernel<MODE_DUAL, false, 0>, dim3(A.grid), dim3(TPB),
                               A.view(), (const double *)h->xbar, (const int2 *)A.blks, (const int4 *)A.ext, A.nblk, A.per_xcd, rm, rx, dual_epi));
    }
  }
  if (G.n_dual_long)
    HIP_TRY(graph_set_kernel(G.exec, G.n_dual_long, (const void *)spmv_long_final_kernel<MODE_DUAL>,
                             dim3(A.long_grid), dim3(TPB), (const int *)A.long_row, (const int *)A.long_chunk_ptr,
                             A.nlong, (const double *)A.chunk_partial, dual_epi, A.grid));
  return 0;
}

int graph_build(pdhg_handle *h, pdhg_handle::TrialGraph &G, double tau, double theta, double sigma) {
  graph_destroy(G);
  int rcw = ensure_result_word(h);
  if (rcw) return rcw;
  HIP_TRY(hipGraphCreate(&G.graph, 0));
  GraphArgs a(h, sigma);
  const double *nullq = nullptr;
  // K1+K2
  HIP_TRY(graph_add_kernel(G.graph, &G.n_primal, {}, (const void *)primal_kernel<false, true>, a.primal_grid, dim3(TPB),
                           a.n, (const double *)h->x, (const double *)h->c, (const double *)h->aty, nullq,
                           (const double *)h->lb, (const double *)h->ub, tau, theta, h->x_next, h->xbar,
                           h->pend_w, h->pend_x ? h->sum_x : (double *)nullptr));
  // K3+K4 on CSR(A), K5+K6 on CSR(A'): the stream kernel (or its column-slab passes, a
  // chain) and the long-row pair are independent branches
  std::vector<hipGraphNode_t> dual_done, aty_done;
  {
    int rc = graph_add_spmv<MODE_DUAL, 0>(h, G.graph, h->A, h->xbar, a.dual_epi, {G.n_primal}, dual_done, &G.n_dual, &G.n_dual_long);
    if (rc) return rc;
  }
  if (dual_done.empty()) dual_done.push_back(G.n_primal);
  const CsrDev &A = h->A;
  const CsrDev &T = h->At;
  EpiArgs te{};
  te.x = h->x; te.x_next = h->x_next; te.aty = h->aty; te.aty_next = h->aty_next;
  te.partials = h->pAt; te.stride = h->pAt_stride; te.lo_offset = 3 * h->pAt_stride;
  {
    int rc = graph_add_spmv<MODE_ATY, 1>(h, G.graph, T, h->y_next, te, dual_done, aty_done, nullptr, nullptr);
    if (rc) return rc;
  }
  if (aty_done.empty()) aty_done = dual_done;
  // K6b -> pinned host memory + sequence number
  FinalSpec sp{};
  sp.ptr[0] = h->pAt;                       sp.count[0] = T.slots();
  sp.ptr[1] = h->pAt + h->pAt_stride;       sp.count[1] = T.slots();
  sp.ptr[2] = h->pA;                        sp.count[2] = A.slots();
  sp.ptr[3] = h->pAt + 2 * h->pAt_stride;   sp.count[3] = T.slots();
  sp.ptr[4] = h->pQ;                        sp.count[4] = 0;
  for (int q : {0, 1, 3}) sp.ptr_lo[q] = sp.ptr[q] + 3 * h->pAt_stride;
  sp.ptr_lo[2] = h->pA + A.slots();
  sp.ptr_lo[4] = h->pQ + h->ew_grid_n;
  sp.out = nullptr;
  hipGraphNode_t fin = nullptr;
  HIP_TRY(graph_add_kernel(G.graph, &fin, aty_done, (const void *)final_reduce_host_kernel, dim3(1), dim3(FINAL_TPB), sp,
                           h->seq_dev, h->res_host));
  HIP_TRY(hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0));
  G.x = h->x; G.y = h->y; G.aty = h->aty;
  G.tau = tau; G.theta = theta; G.sigma = sigma;
  G.add_x = h->pend_x; G.add_wx = h->pend_w;
  G.add_y = h->pend_y; G.add_wy = h->pend_w;
  return 0;
}

int graph_trial(pdhg_handle *h, double step_size, double primal_weight, double theta, double out[5]) {
  const double tau = step_size / primal_weight, sigma = primal_weight * step_size;
  pdhg_handle::TrialGraph *G = nullptr;
  for (int k = 0; k < 2; ++k)
    if (h->tgraph[k].exec && h->tgraph[k].x == h->x && h->tgraph[k].y == h->y && h->tgraph[k].aty == h->aty)
      G = &h->tgraph[k];
  if (!G) {
    G = !h->tgraph[0].exec ? &h->tgraph[0] : (!h->tgraph[1].exec ? &h->tgraph[1] : &h->tgraph[0]);
    int rc = graph_build(h, *G, tau, theta, sigma);
    if (rc) return rc;
  } else {
    const auto c0 = std::chrono::steady_clock::now();
    GraphArgs a(h, sigma);
    if (G->tau != tau || G->theta != theta || G->add_x != h->pend_x || (h->pend_x && G->add_wx != h->pend_w)) {
      const double *nullq = nullptr;
      HIP_TRY(graph_set_kernel(G->exec, G->n_primal, (const void *)primal_kernel<false, true>, a.primal_grid, dim3(TPB),
                               a.n, (const double *)h->x, (const double *)h->c, (const double *)h->aty, nullq,
                               (const double *)h->lb, (const double *)h->ub, tau, theta, h->x_next, h->xbar,
                               h->pend_w, h->pend_x ? h->sum_x : (double *)nullptr));
      G->tau = tau; G->theta = theta;
      G->add_x = h->pend_x; G->add_wx = h->pend_w;
    }
    if (G->sigma != sigma || G->add_y != h->pend_y || (h->pend_y && G->add_wy != h->pend_w)) {
      int rc = graph_set_dual(h, *G, a.dual_epi);
      if (rc) return rc;
      G->sigma = sigma;
      G->add_y = h->pend_y; G->add_wy = h->pend_w;
    }
    h->t_set += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
  }
  h->seq_expected += 1;
  const auto c1 = std::chrono::steady_clock::now();
  HIP_TRY(hipGraphLaunch(G->exec, h->stream));
#ifdef PDHG_PROBE_GRAPH_TWICE      // latency probe only (the deferred sums are applied twice): the boundary between two queued graphs, tools/archive/r6_w.sh
  HIP_TRY(hipGraphLaunch(G->exec, h->stream));
  h->seq_expected += 1;
#endif
  const auto c2 = std::chrono::steady_clock::now();
  h->t_launch += std::chrono::duration<double>(c2 - c1).count();
  h->n_graph_trials += 1;
  h->pend_x = h->pend_y = false;     // the launch carries the deferred average update
  int rcw = wait_result_word(h, out);
  h->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - c2).count();
  return rcw;
}

// queues_work: the entry point may put work on the shards' own streams (everything but a trial step and a lazy accept)
int check_handle(pdhg_handle *h, bool queues_work = true) {
  if (!h) return fail(-1, "null handle");
  if (queues_work && h->grp && !h->grp->coop_dev.empty()) {
    DistGroup &g = *h->grp;
    if (g.join_pending) {          // the members' streams wait for the last persistent group launch (group_kernel.hpp)
      for (GroupDevLaunch &D : g.coop_dev) {
        HIP_TRY(hipSetDevice(D.device));
        for (int i : D.members)
          if (g.sh[(size_t)i]->stream != D.stream) HIP_TRY(hipStreamWaitEvent(g.sh[(size_t)i]->stream, D.ev_done, 0));
      }
      g.join_pending = false;
    }
    g.members_dirty = true;
  }
  HIP_TRY(hipSetDevice(h->device));
  return 0;
}

int sync_all(const Shards &L) {
  FOR_SHARDS(L, s) HIP_TRY(hipStreamSynchronize(s->stream));
  return 0;
}

// Settle a deferred K7 (lazy accept): sum_x += w * x and / or sum_y += w * y on the
// iterate that is current now.  Called by every entry point other than the trial itself.
int flush_pending(const Shards &L) {
  FOR_SHARDS(L, h) {
    if (!h->pend_x && !h->pend_y) continue;
    ProfScope ps(h, PDHG_K_ACCEPT);
    const int64_t o = h->clo;
    const int nn = h->pend_x ? (int)h->cn : 0, mm = h->pend_y ? (int)h->m : 0;
    hipLaunchKernelGGL(accept_kernel, dim3(ew_grid(std::max<int64_t>(std::max(nn, mm), 1))), dim3(TPB), 0, h->stream, nn, mm,
                       h->pend_w, h->x + o, h->sum_x + o, h->y, h->sum_y);
    HIP_TRY(hipGetLastError());
    h->pend_x = h->pend_y = false;
  }
  return 0;
}

void bump_version(const Shards &L) {   // x, y, the running sums or A change: cached A*x / A'*y are stale
  for (int i = 0; i < L.count; ++i) L.p[i]->state_version += 1;
}
