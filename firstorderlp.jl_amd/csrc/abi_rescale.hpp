// abi_rescale.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// C ABI: rescaling on the device (N2).

// ---- rescaling on the device (N2) --------------------------------------------

static int row_grid(int rows) { return std::max(1, (rows + (TPB / WAVE) - 1) / (TPB / WAVE)); }

// scratch vectors of one pdhg_rescale call, per shard: row factors have the
// shard's m entries, column factors all n (n_alloc: they are reduced over ranks)
struct RescaleTmp {
  double *ev = nullptr, *dv = nullptr, *inv_e = nullptr, *inv_d = nullptr, *cum_e = nullptr, *cum_d = nullptr;
  double *tmp_e = nullptr, *tmp_d = nullptr;
};

// a row statistic of one resident CSR: short rows one wave each, long rows chunk by chunk
extern "C++" {
template <int OP>
static void launch_row_op(pdhg_handle *h, const CsrDev &D, int cols, double pexp, const double *inv_scale, double *out) {
  if (!D.segs.empty()) {        // row segments (layout.hpp): the statistic is per row, segment by segment
    for (const CsrDev &S : D.segs) launch_row_op<OP>(h, S, cols, pexp, inv_scale ? inv_scale + S.row0 : inv_scale, out + S.row0);
    return;
  }
  hipLaunchKernelGGL(row_op_kernel<OP>, dim3(row_grid(D.rows)), dim3(TPB), 0, h->stream, D.view(), cols, pexp, inv_scale, out,
                     D.long_thr);
  if (D.nlong > 0) {
    hipLaunchKernelGGL(row_op_long_partial_kernel<OP>, dim3(D.nchunks), dim3(TPB), 0, h->stream, D.view(),
                       (const int *)D.chunk_row, (const int *)D.chunk_off, pexp, inv_scale, D.chunk_partial);
    hipLaunchKernelGGL(row_op_long_final_kernel<OP>, dim3(D.long_grid), dim3(TPB), 0, h->stream, D.view(),
                       (const int *)D.long_row, (const int *)D.long_chunk_ptr, D.nlong,
                       (const double *)D.chunk_partial, cols, pexp, out);
  }
}
}  // extern "C++"

// one scale_problem step on every resident layout + the vectors of one shard
static int apply_scaling(pdhg_handle *h, RescaleTmp &t) {
  const int n = (int)h->n, m = (int)h->m;
  hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev, t.inv_e, 0);
  hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv, t.inv_d, 0);
  // every resident copy of one matrix (k = 0: CSR(A), rows -> E; k = 1: CSR(A'), rows -> D); a row segment takes the
  // row-indexed factor from its first row on
  std::function<void(CsrDev &, const double *, const double *, int)> scale_one =
      [&](CsrDev &D, const double *inv_e, const double *inv_d, int k) {
    for (CsrDev &S : D.segs) scale_one(S, k == 0 ? inv_e + S.row0 : inv_e, k == 1 ? inv_d + S.row0 : inv_d, k);
    if (!D.segs.empty() || D.nnz == 0) return;
    hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(D.rows)), dim3(TPB), 0, h->stream, D.rows, D.rowptr,
                       D.col, D.val, inv_e, inv_d, k, D.long_thr);
    if (D.nlong > 0)
      hipLaunchKernelGGL(scale_long_kernel, dim3(D.nchunks), dim3(TPB), 0, h->stream, (const int *)D.rowptr,
                         (const int *)D.col, D.val, (const int *)D.chunk_row, (const int *)D.chunk_off,
                         inv_e, inv_d, k);
    if (D.tiled && D.nwaves > 0)
      hipLaunchKernelGGL(scale_tiled_kernel, dim3(row_grid(D.nwaves)), dim3(TPB), 0, h->stream, D.wave_rows,
                         D.wave_ent, D.wave_step_off, D.step_tile, D.wg_step_off, D.nwaves, D.tile_shift,
                         D.pk, D.tv, inv_e, inv_d, k);
    for (const SlabDev &S : D.slabs)
      if (S.nnz > 0)
        hipLaunchKernelGGL(scale_csr_kernel, dim3(row_grid(D.rows)), dim3(TPB), 0, h->stream, D.rows, S.rowptr,
                           S.col, S.val, inv_e, inv_d, k, 0);
    // the sliced jagged copies (sj_kernels.hpp) hold the same entries in another order: copied again from the CSR arrays
    // just scaled, so that they carry the same bits (two multiplications in a fixed order per entry, done once)
    auto refill = [&](const SjDev &J, const int *rowptr, const int *col, const double *val) {
      if (J.on() && J.nnz > 0)
        hipLaunchKernelGGL(sj_fill_kernel, dim3((J.nslices + TPB / WAVE - 1) / (TPB / WAVE)), dim3(TPB), 0, h->stream, J.nslices, (TPB / WAVE) * J.G,
                           (const unsigned *)J.meta, (const int *)J.slice_off, rowptr, col, val, J.col, J.val);
    };
    refill(D.sj, D.rowptr, D.col, D.val);
    for (const SlabDev &S : D.slabs) refill(S.sj, S.rowptr, S.col, S.val);
  };
  scale_one(h->A, t.inv_e, t.inv_d, 0);
  scale_one(h->At, t.inv_e, t.inv_d, 1);
  if (!h->Achunk.empty() && h->grp) {
    // the column-chunk layouts of a shard group (dist.hpp) index their columns INTO THE CHUNK: the column factors in chunk
    // layout (xchunk is free between trials: the next trial packs xbar into it again)
    DistGroup &g = *h->grp;
    int rc2 = launch_chunk_pack(g, h, t.inv_d, h->xchunk, 0, g.world, h->n, h->stream);
    if (rc2) return rc2;
    const int64_t W = (int64_t)g.world * g.ag_sub;
    for (size_t c = 0; c < h->Achunk.size(); ++c) scale_one(h->Achunk[c], t.inv_e, h->xchunk + (int64_t)c * W, 0);
  }
  if (h->has_q) {
    // objective_matrix = (D^-1 Q) D^-1 (preprocess.jl:562-564); Qt holds Q' entry by entry, so the
    // "transposed" order reproduces the same two roundings on it
    // Both through scale_one like A and A': every resident copy of Q and Q' -- CSR, long-row chunks, column slabs, a
    // tiled copy, and the sliced jagged copies (refilled from the scaled CSR arrays; round 5 scaled only the CSR and slab
    // arrays here, so a QP whose Q ran spmv_sj_kernel multiplied by the UNSCALED Hessian after pdhg_rescale).
    scale_one(h->Q, t.inv_d, t.inv_d, 0);
    scale_one(h->Qt, t.inv_d, t.inv_d, 1);
  }
  hipLaunchKernelGGL(resc_apply_vectors_kernel, dim3(h->ew_grid_nm), dim3(TPB), 0, h->stream, n, m, t.dv, t.ev,
                     h->c, h->lb, h->ub, h->b, t.cum_d, t.cum_e);
  HIP_TRY(hipGetLastError());
  return 0;
}

int pdhg_rescale(pdhg_handle *h0, int l_inf_ruiz_iterations, int l2_norm_rescaling,
                 int use_pock_chambolle, double pock_chambolle_alpha,
                 double *constraint_rescaling_out, double *variable_rescaling_out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  if (use_pock_chambolle && !(pock_chambolle_alpha >= 0.0 && pock_chambolle_alpha <= 2.0))
    return fail(-1, "pock_chambolle_alpha must be in [0, 2]");
  const Shards L = shards_of(h0);
  if ((rc = flush_pending(L))) return rc;
  bump_version(L);
  for (int i = 0; i < L.count; ++i) L.p[i]->matrix_version += 1;
  std::vector<RescaleTmp> T((size_t)L.count);
  auto cleanup = [&]() {
    for (int i = 0; i < L.count; ++i) {
      (void)hipSetDevice(L.p[i]->device);
      RescaleTmp &t = T[(size_t)i];
      for (double *p : {t.ev, t.dv, t.inv_e, t.inv_d, t.cum_e, t.cum_d, t.tmp_e, t.tmp_d}) if (p) (void)hipFree(p);
    }
  };
#define RS(expr) do { int _r = (expr); if (_r) { cleanup(); return _r; } } while (0)
#define EACH(h, t) for (int _i = 0; _i < L.count; ++_i) if (pdhg_handle *h = L.p[_i]) \
    if (hipError_t _sde = hipSetDevice(h->device); _sde != hipSuccess) { cleanup(); return fail_hip(_sde, "hipSetDevice (rescale)"); } \
    else if (RescaleTmp *_tp = &T[(size_t)_i]) if (RescaleTmp &t = *_tp; true)
  EACH(h, t) {
    RS(alloc_zero(&t.ev, h->m)); RS(alloc_zero(&t.dv, h->n_alloc)); RS(alloc_zero(&t.inv_e, h->m)); RS(alloc_zero(&t.inv_d, h->n));
    RS(alloc_zero(&t.cum_e, h->m)); RS(alloc_zero(&t.cum_d, h->n)); RS(alloc_zero(&t.tmp_e, h->m)); RS(alloc_zero(&t.tmp_d, h->n_alloc));
    hipLaunchKernelGGL(fill_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, (int)h->m, 1.0, t.cum_e);
    hipLaunchKernelGGL(fill_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, (int)h->n, 1.0, t.cum_d);
  }
  // Column statistics of A are reductions over the row shards: every shard reduces its
  // rows, then max / sum over ranks (reduce-scatter + all-gather: the same bits everywhere).
  // Row statistics are complete on the shard that owns the row.
  auto reduce_cols = [&](bool use_tmp, bool maxop) -> int {
    if (!L.g) return 0;
    std::vector<double *> ptr((size_t)L.g->world, nullptr);
    for (int i = 0; i < L.count; ++i) ptr[(size_t)L.p[i]->rank] = use_tmp ? T[(size_t)i].tmp_d : T[(size_t)i].dv;
    return dist_all_reduce(*L.g, [&](pdhg_handle *s) { return ptr[(size_t)s->rank]; }, L.g->S, maxop);
  };
  // ruiz_rescaling, p = Inf (preprocess.jl:412-477): sqrt of the row / column max |a|, zeros -> 1
  for (int it = 0; it < l_inf_ruiz_iterations; ++it) {
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      launch_row_op<ROP_MAXABS>(h, h->At, m, 0.0, (const double *)nullptr, t.dv);
      launch_row_op<ROP_MAXABS>(h, h->A, n, 0.0, (const double *)nullptr, t.ev);
    }
    RS(reduce_cols(false, true));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      if (h->has_q) {   // QP: column max over the constraint AND the objective matrix (preprocess.jl:425-433)
        launch_row_op<ROP_MAXABS>(h, h->Qt, n, 0.0, (const double *)nullptr, t.tmp_d);
        hipLaunchKernelGGL(resc_max_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv, t.tmp_d);
      }
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv);
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev);
      RS(apply_scaling(h, t));
    }
  }
  // l2_norm_rescaling (preprocess.jl:358-372): sqrt of the row / column L2 norms, zeros -> 1
  if (l2_norm_rescaling) {
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      launch_row_op<ROP_MAXABS>(h, h->At, m, 0.0, (const double *)nullptr, t.tmp_d);
      launch_row_op<ROP_MAXABS>(h, h->A, n, 0.0, (const double *)nullptr, t.tmp_e);
    }
    RS(reduce_cols(true, true));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.tmp_d, t.inv_d, 1);
      hipLaunchKernelGGL(resc_zero_to_one_inv_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.tmp_e, t.inv_e, 1);
      launch_row_op<ROP_SUMSQ_SCALED>(h, h->At, m, 0.0, t.inv_d, t.dv);
      launch_row_op<ROP_SUMSQ_SCALED>(h, h->A, n, 0.0, t.inv_e, t.ev);
    }
    RS(reduce_cols(false, false));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      hipLaunchKernelGGL(resc_l2norm_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.tmp_d, t.dv);
      hipLaunchKernelGGL(resc_l2norm_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.tmp_e, t.ev);
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv);   // norm 0 -> sqrt 0 -> 1
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev);
      RS(apply_scaling(h, t));
    }
  }
  // pock_chambolle_rescaling (preprocess.jl:508-539)
  if (use_pock_chambolle) {
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      launch_row_op<ROP_SUMPOW>(h, h->At, m, 2.0 - pock_chambolle_alpha, (const double *)nullptr, t.dv);
      launch_row_op<ROP_SUMPOW>(h, h->A, n, pock_chambolle_alpha, (const double *)nullptr, t.ev);
    }
    RS(reduce_cols(false, false));
    EACH(h, t) {
      const int n = (int)h->n, m = (int)h->m;
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_n), dim3(TPB), 0, h->stream, n, t.dv);
      hipLaunchKernelGGL(resc_sqrt_kernel, dim3(h->ew_grid_m), dim3(TPB), 0, h->stream, m, t.ev);
      RS(apply_scaling(h, t));
    }
  }
  EACH(h, t) {
    (void)t;
    hipError_t e1 = hipGetLastError();
    if (e1 != hipSuccess) { cleanup(); return fail((int)e1, hipGetErrorString(e1)); }
  }
  if (constraint_rescaling_out && h0->m_global > 0) {
    std::vector<double *> ptr((size_t)h0->world, nullptr);
    for (int i = 0; i < L.count; ++i) ptr[(size_t)L.p[i]->rank] = T[(size_t)i].cum_e;
    RS(rows_to_host(L, [&](pdhg_handle *s) { return ptr[(size_t)s->rank]; }, constraint_rescaling_out));
  }
  if (variable_rescaling_out && h0->n > 0) {
    (void)hipSetDevice(h0->device);
    (void)hipMemcpyAsync(variable_rescaling_out, T[0].cum_d, sizeof(double) * (size_t)h0->n, hipMemcpyDeviceToHost, h0->stream);
  }
  rc = sync_all(L);
  cleanup();
#undef RS
#undef EACH
  return rc;
}

int pdhg_get_problem_vectors(pdhg_handle *h0, double *c, double *b, double *lb, double *ub) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  pdhg_handle *h = h0;   // column vectors of the problem are stored in full on every shard
  if (c) HIP_TRY(hipMemcpyAsync(c, h->c, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (lb) HIP_TRY(hipMemcpyAsync(lb, h->lb, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (ub) HIP_TRY(hipMemcpyAsync(ub, h->ub, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (b && (rc = rows_to_host(L, [](pdhg_handle *s) { return s->b; }, b))) return rc;
  return sync_all(L);
}

int pdhg_matrix_max_abs(pdhg_handle *h0, double *out) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  FOR_SHARDS(L, h) {
    if ((rc = ev_alloc(h))) return rc;
    if (!h->At.segs.empty()) {            // row segments: the max over the segments' maxima (single handle: L is this one)
      double best = 0.0;
      for (const CsrDev &S : h->At.segs) {
        hipLaunchKernelGGL(maxabs_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int64_t)S.nnz, S.val, h->ev_partials, h->ev_grid);
        HIP_TRY(hipGetLastError());
        double part = 0.0;
        if ((rc = ev_finish(L, 0, 1, &part))) return rc;
        best = std::max(best, part);
      }
      *out = best;
      return 0;
    }
    hipLaunchKernelGGL(maxabs_kernel, dim3(h->ev_grid), dim3(TPB), 0, h->stream, (int64_t)h->At.nnz, h->At.val,
                       h->ev_partials, h->ev_grid);
    HIP_TRY(hipGetLastError());
  }
  return ev_finish(L, 0, 1, out);
}

