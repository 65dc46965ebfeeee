// abi_dist.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// C ABI: row-partitioned multi-GPU handles (pdhg_create_multi / _dist / _dist_rows, partition, info).

// ---- row-partitioned multi-GPU handles -------------------------------------------

int pdhg_dist_get_unique_id(void *id) {
  if (!id) return fail(-1, "id == NULL");
  static_assert(sizeof(ncclUniqueId) <= PDHG_UNIQUE_ID_BYTES, "unique id does not fit the ABI's buffer");
  RCCL_API(R);
  ncclUniqueId u;
  NCCL_TRY(R->GetUniqueId(&u));
  memset(id, 0, PDHG_UNIQUE_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return 0;
}

int pdhg_create_dist(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                     const int64_t *colptr, const int64_t *rowval, const double *nzval,
                     int index_base, const double *c, const double *b, const double *lb,
                     const double *ub, int64_t num_equalities, int device_id, void *stream,
                     const void *unique_id, int rank, int world) {
  if (!out) return fail(-1, "out == NULL");
  *out = nullptr;
  if (!unique_id) return fail(-1, "unique_id == NULL");
  if (rank < 0 || rank >= world) return fail(-1, "rank out of range");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m) return fail(-1, "num_equalities out of range");
  if (!colptr || (nnz > 0 && (!rowval || !nzval))) return fail(-1, "null input array");
  if (colptr[0] != index_base || colptr[n] - index_base != nnz) return fail(-1, "colptr does not match nnz / index_base");
  DistGroup *g = new DistGroup();
  int rc = init_group_geometry(g, m, n, colptr, rowval, index_base, num_equalities, world);
  if (rc) { delete g; return rc; }
  g->backend = COMM_RCCL;
  choose_exchange_pattern(g);
  pdhg_handle *s = nullptr;
  rc = create_rank_shard(g, rank, n, colptr, rowval, nzval, index_base, c, b, lb, ub, device_id, stream, &s);
  if (rc) { delete g; return rc; }
  g->sh.push_back(s);
  g->comm.assign(1, nullptr);
  ncclUniqueId u;
  memcpy(&u, unique_id, sizeof(u));
  const RcclApi *R = rccl();
  if (!R) { destroy_group(g); return 2999; }
  ncclResult_t nr = R->CommInitRank(&g->comm[0], world, u, rank);
  if (nr != ncclSuccess) {
    g_last_error = std::string("ncclCommInitRank: ") + R->GetErrorString(nr);
    destroy_group(g);
    return 2000 + (int)nr;
  }
  *out = s;
  return 0;
}

int pdhg_partition_rows(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int index_base,
                        int world, int64_t *row_bounds) {
  if (!colptr || !row_bounds || m < 0 || n < 0) return fail(-1, "null / negative argument");
  if (world < 1 || world > DIST_MAX_WORLD) return fail(-1, "world size out of range (1..64)");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (colptr[0] != index_base) return fail(-1, "colptr[0] != index_base");
  if (colptr[n] - index_base > 0 && !rowval) return fail(-1, "null input array");
  std::vector<int64_t> b;
  partition_rows_by_nnz(m, n, colptr, rowval, index_base, world, b);
  for (int p = 0; p <= world; ++p) row_bounds[p] = b[(size_t)p];
  return 0;
}

int pdhg_create_dist_rows(pdhg_handle **out, int64_t m_global, int64_t n, const int64_t *row_bounds,
                          int64_t local_nnz, const int64_t *colptr, const int64_t *rowval, const double *nzval,
                          int index_base, const double *c, const double *b_local, const double *lb,
                          const double *ub, int64_t num_equalities, int device_id, void *stream,
                          const void *unique_id, int rank, int world) {
  if (!out) return fail(-1, "out == NULL");
  *out = nullptr;
  if (!unique_id) return fail(-1, "unique_id == NULL");
  if (rank < 0 || rank >= world) return fail(-1, "rank out of range");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m_global) return fail(-1, "num_equalities out of range");
  if (!row_bounds || !colptr || (local_nnz > 0 && (!rowval || !nzval))) return fail(-1, "null input array");
  if (colptr[0] != index_base || colptr[n] - index_base != local_nnz) return fail(-1, "colptr does not match local_nnz / index_base");
  DistGroup *g = new DistGroup();
  int rc = init_group_geometry(g, m_global, n, nullptr, nullptr, index_base, num_equalities, world, row_bounds);
  if (rc) { delete g; return rc; }
  g->backend = COMM_RCCL;
  choose_exchange_pattern(g);
  pdhg_handle *s = nullptr;
  rc = create_rank_shard_local(g, rank, n, colptr, rowval, nzval, index_base, c, b_local, lb, ub, device_id, stream, &s);
  if (rc) { delete g; return rc; }
  g->sh.push_back(s);
  g->comm.assign(1, nullptr);
  ncclUniqueId u;
  memcpy(&u, unique_id, sizeof(u));
  const RcclApi *R = rccl();
  if (!R) { destroy_group(g); return 2999; }
  ncclResult_t nr = R->CommInitRank(&g->comm[0], world, u, rank);
  if (nr != ncclSuccess) {
    g_last_error = std::string("ncclCommInitRank: ") + R->GetErrorString(nr);
    destroy_group(g);
    return 2000 + (int)nr;
  }
  *out = s;
  return 0;
}

int pdhg_rccl_info(int *compiled_version, int *runtime_version, char *path, int path_len) {
  if (compiled_version) *compiled_version = NCCL_VERSION_CODE;
  if (runtime_version) *runtime_version = 0;
  if (path && path_len > 0) path[0] = 0;
  RcclLoader &L = rccl_loader();
  if (runtime_version) *runtime_version = L.api.runtime_version;
  if (path && path_len > 0) snprintf(path, (size_t)path_len, "%s", L.api.path.c_str());
  if (!L.ok) { g_last_error = L.api.error; return 2999; }
  return 0;
}

int pdhg_host_issue_stats(pdhg_handle *h, int64_t *trials, double *issue_seconds, double *wait_seconds) {
  if (!h || !trials || !issue_seconds || !wait_seconds) return fail(-1, "null argument");
  if (h->grp) { *trials = h->grp->n_trials; *issue_seconds = h->grp->t_issue; *wait_seconds = h->grp->t_wait; }
  else { *trials = h->n_graph_trials; *issue_seconds = h->t_set + h->t_launch; *wait_seconds = h->t_wait; }
  return 0;
}

int pdhg_create_multi(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                      const int64_t *colptr, const int64_t *rowval, const double *nzval,
                      int index_base, const double *c, const double *b, const double *lb,
                      const double *ub, int64_t num_equalities, int n_devices, const int *device_ids) {
  return create_multi_impl(out, m, n, nnz, colptr, rowval, nzval, index_base, c, b, lb, ub, num_equalities,
                           n_devices, device_ids, nullptr);
}

static int create_multi_impl(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                             const int64_t *colptr, const int64_t *rowval, const double *nzval,
                             int index_base, const double *c, const double *b, const double *lb,
                             const double *ub, int64_t num_equalities, int n_devices, const int *device_ids,
                             const int64_t *row_bounds) {
  if (!out) return fail(-1, "out == NULL");
  *out = nullptr;
  if (n_devices < 1 || !device_ids) return fail(-1, "n_devices < 1 or device_ids == NULL");
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m) return fail(-1, "num_equalities out of range");
  if (!colptr || (nnz > 0 && (!rowval || !nzval))) return fail(-1, "null input array");
  if (colptr[0] != index_base || colptr[n] - index_base != nnz) return fail(-1, "colptr does not match nnz / index_base");
  DistGroup *g = new DistGroup();
  int rc = init_group_geometry(g, m, n, colptr, rowval, index_base, num_equalities, n_devices, row_bounds);
  if (rc) { delete g; return rc; }
  // Back end: RCCL (ncclCommInitAll) when every shard has its own device; direct peer
  // kernels when devices repeat (several shards on one GPU: tests, oversubscription)
  // or when PDHG_COMM=p2p asks for them.
  bool distinct = true;
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < i; ++j) if (device_ids[i] == device_ids[j]) distinct = false;
  const char *cm = getenv("PDHG_COMM");
  g->backend = (!distinct || (cm && !strcmp(cm, "p2p"))) ? COMM_P2P : COMM_RCCL;
  if (g->backend == COMM_P2P && n_devices > P2P_MAX_WORLD) { delete g; return fail(-1, "peer back end supports at most 16 shards"); }
  choose_exchange_pattern(g);
  for (int r = 0; r < n_devices; ++r) {
    pdhg_handle *s = nullptr;
    rc = create_rank_shard(g, r, n, colptr, rowval, nzval, index_base, c, b, lb, ub, device_ids[r], nullptr, &s);
    if (rc) { destroy_group(g); return rc; }
    g->sh.push_back(s);
  }
  if (g->backend == COMM_RCCL) {
    g->comm.assign((size_t)n_devices, nullptr);
    const RcclApi *R = rccl();
    if (!R) { destroy_group(g); return 2999; }
    ncclResult_t nr = R->CommInitAll(g->comm.data(), n_devices, device_ids);
    if (nr != ncclSuccess) {
      g_last_error = std::string("ncclCommInitAll: ") + R->GetErrorString(nr);
      destroy_group(g);
      return 2000 + (int)nr;
    }
  } else {
    for (int f = 0; f < 2; ++f) g->ev[f].assign((size_t)n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
      (void)hipSetDevice(device_ids[i]);
      for (int j = 0; j < n_devices; ++j)
        if (device_ids[j] != device_ids[i]) {
          hipError_t e = hipDeviceEnablePeerAccess(device_ids[j], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            destroy_group(g);
            return fail((int)e, "hipDeviceEnablePeerAccess failed");
          }
          (void)hipGetLastError();
        }
      for (int f = 0; f < 2; ++f)
        if (hipEventCreateWithFlags(&g->ev[f][(size_t)i], hipEventDisableTiming) != hipSuccess) {
          destroy_group(g);
          return fail(999, "hipEventCreate failed");
        }
    }
  }
  // one issuing host thread per shard for the trial steps (dist.hpp, ShardPool)
  {
    const char *ev = getenv("PDHG_SHARD_THREADS");
    if (n_devices > 1 && !(ev && ev[0] == '0')) g->pool = new ShardPool(n_devices);
  }
  *out = g->sh[0];
  return 0;
}

int pdhg_dist_info(pdhg_handle *h, int64_t info[8]) {
  if (!h || !info) return fail(-1, "null argument");
  info[0] = h->world;
  info[1] = h->grp ? (int64_t)h->grp->sh.size() : 1;
  info[2] = h->rank;
  info[3] = h->grp ? h->grp->backend : -1;
  info[4] = h->row_lo;
  info[5] = h->row_lo + h->m;
  info[6] = h->clo;
  info[7] = h->clo + h->cn;
  return 0;
}

int pdhg_set_objective_matrix(pdhg_handle *h0, int64_t q_nnz, const int64_t *q_colptr,
                              const int64_t *q_rowval, const double *q_nzval, int index_base) {
  int rc = check_handle(h0);
  if (rc) return rc;
  const Shards L = shards_of(h0);
  bump_version(L);
  bool all_zero = true;
  for (int64_t k = 0; k < q_nnz; ++k) if (q_nzval[k] != 0.0) all_zero = false;
  std::vector<int> t_rowptr, rowptr;
  ivec t_col, col;
  dvec t_val, val;
  if (!all_zero) {
    rc = csc_to_both(h0->n, h0->n, q_nnz, q_colptr, q_rowval, q_nzval, index_base, t_rowptr, t_col, t_val, rowptr, col, val);
    if (rc) return rc;
  }
  for (int i = 0; i < L.count; ++i) L.p[i]->matrix_version += 1;
  FOR_SHARDS(L, h) {   // the objective matrix is replicated on every shard (it acts on full n-vectors)
    // the launch paths were decided for the problem without (or with another) Q: decide again at the next trial
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->gsync) { (void)hipFree(h->gsync); h->gsync = nullptr; }
    if (h->coop_trace) { (void)hipFree(h->coop_trace); h->coop_trace = nullptr; }
    h->coop_mode = -1; h->coop_launches = 0; h->coop_epoch = 0;
    h->graph_mode = -1;
    h->small_lp_mode = -1;
    graph_destroy(h->tgraph[0]); graph_destroy(h->tgraph[1]);
    if (h->has_q) { free_csr_dev(h->Q); free_csr_dev(h->Qt); h->has_q = false; }
    if (all_zero) continue;  // iszero(objective_matrix): LP path (pdhg.jl:536)
    if ((rc = build_csr_dev(h->Q, (int)h->n, (int)h->n, rowptr, col, val, h->remap))) return rc;
    if ((rc = build_csr_dev(h->Qt, (int)h->n, (int)h->n, t_rowptr, t_col, t_val, h->remap))) return rc;
    if (!h->qx) { if ((rc = alloc_zero(&h->qx, h->n))) return rc; }
    if (!h->tmp_n2) { if ((rc = alloc_zero(&h->tmp_n2, h->n))) return rc; }
    h->has_q = true;
  }
  return 0;
}

void pdhg_destroy(pdhg_handle *h) {
  if (!h) return;
  if (h->grp) destroy_group(h->grp);
  else destroy_shard(h);
}

