// host_trial_graph.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// the trial step as a HIP graph: node construction, parameter updates, launch (host side).

// ---- the trial step as a HIP graph ---------------------------------------------------

template <typename... Args>
hipError_t graph_add_kernel_lds(hipGraph_t g, hipGraphNode_t *node, const std::vector<hipGraphNode_t> &deps,
                                const void *func, dim3 grid, dim3 block, size_t lds, Args... args) {
  void *params[] = {(void *)&args...};
  hipKernelNodeParams p{};
  p.func = const_cast<void *>(func);
  p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)lds;
  p.kernelParams = params; p.extra = nullptr;
  return hipGraphAddKernelNode(node, g, deps.empty() ? nullptr : deps.data(), deps.size(), &p);
}
template <typename... Args>
hipError_t graph_add_kernel(hipGraph_t g, hipGraphNode_t *node, const std::vector<hipGraphNode_t> &deps,
                            const void *func, dim3 grid, dim3 block, Args... args) {
  return graph_add_kernel_lds(g, node, deps, func, grid, block, 0, args...);
}
template <typename... Args>
hipError_t graph_set_kernel_lds(hipGraphExec_t exec, hipGraphNode_t node, const void *func, dim3 grid, dim3 block,
                                size_t lds, Args... args) {
  void *params[] = {(void *)&args...};
  hipKernelNodeParams p{};
  p.func = const_cast<void *>(func);
  p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)lds;
  p.kernelParams = params; p.extra = nullptr;
  return hipGraphExecKernelNodeSetParams(exec, node, &p);
}
template <typename... Args>
hipError_t graph_set_kernel(hipGraphExec_t exec, hipGraphNode_t node, const void *func, dim3 grid, dim3 block,
                            Args... args) {
  return graph_set_kernel_lds(exec, node, func, grid, block, 0, args...);
}

// pinned, host-coherent result word of the one-launch paths: [0..5) sums, [6] error, [7] sequence number
int ensure_result_word(pdhg_handle *h) {
  if (h->seq_dev) return 0;
  HIP_TRY(hipMalloc((void **)&h->seq_dev, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(h->seq_dev, 0, sizeof(unsigned long long), nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));   // the null stream does not order against h->stream
  HIP_TRY(hipHostMalloc((void **)&h->res_host, 8 * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
  for (int q = 0; q < 8; ++q) h->res_host[q] = 0.0;
  return 0;
}

// wait for launch number seq_expected's results in pinned memory (bounded spin, then the stream).
// checked: the trial kernel publishes without a system-scope fence -- a read counts only when
// the sequence number AND the checksum over the eight words match (trial_kernel.hpp).
int wait_result_word(pdhg_handle *h, double out[5], bool checked = false) {
  const double want = (double)h->seq_expected;
  const volatile unsigned long long *bits = reinterpret_cast<const volatile unsigned long long *>(h->res_host);
  auto ready = [&]() -> bool {
    if (h->res_host[7] != want) return false;
    if (!checked) return true;
    unsigned long long w[8];
    for (int q = 0; q < 8; ++q) w[q] = bits[q];
    unsigned long long ck = RESULT_CHECK_SALT ^ w[6] ^ w[7];
    for (int q = 0; q < 5; ++q) ck ^= w[q];
    if (ck != w[5]) return false;
    for (int q = 0; q < 5; ++q) memcpy(&out[q], &w[q], 8);
    memcpy(&h->res_error, &w[6], 8);
    return true;
  };
  bool seen = false;
  for (long spin = 0; spin < 40000000L; ++spin) {
    if (ready()) { seen = true; break; }
    if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(h->stream) != hipErrorNotReady) break;
  }
  if (!seen) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!ready()) {
      h->seq_expected = (unsigned long long)h->res_host[7];   // resynchronise: the next launch can succeed
      return fail(998, "one-launch trial finished without publishing its results");
    }
  }
  if (!checked) for (int q = 0; q < 5; ++q) out[q] = h->res_host[q];
  out[4] *= 0.5;
  return 0;
}

