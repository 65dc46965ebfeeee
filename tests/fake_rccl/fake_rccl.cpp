// fake_rccl.cpp -- TEST INFRASTRUCTURE, never shipped, never loaded by the product unless a test
// points PDHG_RCCL_LIB at it.
//
// A stand-in for the dozen RCCL entry points csrc/rccl_loader.hpp binds at run time, between
// SEPARATE PROCESSES THAT ALL USE THE SAME DEVICE.  Real RCCL refuses two ranks on one GPU, and
// the builder's boxes have one GPU: without this file pdhg_create_dist / pdhg_create_dist_rows,
// csrc/dist.hpp's one-rank-per-process routes (combine_scalars through ncclAllGather, the per-slice
// ncclReduce on the comm stream, ncclBroadcast of row ranges) and `bench.py --gpus N` would run for
// the first time on the driver's 8-GPU node.
//
// Transport: a file in $FAKE_RCCL_DIR (default /tmp) mapped MAP_SHARED by every rank: a header of
// process-shared atomics + one staging slot per rank.  A collective is
//   hipStreamSynchronize(stream) -> my chunk device-to-host into my slot -> barrier
//   -> read the slots (reductions: added / maxed IN RANK ORDER, the order csrc/dist.hpp's peer
//      kernels use, so a multi-process run must equal the in-process shard group bit for bit)
//   -> host-to-device on `stream` + synchronise -> barrier (the slots may be rewritten).
// Host-synchronous, so stream order is trivially kept; vectors longer than a slot go in chunks.
// Every rank issues the same sequence of collectives (what RCCL demands too), one thread per
// communicator.  ncclGroupStart/End queue the calls and run them in order at the outermost End.
// A barrier that does not complete within FAKE_RCCL_TIMEOUT_S (default 180) returns
// ncclSystemError on every waiting rank instead of hanging the test.
//
// Not implemented (ncclInvalidUsage): ncclCommInitAll (several communicators inside ONE process:
// the library's in-process groups on one device use its own peer-kernel back end).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr size_t HEADER_BYTES = 4096;

struct Header {
  std::atomic<int> attached;
  std::atomic<int> bar_count;
  std::atomic<int> bar_gen;
  std::atomic<int> failed;          // a rank gave up: everybody else stops waiting
  std::atomic<int> detached;
  std::atomic<int64_t> ops;         // collectives completed (rank 0 counts): reported by fake_rccl_stats
  std::atomic<int64_t> bytes;       // payload bytes rank 0 staged
};
static_assert(sizeof(Header) <= HEADER_BYTES, "header");
static_assert(std::atomic<int>::is_always_lock_free, "process-shared atomics must be lock-free");

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double timeout_s() {
  static const double t = [] { const char *e = getenv("FAKE_RCCL_TIMEOUT_S"); return e ? atof(e) : 180.0; }();
  return t;
}

size_t slot_bytes_cfg() {
  const char *e = getenv("FAKE_RCCL_SLOT_MB");
  const size_t mb = e ? (size_t)std::max(1, atoi(e)) : 8;
  return mb << 20;
}

}  // namespace

struct ncclComm {
  int rank = 0, world = 1;
  Header *hdr = nullptr;
  char *base = nullptr;
  size_t map_bytes = 0, slot_bytes = 0;
  bool registered = false;
  std::vector<char> tmp;
  char *slot(int r) const { return base + HEADER_BYTES + (size_t)r * slot_bytes; }
};

namespace {

ncclResult_t barrier(ncclComm *c) {
  Header *h = c->hdr;
  if (c->world == 1) return ncclSuccess;
  const int gen = h->bar_gen.load(std::memory_order_acquire);
  if (h->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    h->bar_count.store(0, std::memory_order_relaxed);
    h->bar_gen.fetch_add(1, std::memory_order_acq_rel);
    return ncclSuccess;
  }
  const double t0 = now_s();
  unsigned spins = 0;
  while (h->bar_gen.load(std::memory_order_acquire) == gen) {
    if (h->failed.load(std::memory_order_relaxed)) return ncclSystemError;
    if ((++spins & 0x3ff) == 0) {
      if (now_s() - t0 > timeout_s()) {
        h->failed.store(1);
        fprintf(stderr, "[fake_rccl] rank %d: barrier timed out after %.0f s\n", c->rank, timeout_s());
        return ncclSystemError;
      }
      if (spins > (1u << 16)) usleep(50);      // the ranks share the host's cores with each other
    }
  }
  return ncclSuccess;
}

// FAKE_RCCL_HOST_BUFFERS=1: the buffers are HOST memory and no HIP call is made -- the CPU-only unit test of this file's
// own protocol (tests/test_fake_rccl_host.py), which runs where there is no GPU
bool host_mode() {
  static const bool on = [] { const char *e = getenv("FAKE_RCCL_HOST_BUFFERS"); return e && e[0] == '1'; }();
  return on;
}
hipError_t stage_in(void *slot, const void *src, size_t nb, hipStream_t s) {
  if (host_mode()) { memcpy(slot, src, nb); return hipSuccess; }
  return hipMemcpyAsync(slot, src, nb, hipMemcpyDeviceToHost, s);
}
hipError_t stage_out(void *dst, const void *slot, size_t nb, hipStream_t s) {
  if (host_mode()) { memcpy(dst, slot, nb); return hipSuccess; }
  return hipMemcpyAsync(dst, slot, nb, hipMemcpyHostToDevice, s);
}
hipError_t stream_sync(hipStream_t s) { return host_mode() ? hipSuccess : hipStreamSynchronize(s); }

size_t dtype_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclDouble: case ncclInt64: case ncclUint64: return 8;
    case ncclFloat: case ncclInt32: case ncclUint32: return 4;
    case ncclInt8: case ncclUint8: return 1;
    default: return 0;
  }
}

#define HIP_OK(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      fprintf(stderr, "[fake_rccl] %s: %s\n", #expr, hipGetErrorString(_e));           \
      c->hdr->failed.store(1);                                                         \
      return ncclUnhandledCudaError;                                                   \
    }                                                                                  \
  } while (0)
#define BAR()                                                                          \
  do {                                                                                 \
    ncclResult_t _b = barrier(c);                                                      \
    if (_b != ncclSuccess) return _b;                                                  \
  } while (0)

// dst[i] = op over ranks q ascending of slot(q)[i]: doubles only (what the library reduces)
void reduce_slots(ncclComm *c, double *dst, size_t count, ncclRedOp_t op) {
  const double *s0 = (const double *)c->slot(0);
  for (size_t i = 0; i < count; ++i) dst[i] = s0[i];
  for (int q = 1; q < c->world; ++q) {
    const double *sq = (const double *)c->slot(q);
    if (op == ncclMax) for (size_t i = 0; i < count; ++i) dst[i] = std::fmax(dst[i], sq[i]);
    else for (size_t i = 0; i < count; ++i) dst[i] = dst[i] + sq[i];
  }
}

struct Op {
  int kind;      // 0 all-gather, 1 reduce-scatter, 2 reduce, 3 broadcast
  const void *send;
  void *recv;
  size_t count;
  ncclDataType_t dt;
  ncclRedOp_t op;
  int root;
  ncclComm *comm;
  hipStream_t stream;
};

void account(ncclComm *c, size_t bytes) {
  if (c->rank == 0) {
    c->hdr->ops.fetch_add(1, std::memory_order_relaxed);
    c->hdr->bytes.fetch_add((int64_t)bytes, std::memory_order_relaxed);
  }
}

ncclResult_t do_all_gather(const Op &o) {
  ncclComm *c = o.comm;
  const size_t es = dtype_bytes(o.dt);
  if (!es) return ncclInvalidArgument;
  HIP_OK(stream_sync(o.stream));
  const size_t total = o.count * es, chunk = c->slot_bytes;
  for (size_t off = 0; off < total || off == 0; off += chunk) {
    const size_t nb = std::min(chunk, total - off);
    if (nb) HIP_OK(stage_in(c->slot(c->rank), (const char *)o.send + off, nb, o.stream));
    HIP_OK(stream_sync(o.stream));
    BAR();
    for (int q = 0; q < c->world && nb; ++q) {
      char *dst = (char *)o.recv + (size_t)q * total + off;
      if (q == c->rank && dst == (const char *)o.send + off) continue;       // in place: my own part is there already
      HIP_OK(stage_out(dst, c->slot(q), nb, o.stream));
    }
    HIP_OK(stream_sync(o.stream));
    BAR();
    if (total == 0) break;
  }
  account(c, total);
  return ncclSuccess;
}

// the reduction of part `part` (count elements at send + part*count) lands at `recv` on rank `part`
ncclResult_t reduce_part(const Op &o, const void *send_part, void *recv, int owner) {
  ncclComm *c = o.comm;
  if (o.dt != ncclDouble || (o.op != ncclSum && o.op != ncclMax)) return ncclInvalidArgument;
  const size_t total = o.count * 8, chunk = c->slot_bytes;
  for (size_t off = 0; off < total || off == 0; off += chunk) {
    const size_t nb = std::min(chunk, total - off);
    if (nb) HIP_OK(stage_in(c->slot(c->rank), (const char *)send_part + off, nb, o.stream));
    HIP_OK(stream_sync(o.stream));
    BAR();
    if (c->rank == owner && nb) {
      if (c->tmp.size() < nb) c->tmp.resize(nb);
      reduce_slots(c, (double *)c->tmp.data(), nb / 8, o.op);
      HIP_OK(stage_out((char *)recv + off, c->tmp.data(), nb, o.stream));
      HIP_OK(stream_sync(o.stream));
    }
    BAR();
    if (total == 0) break;
  }
  return ncclSuccess;
}

ncclResult_t do_reduce_scatter(const Op &o) {
  ncclComm *c = o.comm;
  HIP_OK(stream_sync(o.stream));
  for (int q = 0; q < c->world; ++q) {
    ncclResult_t r = reduce_part(o, (const char *)o.send + (size_t)q * o.count * 8, o.recv, q);
    if (r != ncclSuccess) return r;
  }
  account(c, o.count * 8 * (size_t)c->world);
  return ncclSuccess;
}

ncclResult_t do_reduce(const Op &o) {
  ncclComm *c = o.comm;
  if (o.root < 0 || o.root >= c->world) return ncclInvalidArgument;
  HIP_OK(stream_sync(o.stream));
  ncclResult_t r = reduce_part(o, o.send, o.recv, o.root);
  if (r == ncclSuccess) account(c, o.count * 8);
  return r;
}

ncclResult_t do_broadcast(const Op &o) {
  ncclComm *c = o.comm;
  const size_t es = dtype_bytes(o.dt);
  if (!es || o.root < 0 || o.root >= c->world) return ncclInvalidArgument;
  HIP_OK(stream_sync(o.stream));
  const size_t total = o.count * es, chunk = c->slot_bytes;
  for (size_t off = 0; off < total || off == 0; off += chunk) {
    const size_t nb = std::min(chunk, total - off);
    if (c->rank == o.root && nb) {
      HIP_OK(stage_in(c->slot(o.root), (const char *)o.send + off, nb, o.stream));
      HIP_OK(stream_sync(o.stream));
    }
    BAR();
    if (nb && !(c->rank == o.root && o.recv == o.send)) {
      HIP_OK(stage_out((char *)o.recv + off, c->slot(o.root), nb, o.stream));
      HIP_OK(stream_sync(o.stream));
    }
    BAR();
    if (total == 0) break;
  }
  account(c, total);
  return ncclSuccess;
}

thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;

ncclResult_t run_op(const Op &o) {
  switch (o.kind) {
    case 0: return do_all_gather(o);
    case 1: return do_reduce_scatter(o);
    case 2: return do_reduce(o);
    default: return do_broadcast(o);
  }
}

ncclResult_t submit(const Op &o) {
  if (!o.comm || !o.comm->hdr) return ncclInvalidArgument;
  if (g_depth > 0) { g_queue.push_back(o); return ncclSuccess; }
  return run_op(o);
}

std::atomic<int> g_id_counter{0};

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
  if (!version) return ncclInvalidArgument;
  *version = NCCL_VERSION_CODE;          // the header the product was compiled against: same major by construction
  return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof(id->internal), "fake_rccl_%d_%lld_%ld_%d", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec,
           g_id_counter.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (strncmp(id.internal, "fake_rccl_", 10) != 0 || memchr(id.internal, 0, sizeof(id.internal)) == nullptr) return ncclInvalidArgument;
  const char *dir = getenv("FAKE_RCCL_DIR");
  const std::string path = std::string(dir ? dir : "/tmp") + "/" + id.internal;
  ncclComm *c = new ncclComm();
  c->rank = rank;
  c->world = nranks;
  c->slot_bytes = slot_bytes_cfg();
  c->map_bytes = HEADER_BYTES + (size_t)nranks * c->slot_bytes;
  const int fd = open(path.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) { fprintf(stderr, "[fake_rccl] open %s: %s\n", path.c_str(), strerror(errno)); delete c; return ncclSystemError; }
  // every rank sets the same length: a fresh file reads as zeros, i.e. the header starts in its initial state
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return ncclSystemError; }
  void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->base = (char *)p;
  c->hdr = (Header *)p;
  // pinned staging when the runtime allows it (plain pageable memory works too, only slower)
  c->registered = !host_mode() && hipHostRegister(c->base + HEADER_BYTES, c->map_bytes - HEADER_BYTES, hipHostRegisterDefault) == hipSuccess;
  if (!c->registered) (void)hipGetLastError();
  c->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
  const double t0 = now_s();
  while (c->hdr->attached.load(std::memory_order_acquire) < nranks) {
    if (now_s() - t0 > timeout_s()) {
      fprintf(stderr, "[fake_rccl] rank %d: only %d of %d ranks attached to %s\n", rank, c->hdr->attached.load(), nranks, path.c_str());
      c->hdr->failed.store(1);
      munmap(c->base, c->map_bytes);
      unlink(path.c_str());
      delete c;
      return ncclSystemError;
    }
    usleep(200);
  }
  if (barrier(c) != ncclSuccess) { munmap(c->base, c->map_bytes); delete c; return ncclSystemError; }
  if (rank == 0) unlink(path.c_str());         // everybody holds a mapping: the name can go
  if (getenv("FAKE_RCCL_VERBOSE"))
    fprintf(stderr, "[fake_rccl] rank %d / %d attached (%s, slots of %zu MiB, %s staging)\n", rank, nranks, path.c_str(),
            c->slot_bytes >> 20, c->registered ? "pinned" : "pageable");
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->base) {
    if (c->registered) (void)hipHostUnregister(c->base + HEADER_BYTES);
    c->hdr->detached.fetch_add(1);
    munmap(c->base, c->map_bytes);
  }
  delete c;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
    case ncclSystemError: return "fake_rccl: system error (a rank is missing, failed or timed out)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: not implemented by the test transport";
    default: return "fake_rccl: error";
  }
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> q;
  q.swap(g_queue);
  for (const Op &o : q) {
    ncclResult_t r = run_op(o);
    if (r != ncclSuccess) return r;
  }
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
  return submit(Op{0, sendbuff, recvbuff, sendcount, datatype, ncclSum, 0, comm, stream});
}

ncclResult_t ncclReduceScatter(const void *sendbuff, void *recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op,
                               ncclComm_t comm, hipStream_t stream) {
  return submit(Op{1, sendbuff, recvbuff, recvcount, datatype, op, 0, comm, stream});
}

ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream) {
  return submit(Op{2, sendbuff, recvbuff, count, datatype, op, root, comm, stream});
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
  return submit(Op{3, sendbuff, recvbuff, count, datatype, ncclSum, root, comm, stream});
}

// test aid: collectives completed and payload bytes staged by rank 0 of this communicator's group
int fake_rccl_stats(ncclComm_t c, long long out[2]) {
  if (!c || !c->hdr || !out) return -1;
  out[0] = (long long)c->hdr->ops.load();
  out[1] = (long long)c->hdr->bytes.load();
  return 0;
}

}  // extern "C"
