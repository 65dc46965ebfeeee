// rccl_loader.hpp -- part of the single translation unit pdhg_hip.hip (included by dist.hpp).
// RCCL is bound at RUN time, on the first multi-GPU entry point, not at link time:
//  * a single-GPU user (and the "library loads and exports every symbol" host test) needs no
//    librccl.so at all -- the library used to carry a DT_NEEDED on it plus an rpath fixed at
//    build time;
//  * WHICH librccl gets bound is decided here and reported (pdhg_rccl_info): PDHG_RCCL_LIB if
//    set; else the one the process has already loaded (a host that uses torch.distributed has
//    torch's bundled librccl.so resident -- binding the same instance avoids two RCCL runtimes
//    in one process); else $ROCM_PATH/lib/librccl.so.1; else the loader's default search;
//  * its ncclGetVersion() is compared with the header this file was compiled against
//    (NCCL_VERSION_CODE): a different MAJOR version is refused with a clear error instead of
//    running with mismatched struct layouts; a different minor is accepted (the dozen entry
//    points used here -- communicator creation, all-gather, reduce-scatter, reduce, broadcast,
//    group calls -- have kept their C signatures throughout 2.x) and reported.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void *handle = nullptr;
  std::string path, error;
  int runtime_version = 0;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
};

inline int nccl_major_of(int code) { return code >= 10000 ? code / 10000 : code / 1000; }

// Loads on first use (thread-safe: function-local static).  ok() == false: `error` says why.
struct RcclLoader {
  RcclApi api;
  bool ok = false;
  RcclLoader() {
    std::vector<std::string> tries;
    if (const char *ev = getenv("PDHG_RCCL_LIB")) tries.push_back(ev);
    else {
      tries.push_back("@loaded");                       // whatever librccl the process already holds
      const char *rocm = getenv("ROCM_PATH");
      tries.push_back(std::string(rocm ? rocm : "/opt/rocm") + "/lib/librccl.so.1");
      tries.push_back("librccl.so.1");
      tries.push_back("librccl.so");
    }
    std::string why;
    for (const std::string &t : tries) {
      void *h = nullptr;
      if (t == "@loaded") {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
          h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
          if (h) break;
        }
      } else {
        h = dlopen(t.c_str(), RTLD_NOW | RTLD_GLOBAL);
      }
      if (!h) { if (t != "@loaded") { const char *e = dlerror(); why += t + ": " + (e ? e : "?") + "; "; } continue; }
      api.handle = h;
      break;
    }
    if (!api.handle) { api.error = "no RCCL library could be loaded (" + why + "set PDHG_RCCL_LIB)"; return; }
#define PDHG_RCCL_SYM(field, name)                                                      \
    *(void **)(&api.field) = dlsym(api.handle, name);                                   \
    if (!api.field) { api.error = std::string("librccl lacks ") + name; return; }
    PDHG_RCCL_SYM(GetVersion, "ncclGetVersion")
    PDHG_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    PDHG_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    PDHG_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    PDHG_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    PDHG_RCCL_SYM(GetErrorString, "ncclGetErrorString")
    PDHG_RCCL_SYM(GroupStart, "ncclGroupStart")
    PDHG_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    PDHG_RCCL_SYM(AllGather, "ncclAllGather")
    PDHG_RCCL_SYM(ReduceScatter, "ncclReduceScatter")
    PDHG_RCCL_SYM(Reduce, "ncclReduce")
    PDHG_RCCL_SYM(Broadcast, "ncclBroadcast")
#undef PDHG_RCCL_SYM
    Dl_info info{};
    if (dladdr((void *)api.GetVersion, &info) && info.dli_fname) api.path = info.dli_fname;
    int v = 0;
    if (api.GetVersion(&v) != ncclSuccess) { api.error = "ncclGetVersion failed"; return; }
    api.runtime_version = v;
    if (nccl_major_of(v) != NCCL_MAJOR) {
      api.error = "RCCL major version mismatch: " + api.path + " reports " + std::to_string(v) +
                  ", this library was compiled against rccl.h " + std::to_string(NCCL_VERSION_CODE) +
                  " (set PDHG_RCCL_LIB to a matching librccl)";
      return;
    }
    if (getenv("PDHG_VERBOSE"))
      fprintf(stderr, "[pdhg_hip] RCCL bound at run time: %s, version %d (header %d)\n", api.path.c_str(), v, NCCL_VERSION_CODE);
    ok = true;
  }
};

inline RcclLoader &rccl_loader() {
  static RcclLoader L;
  return L;
}

// nullptr (and g_last_error set) when RCCL is unavailable or refused
inline const RcclApi *rccl() {
  RcclLoader &L = rccl_loader();
  if (!L.ok) { g_last_error = L.api.error; return nullptr; }
  return &L.api;
}

}  // namespace
