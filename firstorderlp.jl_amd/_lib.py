"""ctypes loader for csrc/libpdhg_hip.so (the C ABI in include/pdhg_hip.h).

There is deliberately NO fallback: if the HIP library is missing or cannot be
loaded the product path raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
# PDHG_HIP_LIB: developer override used by tools/variants.sh to time -D variants
LIB_PATH = os.environ.get("PDHG_HIP_LIB") or os.path.join(CSRC, "libpdhg_hip.so")
SRC_PATH = os.path.join(CSRC, "pdhg_hip.hip")

HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off",
               "-std=c++17", "-shared", "-fPIC", "-pthread"]
# RCCL (the library owns the multi-GPU exchange, csrc/dist.hpp) is bound at RUN time by
# csrc/rccl_loader.hpp: no link-time dependency, no rpath
LINK_FLAGS = ["-ldl"]

# every symbol include/pdhg_hip.h declares
EXPORTS = [
    "pdhg_last_error", "pdhg_abi_version", "pdhg_create",
    "pdhg_set_objective_matrix", "pdhg_destroy", "pdhg_trial_step",
    "pdhg_trial_primal", "pdhg_trial_dual", "pdhg_accept", "pdhg_take_step_adaptive", "pdhg_take_steps_adaptive",
    "pdhg_add_current_primal_to_average", "pdhg_get_average_info",
    "pdhg_get_average", "pdhg_reset_average", "pdhg_restart_to_average",
    "pdhg_get_current", "pdhg_set_current", "pdhg_get_trial", "pdhg_spmv",
    "pdhg_spmv_t", "pdhg_dist_get_unique_id", "pdhg_create_dist", "pdhg_create_multi",
    "pdhg_dist_info", "pdhg_profile_enable", "pdhg_profile_read",
    "pdhg_kernel_algorithmic_bytes", "pdhg_kernel_name", "pdhg_layout_info", "pdhg_layout_describe", "pdhg_measure_triad", "pdhg_measure_sweep_ceiling", "pdhg_trial_timeline",
    "pdhg_set_original_problem", "pdhg_eval_point", "pdhg_save_restart_point",
    "pdhg_distance_to_restart", "pdhg_get_point", "pdhg_trust_region_bound", "pdhg_trust_region_bounds",
    "pdhg_point_sumsq", "pdhg_rescale", "pdhg_get_problem_vectors", "pdhg_matrix_max_abs",
    "pdhg_partition_rows", "pdhg_create_dist_rows", "pdhg_rccl_info", "pdhg_host_issue_stats",
    "pdhg_measure_launch_overhead", "pdhg_selftest_wave_sums", "pdhg_layout_checksums",
]

ABI_VERSION = 11
UNIQUE_ID_BYTES = 128
(K_PRIMAL, K_SPMV_DUAL, K_SPMV_ATY, K_FINAL, K_ACCEPT, K_ALLGATHER, K_REDUCE_SCATTER,
 K_INTERACTION, K_COUNT) = range(9)
POINT_CURRENT, POINT_AVERAGE, POINT_RESTART = range(3)


def build(force=False, verbose=False):
    """Cross-compile the HIP library for gfx950 with hipcc (no GPU needed)."""
    hipcc = os.environ.get("HIPCC", "hipcc")
    sources = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))] + \
              [os.path.join(INCLUDE, "pdhg_hip.h")]
    if not force and os.path.exists(LIB_PATH) and \
            os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(f) for f in sources):
        return LIB_PATH
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", INCLUDE, "-o", LIB_PATH, SRC_PATH] + LINK_FLAGS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


# ---- sanitizer builds of the HOST side (the kernels are compiled as usual: -fno-gpu-sanitize) ----------------------
# libpdhg_hip_asan.so: AddressSanitizer + UndefinedBehaviorSanitizer.  tests/test_sanitizer_host.py drives the
# host-only entry points (row partition, argument validation) through it in a child process that preloads the ASan
# runtime; libpdhg_hip_tsan.so: ThreadSanitizer, for the shard pool's issuing threads (tools/archive/r4_tsan_shards.sh, GPU box).
SANITIZED = {"asan": (os.path.join(CSRC, "libpdhg_hip_asan.so"),
                      ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
             "tsan": (os.path.join(CSRC, "libpdhg_hip_tsan.so"), ["-fsanitize=thread"])}


def sanitizer_runtime(kind="asan"):
    """Path of the clang runtime a process must LD_PRELOAD before it loads the sanitized library."""
    hipcc = os.environ.get("HIPCC", "hipcc")
    name = {"asan": "libclang_rt.asan-x86_64.so", "tsan": "libclang_rt.tsan-x86_64.so"}[kind]
    clang = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang")
    out = subprocess.check_output([clang if os.path.exists(clang) else hipcc, f"-print-file-name={name}"]).decode().strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build_sanitized(kind="asan", force=False, verbose=False):
    """hipcc -O1 -g -fsanitize=... -fno-gpu-sanitize: the same translation unit, host code instrumented."""
    hipcc = os.environ.get("HIPCC", "hipcc")
    path, flags = SANITIZED[kind]
    sources = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))] + \
              [os.path.join(INCLUDE, "pdhg_hip.h")]
    if not force and os.path.exists(path) and os.path.getmtime(path) >= max(os.path.getmtime(f) for f in sources):
        return path
    cmd = [hipcc, "-O1", "-g"] + HIPCC_FLAGS[1:] + flags + ["-fno-gpu-sanitize", "-I", INCLUDE, "-o", path, SRC_PATH] + LINK_FLAGS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return path


class PdhgHipError(RuntimeError):
    pass


_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p


def lib():
    """Load the library and declare every prototype.  Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    d, i64, i32 = ctypes.c_double, ctypes.c_int64, ctypes.c_int
    # a stale library (older ABI: different out[] lengths, missing entry points)
    # must fail here, not as a buffer overrun deep inside a solve
    missing = [name for name in EXPORTS if not hasattr(L, name)]
    L.pdhg_abi_version.restype = i32
    L.pdhg_abi_version.argtypes = []
    version = L.pdhg_abi_version() if not missing or "pdhg_abi_version" not in missing else None
    if missing or version != ABI_VERSION:
        raise PdhgHipError(
            f"{LIB_PATH} is stale: abi {version} (need {ABI_VERSION}), missing symbols {missing}; "
            "rebuild with __graft_entry__.build()")
    L.pdhg_last_error.restype = ctypes.c_char_p
    L.pdhg_last_error.argtypes = []
    L.pdhg_abi_version.restype = i32
    L.pdhg_create.restype = i32
    L.pdhg_create.argtypes = [ctypes.POINTER(_vp), i64, i64, i64, _ip, _ip,
                              _dp, i32, _dp, _dp, _dp, _dp, i64, i32, _vp]
    L.pdhg_set_objective_matrix.restype = i32
    L.pdhg_set_objective_matrix.argtypes = [_vp, i64, _ip, _ip, _dp, i32]
    L.pdhg_destroy.restype = None
    L.pdhg_destroy.argtypes = [_vp]
    L.pdhg_trial_step.restype = i32
    L.pdhg_trial_step.argtypes = [_vp, d, d, d, _dp]
    L.pdhg_trial_primal.restype = i32
    L.pdhg_trial_primal.argtypes = [_vp, d, d]
    L.pdhg_trial_dual.restype = i32
    L.pdhg_trial_dual.argtypes = [_vp, d, d, d, _dp]
    L.pdhg_accept.restype = i32
    L.pdhg_accept.argtypes = [_vp, d]
    L.pdhg_take_step_adaptive.restype = i32
    L.pdhg_take_step_adaptive.argtypes = [_vp, d, d, _dp, d, _ip, _dp, ctypes.POINTER(i32)]
    L.pdhg_take_steps_adaptive.restype = i32
    L.pdhg_take_steps_adaptive.argtypes = [_vp, i64, d, d, _dp, d, _ip, _dp, ctypes.POINTER(i32), _ip]
    L.pdhg_add_current_primal_to_average.restype = i32
    L.pdhg_add_current_primal_to_average.argtypes = [_vp, d]
    L.pdhg_get_average_info.restype = i32
    L.pdhg_get_average_info.argtypes = [_vp, _ip, _dp]
    L.pdhg_get_average.restype = i32
    L.pdhg_get_average.argtypes = [_vp, _dp, _dp]
    L.pdhg_reset_average.restype = i32
    L.pdhg_reset_average.argtypes = [_vp]
    L.pdhg_restart_to_average.restype = i32
    L.pdhg_restart_to_average.argtypes = [_vp]
    L.pdhg_get_current.restype = i32
    L.pdhg_get_current.argtypes = [_vp, _dp, _dp, _dp]
    L.pdhg_set_current.restype = i32
    L.pdhg_set_current.argtypes = [_vp, _dp, _dp]
    L.pdhg_get_trial.restype = i32
    L.pdhg_get_trial.argtypes = [_vp, _dp, _dp, _dp]
    L.pdhg_spmv.restype = i32
    L.pdhg_spmv.argtypes = [_vp, _dp, _dp]
    L.pdhg_spmv_t.restype = i32
    L.pdhg_spmv_t.argtypes = [_vp, _dp, _dp]
    L.pdhg_measure_triad.restype = i32
    L.pdhg_measure_triad.argtypes = [_vp, i64, i32, _dp]
    L.pdhg_measure_sweep_ceiling.restype = i32
    L.pdhg_measure_sweep_ceiling.argtypes = [_vp, i64, i64, i64, i32, _dp]
    L.pdhg_trial_timeline.restype = i32
    L.pdhg_trial_timeline.argtypes = [_vp, _dp]
    L.pdhg_layout_checksums.restype = i32
    L.pdhg_layout_checksums.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint64)]
    L.pdhg_measure_launch_overhead.restype = i32
    L.pdhg_measure_launch_overhead.argtypes = [_vp, i32, _dp]
    L.pdhg_selftest_wave_sums.restype = i32
    L.pdhg_selftest_wave_sums.argtypes = [_vp, ctypes.c_int64, _ip]
    L.pdhg_dist_get_unique_id.restype = i32
    L.pdhg_dist_get_unique_id.argtypes = [_vp]
    L.pdhg_create_dist.restype = i32
    L.pdhg_create_dist.argtypes = [ctypes.POINTER(_vp), i64, i64, i64, _ip, _ip,
                                   _dp, i32, _dp, _dp, _dp, _dp, i64, i32, _vp, _vp, i32, i32]
    L.pdhg_create_multi.restype = i32
    L.pdhg_create_multi.argtypes = [ctypes.POINTER(_vp), i64, i64, i64, _ip, _ip,
                                    _dp, i32, _dp, _dp, _dp, _dp, i64, i32, ctypes.POINTER(i32)]
    L.pdhg_partition_rows.restype = i32
    L.pdhg_partition_rows.argtypes = [i64, i64, _ip, _ip, i32, i32, _ip]
    L.pdhg_create_dist_rows.restype = i32
    L.pdhg_create_dist_rows.argtypes = [ctypes.POINTER(_vp), i64, i64, _ip, i64, _ip, _ip, _dp, i32,
                                        _dp, _dp, _dp, _dp, i64, i32, _vp, _vp, i32, i32]
    L.pdhg_rccl_info.restype = i32
    L.pdhg_rccl_info.argtypes = [ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.c_char_p, i32]
    L.pdhg_host_issue_stats.restype = i32
    L.pdhg_host_issue_stats.argtypes = [_vp, _ip, _dp, _dp]
    L.pdhg_dist_info.restype = i32
    L.pdhg_dist_info.argtypes = [_vp, _ip]
    L.pdhg_profile_enable.restype = i32
    L.pdhg_profile_enable.argtypes = [_vp, i32]
    L.pdhg_profile_read.restype = i32
    L.pdhg_profile_read.argtypes = [_vp, i32, _ip, _dp]
    L.pdhg_kernel_algorithmic_bytes.restype = i64
    L.pdhg_kernel_algorithmic_bytes.argtypes = [_vp, i32]
    L.pdhg_layout_describe.restype = i32
    L.pdhg_layout_describe.argtypes = [_vp, ctypes.c_char_p, i32]
    L.pdhg_kernel_name.restype = ctypes.c_char_p
    L.pdhg_kernel_name.argtypes = [_vp, i32]
    L.pdhg_layout_info.restype = i32
    L.pdhg_layout_info.argtypes = [_vp, _ip]
    L.pdhg_set_original_problem.restype = i32
    L.pdhg_set_original_problem.argtypes = [_vp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.pdhg_eval_point.restype = i32
    L.pdhg_eval_point.argtypes = [_vp, i32, _dp]
    L.pdhg_save_restart_point.restype = i32
    L.pdhg_save_restart_point.argtypes = [_vp]
    L.pdhg_distance_to_restart.restype = i32
    L.pdhg_distance_to_restart.argtypes = [_vp, i32, _dp]
    L.pdhg_point_sumsq.restype = i32
    L.pdhg_point_sumsq.argtypes = [_vp, i32, _dp]
    L.pdhg_get_point.restype = i32
    L.pdhg_get_point.argtypes = [_vp, i32, _dp, _dp]
    L.pdhg_rescale.restype = i32
    L.pdhg_rescale.argtypes = [_vp, i32, i32, i32, d, _dp, _dp]
    L.pdhg_get_problem_vectors.restype = i32
    L.pdhg_get_problem_vectors.argtypes = [_vp, _dp, _dp, _dp, _dp]
    L.pdhg_matrix_max_abs.restype = i32
    L.pdhg_matrix_max_abs.argtypes = [_vp, _dp]
    L.pdhg_trust_region_bound.restype = i32
    L.pdhg_trust_region_bound.argtypes = [_vp, i32, d, d, d, i32, i32, _dp]
    L.pdhg_trust_region_bounds.restype = i32
    L.pdhg_trust_region_bounds.argtypes = [_vp, i32, _vp, d, d, _dp, _vp, i32, _dp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().pdhg_last_error().decode("utf-8", "replace")
        raise PdhgHipError(f"pdhg_hip error {rc}: {msg}")
