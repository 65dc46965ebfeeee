"""Build recipe of the test-only stand-in for RCCL (tests/fake_rccl/fake_rccl.cpp): TEST INFRASTRUCTURE.

``build()`` compiles ``libfake_rccl.so`` next to the source (hipcc: host code + the HIP runtime for the staging copies);
``env(base)`` returns an environment in which the product library binds it instead of librccl (``PDHG_RCCL_LIB``,
csrc/rccl_loader.hpp).  Nothing in ``firstorderlp.jl_amd/`` knows this file exists."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.cpp")
LIB_PATH = os.path.join(HERE, "libfake_rccl.so")


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(SRC):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", SRC, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def env(base=None, **extra):
    """Environment for processes whose pdhg_create_dist* calls must go through the stand-in."""
    e = dict(os.environ if base is None else base)
    e["PDHG_RCCL_LIB"] = build()
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update({k: str(v) for k, v in extra.items()})
    return e
