"""The Julia `ccall` shim (julia/FirstOrderLpHIP.jl) cannot be executed here (no
Julia in the image); what can be checked is its coverage of the ABI: every entry
point include/pdhg_hip.h declares has a ccall in the shim, and nothing is bound
that the header does not declare."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_header_export_has_a_ccall_in_the_julia_shim():
    header = open(os.path.join(ROOT, "include", "pdhg_hip.h")).read()
    declared = set(re.findall(r"\b(pdhg_[a-z_0-9]+)\s*\(", header))
    shim = open(os.path.join(ROOT, "julia", "FirstOrderLpHIP.jl")).read()
    bound = set(re.findall(r"ccall\(\(:(pdhg_[a-z_0-9]+), LIB\)", shim))
    assert declared - bound == set(), f"exports without a ccall: {sorted(declared - bound)}"
    assert bound - declared == set(), f"ccalls to undeclared symbols: {sorted(bound - declared)}"


def test_julia_shim_defines_the_reference_methods():
    shim = open(os.path.join(ROOT, "julia", "FirstOrderLpHIP.jl")).read()
    for needle in ("function FirstOrderLp.take_step(step_params::FirstOrderLp.AdaptiveStepsizeParams",
                   "function FirstOrderLp.take_step(step_params::FirstOrderLp.ConstantStepsizeParams",
                   "function FirstOrderLp.take_step(step_params::FirstOrderLp.MalitskyPockStepsizeParameters",
                   "function FirstOrderLp.optimize(hp::HipPdhgParameters"):
        assert needle in shim, needle
    abi = int(re.search(r"const ABI_VERSION = (\d+)", shim).group(1))
    from firstorderlp_jl_amd import _lib
    assert abi == _lib.ABI_VERSION
