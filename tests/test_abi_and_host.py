"""CPU-only checks: the C-ABI library builds/loads and exports every symbol
include/pdhg_hip.h declares; host-side helpers pinned by the reference's
exact-equality unit tests (test/test_saddle_point.jl, test_iteration_stats.jl)."""
import ctypes
import os
import re

import numpy as np
import pytest

from firstorderlp_jl_amd import _lib, linear_programming_problem
from firstorderlp_jl_amd.iteration_stats_utils import (compute_dual_stats,
                                                       max_primal_violation,
                                                       primal_obj)
from firstorderlp_jl_amd.saddle_point import (compute_lagrangian_value,
                                              select_initial_primal_weight)
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    header = open(os.path.join(ROOT, "include", "pdhg_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(pdhg_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations found"
    assert sorted(_lib.EXPORTS) == declared
    L = ctypes.CDLL(_lib.LIB_PATH)       # loads without a GPU
    for name in declared:
        assert hasattr(L, name), f"{name} not exported by libpdhg_hip.so"
    assert _lib.lib().pdhg_abi_version() == _lib.ABI_VERSION == 11


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from firstorderlp_jl_amd import HipPdhgEngine
    with pytest.raises(_lib.PdhgHipError):
        HipPdhgEngine.from_problem(H.example_lp())


def test_select_initial_primal_weight():
    """test/test_saddle_point.jl:43-63 (exact ==)."""
    lp = H.example_lp()
    pw = select_initial_primal_weight(lp, np.ones(4), np.ones(3), 1.0, 0)
    assert pw == np.sqrt(31.0) / np.sqrt(194.0) or abs(pw - np.sqrt(31.0 / 194.0)) < 1e-16
    lp.objective_vector = np.zeros(4)
    assert select_initial_primal_weight(lp, np.ones(4), np.ones(3), 1.0, 0) == 1.0
    assert select_initial_primal_weight(lp, np.ones(4), np.ones(3), 2.5, 0) == 2.5
    # the reference's own three cases, primal_importance 1.3 (:33-63)
    lp1 = H.example_lp()
    want = 1.3 * np.linalg.norm([5.0, 2.0, 1.0, 1.0]) / np.linalg.norm([12.0, 7.0, 1.0])
    assert abs(select_initial_primal_weight(lp1, np.ones(4), np.ones(3), 1.3, 0) - want) <= 1e-16
    lp3 = H.example_lp()
    lp3.right_hand_side = np.zeros(3)
    assert select_initial_primal_weight(lp3, np.ones(4), np.ones(3), 1.3, 0) == 1.3


def test_compute_lagrangian_value():
    """test/test_saddle_point.jl:66-74: zero point -> objective_constant."""
    lp = H.example_lp()
    assert compute_lagrangian_value(lp, np.zeros(4), np.zeros(3)) == -14.0
    qp = H.example_qp()
    assert compute_lagrangian_value(qp, np.zeros(2), np.zeros(1)) == 0.0
    assert compute_lagrangian_value(qp, np.array([1.0, 1.0]), np.array([0.0])) == 0.5      # :70-73
    assert compute_lagrangian_value(qp, np.array([1.0, 1.0]), np.array([1.0])) == 1.5
    assert compute_lagrangian_value(qp, np.array([0.25, 0.0]), np.array([0.0])) == -0.125
    # L(x,y) = c'x - y'(Ax) + b'y + const at the LP optimum equals the optimum value
    x, y = np.array([1.0, 0.0, 6.0, 2.0]), np.array([0.5, 4.0, 0.0])
    assert abs(compute_lagrangian_value(lp, x, y) - (-1.0)) < 1e-14


def test_residual_and_objective_helpers():
    """test/test_iteration_stats.jl:17-45 style checks on example_lp."""
    lp = H.example_lp()
    x_opt = np.array([1.0, 0.0, 6.0, 2.0])
    assert max_primal_violation(lp, x_opt) == 0.0
    assert primal_obj(lp, x_opt) == -1.0
    assert max_primal_violation(lp, np.zeros(4)) == 12.0
    ds = compute_dual_stats(lp, x_opt, np.array([0.5, 4.0, 0.0]))
    assert abs(ds.dual_objective - (-1.0)) < 1e-14
    assert np.max(np.abs(ds.dual_residual)) == 0.0


def test_partition_rows_helper_needs_no_gpu_and_matches_the_host_mirror():
    """pdhg_partition_rows (host-only export): the nnz-balanced contiguous row partition
    pdhg_create_dist / pdhg_create_multi use, which a host needs to slice a matrix for
    pdhg_create_dist_rows."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import HipPdhgEngine
    from firstorderlp_jl_amd.distributed import partition_rows, row_shard_of
    from firstorderlp_jl_amd.generators import random_lp
    A = sp.random(1000, 700, density=0.01, format="csc", random_state=1)
    for world in (1, 2, 3, 8):
        got = HipPdhgEngine.partition_rows(A, world)
        ref = partition_rows(A, world)
        assert list(got) == [ref[0][0]] + [hi for _, hi in ref]
    p = random_lp(300, 200, 5, seed=1)
    bounds = HipPdhgEngine.partition_rows(p.constraint_matrix, 3)
    total = 0
    for r in range(3):
        sh = row_shard_of(p, bounds, r)
        assert sh["constraint_rows"].shape == (bounds[r + 1] - bounds[r], 200)
        assert sh["right_hand_side_rows"].shape == (bounds[r + 1] - bounds[r],)
        total += sh["constraint_rows"].nnz
    assert total == p.constraint_matrix.nnz


def test_library_has_no_link_time_rccl_dependency():
    """RCCL is bound at run time (csrc/rccl_loader.hpp): the shared object must load where no
    librccl exists, so it may not carry a DT_NEEDED on it."""
    import subprocess
    from firstorderlp_jl_amd import _lib
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    if not out:
        import pytest
        pytest.skip("readelf not available")
    needed = [ln for ln in out.splitlines() if "NEEDED" in ln]
    assert needed and not any("rccl" in ln for ln in needed), needed


def test_julia_min_propagates_nan():
    import math
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import julia_min
    assert julia_min(1.0, 2.0) == 1.0 and julia_min(2.0, 1.0) == 1.0
    assert math.isnan(julia_min(math.nan, 1.0)) and math.isnan(julia_min(1.0, math.nan))
    assert julia_min(math.inf, 3.0) == 3.0


def test_take_steps_host_loop_stops_on_numerical_error():
    """take_steps with an engine that has no batched entry point: the plain loop, ending
    on the step that set numerical_error (what optimize() relies on between evaluations)."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (ConstantStepsizeParams,
                                                                 PdhgSolverState, take_steps)

    class Eng:
        calls = 0

        def trial_step(self, *a):
            Eng.calls += 1
            return [0.0] * 5

        def accept(self, w):
            pass

    st = PdhgSolverState(Eng(), step_size=0.5, primal_weight=1.0)
    assert take_steps(ConstantStepsizeParams(), st, 7) == 7 and Eng.calls == 7
    st.numerical_error = True
    assert take_steps(ConstantStepsizeParams(), st, 7) == 1
