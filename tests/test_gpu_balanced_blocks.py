"""Row blocks of equal cost (layout.hpp: balanced_block_target, opt-in with PDHG_BALANCED_BLOCKS=1): the same rows in
the same order with other block boundaries.  Row sums keep their bits and the block partials are exactly rounded
double-double sums, so whole trajectories must be bitwise those of the greedy (filled) blocks on every launch path."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(p, monkeypatch, balanced, device_loop):
    monkeypatch.setenv("PDHG_BALANCED_BLOCKS", "1" if balanced else "0")
    monkeypatch.setenv("PDHG_BALANCED_CUS", "40")       # ~120 greedy blocks: between one and four per "CU"
    monkeypatch.setenv("PDHG_DEVICE_LOOP", device_loop)
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    sizes = []
    for k in (7, 30, 30):
        assert take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, k) == k
        sizes.append(st.step_size)
    x, y = eng.get_current()
    xa, ya = eng.get_average()
    raw = eng.trial_step(st.step_size, pw, 1.0)
    eng.close()
    return info, (np.array(sizes), x, y, xa, ya, np.array(raw), st.total_number_iterations)


@pytest.mark.parametrize("device_loop", ["0", "1"], ids=["launch_per_trial", "multi_step_kernel"])
def test_equal_cost_blocks_are_bitwise_the_filled_blocks(gpu_required, monkeypatch, device_loop):
    p = random_lp(30000, 25000, 8, seed=3)
    info0, ref = _run(p, monkeypatch, False, device_loop)
    info1, got = _run(p, monkeypatch, True, device_loop)
    assert info1["A_blocks"] > info0["A_blocks"] and info1["A_blocks"] % 8 == 0      # the cut happened: more, smaller blocks
    assert info1["At_blocks"] > info0["At_blocks"]
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k
