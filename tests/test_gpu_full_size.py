"""BASELINE configs[4] at FULL size (m = n = 10M, nnz = 100M) on the GPU:
 * bit-exact A*x and A'*y against the oracle's sequential loops (tiled layout);
 * the adjoint identity <A x, y> = <x, A'y> tying K3 and K5 together;
 * linearity A(ax + bz) = a Ax + b Az to 1e-13 * |A||.|;
 * three adaptive PDHG steps: trial vectors bit-exact, scalars to 1e-12."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_config_s_full_size(gpu_required):
    m = n = 10_000_000
    p = random_lp(m, n, 10, 12345)
    A = p.constraint_matrix
    assert A.nnz == 100_000_000
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_tiled_waves"] > 0 and info["At_tiled_waves"] > 0   # the v2 layout is the one under test
    rng = np.random.default_rng(0)
    x, z, y = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(m)
    ax, aty = eng.spmv(x), eng.spmv_t(y)
    assert np.array_equal(ax, orc.spmv(m, n, A.indptr, A.indices, A.data, x))
    assert np.array_equal(aty, orc.spmv_t(m, n, A.indptr, A.indices, A.data, y))
    lhs, rhs = float(ax @ y), float(x @ aty)
    assert abs(lhs - rhs) <= 1e-10 * np.linalg.norm(ax) * np.linalg.norm(y)
    az = eng.spmv(z)
    comb = eng.spmv(0.7 * x - 1.3 * z)
    bound = 0.7 * np.abs(ax) + 1.3 * np.abs(az) + 1.0
    assert np.all(np.abs(comb - (0.7 * ax - 1.3 * az)) <= 1e-12 * bound)

    st = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    for it in range(3):
        raw = eng.trial_step(step, pw, 1.0)
        raw_o, xn, yn, an = st.trial_step(step, pw, 1.0)
        gx, gy, ga = eng.get_trial()
        assert np.array_equal(gx, xn) and np.array_equal(gy, yn) and np.array_equal(ga, an), it
        assert np.all(np.abs(raw[:4] - raw_o[:4]) <= 1e-12 * np.abs(raw_o[:4]) + 1e-300), (raw, raw_o)
        st.step_size = step
        st.accept(xn, yn, an)
        eng.accept(step)
        step *= 1.1
    xs, ys = eng.get_current()
    assert np.array_equal(xs, st.x) and np.array_equal(ys, st.y)
