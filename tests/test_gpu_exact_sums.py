"""Free-running trajectories against the CPU restatement, bitwise.

Everything a trial writes is bit-exact with the oracle GIVEN the same step sizes (elementwise
updates, SpMV rows of <= 2048 entries in strict row order, the running averages).  What used to
differ were the three step-acceptance sums (pdhg.jl:527-549: the reference takes them with BLAS
dot / nrm2, whose order is implementation-defined; the oracle adds sequentially, the kernels in
trees), and the discontinuous step-size rule amplified those last-bit differences: iterates
agreed to 1e-10 for 10-60 free-running steps only.  The library now accumulates these sums in
double-double (csrc/common.hpp: Acc3, dd_add) -- the correctly rounded exact sum of the same terms,
independent of the order of the additions -- and the oracle has an exact-sums mode that adds the
same terms the same way (a test aid, oracle/pdhg_oracle.c::oracle_set_exact_sums).  With it both
sides compute bitwise the same scalars, hence take the same decisions and produce the same bits,
for as long as one cares to run: north_star's "same iterates within a stated fp64 tolerance" holds
free-running at tolerance ZERO on these problems."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, MalitskyPockStepsizeParameters,
                                                             PdhgSolverState, take_step)
from tests import helpers as H
from tests.test_gpu_netlib_like import netlib_like

pytestmark = pytest.mark.gpu

MAKERS = {
    # configs[1] SUBSTITUTES: seeded LPs with the shape of Netlib afiro / adlittle (the files are not available offline)
    "afiro_like": lambda: netlib_like(27, 32, 83, 8, seed=27),
    "adlittle_like": lambda: netlib_like(56, 97, 383, 15, seed=56),
    "random_5000": lambda: random_lp(5000, 4000, 8, seed=7),
    "random_200k_sweep": lambda: random_lp(200_000, 180_000, 9, seed=3),
    "pagerank_20000": lambda: pagerank_lp(20000, seed=2),           # dense row beyond 2048 entries: tolerance below
    "example_lp": lambda: H.example_lp(),
}


def _gpu_run(p, steps, policy, monkeypatch, path):
    monkeypatch.setenv("PDHG_ROW_ORDER", "strict")
    monkeypatch.setenv("PDHG_GRAPH", "0" if path == "plain" else "1")
    monkeypatch.setenv("PDHG_COOP", "1" if path == "one_kernel" else "0")
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw, ratio_step_sizes=1.0)
    sizes = []
    for _ in range(steps):
        take_step(policy, st)
        sizes.append(st.step_size)
    x, y = eng.get_current()
    xa, ya = eng.get_average()
    eng.close()
    return np.array(sizes), x, y, xa, ya, st.total_number_iterations


def _oracle_run(p, steps, policy):
    o = H.oracle_from_problem(p)
    o.exact_sums = True
    step, pw = H.initial_step_and_weight(p)
    o.step_size, o.primal_weight, o.ratio_step_sizes = step, pw, 1.0
    sizes = []
    for _ in range(steps):
        if isinstance(policy, AdaptiveStepsizeParams):
            o.take_step_adaptive(policy.reduction_exponent, policy.growth_exponent)
        else:
            o.take_step_malitsky_pock(policy.downscaling_factor, policy.breaking_factor, policy.interpolation_coefficient)
        sizes.append(o.step_size)
    xa, ya = o.compute_average()
    return np.array(sizes), o.x, o.y, xa, ya, o.total_number_iterations


@pytest.mark.parametrize("path", ["plain", "one_kernel"])
@pytest.mark.parametrize("name", sorted(MAKERS))
def test_free_running_adaptive_trajectory_is_bitwise_the_oracles(gpu_required, monkeypatch, name, path):
    p = MAKERS[name]()
    steps = 1000 if p.constraint_matrix.nnz < 100_000 else 300
    pol = AdaptiveStepsizeParams(0.3, 0.6)
    g = _gpu_run(p, steps, pol, monkeypatch, path)
    o = _oracle_run(p, steps, pol)
    long_rows = max(np.diff(p.constraint_matrix.indptr).max(), np.diff(p.constraint_matrix.tocsr().indptr).max()) > 2048
    if long_rows:
        # rows beyond 2048 entries are summed in chunks (1e-13 * sum |a x| per product): same decisions for a while
        k = 40
        g, o = _gpu_run(p, k, pol, monkeypatch, path), _oracle_run(p, k, pol)
        assert g[5] == o[5]
        for a, b in zip(g[:5], o[:5]):
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
        return
    assert g[5] == o[5], "accept / reject decisions differ"
    for a, b in zip(g[:5], o[:5]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["afiro_like", "random_5000"])
def test_free_running_malitsky_pock_trajectory_is_bitwise_the_oracles(gpu_required, monkeypatch, name):
    p = MAKERS[name]()
    pol = MalitskyPockStepsizeParameters(downscaling_factor=0.7, breaking_factor=0.99, interpolation_coefficient=1.0)
    g = _gpu_run(p, 400, pol, monkeypatch, "one_kernel")
    o = _oracle_run(p, 400, pol)
    assert g[5] == o[5]
    for a, b in zip(g[:5], o[:5]):
        assert np.array_equal(a, b)


def test_the_sums_do_not_depend_on_the_layout(gpu_required, monkeypatch):
    """Exactly rounded sums: the stream layout, the sweep at two tile widths and a shard-free plain path give the
    same five scalars, bit for bit (they used to differ in the last bits with the order of the block partials)."""
    p = random_lp(60_000, 50_000, 8, seed=5)
    step, pw = H.initial_step_and_weight(p)
    outs = []
    for env in ({"PDHG_SPMV": "stream"}, {"PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "4096"},
                {"PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "16384"}, {"PDHG_SPMV": "stream", "PDHG_GRAPH": "0"}):
        for k in ("PDHG_SPMV", "PDHG_TILE_COLS", "PDHG_GRAPH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = HipPdhgEngine.from_problem(p)
        raws = []
        for _ in range(5):
            raws.append(eng.trial_step(step, pw, 1.0).copy())
            eng.accept(step)
        outs.append(np.array(raws))
        eng.close()
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
