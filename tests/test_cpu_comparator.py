"""The OpenMP CPU comparator of bench.py's cpu_baseline leg (oracle/pdhg_cpu_omp.c: measurement infrastructure) against
the literal single-thread oracle: the same adaptive trajectory (pdhg.jl:653-731) to 1e-9 -- only the order of the three
reductions differs -- with 64-bit and 32-bit indices, nnz-balanced row cuts on a skewed matrix, prefetch on and off."""
import numpy as np
import pytest

from firstorderlp_jl_amd.generators import random_lp
from oracle.oracle import OmpCpuState, OracleState
from tests import helpers as H


def _args(p):
    A = p.constraint_matrix
    return (A.shape[0], A.shape[1], A.indptr, A.indices, A.data, p.objective_vector, p.right_hand_side,
            p.variable_lower_bound, p.variable_upper_bound, p.num_equalities)


@pytest.mark.parametrize("problem", ["random", "skewed"])
@pytest.mark.parametrize("idx64,prefetch", [("1", 0), ("0", 0), ("0", 16)])
def test_openmp_comparator_follows_the_oracle(monkeypatch, problem, idx64, prefetch):
    monkeypatch.setenv("PDHG_CPU_IDX64", idx64)
    p = random_lp(3000, 2500, 7, seed=3) if problem == "random" else H.skewed_lp(2000, 4000, seed=5, dense_rows=3, dense_cols=3)
    step, pw = H.initial_step_and_weight(p)
    st = OracleState(*_args(p))
    st.step_size, st.primal_weight = step, pw
    om = OmpCpuState(*_args(p), cpus=[0, 1, 2])
    assert om.index_bytes() == (8 if idx64 == "1" else 4) and om.threads() == 3
    om.set_scalars(step, pw)
    om.set_prefetch(prefetch)
    for _ in range(40):
        st.take_step_adaptive(0.3, 0.6)
        om.take_step_adaptive(0.3, 0.6)
    assert om.total_number_iterations == st.total_number_iterations
    assert abs(om.step_size - st.step_size) <= 1e-9 * st.step_size
    x, y = om.xy()
    np.testing.assert_allclose(x, st.x, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(y, st.y, rtol=1e-9, atol=1e-9)
    A = p.constraint_matrix
    assert om.bytes_per_trial() == 2 * A.nnz * (8 + om.index_bytes()) + 8 * (A.shape[0] + A.shape[1] + 2) + 8 * (13 * A.shape[1] + 6 * A.shape[0])
    om.close()
    st.close()
