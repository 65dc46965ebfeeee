#!/usr/bin/env python3
"""Generates tests/golden/*.npz: per-iteration PDHG iterates from the CPU oracle
(oracle/pdhg_oracle.c) AFTER it passed every reference KAT
(tests/test_kat_oracle.py).  The reference itself is Julia and cannot run in
this environment, so these are oracle-generated golden vectors, not
reference-generated ones ("oracle").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import folp_loader  # noqa: E402

folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp  # noqa: E402
from tests import helpers as H  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K = 60

CASES = {
    "example_lp": (H.example_lp, "adaptive"),
    "example_cc_lp": (H.example_cc_lp, "adaptive"),
    "example_lp_without_bounds": (H.example_lp_without_bounds, "adaptive"),
    "random_lp_40x30_seed3": (lambda: random_lp(40, 30, 4, seed=3), "adaptive"),
    "random_lp_40x30_seed3_mp": (lambda: random_lp(40, 30, 4, seed=3), "malitsky-pock"),
    "random_lp_40x30_seed3_const": (lambda: random_lp(40, 30, 4, seed=3), "constant"),
}


def run(maker, policy):
    p = maker()
    st = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    if policy == "constant":
        step *= 0.1
    st.step_size, st.primal_weight, st.ratio_step_sizes = step, pw, 1.0
    xs, ys, atys, steps, trials = [], [], [], [], []
    for _ in range(K):
        if st.numerical_error:
            break
        if policy == "adaptive":
            st.take_step_adaptive(0.3, 0.6)
        elif policy == "constant":
            st.take_step_constant()
        else:
            st.take_step_malitsky_pock(0.7, 0.99, 1.0)
        xs.append(st.x); ys.append(st.y); atys.append(st.aty)
        steps.append(st.step_size); trials.append(st.total_number_iterations)
    xa, ya = st.compute_average()
    return dict(x=np.array(xs), y=np.array(ys), aty=np.array(atys),
                step_size=np.array(steps), total_number_iterations=np.array(trials),
                x_avg=xa, y_avg=ya, initial_step_size=step, primal_weight=pw)


if __name__ == "__main__":
    for name, (maker, policy) in CASES.items():
        np.savez(os.path.join(HERE, name + ".npz"), **run(maker, policy))
        print("wrote", name)
