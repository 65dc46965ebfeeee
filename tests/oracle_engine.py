"""The CPU oracle behind the engine interface (``HipPdhgEngine``'s methods),
so that the host driver's control logic -- optimize(), restarts, termination,
the row-partitioned exchange -- can be exercised on a box without a GPU.

TEST INFRASTRUCTURE: lives under tests/, is injected through
``optimize(..., engine_factory=...)`` and never ships in the product path."""
import numpy as np

from oracle import oracle as orc


class OracleEngine:
    def __init__(self, constraint_matrix, objective_vector, right_hand_side,
                 variable_lower_bound, variable_upper_bound, num_equalities,
                 objective_matrix=None):
        A = constraint_matrix
        self.m, self.n = A.shape
        self._A = A
        self._Q = None
        q = (None, None, None)
        if objective_matrix is not None and objective_matrix.nnz > 0:
            Q = objective_matrix
            self._Q = Q
            q = (Q.indptr, Q.indices, Q.data)
        self.st = orc.OracleState(self.m, self.n, A.indptr, A.indices, A.data,
                                  objective_vector, right_hand_side,
                                  variable_lower_bound, variable_upper_bound,
                                  num_equalities, *q)
        self._trial = None
        self._exchange = np.zeros(self.n + 1)
        self._dist_pending = None

    @classmethod
    def from_problem(cls, p):
        return cls(p.constraint_matrix, p.objective_vector, p.right_hand_side,
                   p.variable_lower_bound, p.variable_upper_bound,
                   p.num_equalities, p.objective_matrix)

    def close(self):
        self.st.close()

    # ---- hot path ----
    def trial_step(self, step_size, primal_weight, theta=1.0):
        raw, xn, yn, an = self.st.trial_step(step_size, primal_weight, theta)
        self._trial = (xn, yn, an)
        return raw

    def trial_primal(self, step_size, primal_weight):
        self.st.trial_primal(step_size, primal_weight)

    def trial_dual(self, step_size, primal_weight, theta):
        raw, xn, yn, an = self.st.trial_dual(step_size, primal_weight, theta)
        self._trial = (xn, yn, an)
        return raw

    def accept(self, avg_weight):
        # oracle_update_solution reads the weight from state.step_size (Q1)
        self.st.step_size = avg_weight
        self.st.accept(*self._trial)

    def add_current_primal_to_average(self, weight):
        self.st.add_to_primal_average(self.st.x, weight)

    # ---- average / restart ----
    def average_info(self):
        return self.st.average_counts()

    def get_average(self):
        return self.st.compute_average()

    def reset_average(self):
        self.st.reset_average()

    def restart_to_average(self):
        xa, ya = self.st.compute_average()
        self.st.x, self.st.y = xa, ya
        self.st.recompute_dual_product()

    # ---- iterate I/O ----
    def get_current(self):
        return self.st.x, self.st.y

    def get_dual_product(self):
        return self.st.aty

    def set_current(self, x=None, y=None):
        if x is not None:
            self.st.x = x
        if y is not None:
            self.st.y = y
        self.st.recompute_dual_product()

    def get_trial(self):
        return self._trial

    def spmv(self, x):
        A = self._A
        return orc.spmv(self.m, self.n, A.indptr, A.indices, A.data, x)

    def spmv_t(self, y):
        A = self._A
        return orc.spmv_t(self.m, self.n, A.indptr, A.indices, A.data, y)

    # ---- row-partitioned form (same contract as include/pdhg_hip.h) ----
    def dist_trial_begin(self, step_size, primal_weight, theta=1.0):
        raw, xn, yn, an = self.st.trial_step(step_size, primal_weight, theta)
        self._exchange[:self.n] = an          # local partial A_p' y'_p
        self._exchange[self.n] = raw[2]       # local sum dy_p^2
        self._dist_pending = (xn, yn)

    def dist_trial_dual_begin(self, step_size, primal_weight, theta):
        raw, xn, yn, an = self.st.trial_dual(step_size, primal_weight, theta)
        self._exchange[:self.n] = an
        self._exchange[self.n] = raw[2]
        self._dist_pending = (xn, yn)

    def exchange_array(self):
        """t_p = A_p' y'_p, n doubles, reduced in place by the caller."""
        return self._exchange[:self.n]

    def dist_trial_end_slice(self, lo, hi):
        """After the exchange: interaction/movement sums over the OWNED column
        slice [lo, hi) plus this shard's rows (sum dy^2); the QP term acts on
        the replicated full vectors and is identical on every rank."""
        xn, yn = self._dist_pending
        an = self._exchange[:self.n].copy()
        dx = xn - self.st.x
        dd = an - self.st.aty
        qterm = 0.5 * float(dx @ (self._Q @ dx)) if self._Q is not None else 0.0
        sl = slice(lo, hi)
        raw = np.array([float(dx[sl] @ dd[sl]), float(dx[sl] @ dx[sl]), float(self._exchange[self.n]),
                        float(dd[sl] @ dd[sl]), qterm])
        self._trial = (xn, yn, an)
        return raw

    def dist_dual_product_begin(self):
        self.st.recompute_dual_product()
        self._exchange[:self.n] = self.st.aty
        self._exchange[self.n] = 0.0

    def dist_dual_product_end(self):
        self.st.aty = self._exchange[:self.n].copy()
