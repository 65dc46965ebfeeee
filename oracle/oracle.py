"""ctypes wrapper over oracle/libpdhg_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY -- see the header of pdhg_oracle.c.  Nothing under
``firstorderlp.jl_amd/`` imports this module; it is loaded by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpdhg_oracle.so")

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_i64_p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pdhg_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpdhg_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_create.restype = ctypes.c_void_p
        for name in ("step_size", "primal_weight", "ratio_step_sizes",
                     "cumulative_kkt_passes"):
            getattr(_lib, "oracle_get_" + name).restype = ctypes.c_double
            getattr(_lib, "oracle_get_" + name).argtypes = [ctypes.c_void_p]
            getattr(_lib, "oracle_set_" + name).argtypes = [ctypes.c_void_p,
                                                            ctypes.c_double]
        _lib.oracle_get_numerical_error.restype = ctypes.c_int
        _lib.oracle_get_numerical_error.argtypes = [ctypes.c_void_p]
        _lib.oracle_set_numerical_error.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.oracle_set_exact_sums.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.oracle_get_exact_sums.restype = ctypes.c_int
        _lib.oracle_get_exact_sums.argtypes = [ctypes.c_void_p]
        _lib.oracle_get_total_number_iterations.restype = ctypes.c_int64
        _lib.oracle_get_total_number_iterations.argtypes = [ctypes.c_void_p]
        _lib.oracle_take_step_adaptive.argtypes = [ctypes.c_void_p,
                                                   ctypes.c_double,
                                                   ctypes.c_double]
        _lib.oracle_take_step_constant.argtypes = [ctypes.c_void_p]
        _lib.oracle_take_step_malitsky_pock.argtypes = [ctypes.c_void_p] + \
            [ctypes.c_double] * 3
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _dp(a):
    return a.ctypes.data_as(_c_double_p)


def _ip(a):
    return a.ctypes.data_as(_c_i64_p)


def spmv(m, n, colptr, rowval, nzval, x):
    """A*x, Julia SparseMatrixCSC order (0-based CSC arrays)."""
    colptr, rowval, nzval, x = _i(colptr), _i(rowval), _d(nzval), _d(x)
    out = np.empty(m, dtype=np.float64)
    lib().oracle_spmv(ctypes.c_int64(m), ctypes.c_int64(n), _ip(colptr),
                      _ip(rowval), _dp(nzval), _dp(x), _dp(out))
    return out


def spmv_t(m, n, colptr, rowval, nzval, y):
    """A'*y, Julia Adjoint{SparseMatrixCSC} order (0-based CSC arrays)."""
    colptr, rowval, nzval, y = _i(colptr), _i(rowval), _d(nzval), _d(y)
    out = np.empty(n, dtype=np.float64)
    lib().oracle_spmv_t(ctypes.c_int64(m), ctypes.c_int64(n), _ip(colptr),
                        _ip(rowval), _dp(nzval), _dp(y), _dp(out))
    return out


class OracleState:
    """PdhgSolverState + problem held by the C oracle.

    Arguments are the *already rescaled* problem, CSC 0-based: the same data
    the product's C-ABI ``pdhg_create`` ingests.
    """

    def __init__(self, m, n, colptr, rowval, nzval, c, b, lb, ub,
                 num_equalities, q_colptr=None, q_rowval=None, q_nzval=None):
        L = lib()
        self.m, self.n = int(m), int(n)
        colptr, rowval, nzval = _i(colptr), _i(rowval), _d(nzval)
        c, b, lb, ub = _d(c), _d(b), _d(lb), _d(ub)
        assert colptr.shape == (n + 1,) and c.shape == (n,) and b.shape == (m,)
        if q_colptr is None:
            q_colptr = np.zeros(n + 1, dtype=np.int64)
            q_rowval = np.zeros(0, dtype=np.int64)
            q_nzval = np.zeros(0, dtype=np.float64)
        q_colptr, q_rowval, q_nzval = _i(q_colptr), _i(q_rowval), _d(q_nzval)
        self._h = ctypes.c_void_p(L.oracle_create(
            ctypes.c_int64(m), ctypes.c_int64(n), _ip(colptr), _ip(rowval),
            _dp(nzval), ctypes.c_int64(len(q_nzval)), _ip(q_colptr),
            _ip(q_rowval), _dp(q_nzval), _dp(c), _dp(b), _dp(lb), _dp(ub),
            ctypes.c_int64(num_equalities)))
        self._L = L

    def close(self):
        if self._h:
            self._L.oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- vectors ------------------------------------------------------------
    def _get(self, name, length):
        out = np.empty(length, dtype=np.float64)
        getattr(self._L, "oracle_get_" + name)(self._h, _dp(out))
        return out

    def _set(self, name, arr, length):
        arr = _d(arr)
        assert arr.shape == (length,)
        getattr(self._L, "oracle_set_" + name)(self._h, _dp(arr))

    x = property(lambda s: s._get("x", s.n), lambda s, v: s._set("x", v, s.n))
    y = property(lambda s: s._get("y", s.m), lambda s, v: s._set("y", v, s.m))
    aty = property(lambda s: s._get("aty", s.n),
                   lambda s, v: s._set("aty", v, s.n))
    delta_x = property(lambda s: s._get("delta_x", s.n))
    delta_y = property(lambda s: s._get("delta_y", s.m))
    sum_x = property(lambda s: s._get("sum_x", s.n))
    sum_y = property(lambda s: s._get("sum_y", s.m))
    x_next = property(lambda s: s._get("x_next", s.n))
    y_next = property(lambda s: s._get("y_next", s.m))
    aty_next = property(lambda s: s._get("aty_next", s.n))

    # -- scalars ------------------------------------------------------------
    step_size = property(lambda s: s._L.oracle_get_step_size(s._h),
                         lambda s, v: s._L.oracle_set_step_size(s._h, float(v)))
    primal_weight = property(
        lambda s: s._L.oracle_get_primal_weight(s._h),
        lambda s, v: s._L.oracle_set_primal_weight(s._h, float(v)))
    ratio_step_sizes = property(
        lambda s: s._L.oracle_get_ratio_step_sizes(s._h),
        lambda s, v: s._L.oracle_set_ratio_step_sizes(s._h, float(v)))
    cumulative_kkt_passes = property(
        lambda s: s._L.oracle_get_cumulative_kkt_passes(s._h),
        lambda s, v: s._L.oracle_set_cumulative_kkt_passes(s._h, float(v)))
    numerical_error = property(
        lambda s: bool(s._L.oracle_get_numerical_error(s._h)),
        lambda s, v: s._L.oracle_set_numerical_error(s._h, int(bool(v))))
    # test aid (not the reference's arithmetic): the three step-acceptance sums in double-double,
    # i.e. exactly rounded and independent of the order of the additions -- what the HIP library does
    exact_sums = property(lambda s: bool(s._L.oracle_get_exact_sums(s._h)),
                          lambda s, v: s._L.oracle_set_exact_sums(s._h, int(bool(v))))
    total_number_iterations = property(
        lambda s: int(s._L.oracle_get_total_number_iterations(s._h)))

    def average_counts(self):
        counts = np.zeros(2, dtype=np.int64)
        weights = np.zeros(2, dtype=np.float64)
        self._L.oracle_get_average_counts(self._h, _ip(counts), _dp(weights))
        return int(counts[0]), int(counts[1]), float(weights[0]), float(weights[1])

    # -- hot path pieces ------------------------------------------------------
    def trial_primal(self, step_size, primal_weight):
        """compute_next_primal_solution into the state's x_next."""
        xn = np.empty(self.n)
        self._L.oracle_compute_next_primal(self._h, ctypes.c_double(step_size),
                                           ctypes.c_double(primal_weight), _dp(xn))
        self._set("x_next", xn, self.n)
        return xn

    def trial_dual(self, step_size, primal_weight, theta):
        """compute_next_dual_solution from the stored x_next; returns the raw
        sums (same 5 entries as the C-ABI's out[5]) and the trial vectors."""
        L, h = self._L, self._h
        xn = self.x_next
        yn = np.empty(self.m)
        an = np.empty(self.n)
        L.oracle_compute_next_dual(h, _dp(xn), ctypes.c_double(step_size),
                                   ctypes.c_double(primal_weight),
                                   ctypes.c_double(theta), _dp(yn), _dp(an))
        inter = ctypes.c_double()
        move = ctypes.c_double()
        raw = np.empty(5)
        L.oracle_interaction_and_movement(h, _dp(xn), _dp(yn), _dp(an),
                                          ctypes.byref(inter),
                                          ctypes.byref(move), _dp(raw))
        self._set("y_next", yn, self.m)
        self._set("aty_next", an, self.n)
        return raw, xn, yn, an

    def trial_step(self, step_size, primal_weight, theta=1.0):
        """One full trial (primal then dual)."""
        self.trial_primal(step_size, primal_weight)
        return self.trial_dual(step_size, primal_weight, theta)

    def accept(self, xn, yn, an):
        self._L.oracle_update_solution(self._h, _dp(_d(xn)), _dp(_d(yn)),
                                       _dp(_d(an)))

    def take_step_adaptive(self, reduction_exponent=0.3, growth_exponent=0.6):
        return self._L.oracle_take_step_adaptive(self._h, reduction_exponent,
                                                 growth_exponent)

    def take_step_constant(self):
        return self._L.oracle_take_step_constant(self._h)

    def take_step_malitsky_pock(self, downscaling_factor, breaking_factor,
                                interpolation_coefficient):
        return self._L.oracle_take_step_malitsky_pock(
            self._h, downscaling_factor, breaking_factor,
            interpolation_coefficient)

    def add_to_primal_average(self, x, weight):
        self._L.oracle_add_to_primal_average(self._h, _dp(_d(x)),
                                             ctypes.c_double(weight))

    def reset_average(self):
        self._L.oracle_reset_average(self._h)

    def compute_average(self):
        xa = np.empty(self.n)
        ya = np.empty(self.m)
        self._L.oracle_compute_average(self._h, _dp(xa), _dp(ya))
        return xa, ya

    def recompute_dual_product(self):
        self._L.oracle_recompute_dual_product(self._h)


# ---- multi-threaded CPU comparator (pdhg_cpu_omp.c): measurement only -------------

_OMP_PATH = os.path.join(_HERE, "libpdhg_cpu_omp.so")
_omp = None


def build_omp(force=False):
    src = os.path.join(_HERE, "pdhg_cpu_omp.c")
    if force or not os.path.exists(_OMP_PATH) or os.path.getmtime(_OMP_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpdhg_cpu_omp.so"], stdout=subprocess.DEVNULL)
    return _OMP_PATH


def omp_lib():
    global _omp
    if _omp is None:
        build_omp()
        _omp = ctypes.CDLL(_OMP_PATH)
        _omp.omp_create.restype = ctypes.c_void_p
        _omp.omp_get_step_size.restype = ctypes.c_double
        _omp.omp_get_step_size.argtypes = [ctypes.c_void_p]
        _omp.omp_get_total_iterations.restype = ctypes.c_int64
        _omp.omp_get_total_iterations.argtypes = [ctypes.c_void_p]
        _omp.omp_set_scalars.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
        _omp.omp_take_step_adaptive.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
        _omp.omp_destroy.argtypes = [ctypes.c_void_p]
        _omp.omp_get_xy.argtypes = [ctypes.c_void_p, _c_double_p, _c_double_p]
        _omp.omp_set_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _omp.omp_index_bytes.argtypes = [ctypes.c_void_p]
        _omp.omp_bytes_per_trial.argtypes = [ctypes.c_void_p]
        _omp.omp_bytes_per_trial.restype = ctypes.c_double
    return _omp


class OmpCpuState:
    """OpenMP adaptive-step PDHG on an LP (CSC 0-based arrays); bench comparator."""

    def __init__(self, m, n, colptr, rowval, nzval, c, b, lb, ub, num_equalities, cpus=None):
        L = omp_lib()
        if cpus:
            arr = (ctypes.c_int * len(cpus))(*cpus)
            L.omp_configure(len(cpus), arr)
        self.m, self.n = int(m), int(n)
        colptr, rowval, nzval = _i(colptr), _i(rowval), _d(nzval)
        c, b, lb, ub = _d(c), _d(b), _d(lb), _d(ub)
        self._h = ctypes.c_void_p(L.omp_create(
            ctypes.c_int64(m), ctypes.c_int64(n), _ip(colptr), _ip(rowval), _dp(nzval),
            _dp(c), _dp(b), _dp(lb), _dp(ub), ctypes.c_int64(num_equalities)))
        self._L = L

    def threads(self):
        return int(self._L.omp_threads())

    def set_prefetch(self, entries_ahead):
        self._L.omp_set_prefetch(self._h, int(entries_ahead))

    def index_bytes(self):
        return int(self._L.omp_index_bytes(self._h))

    def bytes_per_trial(self):
        """Bytes one trial streams at the least (both matrix copies once + SURVEY 8d's vector traffic)."""
        self._L.omp_bytes_per_trial.restype = ctypes.c_double
        return float(self._L.omp_bytes_per_trial(self._h))

    def set_scalars(self, step_size, primal_weight):
        self._L.omp_set_scalars(self._h, step_size, primal_weight)

    def take_step_adaptive(self, reduction_exponent=0.3, growth_exponent=0.6):
        return self._L.omp_take_step_adaptive(self._h, reduction_exponent, growth_exponent)

    step_size = property(lambda s: s._L.omp_get_step_size(s._h))
    # test aid (not the reference's arithmetic): the three step-acceptance sums in double-double,
    # i.e. exactly rounded and independent of the order of the additions -- what the HIP library does
    exact_sums = property(lambda s: bool(s._L.oracle_get_exact_sums(s._h)),
                          lambda s, v: s._L.oracle_set_exact_sums(s._h, int(bool(v))))
    total_number_iterations = property(lambda s: int(s._L.omp_get_total_iterations(s._h)))

    def xy(self):
        x, y = np.empty(self.n), np.empty(self.m)
        self._L.omp_get_xy(self._h, _dp(x), _dp(y))
        return x, y

    def close(self):
        if self._h:
            self._L.omp_destroy(self._h)
            self._h = None
