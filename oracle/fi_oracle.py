"""ctypes front end of oracle/libfi_oracle.so (the C++ restatement in fi_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- see the header of fi_oracle.cpp for the parity status
("parity unpinned by reference tests") and the reference file:line citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VALUE_NEAREST, VALUE_LINEAR = 0, 1                      # field_interpolation.hpp:47-51
GRAD_NEAREST, GRAD_CELL_EDGES, GRAD_LINEAR = 0, 1, 2    # field_interpolation.hpp:54-59


class Weights(C.Structure):
    """field_interpolation.hpp:75-95 (same defaults)."""
    _fields_ = [("data_pos", C.c_float), ("data_gradient", C.c_float),
                ("model_0", C.c_float), ("model_1", C.c_float), ("model_2", C.c_float),
                ("model_3", C.c_float), ("model_4", C.c_float),
                ("gradient_smoothness", C.c_float),
                ("value_kernel", C.c_int), ("gradient_kernel", C.c_int)]

    def __init__(self, **kw):
        super().__init__()
        self.data_pos, self.data_gradient = 1.0, 1.0
        self.model_0 = self.model_1 = self.model_3 = self.model_4 = 0.0
        self.model_2 = 0.5
        self.gradient_smoothness = 0.0
        self.value_kernel, self.gradient_kernel = VALUE_LINEAR, GRAD_CELL_EDGES
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class SolveOptions(C.Structure):
    """sparse_linear.hpp:66-73 (same defaults)."""
    _fields_ = [("tile", C.c_int), ("tile_size", C.c_int), ("cg", C.c_int),
                ("max_iterations", C.c_int), ("error_tolerance", C.c_float)]

    def __init__(self, **kw):
        super().__init__()
        self.tile, self.tile_size, self.cg, self.max_iterations = 0, 16, 1, 0
        self.error_tolerance = 1e-3
        for k, v in kw.items():
            setattr(self, k, v)


def build(force=False):
    so = os.path.join(_HERE, "libfi_oracle.so")
    src = os.path.join(_HERE, "fi_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libfi_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)
        vp = C.c_void_p
        L.fio_field_new.restype = vp
        L.fio_field_new.argtypes = [C.c_int, ip]
        L.fio_field_free.argtypes = [vp]
        L.fio_num_rows.restype = C.c_long
        L.fio_num_rows.argtypes = [vp]
        L.fio_num_triplets.restype = C.c_long
        L.fio_num_triplets.argtypes = [vp]
        L.fio_get.argtypes = [vp, ip, ip, fp, fp]
        L.fio_push_triplet.argtypes = [vp, C.c_int, C.c_int, C.c_float]
        L.fio_push_rhs.argtypes = [vp, C.c_float]
        L.fio_add_equation.argtypes = [vp, C.c_float, C.c_float, C.c_int, ip, fp]
        L.fio_add_value_constraint.argtypes = [vp, fp, C.c_float, C.c_float]
        L.fio_add_value_constraints.argtypes = [vp, C.c_int, fp, fp, fp, C.c_float]
        L.fio_add_value_constraint_nearest_neighbor.argtypes = [vp, fp, fp, C.c_float, C.c_float]
        L.fio_add_gradient_constraint.argtypes = [vp, fp, fp, C.c_float, C.c_int]
        L.fio_add_field_constraints.argtypes = [vp, C.POINTER(Weights)]
        L.fio_add_points.argtypes = [vp, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, fp, fp, fp]
        L.fio_sdf_from_points.restype = vp
        L.fio_sdf_from_points.argtypes = [C.c_int, ip, C.POINTER(Weights), C.c_int, fp, fp, fp]
        L.fio_error_map.argtypes = [vp, C.c_long, fp, fp]
        L.fio_upscale_field.argtypes = [fp, C.c_int, ip, ip, fp]
        L.fio_solve_exact.argtypes = [vp, C.c_int, fp]
        L.fio_solve_fast.argtypes = [vp, C.c_int, fp]
        L.fio_solve_exact_f64.argtypes = [vp, C.c_int, dp]
        L.fio_solve_with_guess.argtypes = [vp, C.c_int, fp, C.c_int, C.c_float, fp, ip, fp]
        L.fio_solve_pcg.argtypes = [vp, C.c_int, fp, C.c_int, C.c_double, C.c_int, dp, ip, dp]
        L.fio_solve_pcg_f64_mt.argtypes = [vp, C.c_int, dp, C.c_int, C.c_double, C.c_int, C.c_int, dp, ip, dp]
        L.fio_solve_pcg_rows_omp.argtypes = [vp, C.c_int, fp, C.c_int, C.c_double, C.c_int, fp, ip, dp, dp]
        L.fio_jacobi_iterations.argtypes = [vp, C.c_int, fp, C.c_int, C.c_float, fp]
        L.fio_solve_tiled_with_guess.argtypes = [vp, C.c_long, fp, C.c_int, ip, C.POINTER(SolveOptions),
                                                 fp, ip, fp]
        L.fio_normal_equations_f64.restype = C.c_long
        L.fio_normal_equations_f64.argtypes = [vp, C.c_int, ip, ip, dp, dp, dp]
        L.fio_apply_normal_f64.argtypes = [vp, C.c_int, dp, dp]
        L.fio_apply_transpose_rhs_f64.argtypes = [vp, C.c_int, dp]
        _LIB = L
    return _LIB


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _d(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class LatticeField:
    """field_interpolation.hpp:97-114.  `.triplets()` / `.rhs()` expose `eq`."""

    def __init__(self, sizes, _handle=None):
        self.sizes = [int(s) for s in sizes]
        self._sz = np.asarray(self.sizes, dtype=np.int32)
        self._h = _handle if _handle is not None else lib().fio_field_new(len(self.sizes), _i(self._sz))
        if not self._h:
            raise ValueError("bad lattice")

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.fio_field_free(self._h)
            self._h = None

    @property
    def num_unknowns(self):
        return int(np.prod(self.sizes)) if self.sizes else 1

    @property
    def num_rows(self):
        return lib().fio_num_rows(self._h)

    @property
    def num_triplets(self):
        return lib().fio_num_triplets(self._h)

    def get(self):
        nt, nr = self.num_triplets, self.num_rows
        rows, cols = np.empty(nt, np.int32), np.empty(nt, np.int32)
        vals, rhs = np.empty(nt, np.float32), np.empty(nr, np.float32)
        lib().fio_get(self._h, _i(rows), _i(cols), _f(vals), _f(rhs))
        return rows, cols, vals, rhs

    # -- row builders -------------------------------------------------------------------------
    def push_triplet(self, row, col, v):
        lib().fio_push_triplet(self._h, row, col, v)

    def push_rhs(self, v):
        lib().fio_push_rhs(self._h, v)

    def add_equation(self, weight, rhs, pairs):
        cols = np.asarray([p[0] for p in pairs], np.int32)
        coef = np.asarray([p[1] for p in pairs], np.float32)
        lib().fio_add_equation(self._h, weight, rhs, len(pairs), _i(cols), _f(coef))

    def add_value_constraint(self, pos, value, weight):
        p = _f32(np.atleast_1d(pos))
        return bool(lib().fio_add_value_constraint(self._h, _f(p), value, weight))

    def add_value_constraints(self, positions, values, weight, point_weights=None):
        """add_value_constraint for every point, in order; returns how many were accepted."""
        pos, val, pw = _f32(positions), _f32(values), _f32(point_weights)
        return lib().fio_add_value_constraints(self._h, val.size, _f(pos), _f(val), _f(pw), weight)

    def add_value_constraint_nearest_neighbor(self, pos, gradient, value, weight):
        p, g = _f32(np.atleast_1d(pos)), _f32(np.atleast_1d(gradient))
        return bool(lib().fio_add_value_constraint_nearest_neighbor(self._h, _f(p), _f(g), value, weight))

    def add_gradient_constraint(self, pos, gradient, weight, kernel):
        p, g = _f32(np.atleast_1d(pos)), _f32(np.atleast_1d(gradient))
        r = lib().fio_add_gradient_constraint(self._h, _f(p), _f(g), weight, kernel)
        if r < 0:
            raise ValueError("Unknown gradient kernel")   # reference ABORT_F, cpp:238
        return bool(r)

    def add_field_constraints(self, weights):
        lib().fio_add_field_constraints(self._h, C.byref(weights))

    def add_points(self, value_weight, value_kernel, gradient_weight, gradient_kernel, positions,
                   normals=None, point_weights=None):
        pos = _f32(positions)
        nrm, pw = _f32(normals), _f32(point_weights)
        n = pos.size // max(1, len(self.sizes))
        if lib().fio_add_points(self._h, value_weight, value_kernel, gradient_weight, gradient_kernel, n,
                                _f(pos), _f(nrm), _f(pw)) != 0:
            raise ValueError("add_points: check failed (normals required / unknown kernel)")

    # -- post-processing ----------------------------------------------------------------------
    def error_map(self, solution):
        sol = _f32(solution)
        out = np.empty_like(sol)
        lib().fio_error_map(self._h, sol.size, _f(sol), _f(out))
        return out

    # -- solvers (None <=> the reference returns an empty vector) ------------------------------
    def solve_exact(self, ncols=None):
        n = ncols or self.num_unknowns
        out = np.empty(n, np.float32)
        return out if lib().fio_solve_exact(self._h, n, _f(out)) else None

    def solve_fast(self, ncols=None):
        n = ncols or self.num_unknowns
        out = np.empty(n, np.float32)
        return out if lib().fio_solve_fast(self._h, n, _f(out)) else None

    def solve_exact_f64(self, ncols=None):
        n = ncols or self.num_unknowns
        out = np.empty(n, np.float64)
        return out if lib().fio_solve_exact_f64(self._h, n, _d(out)) else None

    def solve_with_guess(self, guess, max_iterations=0, error_tolerance=0.0):
        g = _f32(guess)
        out = np.empty_like(g)
        it, err = C.c_int(0), C.c_float(0)
        ok = lib().fio_solve_with_guess(self._h, g.size, _f(g), max_iterations, error_tolerance, _f(out),
                                        C.byref(it), C.byref(err))
        return (out, it.value, err.value) if ok else None

    def solve_pcg(self, guess, max_iterations=0, tol=0.0, use_double=True):
        g = _f32(guess)
        out = np.empty(g.size, np.float64)
        it, err = C.c_int(0), C.c_double(0)
        ok = lib().fio_solve_pcg(self._h, g.size, _f(g), max_iterations, tol, int(use_double), _d(out),
                                 C.byref(it), C.byref(err))
        return (out, it.value, err.value) if ok else None

    def solve_pcg_f64_mt(self, guess=None, max_iterations=0, tol=0.0, threads=0, report=0):
        """The fp64 Jacobi-PCG of solve_pcg on `threads` cores (row gathers over the symmetric AtA: the iterates do not
        depend on the thread count), fp64 guess: golden solutions at benchmark sizes.  -> (x, iterations, residual)."""
        n = self.num_unknowns
        g = np.zeros(n, np.float64) if guess is None else np.ascontiguousarray(guess, np.float64)
        out = np.empty(n, np.float64)
        it, err = C.c_int(0), C.c_double(0)
        ok = lib().fio_solve_pcg_f64_mt(self._h, n, _d(g), max_iterations, tol, threads, report, _d(out),
                                        C.byref(it), C.byref(err))
        return (out, it.value, err.value) if ok else None

    def solve_pcg_rows_omp(self, guess, max_iterations=0, tol=0.0, threads=0):
        """Jacobi-PCG on A^T(A x) from the compressed rows / columns of A, OpenMP on `threads` cores (0: all): the
        "best-effort CPU" figure of bench.py -- not the reference's algorithm.  Returns (x, iterations, relative
        residual, seconds building the compressed forms, seconds iterating)."""
        g = _f32(guess)
        out = np.empty_like(g)
        it, err = C.c_int(0), C.c_double(0)
        sec = (C.c_double * 2)()
        ok = lib().fio_solve_pcg_rows_omp(self._h, g.size, _f(g), max_iterations, tol, threads, _f(out), C.byref(it),
                                          C.byref(err), sec)
        return (out, it.value, err.value, sec[0], sec[1]) if ok else None

    def jacobi_iterations(self, guess, num_iterations, weight):
        g = _f32(guess)
        out = np.empty_like(g)
        ok = lib().fio_jacobi_iterations(self._h, g.size, _f(g), num_iterations, weight, _f(out))
        return out if ok else None

    def solve_tiled_with_guess(self, guess, sizes, options):
        g = _f32(guess)
        sz = np.asarray(sizes, np.int32)
        out = np.empty(int(np.prod(sz)), np.float32)
        it, err = C.c_int(0), C.c_float(0)
        ok = lib().fio_solve_tiled_with_guess(self._h, g.size, _f(g), len(sz), _i(sz), C.byref(options),
                                              _f(out), C.byref(it), C.byref(err))
        return (out, it.value, err.value) if ok else None

    # -- operator extraction ------------------------------------------------------------------
    def normal_equations(self, ncols=None):
        """(AtA as scipy CSC float64, Atb, diag) with duplicates summed."""
        import scipy.sparse as sp
        n = ncols or self.num_unknowns
        nnz = lib().fio_normal_equations_f64(self._h, n, None, None, None, None, None)
        if nnz < 0:
            raise ValueError("index out of range")
        ptr, idx = np.empty(n + 1, np.int32), np.empty(nnz, np.int32)
        val, atb, diag = np.empty(nnz, np.float64), np.empty(n, np.float64), np.empty(n, np.float64)
        lib().fio_normal_equations_f64(self._h, n, _i(ptr), _i(idx), _d(val), _d(atb), _d(diag))
        return sp.csc_matrix((val, idx, ptr), shape=(n, n)), atb, diag

    def apply_normal(self, x, ncols=None):
        n = ncols or self.num_unknowns
        xx = np.ascontiguousarray(x, np.float64)
        y = np.empty(n, np.float64)
        if not lib().fio_apply_normal_f64(self._h, n, _d(xx), _d(y)):
            raise ValueError("index out of range")
        return y


    def apply_transpose_rhs(self, ncols=None):
        """A^T b in float64 from the rows (no compressed matrix)."""
        n = ncols or self.num_unknowns
        y = np.empty(n, np.float64)
        if not lib().fio_apply_transpose_rhs_f64(self._h, n, _d(y)):
            raise ValueError("index out of range")
        return y


def sdf_from_points(sizes, weights, positions, normals=None, point_weights=None):
    """field_interpolation.cpp:373-400."""
    sz = np.asarray(sizes, np.int32)
    pos, nrm, pw = _f32(positions), _f32(normals), _f32(point_weights)
    n = pos.size // len(sz)
    h = lib().fio_sdf_from_points(len(sz), _i(sz), C.byref(weights), n, _f(pos), _f(nrm), _f(pw))
    if not h:
        raise ValueError("sdf_from_points: check failed")
    return LatticeField(sizes, _handle=h)


def upscale_field(small, small_sizes, large_sizes):
    """field_interpolation.cpp:431-485."""
    s = _f32(small)
    ss, ls = np.asarray(small_sizes, np.int32), np.asarray(large_sizes, np.int32)
    out = np.empty(int(np.prod(ls)), np.float32)
    lib().fio_upscale_field(_f(s), len(ss), _i(ss), _i(ls), _f(out))
    return out
