"""Host-side mirror of the reference's C++ interface for the hot path, over the C ABI.

Names, argument meaning, defaults and error behaviour follow
  field_interpolation/field_interpolation.hpp:44-183 and field_interpolation/sparse_linear.hpp:8-80
so the parity tests read like calls into the reference.  Differences that the matrix-free design
forces are stated where they occur:
  * `LatticeField.eq` (the materialised triplet list, hpp:99) does not exist: rows go straight to
    the GPU, so the solver functions take the field instead of `field.eq`;
  * solvers return None where the reference returns an empty vector (sparse_linear.cpp:137-149,
    169-181, 402-405).
Buffers may be numpy arrays (host) or torch CUDA tensors (device pointers are passed through).
"""
import ctypes as C
import enum
import math

import numpy as np

from . import _capi
from ._capi import FI_DEVICE, FI_F32, FI_F64, FI_HOST, FiError, FiStats, FiWeights, check

MAX_DIM = 3  # field_interpolation.hpp:44


class ValueKernel(enum.IntEnum):      # field_interpolation.hpp:47-51
    kNearestNeighbor = 0
    kLinearInterpolation = 1


class GradientKernel(enum.IntEnum):   # field_interpolation.hpp:54-59
    kNearestNeighbor = 0
    kCellEdges = 1
    kLinearInterpolation = 2


class Weights:
    """field_interpolation.hpp:75-95, same field names and defaults."""

    def __init__(self, data_pos=1.0, data_gradient=1.0, model_0=0.0, model_1=0.0, model_2=0.5, model_3=0.0,
                 model_4=0.0, gradient_smoothness=0.0, value_kernel=ValueKernel.kLinearInterpolation,
                 gradient_kernel=GradientKernel.kCellEdges):
        self.data_pos, self.data_gradient = data_pos, data_gradient
        self.model_0, self.model_1, self.model_2 = model_0, model_1, model_2
        self.model_3, self.model_4 = model_3, model_4
        self.gradient_smoothness = gradient_smoothness
        self.value_kernel, self.gradient_kernel = value_kernel, gradient_kernel

    def _c(self):
        return FiWeights(self.data_pos, self.data_gradient, self.model_0, self.model_1, self.model_2,
                         self.model_3, self.model_4, self.gradient_smoothness, int(self.value_kernel),
                         int(self.gradient_kernel))


class SolveOptions:
    """sparse_linear.hpp:66-73."""

    def __init__(self, tile=False, tile_size=16, cg=True, max_iterations=0, error_tolerance=1e-3):
        self.tile, self.tile_size, self.cg = tile, tile_size, cg
        self.max_iterations, self.error_tolerance = max_iterations, error_tolerance


def _buf(a, dtype=np.float32):
    """(pointer, memory kind, keep-alive object) of a numpy array / torch tensor / None."""
    if a is None:
        return None, None, None
    if hasattr(a, "data_ptr"):           # torch tensor
        import torch
        want = {np.float32: torch.float32, np.float64: torch.float64}[dtype]
        t = a.contiguous()
        if t.dtype != want:
            t = t.to(want)
        return C.c_void_p(t.data_ptr()), (FI_DEVICE if t.is_cuda else FI_HOST), t
    arr = np.ascontiguousarray(a, dtype=dtype)
    return C.c_void_p(arr.ctypes.data), FI_HOST, arr


def _same_memory(*kinds):
    ks = {k for k in kinds if k is not None}
    if len(ks) > 1:
        raise ValueError("all buffers of one call must live in the same memory (all host or all device)")
    return ks.pop() if ks else FI_HOST


class LatticeField:
    """field_interpolation.hpp:97-114 `LatticeField{sizes}`: sizes[0] (x) is the fastest axis.

    dtype: "f32" (vectors fp32, reductions fp64) or "f64".  rank/nranks select a slab of the slowest
    axis (one process per GPU)."""

    def __init__(self, sizes, dtype="f32", rank=0, nranks=1, _borrowed=None):
        self.sizes = [int(s) for s in sizes]
        self.dtype = dtype
        self.rank, self.nranks = rank, nranks
        self.strides = []
        s = 1
        for n in self.sizes:
            self.strides.append(s)
            s *= n
        self._h = C.c_void_p()
        self._borrowed = _borrowed is not None
        sz = (C.c_int * len(self.sizes))(*self.sizes)
        code = {"f32": FI_F32, "f64": FI_F64}[dtype]
        if _borrowed is not None:       # a member of a LatticeGroup: the group owns the context
            self._h = C.c_void_p(_borrowed)
        elif nranks == 1:
            check(_capi.lib().fi_ctx_create(C.byref(self._h), len(self.sizes), sz, code))
        else:
            check(_capi.lib().fi_ctx_create_slab(C.byref(self._h), len(self.sizes), sz, code, rank, nranks))
        self._weights = Weights()
        self._dirty = True
        lo, hi = C.c_int(0), C.c_int(0)
        check(_capi.lib().fi_slab_range(self._h, C.byref(lo), C.byref(hi)))
        self.slab = (lo.value, hi.value)

    def point_range(self):
        """[lo, hi) of the slowest coordinate of the data points this rank has to be given (fi_slab_point_range:
        covers the cells touching the slab on every level set so far -- call after set_levels)."""
        lo, hi = C.c_float(0), C.c_float(0)
        check(_capi.lib().fi_slab_point_range(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _capi._LIB is not None and not getattr(self, "_borrowed", False):
            _capi._LIB.fi_ctx_destroy(h)
        self._h = None

    def num_dim(self):
        return len(self.sizes)

    @property
    def num_unknowns(self):
        return int(np.prod(self.sizes))

    @property
    def num_owned(self):
        n = 1
        for s in self.sizes[:-1]:
            n *= s
        return n * (self.slab[1] - self.slab[0])

    # ---- model -----------------------------------------------------------------------------
    def add_field_constraints(self, weights):
        """add_field_constraints (field_interpolation.cpp:326-341), matrix-free."""
        self._weights = weights
        w = weights._c()
        check(_capi.lib().fi_set_model(self._h, C.byref(w)))
        self._dirty = True

    # ---- data ------------------------------------------------------------------------------
    def _inside_ext(self, pos):
        fl = [math.floor(float(np.float32(p))) for p in pos]
        return all(-1 <= f <= n - 1 for f, n in zip(fl, self.sizes))

    def _cell_valid(self, pos):
        fl = [math.floor(float(np.float32(p))) for p in pos]
        return all(0 <= f and f + 1 < n for f, n in zip(fl, self.sizes))

    def add_value_constraint(self, pos, value, weight):
        """add_value_constraint (field_interpolation.cpp:57-80).  False if the position was ignored."""
        pos = np.atleast_1d(np.asarray(pos, np.float32))
        if weight == 0 or not self._inside_ext(pos):
            return False
        self.add_points(weight, ValueKernel.kLinearInterpolation, 0.0, GradientKernel.kCellEdges, pos, None, None,
                        values=np.asarray([value], np.float32))
        return True

    def add_value_constraint_nearest_neighbor(self, pos, gradient, value, weight):
        """add_value_constraint_nearest_neighbor (field_interpolation.cpp:82-107)."""
        pos = np.atleast_1d(np.asarray(pos, np.float32))
        for p, n in zip(pos, self.sizes):
            q = math.floor(abs(float(p)) + 0.5) * (1 if p >= 0 else -1)     # std::round
            if q < 0 or n <= q:
                return False
        self.add_points(weight, ValueKernel.kNearestNeighbor, 0.0, GradientKernel.kCellEdges, pos,
                        np.atleast_1d(np.asarray(gradient, np.float32)), None, values=np.asarray([value], np.float32))
        return True

    def add_gradient_constraint(self, pos, gradient, weight, kernel):
        """add_gradient_constraint (field_interpolation.cpp:123-240)."""
        if int(kernel) not in (0, 1, 2):
            raise ValueError("Unknown gradient kernel: %d" % int(kernel))    # ABORT_F, cpp:238
        pos = np.atleast_1d(np.asarray(pos, np.float32))
        if weight == 0:
            return False
        if int(kernel) != GradientKernel.kLinearInterpolation and not self._cell_valid(pos):
            return False
        self.add_points(0.0, ValueKernel.kLinearInterpolation, weight, kernel, pos,
                        np.atleast_1d(np.asarray(gradient, np.float32)), None)
        return True

    def add_border_prior(self, weight):
        """The border prior of the reference's SDF application (src/sdf_field.cpp:218-246): every border lattice point
        gets the row [1] * weight = (distance to the nearest data point added so far) * weight.  After add_points."""
        check(_capi.lib().fi_add_border_prior(self._h, float(weight)))
        self._dirty = True

    def add_points(self, value_weight, value_kernel, gradient_weight, gradient_kernel, positions, normals=None,
                   point_weights=None, values=None):
        """add_points (field_interpolation.cpp:343-371); `values` (optional) generalises the fixed 0 target."""
        pp, km, _k1 = _buf(positions)
        np_, kn, _k2 = _buf(normals)
        pw, kw, _k3 = _buf(point_weights)
        pv, kv, _k4 = _buf(values)
        mem = _same_memory(km, kn, kw, kv)
        count = (_k1.numel() if hasattr(_k1, "numel") else _k1.size) // len(self.sizes)
        check(_capi.lib().fi_add_points(self._h, count, pp, np_, pw, pv, float(value_weight), int(value_kernel),
                                        float(gradient_weight), int(gradient_kernel), mem))
        self._dirty = True

    def add_rows_coo(self, rows, cols, values, rhs):
        """Arbitrary rows of a `LinearEquation` (sparse_linear.hpp:18-22): triplets (row, col, value) with rows
        numbered from 0 within this call, and one rhs per row.  Duplicate (row, col) entries are summed."""
        trip = np.empty(len(values), dtype=[("row", np.int32), ("col", np.int32), ("value", np.float32)])
        trip["row"], trip["col"], trip["value"] = rows, cols, values
        b = np.ascontiguousarray(rhs, np.float32)
        check(_capi.lib().fi_add_rows_coo(self._h, b.size, trip.size, C.c_void_p(trip.ctypes.data),
                                          C.c_void_p(b.ctypes.data), FI_HOST))
        self._dirty = True

    def clear_points(self):
        check(_capi.lib().fi_clear_points(self._h))
        self._dirty = True

    # ---- distributed -----------------------------------------------------------------------
    def comm_init(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(_capi.lib().fi_comm_init(self._h, buf))

    def comm_info(self):
        """fi_comm_info: what the slab transport looks like from inside -- the ranks RCCL itself counts (ncclCommCount), this
        rank's index and device there, the kind of transport, ghost planes per exchange and their bytes."""
        out = (C.c_long * 7)()
        check(_capi.lib().fi_comm_info(self._h, out))
        kind = {0: "none", 1: "rccl", 2: "host-staged test transport"}.get(out[3], "?")
        return {"ranks": out[0], "rank": out[1], "device": out[2], "transport": kind, "halo_planes": out[4],
                "halo_planes_stored": out[5], "plane_bytes": out[6], "halo_bytes_per_exchange": out[4] * out[6]}

    def comm_init_host(self, name, create):
        """fi_comm_init_host: the host-staged TEST transport for ranks that share one GPU."""
        check(_capi.lib().fi_comm_init_host(self._h, name.encode(), 1 if create else 0))

    # ---- assemble / solve --------------------------------------------------------------------
    def assemble(self):
        """as_sparse_matrix_float + make_square + A^T b (sparse_linear.cpp:59-70,105-113,120) on the GPU."""
        check(_capi.lib().fi_assemble(self._h))
        self._dirty = False

    def _ready(self):
        if self._dirty:
            self.assemble()

    def _out(self, like):
        if like is not None and hasattr(like, "data_ptr") and like.is_cuda:
            import torch
            return torch.empty(self.num_owned, dtype=torch.float32, device=like.device)
        return np.empty(self.num_owned, np.float32)

    def solve_cg(self, guess=None, max_iterations=0, error_tolerance=0.0, out=None):
        """-> (x, iterations, relative residual) or None on solver breakdown."""
        self._ready()
        g, kg, _kg = _buf(guess)
        if out is None:
            out = self._out(guess)
        o, ko, _ko = _buf(out)
        mem = _same_memory(kg, ko)
        it, rel = C.c_int(0), C.c_float(0)
        try:
            check(_capi.lib().fi_solve_cg(self._h, g, int(max_iterations), float(error_tolerance), o, C.byref(it),
                                          C.byref(rel), mem))
        except FiError as e:
            if e.code == 6:     # FI_ERR_BREAKDOWN: the reference logs a warning and returns {}
                return None
            raise
        return out, it.value, rel.value

    def set_verify_residual(self, on):
        """FI_OPT_VERIFY_RESIDUAL: True (default) checks b - A x at convergence and restarts CG if fp32 drift
        left it above the tolerance; False stops on the recurrence residual alone, like the reference."""
        check(_capi.lib().fi_set_option(self._h, 1, 1.0 if on else 0.0))

    def set_levels(self, levels, coarse_tolerance=None):
        """FI_OPT_LEVELS: build `levels` coarser replicas at the next assemble; solve_cg(guess=None) then starts
        from a coarse-to-fine cascade (src/sdf_field.cpp:272-288 generalised, on the device)."""
        check(_capi.lib().fi_set_option(self._h, 2, float(levels)))
        if coarse_tolerance is not None:
            check(_capi.lib().fi_set_option(self._h, 3, float(coarse_tolerance)))
        self._dirty = True

    def set_multigrid(self, on=True):
        """FI_OPT_MULTIGRID: V-cycle preconditioned CG over the levels of set_levels()."""
        check(_capi.lib().fi_set_option(self._h, 4, 1.0 if on else 0.0))
        self._dirty = True

    def set_mixed_precision(self, on=True):
        """FI_OPT_MIXED_PRECISION (dtype="f64" fields with levels + multigrid): CG in fp64, V-cycle in fp32."""
        check(_capi.lib().fi_set_option(self._h, 5, 1.0 if on else 0.0))
        self._dirty = True

    def set_field_tolerance(self, tol):
        """FI_OPT_FIELD_TOLERANCE (V-cycle PCG; undivided lattices and up to 16 slabs): stop when the field is within `tol` (relative, maximum
        norm) of the converged solution by the solver's own measure -- the last step times sigma / (1 - sigma), sigma the
        slowest mean decay of the residual norm over the recent windows and the whole solve, doubled (include/fi_hip.h) --
        instead of at a residual; the `tol` of solve_cg is then ignored.  0: the residual rule.  stats(): field_estimate, field_per_residual, stop_residual."""
        check(_capi.lib().fi_set_option(self._h, 12, float(tol)))

    def set_cheb_smoother(self, degree=0, ratio=0.0):
        """FI_OPT_MG_CHEB_DEGREE / FI_OPT_MG_CHEB_RATIO: the full-operator Chebyshev smoother's degree and interval; 0: by the dimension."""
        check(_capi.lib().fi_set_option(self._h, 14, float(degree)))
        check(_capi.lib().fi_set_option(self._h, 15, float(ratio)))
        self._dirty = True

    def set_kcycle(self, levels):
        """FI_OPT_MG_KCYCLE: the first `levels` coarse levels corrected by two flexible-CG steps each (a K-cycle); 0: the V-cycle."""
        check(_capi.lib().fi_set_option(self._h, 13, float(levels)))
        self._dirty = True

    def set_polynomial(self, terms, ratio=None):
        """FI_OPT_POLY_TERMS / FI_OPT_POLY_RATIO: CG preconditioned by a Chebyshev polynomial of `terms` terms
        (0: the Jacobi diagonal).  No re-assembly needed."""
        check(_capi.lib().fi_set_option(self._h, 6, float(terms)))
        if ratio is not None:
            check(_capi.lib().fi_set_option(self._h, 7, float(ratio)))

    def set_mg_smoother(self, polynomial=True, safe_factor=None, terms=None, ratio=None):
        """FI_OPT_MG_SMOOTHER / FI_OPT_MG_SAFE_FACTOR / FI_OPT_MG_TERMS / FI_OPT_MG_RATIO: the V-cycle's smoother on fp32 3-D
        levels -- the polynomial in A_model + f diag(A_data) (default; `terms` terms over [hi / ratio, hi]) or the Chebyshev
        polynomial in the full operator."""
        check(_capi.lib().fi_set_option(self._h, 8, 1.0 if polynomial else 0.0))
        if safe_factor is not None:
            check(_capi.lib().fi_set_option(self._h, 9, float(safe_factor)))
        if terms is not None:
            check(_capi.lib().fi_set_option(self._h, 10, float(terms)))
        if ratio is not None:
            check(_capi.lib().fi_set_option(self._h, 11, float(ratio)))
        self._dirty = True

    def jacobi(self, guess, num_iterations, weight):
        self._ready()
        g, kg, _kg = _buf(guess)
        out = self._out(guess)
        o, ko, _ko = _buf(out)
        check(_capi.lib().fi_jacobi(self._h, g, int(num_iterations), float(weight), o, _same_memory(kg, ko)))
        return out

    def error_map(self, solution):
        """generate_error_map (field_interpolation.cpp:402-429) on the device: fi_error_map."""
        self._ready()
        g, kg, _kg = _buf(solution)
        out = self._out(solution)
        o, ko, _ko = _buf(out)
        check(_capi.lib().fi_error_map(self._h, g, o, _same_memory(kg, ko)))
        return out

    def tile_pass(self, guess, tile_size=16):
        """tile_solver_square (sparse_linear.cpp:246-390) on the device: fi_tile_pass."""
        self._ready()
        g, kg, _kg = _buf(guess)
        out = self._out(guess)
        o, ko, _ko = _buf(out)
        check(_capi.lib().fi_tile_pass(self._h, g, int(tile_size), o, _same_memory(kg, ko)))
        return out

    def solution_f64(self):
        out = np.empty(self.num_owned, np.float64)
        check(_capi.lib().fi_get_solution_f64(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def true_residual(self):
        r = C.c_double(0)
        check(_capi.lib().fi_true_residual(self._h, C.byref(r)))
        return r.value

    # ---- test / measurement hooks --------------------------------------------------------------
    def apply_AtA(self, x):
        self._ready()
        xx = np.ascontiguousarray(x, np.float64)
        y = np.empty(self.num_owned, np.float64)
        dp = C.POINTER(C.c_double)
        check(_capi.lib().fi_apply_AtA_f64(self._h, xx.ctypes.data_as(dp), y.ctypes.data_as(dp)))
        return y

    def Atb(self):
        self._ready()
        out = np.empty(self.num_owned, np.float64)
        check(_capi.lib().fi_get_Atb_f64(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def diag(self):
        self._ready()
        out = np.empty(self.num_owned, np.float64)
        check(_capi.lib().fi_get_diag_f64(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def stats(self):
        s = FiStats()
        check(_capi.lib().fi_get_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in FiStats._fields_}

    def time_apply(self, reps=20):
        self._ready()
        ms = C.c_double(0)
        check(_capi.lib().fi_time_apply(self._h, reps, C.byref(ms)))
        return ms.value


class LatticeGroup:
    """All slabs of a decomposition in one process on one GPU (loop-back test facility, fi_hip.h
    "loop-back group"): same kernels and slab rules as the one-process-per-GPU RCCL path."""

    def __init__(self, sizes, nranks, dtype="f32"):
        self.sizes = [int(s) for s in sizes]
        self.dtype = dtype
        self._g = C.c_void_p()
        sz = (C.c_int * len(self.sizes))(*self.sizes)
        check(_capi.lib().fi_group_create(C.byref(self._g), len(self.sizes), sz, {"f32": FI_F32, "f64": FI_F64}[dtype],
                                          nranks))
        self.members = [LatticeField(self.sizes, dtype=dtype, rank=r, nranks=nranks,
                                     _borrowed=_capi.lib().fi_group_rank(self._g, r)) for r in range(nranks)]

    def __del__(self):
        g = getattr(self, "_g", None)
        if g and _capi._LIB is not None:
            for m in getattr(self, "members", []):
                m._h = None
            _capi._LIB.fi_group_destroy(g)
            self._g = None

    @property
    def num_unknowns(self):
        return int(np.prod(self.sizes))

    def add_field_constraints(self, weights):
        for m in self.members:
            m.add_field_constraints(weights)

    def add_points(self, *a, **kw):
        for m in self.members:          # every rank sees every point and keeps the cells touching its slab
            m.add_points(*a, **kw)

    def set_levels(self, levels, coarse_tolerance=None):
        for m in self.members:
            m.set_levels(levels, coarse_tolerance)

    def set_multigrid(self, on=True):
        for m in self.members:
            m.set_multigrid(on)

    def set_mixed_precision(self, on=True):
        for m in self.members:
            m.set_mixed_precision(on)

    def set_field_tolerance(self, tol):
        for m in self.members:
            m.set_field_tolerance(tol)

    def set_polynomial(self, terms, ratio=None):
        for m in self.members:
            m.set_polynomial(terms, ratio)

    def set_mg_smoother(self, polynomial=True, safe_factor=None, terms=None, ratio=None):
        for m in self.members:
            m.set_mg_smoother(polynomial, safe_factor, terms, ratio)

    def assemble(self):
        check(_capi.lib().fi_group_assemble(self._g))
        for m in self.members:
            m._dirty = False

    def apply_AtA(self, x):
        xx = np.ascontiguousarray(x, np.float64)
        y = np.empty(self.num_unknowns, np.float64)
        dp = C.POINTER(C.c_double)
        check(_capi.lib().fi_group_apply_AtA_f64(self._g, xx.ctypes.data_as(dp), y.ctypes.data_as(dp)))
        return y

    def Atb(self):
        return np.concatenate([m.Atb() for m in self.members])

    def diag(self):
        return np.concatenate([m.diag() for m in self.members])

    def solve_cg(self, guess=None, max_iterations=0, error_tolerance=0.0):
        g = None if guess is None else np.ascontiguousarray(guess, np.float32)
        out = np.empty(self.num_unknowns, np.float32)
        it, rel = C.c_int(0), C.c_float(0)
        check(_capi.lib().fi_group_solve_cg(self._g, None if g is None else C.c_void_p(g.ctypes.data), int(max_iterations),
                                            float(error_tolerance), C.c_void_p(out.ctypes.data), C.byref(it), C.byref(rel)))
        return out, it.value, rel.value

    def stats(self):
        return self.members[0].stats()

    def tile_pass(self, guess, tile_size=16):
        g = np.ascontiguousarray(guess, np.float32)
        out = np.empty(self.num_unknowns, np.float32)
        fp = C.POINTER(C.c_float)
        check(_capi.lib().fi_group_tile_pass(self._g, g.ctypes.data_as(fp), int(tile_size), out.ctypes.data_as(fp)))
        return out

    def error_map(self, solution):
        s = np.ascontiguousarray(solution, np.float32)
        out = np.empty(self.num_unknowns, np.float32)
        fp = C.POINTER(C.c_float)
        check(_capi.lib().fi_group_error_map(self._g, s.ctypes.data_as(fp), out.ctypes.data_as(fp)))
        return out

    def solution_f64(self):
        out = np.empty(self.num_unknowns, np.float64)
        check(_capi.lib().fi_group_get_solution_f64(self._g, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def true_residual(self):
        r = C.c_double(0)
        check(_capi.lib().fi_group_true_residual(self._g, C.byref(r)))
        return r.value


# ---- free functions with the reference's names ---------------------------------------------------

def sdf_from_points(sizes, weights, positions, normals=None, point_weights=None, dtype="f32", rank=0, nranks=1):
    """sdf_from_points (field_interpolation.cpp:373-400): model rows, then add_points with target 0."""
    if positions is None:
        raise ValueError("positions is null")            # CHECK_NOTNULL_F, cpp:382
    field = LatticeField(sizes, dtype=dtype, rank=rank, nranks=nranks)
    field.add_field_constraints(weights)
    field.add_points(weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel,
                     positions, normals, point_weights)
    return field


def solve_sparse_linear_with_guess(field, guess, max_iterations=0, error_tolerance=0.0):
    """solve_sparse_linear_with_guess (sparse_linear.cpp:186-212).  0 => defaults (2N iterations, fp32 eps)."""
    res = field.solve_cg(guess, max_iterations, error_tolerance)
    return None if res is None else res[0]


def jacobi_iterations(field, guess, num_iterations, weight):
    """jacobi_iterations (sparse_linear.cpp:214-241); num_iterations <= 0 returns the guess (:220)."""
    if num_iterations <= 0:
        return np.array(guess, np.float32, copy=True) if not hasattr(guess, "data_ptr") else guess.clone()
    return field.jacobi(guess, num_iterations, weight)


def generate_error_map(field, solution):
    """generate_error_map(field.eq.triplets, solution, field.eq.rhs) (field_interpolation.cpp:402-429); the rows
    live on the device, so the field stands in for its triplet list."""
    return field.error_map(solution)


def solve_tiled_with_guess(field, guess, sizes, options):
    """solve_tiled_with_guess (sparse_linear.cpp:392-443): wrong guess length -> None (:402-405); optional tile
    pre-pass (options.tile, :415-425), then optional CG (options.cg, :427-440)."""
    n = int(np.prod(sizes))
    glen = guess.numel() if hasattr(guess, "numel") else np.asarray(guess).size
    if glen != n:
        return None
    if options.tile:
        guess = field.tile_pass(guess, options.tile_size)
    if not options.cg:
        return guess if options.tile else np.array(guess, np.float32, copy=True)
    res = field.solve_cg(guess, options.max_iterations, options.error_tolerance)
    return None if res is None else res[0]


def solve_sparse_linear_exact(field, num_columns=None, tolerance=1e-12, max_iterations=0):
    """Stands in for solve_sparse_linear_exact (sparse_linear.cpp:154-184): the reference factorises AtA
    (sparse Cholesky, double).  On the GPU the same system is iterated to `tolerance` in fp64; create the
    field with dtype="f64" for this.  Returns None where the reference returns {}: an unknown without any equation (the
    zero pivot that stops the factorisation) or a solver breakdown.  A well-posed system whose attainable residual
    (about eps * kappa) stays above `tolerance` returns its best iterate with a warning -- the factorisation would
    still answer."""
    import warnings
    if field.dtype != "f64":
        raise ValueError("solve_sparse_linear_exact needs a dtype='f64' field")
    field._ready()
    if not (field.diag() > 0).all():
        return None
    if max_iterations <= 0:
        # CG in floating point can need several times N steps on these kappa ~ side^4 systems
        max_iterations = max(2000, 20 * field.num_unknowns)
    res = field.solve_cg(None, max_iterations, tolerance)
    if res is None:
        return None
    x, it, rel = res
    if not (rel <= tolerance * 1.0001):
        warnings.warn("solve_sparse_linear_exact: relative residual %g after %d iterations (asked for %g)" % (rel, it, tolerance))
    return x


def upscale_field(field, small_sizes, large_sizes):
    """upscale_field (field_interpolation.cpp:431-485)."""
    src, mem, keep = _buf(field)
    ss = (C.c_int * len(small_sizes))(*[int(s) for s in small_sizes])
    ls = (C.c_int * len(large_sizes))(*[int(s) for s in large_sizes])
    n = int(np.prod(large_sizes))
    if mem == FI_DEVICE:
        import torch
        out = torch.empty(n, dtype=torch.float32, device=keep.device)
        o = C.c_void_p(out.data_ptr())
    else:
        out = np.empty(n, np.float32)
        o = C.c_void_p(out.ctypes.data)
    check(_capi.lib().fi_upscale_field(src, len(small_sizes), ss, ls, o, mem))
    return out
