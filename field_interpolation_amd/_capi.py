"""ctypes binding of include/fi_hip.h (libfi_hip.so).  Fails loudly when the library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FI_HIP_LIB", os.path.join(HERE, "libfi_hip.so"))   # override: experiments only

FI_OK = 0
FI_HOST, FI_DEVICE = 0, 1
FI_F32, FI_F64 = 0, 1
ERR_NAMES = {1: "FI_ERR_INVALID", 2: "FI_ERR_HIP", 3: "FI_ERR_STATE", 4: "FI_ERR_COMM",
             5: "FI_ERR_UNSUPPORTED", 6: "FI_ERR_BREAKDOWN", 7: "FI_ERR_TIMEOUT"}

# every symbol include/fi_hip.h declares
SYMBOLS = [
    "fi_last_error", "fi_device_count", "fi_ctx_create", "fi_ctx_create_slab", "fi_ctx_destroy",
    "fi_slab_range", "fi_slab_point_range", "fi_slab_partition", "fi_halo_width", "fi_comm_unique_id", "fi_comm_init", "fi_comm_init_host", "fi_comm_info", "fi_comm_self_test", "fi_set_model", "fi_add_points", "fi_add_border_prior",
    "fi_add_rows_coo", "fi_assemble", "fi_clear_points", "fi_set_option", "fi_solve_cg", "fi_jacobi", "fi_tile_pass", "fi_error_map",
    "fi_get_solution_f64", "fi_true_residual", "fi_apply_AtA_f64", "fi_get_Atb_f64", "fi_get_diag_f64",
    "fi_get_stats", "fi_memory_pool", "fi_time_apply", "fi_upscale_field",
    "fi_group_create", "fi_group_destroy", "fi_group_size", "fi_group_rank", "fi_group_assemble",
    "fi_group_solve_cg", "fi_group_apply_AtA_f64", "fi_group_true_residual", "fi_group_get_solution_f64", "fi_group_tile_pass", "fi_group_error_map",
]


class FiWeights(C.Structure):
    _fields_ = [("data_pos", C.c_float), ("data_gradient", C.c_float),
                ("model_0", C.c_float), ("model_1", C.c_float), ("model_2", C.c_float),
                ("model_3", C.c_float), ("model_4", C.c_float), ("gradient_smoothness", C.c_float),
                ("value_kernel", C.c_int), ("gradient_kernel", C.c_int)]


class FiSolveOptions(C.Structure):
    _fields_ = [("tile", C.c_int), ("tile_size", C.c_int), ("cg", C.c_int),
                ("max_iterations", C.c_int), ("error_tolerance", C.c_float)]


class FiTriplet(C.Structure):
    _fields_ = [("row", C.c_int), ("col", C.c_int), ("value", C.c_float)]


class FiStats(C.Structure):
    _fields_ = [("num_unknowns", C.c_long), ("num_data_rows", C.c_long), ("num_cells", C.c_long),
                ("num_generic_rows", C.c_long), ("iterations", C.c_int), ("converged", C.c_int),
                ("rel_residual", C.c_double), ("assemble_ms", C.c_double), ("solve_ms", C.c_double),
                ("spmv_ms_avg", C.c_double), ("spmv_samples", C.c_int), ("spmv_bytes", C.c_double),
                ("restarts", C.c_int), ("verified_residual", C.c_double),
                ("num_levels", C.c_int), ("coarse_iterations", C.c_int),
                ("prec_ms_avg", C.c_double), ("prec_samples", C.c_int), ("prec_bytes", C.c_double),
                ("operator_applies", C.c_int), ("halo_exchanges", C.c_int), ("reductions", C.c_int),
                ("coarse_unconverged", C.c_int), ("field_estimate", C.c_double), ("field_per_residual", C.c_double),
                ("stop_residual", C.c_double), ("field_rounds", C.c_int)]


class FiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, "error %d" % code), msg))
        self.code = code


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("field_interpolation_amd: %s is missing -- run `python -m field_interpolation_amd.build` "
                          "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ip, fp, dp = C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_double)
    L.fi_last_error.restype = C.c_char_p
    L.fi_device_count.argtypes = [ip]
    L.fi_ctx_create.argtypes = [C.POINTER(vp), C.c_int, ip, C.c_int]
    L.fi_ctx_create_slab.argtypes = [C.POINTER(vp), C.c_int, ip, C.c_int, C.c_int, C.c_int]
    L.fi_ctx_destroy.argtypes = [vp]
    L.fi_slab_range.argtypes = [vp, ip, ip]
    L.fi_slab_point_range.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.fi_slab_partition.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
    L.fi_halo_width.argtypes = [C.POINTER(FiWeights), ip]
    L.fi_comm_unique_id.argtypes = [vp]
    L.fi_comm_init.argtypes = [vp, vp]
    L.fi_comm_info.argtypes = [vp, C.POINTER(C.c_long)]
    L.fi_comm_init_host.argtypes = [vp, C.c_char_p, C.c_int]
    L.fi_set_model.argtypes = [vp, C.POINTER(FiWeights)]
    L.fi_add_points.argtypes = [vp, C.c_long, fp, fp, fp, fp, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int]
    L.fi_add_border_prior.argtypes = [vp, C.c_float]
    L.fi_add_rows_coo.argtypes = [vp, C.c_long, C.c_long, vp, fp, C.c_int]
    L.fi_assemble.argtypes = [vp]
    L.fi_clear_points.argtypes = [vp]
    L.fi_solve_cg.argtypes = [vp, fp, C.c_int, C.c_float, fp, ip, C.POINTER(C.c_float), C.c_int]
    L.fi_set_option.argtypes = [vp, C.c_int, C.c_double]
    L.fi_jacobi.argtypes = [vp, fp, C.c_int, C.c_float, fp, C.c_int]
    L.fi_tile_pass.argtypes = [vp, fp, C.c_int, fp, C.c_int]
    L.fi_comm_self_test.argtypes = [C.c_int, C.c_long]
    L.fi_error_map.argtypes = [vp, fp, fp, C.c_int]
    L.fi_get_solution_f64.argtypes = [vp, dp]
    L.fi_true_residual.argtypes = [vp, dp]
    L.fi_apply_AtA_f64.argtypes = [vp, dp, dp]
    L.fi_get_Atb_f64.argtypes = [vp, dp]
    L.fi_get_diag_f64.argtypes = [vp, dp]
    L.fi_get_stats.argtypes = [vp, C.POINTER(FiStats)]
    L.fi_time_apply.argtypes = [vp, C.c_int, dp]
    L.fi_memory_pool.argtypes = [C.c_longlong, C.POINTER(C.c_longlong)]
    L.fi_upscale_field.argtypes = [fp, C.c_int, ip, ip, fp, C.c_int]
    L.fi_group_create.argtypes = [C.POINTER(vp), C.c_int, ip, C.c_int, C.c_int]
    L.fi_group_destroy.argtypes = [vp]
    L.fi_group_size.argtypes = [vp]
    L.fi_group_rank.restype = vp
    L.fi_group_rank.argtypes = [vp, C.c_int]
    L.fi_group_assemble.argtypes = [vp]
    L.fi_group_solve_cg.argtypes = [vp, fp, C.c_int, C.c_float, fp, ip, C.POINTER(C.c_float)]
    L.fi_group_apply_AtA_f64.argtypes = [vp, dp, dp]
    L.fi_group_true_residual.argtypes = [vp, dp]
    L.fi_group_get_solution_f64.argtypes = [vp, dp]
    L.fi_group_tile_pass.argtypes = [vp, fp, C.c_int, fp]
    L.fi_group_error_map.argtypes = [vp, fp, fp]
    for name in SYMBOLS:
        getattr(L, name)          # AttributeError if the .so lacks a declared symbol
    _LIB = L
    return L


def check(code):
    if code != FI_OK:
        raise FiError(code, lib().fi_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib().fi_device_count(C.byref(n)))
    return n.value


def memory_pool(keep_bytes=-1):
    """fi_memory_pool: device memory of destroyed contexts kept for the next one, on the current device.  Frees it down
    to keep_bytes (0: all of it; negative: nothing) and returns the bytes that stay cached."""
    left = C.c_longlong(0)
    check(lib().fi_memory_pool(int(keep_bytes), C.byref(left)))
    return left.value
