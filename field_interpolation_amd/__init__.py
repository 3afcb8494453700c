"""field_interpolation_amd -- MI355X-native solver core for emilk/field_interpolation's hot path.

Host-side mirror of the reference interface (field_interpolation.hpp / sparse_linear.hpp) over the
C ABI of libfi_hip.so (include/fi_hip.h).  All arithmetic runs in hand-written HIP kernels on the
GPU; importing the API without the built library raises ImportError (no CPU fallback).
"""
from .api import (GradientKernel, LatticeField, LatticeGroup, SolveOptions, ValueKernel, Weights,  # noqa: F401
                  generate_error_map, jacobi_iterations, sdf_from_points, solve_sparse_linear_exact,
                  solve_sparse_linear_with_guess, solve_tiled_with_guess, upscale_field)
from ._capi import FiError, memory_pool  # noqa: F401
