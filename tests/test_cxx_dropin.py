"""The C++ drop-in (include/field_interpolation/*.hpp, libfield_interpolation.so): it must compile and link
against the reference-compatible headers (CPU), and tests/cxx/test_dropin.cpp -- reference-style call
sequences -- must pass on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "field_interpolation_amd")
EXE = os.path.join(ROOT, "tests", "cxx", "test_dropin")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "cxx")])
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "test_dropin.cpp"), "-o", EXE,
                           "-L", PKG, "-lfield_interpolation", "-lfi_hip", "-Wl,-rpath," + PKG,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_dropin_headers_compile_and_link():
    if not os.path.exists(os.path.join(PKG, "libfi_hip.so")):
        pytest.skip("libfi_hip.so not built")
    exe = _build()
    assert os.path.exists(exe)
    # every reference entry point is exported with C++ linkage from the drop-in library
    syms = subprocess.check_output(["nm", "-DC", os.path.join(PKG, "libfield_interpolation.so")], text=True)
    for name in ("field_interpolation::add_equation", "field_interpolation::add_field_constraints",
                 "field_interpolation::add_value_constraint(", "field_interpolation::add_value_constraint_nearest_neighbor",
                 "field_interpolation::add_gradient_constraint", "field_interpolation::add_points",
                 "field_interpolation::sdf_from_points", "field_interpolation::generate_error_map",
                 "field_interpolation::upscale_field", "field_interpolation::solve_sparse_linear_fast",
                 "field_interpolation::solve_sparse_linear_exact", "field_interpolation::solve_sparse_linear_with_guess",
                 "field_interpolation::jacobi_iterations", "field_interpolation::solve_tiled_with_guess",
                 "field_interpolation::operator<<"):
        assert name in syms, name


@pytest.mark.gpu
def test_dropin_program_runs():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all drop-in checks passed" in out.stdout
