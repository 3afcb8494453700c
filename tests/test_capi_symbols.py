"""The drop-in boundary on a machine without a GPU: libfi_hip.so loads, exports every function include/fi_hip.h
declares (and the ctypes mirror declares exactly those), the status codes and struct layouts of the mirror match
the header, and calls that need a device fail with a status code instead of crashing.  No compute."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fi_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fi_[a-z0-9_A-Z]+)\s*\(", text)))


def test_library_exports_every_declared_function():
    from field_interpolation_amd import _capi
    lib = C.CDLL(_capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), "libfi_hip.so lacks %s" % name


def test_ctypes_mirror_declares_the_same_functions():
    from field_interpolation_amd import _capi
    assert sorted(_capi.SYMBOLS) == _declared()


def test_struct_mirrors_match_the_header():
    from field_interpolation_amd import _capi
    text = open(HEADER).read()
    for cname, mirror in (("fi_weights", _capi.FiWeights), ("fi_stats", _capi.FiStats)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in re.findall(r"\b(?:float|double|int|long)\s+([\w\s,]+);", body):
            fields += [name.strip() for name in decl.split(",")]
        assert fields == [f for f, _ in mirror._fields_], cname
    for code, name in _capi.ERR_NAMES.items():
        assert re.search(r"#define\s+%s\s+%d\b" % (name, code), text), name


def test_no_device_is_a_status_code_not_a_crash():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from field_interpolation_amd import _capi
    lib = _capi.lib()
    h = C.c_void_p()
    sizes = (C.c_int * 2)(8, 8)
    rc = lib.fi_ctx_create(C.byref(h), 2, sizes, _capi.FI_F32)
    assert rc != 0 and lib.fi_last_error()
    assert lib.fi_ctx_create(C.byref(h), 7, sizes, _capi.FI_F32) != 0        # bad dimensionality: FI_ERR_INVALID


def test_cxx_dropin_exports_the_reference_api():
    path = os.path.join(ROOT, "field_interpolation_amd", "libfield_interpolation.so")
    if not os.path.exists(path):
        pytest.skip("C++ drop-in not built")
    import subprocess
    out = subprocess.run(["nm", "-DC", "--defined-only", path], capture_output=True, text=True).stdout
    for sym in ("field_interpolation::add_field_constraints", "field_interpolation::add_points",
                "field_interpolation::sdf_from_points", "field_interpolation::add_equation",
                "field_interpolation::solve_sparse_linear_with_guess", "field_interpolation::jacobi_iterations",
                "field_interpolation::solve_tiled_with_guess", "field_interpolation::upscale_field",
                "field_interpolation::generate_error_map", "field_interpolation::GpuLatticeField::solve"):
        assert sym in out, sym


def test_shipped_library_has_no_timing_hooks():
    """The FI_DBG timing modes (results wrong by construction) and the ablation knobs exist only in timing builds
    (-DFI_TIMING_BUILD, tools/build_variant.sh): the shipped library does not even contain their names."""
    import os
    from field_interpolation_amd import _capi
    blob = open(_capi.LIB_PATH, "rb").read()
    for name in (b"FI_DBG", b"FI_TXT", b"FI_RUN_CAP", b"FI_NO_SPLIT", b"FI_NO_FOLD", b"FI_MG_DEGREE", b"FI_MG_RATIO", b"FI_MG_POLY",
                 b"FI_TIME_STEP", b"FI_NO_SAMPLES", b"FI_POLY_UNFOLDED", b"FI_DUMMY"):
        assert name not in blob, name
    for name in (b"FI_NO_FUSE", b"FI_NO_MARCH", b"FI_ZC", b"FI_SOLVE_TIMEOUT_S", b"FI_NO_FUSED_SMOOTHER", b"FI_NO_Z0_ON_LOAD",
                 b"FI_SERIAL_LEVELS", b"FI_LINEAR_START", b"FI_START_ONLY", b"FI_NO_POOL", b"FI_NO_LAMBDA_CACHE", b"FI_SERIAL_LEVEL_CHAINS"):                              # the switches the tests use are there
        assert name in blob, name
