"""The LDS-tiled 2-D kernel (fi_stencil2d.hip: model_0/1/2 + data cells fused through 4 corner planes) against
the oracle's explicit float64 normal equations, against the generic kernel of the same library (FI_NO_TILE2D),
over slabs, and for run-to-run bitwise reproducibility.  fp64 1e-12, fp32 2e-6 of |AtA||x|."""
import os

import numpy as np
import pytest

from util import build_pair, random_points, rel_inf, sphere_points

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 2e-6}


@pytest.fixture(scope="module")
def fi():
    import field_interpolation_amd as fi
    from field_interpolation_amd import _capi
    assert _capi.device_count() >= 1
    return fi


SIZES = [[64, 16], [68, 40], [128, 33], [36, 20], [4, 3], [8, 1], [200, 7], [12, 130],
         [66, 20], [13, 9], [131, 5], [65, 17], [7, 40]]          # rows that are not a multiple of the 16-byte group


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes", SIZES)
@pytest.mark.parametrize("kw", [dict(), dict(model_2=0.0, model_1=0.7), dict(model_0=0.2, model_1=0.4, model_2=0.9)])
def test_against_oracle(oracle, fi, dtype, sizes, kw):
    rng = np.random.default_rng(sizes[0] * 7 + sizes[1])
    pos, nrm, pw, val = random_points(rng, sizes, 400, margin=1.2)
    w = fi.Weights(data_pos=0.8, data_gradient=1.25, **kw)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, val, dtype=dtype)
    fg.assemble()
    AtA, atb, diag = fo.normal_equations()
    absA = abs(AtA)
    n = fo.num_unknowns
    for k in range(2):
        x = rng.normal(size=n) if k == 0 else np.linspace(-50, 50, n) + rng.normal(size=n)
        y = fg.apply_AtA(x)
        scale = (absA @ np.abs(x)).max()
        assert np.abs(y - AtA @ x).max() <= TOL[dtype] * scale


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_equals_generic_kernel_and_is_reproducible(fi, dtype):
    sizes = [260, 150]
    rng = np.random.default_rng(5)
    pos, nrm = sphere_points(rng, sizes, 3000)
    w = fi.Weights(model_1=0.05)

    def build():
        f = fi.LatticeField(sizes, dtype=dtype)
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        return f

    tiled = build()
    os.environ["FI_NO_TILE2D"] = "1"
    try:
        generic = build()
    finally:
        del os.environ["FI_NO_TILE2D"]
    x = rng.normal(size=int(np.prod(sizes)))
    yt, yg = tiled.apply_AtA(x), generic.apply_AtA(x)
    assert np.abs(yt - yg).max() <= (1e-12 if dtype == "f64" else 3e-6) * np.abs(yg).max()
    # no atomics in the tile kernel: the same bits every time, also from a second context
    again = build()
    for _ in range(3):
        assert np.array_equal(tiled.apply_AtA(x), yt)
    assert np.array_equal(again.apply_AtA(x), yt)
    tol = 1e-9 if dtype == "f64" else 1e-4
    x1, it1, r1 = tiled.solve_cg(None, 0, tol)
    x2, it2, r2 = again.solve_cg(None, 0, tol)
    assert it1 == it2 and np.array_equal(x1, x2)
    xg, itg, rg = generic.solve_cg(None, 0, tol)
    assert r1 <= tol and rg <= tol
    assert rel_inf(tiled.solution_f64(), generic.solution_f64()) <= (1e-5 if dtype == "f64" else 2e-2)


@pytest.mark.parametrize("sizes,nranks", [([64, 50], 3), ([100, 37], 4), ([32, 16], 8)])
def test_slabs_equal_undivided(fi, sizes, nranks):
    rng = np.random.default_rng(nranks)
    pos, nrm, pw, val = random_points(rng, sizes, 500, margin=0.8)
    w = fi.Weights(model_0=0.1, model_1=0.3)
    one = fi.LatticeField(sizes, dtype="f64")
    grp = fi.LatticeGroup(sizes, nranks, dtype="f64")
    for f in (one, grp):
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, pw, values=val)
        f.assemble()
    x = rng.normal(size=int(np.prod(sizes)))
    y1, yg = one.apply_AtA(x), grp.apply_AtA(x)
    assert np.abs(yg - y1).max() <= 1e-12 * np.abs(y1).max()
