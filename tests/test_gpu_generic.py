"""Generic sparse rows on the GPU (fi_generic.hip): hand-built `LinearEquation`s as in src/bipolar_2d.cpp /
src/line_2d.cpp, and GradientKernel::kLinearInterpolation (field_interpolation.cpp:188-236), whose rows are
not cell-local.  Operator / A^T b / diag against the oracle's float64 normal equations; CG against its direct
solution."""
import numpy as np
import pytest

from util import build_pair, oracle_weights, random_points, rel_inf

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 2e-6}


@pytest.fixture(scope="module")
def fi():
    import field_interpolation_amd as fi
    from field_interpolation_amd import _capi
    assert _capi.device_count() >= 1
    return fi


def _check(fo, fg, dtype, seed=0):
    AtA, atb, diag = fo.normal_equations()
    n = fo.num_unknowns
    rng = np.random.default_rng(seed)
    assert rel_inf(fg.Atb(), atb) <= TOL[dtype]
    assert rel_inf(fg.diag(), diag) <= TOL[dtype]
    x = rng.normal(size=n)
    scale = (abs(AtA) @ np.abs(x)).max()
    y1, y2 = fg.apply_AtA(x), fg.apply_AtA(x)
    np.testing.assert_array_equal(y1, y2)                     # no atomics: bitwise reproducible
    assert np.abs(y1 - AtA @ x).max() <= TOL[dtype] * scale


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes", [[30], [13, 11], [16, 7, 6], [9, 8, 7]])
def test_gradient_linear_interpolation_kernel(oracle, fi, dtype, sizes):
    rng = np.random.default_rng(len(sizes))
    pos, nrm, pw, val = random_points(rng, sizes, 150, margin=1.2)
    w = fi.Weights(data_gradient=0.9, gradient_kernel=fi.GradientKernel.kLinearInterpolation)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, val, dtype=dtype)
    _check(fo, fg, dtype)
    # return values of the single-constraint form (field_interpolation.cpp:219: all samples dropped -> false)
    f1 = oracle.LatticeField(sizes)
    for p in pos[:40]:
        g1 = fi.LatticeField(sizes)
        assert g1.add_gradient_constraint(p, nrm[0], 1.0, 2) in (True, f1.add_gradient_constraint(p, nrm[0], 1.0, 2))


def _line_2d_like(oracle, n=60, seed=0):
    """Rows in the spirit of src/line_2d.cpp:49-104: unknowns 2i+d (interleaved xy of a polyline), data rows
    pinning noisy points, second-difference smoothness rows; duplicates on purpose."""
    rng = np.random.default_rng(seed)
    f = oracle.LatticeField([2 * n])
    t = np.linspace(0, 4 * np.pi, n)
    pts = np.stack([t * np.cos(t), t * np.sin(t)], 1) + rng.normal(scale=0.3, size=(n, 2))
    for i in range(n):
        for d in range(2):
            f.add_equation(1.0, float(pts[i, d]), [(2 * i + d, 1.0)])
    for i in range(1, n - 1):
        for d in range(2):
            f.add_equation(3.0, 0.0, [(2 * (i - 1) + d, 1.0), (2 * i + d, -2.0), (2 * (i + 1) + d, 1.0)])
    # raw triplets, duplicated entry (bipolar_2d.cpp:251,260-261 pushes triplets directly)
    row = f.num_rows
    f.push_triplet(row, 5, 0.25)
    f.push_triplet(row, 5, 0.5)
    f.push_triplet(row, 9, -0.75)
    f.push_rhs(0.1)
    return f


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_hand_built_linear_equation(oracle, fi, dtype):
    fo = _line_2d_like(oracle)
    rows, cols, vals, rhs = fo.get()
    fg = fi.LatticeField([fo.num_unknowns], dtype=dtype)
    fg.add_field_constraints(fi.Weights(model_2=0.0))          # no lattice model: the rows are everything
    half = len(rhs) // 2                                        # two calls: row numbering restarts per call
    first = rows < half
    fg.add_rows_coo(rows[first], cols[first], vals[first], rhs[:half])
    fg.add_rows_coo(rows[~first] - half, cols[~first], vals[~first], rhs[half:])
    _check(fo, fg, dtype)
    if dtype == "f64":
        x = fi.solve_sparse_linear_exact(fg)
        assert x is not None
        assert rel_inf(fg.solution_f64(), fo.solve_exact_f64()) <= 1e-7


def test_generic_rows_on_top_of_a_lattice(oracle, fi):
    """Torus-wrap smoothness rows (bipolar_2d.cpp style) added to a lattice that also has model + point rows."""
    sizes = [12, 10]
    rng = np.random.default_rng(4)
    pos, nrm, pw, val = random_points(rng, sizes, 60, margin=0.5)
    w = fi.Weights(model_1=0.2)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, val, dtype="f64")
    rows, cols, vals, rhs = [], [], [], []
    for y in range(sizes[1]):                                   # wrap x = 11 -> x = 0
        r = len(rhs)
        a, b = y * 12 + 11, y * 12 + 0
        fo.add_equation(0.7, 0.0, [(a, -1.0), (b, 1.0)])
        rows += [r, r]; cols += [a, b]; vals += [np.float32(-0.7), np.float32(0.7)]; rhs.append(0.0)
    fg.add_rows_coo(np.array(rows), np.array(cols), np.array(vals, np.float32), np.array(rhs, np.float32))
    _check(fo, fg, "f64")
    x = fi.solve_sparse_linear_exact(fg)
    assert rel_inf(fg.solution_f64(), fo.solve_exact_f64()) <= 1e-6


def test_generic_errors(fi):
    from field_interpolation_amd._capi import FiError
    f = fi.LatticeField([8])
    with pytest.raises(FiError):            # column out of range: reference CHECK_LT_F (sparse_linear.cpp:82)
        f.add_rows_coo(np.array([0]), np.array([8]), np.array([1.0], np.float32), np.array([0.0], np.float32))
    with pytest.raises(FiError):            # row out of range
        f.add_rows_coo(np.array([1]), np.array([0]), np.array([1.0], np.float32), np.array([0.0], np.float32))
    g = fi.LatticeGroup([8, 8], 2)
    with pytest.raises(FiError):            # generic rows need an undivided lattice
        g.members[0].add_rows_coo(np.array([0]), np.array([0]), np.array([1.0], np.float32), np.array([0.0], np.float32))
