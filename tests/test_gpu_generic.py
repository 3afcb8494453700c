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


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes,nranks", [([13, 22], 3), ([16, 7, 24], 4), ([9, 8, 17], 2)])
def test_gradient_linear_interpolation_kernel_over_slabs(oracle, fi, dtype, sizes, nranks):
    """GradientKernel::kLinearInterpolation rows (three planes wide, kept as triplets) over slabs: a rank keeps the rows
    that touch one of its planes, whole, with local column numbers, and applies them to its own columns -- like a data cell.
    The decomposed operator, right-hand side and diagonal equal the oracle's explicit normal equations; the error map and
    the solve equal the undivided ones.  (Hand-built rows, fi_add_rows_coo, stay with undivided lattices.)"""
    rng = np.random.default_rng(len(sizes) + nranks)
    pos, nrm, pw, val = random_points(rng, sizes, 200, margin=1.2)
    w = fi.Weights(data_gradient=0.9, gradient_kernel=fi.GradientKernel.kLinearInterpolation)
    fo, one = build_pair(oracle, fi, sizes, w, pos, nrm, pw, val, dtype=dtype)
    grp = fi.LatticeGroup(sizes, nranks, dtype=dtype)
    grp.add_field_constraints(w)
    grp.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, pw, values=val)
    grp.assemble()
    AtA, atb, diag = fo.normal_equations()
    tol = TOL[dtype]
    assert np.abs(grp.Atb() - atb).max() <= tol * np.abs(atb).max()
    assert np.abs(grp.diag() - diag).max() <= tol * np.abs(diag).max()
    x = rng.normal(size=fo.num_unknowns)
    y = grp.apply_AtA(x)
    assert np.abs(y - AtA @ x).max() <= tol * (abs(AtA) @ np.abs(x)).max()
    np.testing.assert_array_equal(y, grp.apply_AtA(x))
    sol = rng.normal(size=fo.num_unknowns).astype(np.float32)
    e1, eg = one.error_map(sol), grp.error_map(sol)
    assert np.abs(eg - e1).max() <= 1e-4 * np.abs(e1).max()
    t = 1e-9 if dtype == "f64" else 1e-5
    x1, it1, r1 = one.solve_cg(None, 0, t)
    xg, itg, rg = grp.solve_cg(None, 0, t)
    assert r1 <= t and rg <= t and abs(itg - it1) <= max(3, it1 // 10)
    assert rel_inf(grp.solution_f64(), one.solution_f64()) <= (1e-6 if dtype == "f64" else 2e-2)


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


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes,ts", [([20, 18], 8), ([12, 9, 10], 4), ([50], 16)])
def test_tile_pass_on_materialised_rows(oracle, fi, dtype, sizes, ts):
    """solve_tiled_with_guess(eq, ...) with SolveOptions.tile (sparse_linear.cpp:415-425) for rows handed over as
    triplets (fi_add_rows_coo): the rows of a lattice problem taken from the oracle, all but one corner region of
    the lattice, so that one tile holds no entry at all and keeps the guess (tile_solver_square skips it).  Against a
    dense float64 re-derivation (couplings to other tiles moved to the rhs TWICE, 1e-6 on the diagonal) and the
    oracle's per-tile Cholesky."""
    rng = np.random.default_rng(ts + len(sizes))
    n = int(np.prod(sizes))
    fo = oracle.LatticeField(sizes)
    fo.add_field_constraints(oracle.Weights(model_1=0.4, model_2=0.3))
    pos = np.stack([rng.uniform(0, s - 1, 60) for s in sizes], 1).astype(np.float32)
    fo.add_value_constraints(pos, rng.normal(size=60).astype(np.float32), 1.0)
    rows, cols, vals, rhs = fo.get()
    # drop every row that touches the last tile along every axis: that tile stays empty
    coords = np.stack(np.unravel_index(np.arange(n), sizes[::-1])[::-1], 1)
    last = np.all(coords // ts == (np.array(sizes) - 1) // ts, axis=1)
    bad_rows = np.unique(rows[last[cols]])
    keep = ~np.isin(rows, bad_rows)
    remap = -np.ones(len(rhs), np.int64)
    kept_rows = np.setdiff1d(np.arange(len(rhs)), bad_rows)
    remap[kept_rows] = np.arange(len(kept_rows))
    rows2, cols2, vals2, rhs2 = remap[rows[keep]].astype(np.int32), cols[keep], vals[keep], rhs[kept_rows]
    import scipy.sparse as sp
    A = sp.csr_matrix((vals2.astype(np.float64), (rows2, cols2)), shape=(len(rhs2), n))
    M = (A.T @ A).toarray()
    atb = A.T @ rhs2.astype(np.float64)
    g = rng.normal(size=n).astype(np.float32)
    tile_of = np.zeros(n, np.int64)
    for d in range(len(sizes) - 1, -1, -1):
        tile_of = tile_of * 64 + coords[:, d] // ts
    expect = g.astype(np.float64).copy()
    empty = 0
    for t in np.unique(tile_of):
        mine, other = np.where(tile_of == t)[0], np.where(tile_of != t)[0]
        if not M[np.ix_(mine, mine)].any():
            empty += 1
            continue                                    # nothing but the 1e-6 diagonal: the tile keeps the guess
        r = atb[mine] - 2.0 * M[np.ix_(mine, other)] @ g[other].astype(np.float64)
        expect[mine] = np.linalg.solve(M[np.ix_(mine, mine)] + 1e-6 * np.eye(len(mine)), r)
    assert empty >= 1
    f = fi.LatticeField(sizes, dtype=dtype)
    f.add_field_constraints(fi.Weights(model_2=0.0))
    f.add_rows_coo(rows2, cols2, vals2, rhs2)
    f.assemble()
    x = f.tile_pass(g, ts)
    tol = 1e-6 if dtype == "f64" else 5e-3
    assert np.abs(x - expect).max() <= tol * np.abs(expect).max()
    np.testing.assert_array_equal(x[last], g[last])                        # the empty tile: the guess, untouched
    # the oracle's tile_solver_square on the same rows
    fo2 = oracle.LatticeField(sizes)
    for r, c, v in zip(rows2, cols2, vals2):
        fo2.push_triplet(int(r), int(c), float(v))
    for v in rhs2:
        fo2.push_rhs(float(v))
    xo, _, _ = fo2.solve_tiled_with_guess(g, sizes, oracle.SolveOptions(tile=1, tile_size=ts, cg=0))
    assert np.abs(xo - expect).max() <= 5e-3 * np.abs(expect).max()
