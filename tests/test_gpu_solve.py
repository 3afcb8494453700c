"""P3 parity on the GPU: solutions of the HIP solvers against the oracle.

  * fp64 CG driven to 1e-12 vs the float64 direct solution:  ||x-x*||_inf/||x*||_inf <= 1e-5
    (BASELINE.json: "field values within 1e-5 relative of the CPU reference"); observed ~1e-9.
  * same recurrence on both sides (GPU PCG vs oracle PCG, fp64): iteration counts agree and the
    iterates agree to 1e-9.
  * fp32 CG to the reference's stop rule: true residual within 3x of the requested tolerance.
  * weighted Jacobi vs jacobi_iterations (fp32 on both sides): 2e-5 relative.
  * the SURVEY known answer for the field_1d default input.
"""
import numpy as np
import pytest

from util import build_pair, random_points, rel_inf, sphere_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fi():
    import field_interpolation_amd as fi
    from field_interpolation_amd import _capi
    assert _capi.device_count() >= 1, "no HIP device visible"
    return fi


def _sdf_pair(oracle, fi, sizes, n, dtype, seed=0, **kw):
    rng = np.random.default_rng(seed)
    pos, nrm = sphere_points(rng, sizes, n)
    return build_pair(oracle, fi, sizes, fi.Weights(**kw), pos, nrm, None, None, dtype=dtype)


@pytest.mark.parametrize("sizes,n", [([96], 6), ([40, 36], 150), ([14, 12, 13], 300)])
def test_fp64_cg_matches_direct_solution(oracle, fi, sizes, n):
    fo, fg = _sdf_pair(oracle, fi, sizes, n, "f64")
    x_ref = fo.solve_exact_f64()
    assert x_ref is not None
    x = fi.solve_sparse_linear_exact(fg)
    assert x is not None
    xd = fg.solution_f64()
    assert rel_inf(xd, x_ref) <= 1e-5           # BASELINE.json tolerance
    assert rel_inf(xd, x_ref) <= 1e-7           # what fp64 actually delivers
    assert fg.true_residual() <= 1e-11
    np.testing.assert_allclose(x, xd.astype(np.float32))


def test_field_1d_known_answer(fi):
    """float64 least-squares solution of the field_1d.cpp:20-29 default input."""
    from field_interpolation_amd import synth
    expected = np.array([-0.1846154, -0.1006993, -0.0167832, 0.0671329, 0.1230769, 0.1510490,
                         0.1510490, 0.1230769, 0.0671329, -0.0167832, -0.1006993, -0.1846154])
    sizes, w, pts = synth.config1(12)
    f = fi.LatticeField(sizes, dtype="f64")
    for pos, value, grad in pts:                                   # field_1d.cpp:100-105
        f.add_value_constraint([pos], value, w.data_pos)
        f.add_gradient_constraint([pos], [grad], w.data_gradient, w.gradient_kernel)
    f.add_field_constraints(w)                                     # field_1d.cpp:107
    x = fi.solve_sparse_linear_exact(f)
    np.testing.assert_allclose(x, expected, rtol=0, atol=2e-7)


def test_readme_known_answers(fi):
    """The reference README's worked statements, on the GPU without the oracle: the 6-point example (README.md:11-40:
    least-squares solution of its printed A and b), and "f(0) = 10, f(10) = 0 with a smoothness constraint lets the
    solver figure out that f(20) = -10" (the straight line 10 - x)."""
    f = fi.LatticeField([21], dtype="f64")
    assert f.add_value_constraint([0.0], 10.0, 1.0)
    assert f.add_value_constraint([10.0], 0.0, 1.0)
    f.add_field_constraints(fi.Weights(model_2=0.5))
    x = fi.solve_sparse_linear_exact(f)
    np.testing.assert_allclose(x, 10.0 - np.arange(21), rtol=0, atol=1e-5)
    A = np.array([[1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], [-1, 1, 0, 0, 0, 0], [0, 0, 0, 0, -1, 1],
                  [1, -2, 1, 0, 0, 0], [0, 1, -2, 1, 0, 0], [0, 0, 1, -2, 1, 0], [0, 0, 0, 1, -2, 1]], np.float64)
    b = np.array([4, 2, 1, -1, 0, 0, 0, 0], np.float64)
    g = fi.LatticeField([6], dtype="f64")
    assert g.add_value_constraint([0.0], 4.0, 1.0) and g.add_value_constraint([5.0], 2.0, 1.0)
    assert g.add_gradient_constraint([0.0], [1.0], 1.0, fi.GradientKernel.kNearestNeighbor)
    rows, cols = np.nonzero(A[3:4])
    g.add_rows_coo(rows, cols, A[3, cols], b[3:4])                 # f(5) - f(4) = -1 sits on the border: a raw row
    g.add_field_constraints(fi.Weights(model_2=1.0))
    xg = fi.solve_sparse_linear_exact(g)
    np.testing.assert_allclose(xg, np.linalg.lstsq(A, b, rcond=None)[0], rtol=0, atol=2e-6)


@pytest.mark.parametrize("term,degree,weight", [("model_1", 0, 1e3), ("model_2", 1, 1e3), ("model_3", 2, 1e4),
                                                ("model_4", 3, 1e5)])
def test_header_limit_behaviour_of_the_model_weights(fi, term, degree, weight):
    """field_interpolation.hpp:79-85 on the GPU, without the oracle: a large model_1 averages the data, model_2 fits a
    line, model_3 a quadratic, model_4 a cubic (fp64 CG to 1e-12; the answer is numpy's least-squares polynomial)."""
    rng = np.random.default_rng(degree)
    n = 40
    at = np.sort(rng.choice(n, 12, replace=False))
    val = rng.normal(size=12)
    f = fi.LatticeField([n], dtype="f64")
    for p, v in zip(at, val):
        assert f.add_value_constraint([float(p)], float(v), 1.0)
    f.add_field_constraints(fi.Weights(**{"model_2": 0.0, term: weight}))
    x = fi.solve_sparse_linear_exact(f)
    assert x is not None
    fit = np.polyval(np.polyfit(at, val, degree), np.arange(n))
    assert np.abs(f.solution_f64() - fit).max() <= 2e-3 * (np.abs(val).max() + np.abs(fit).max())


def test_config1_1024(oracle, fi):
    """BASELINE config 1: 1-D lattice, 1024 points, 2 value + 2 gradient constraints."""
    from field_interpolation_amd import synth
    sizes, w, pts = synth.config1(1024)
    fo = oracle.LatticeField(sizes)
    f = fi.LatticeField(sizes, dtype="f64")
    for pos, value, grad in pts:
        assert fo.add_value_constraint([pos], float(value), w.data_pos) == f.add_value_constraint([pos], value, w.data_pos)
        assert (fo.add_gradient_constraint([pos], [grad], w.data_gradient, int(w.gradient_kernel))
                == f.add_gradient_constraint([pos], [grad], w.data_gradient, w.gradient_kernel))
    from util import oracle_weights
    fo.add_field_constraints(oracle_weights(oracle, w))
    f.add_field_constraints(w)
    x_ref = fo.solve_exact_f64()
    x = fi.solve_sparse_linear_exact(f, tolerance=1e-13, max_iterations=200000)
    assert x is not None
    assert rel_inf(f.solution_f64(), x_ref) <= 1e-5


@pytest.mark.parametrize("sizes,n", [([30, 28], 120), ([10, 11, 12], 200)])
def test_same_recurrence_same_iterates(oracle, fi, sizes, n):
    fo, fg = _sdf_pair(oracle, fi, sizes, n, "f64", model_1=0.2)
    N = fo.num_unknowns
    guess = np.random.default_rng(5).normal(size=N).astype(np.float32)
    for max_it in (1, 7, 40):
        xo, ito, erro = fo.solve_pcg(guess, max_it, 1e-30, use_double=True)
        res = fg.solve_cg(guess, max_it, 1e-30)
        assert res is not None
        _, itg, errg = res
        assert itg == ito == max_it
        assert rel_inf(fg.solution_f64(), xo) <= 1e-9
        assert abs(errg - erro) <= 1e-6 * erro
    xo, ito, erro = fo.solve_pcg(guess, 0, 1e-8, use_double=True)
    _, itg, errg = fg.solve_cg(guess, 0, 1e-8)
    assert abs(itg - ito) <= 1 and errg <= 1e-8


@pytest.mark.parametrize("sizes,n", [([48, 40], 200), ([16, 16, 16], 400)])
def test_fp32_cg_reaches_reference_stop_rule(oracle, fi, sizes, n):
    fo, fg = _sdf_pair(oracle, fi, sizes, n, "f32", model_0=0.05)
    AtA, atb, _ = fo.normal_equations()
    tol = 1e-4
    res = fi.solve_sparse_linear_with_guess(fg, np.zeros(fo.num_unknowns, np.float32), 0, tol)
    assert res is not None
    st = fg.stats()
    assert st["converged"] == 1 and st["rel_residual"] <= tol
    true_rel = np.linalg.norm(atb - AtA @ res.astype(np.float64)) / np.linalg.norm(atb)
    assert true_rel <= 3 * tol
    assert abs(fg.true_residual() - true_rel) <= 0.05 * true_rel + 1e-7
    # the reference's own solver (BiCGSTAB restatement, fp32) reaches the same system's solution
    xb, itb, errb = fo.solve_with_guess(np.zeros(fo.num_unknowns), 0, tol)
    x64 = fo.solve_exact_f64()
    assert rel_inf(res, x64) <= 50 * max(rel_inf(xb, x64), 1e-4)


def test_max_iterations_and_defaults(oracle, fi):
    fo, fg = _sdf_pair(oracle, fi, [20, 20], 80, "f32")
    out, it, rel = fg.solve_cg(None, 5, 1e-12)
    assert it == 5 and fg.stats()["converged"] == 0
    # zero rhs -> x = 0, no iterations (Eigen's early return; oracle does the same)
    f = fi.LatticeField([12, 12])
    f.add_field_constraints(fi.Weights())
    out, it, rel = f.solve_cg(np.ones(144, np.float32), 0, 0.0)
    assert it == 0 and not out.any()
    # solve_tiled_with_guess: wrong guess length -> None (sparse_linear.cpp:402-405)
    assert fi.solve_tiled_with_guess(fg, np.zeros(399, np.float32), [20, 20], fi.SolveOptions()) is None
    x = fi.solve_tiled_with_guess(fg, np.zeros(400, np.float32), [20, 20], fi.SolveOptions())
    assert x is not None and fg.stats()["rel_residual"] <= 1e-3      # SolveOptions default tolerance


@pytest.mark.parametrize("sizes,n", [([64], 5), ([18, 17], 90), ([9, 10, 8], 150)])
def test_jacobi_iterations(oracle, fi, sizes, n):
    fo, fg = _sdf_pair(oracle, fi, sizes, n, "f32")
    N = fo.num_unknowns
    guess = np.random.default_rng(2).normal(size=N).astype(np.float32)
    for sweeps, w in ((1, 0.5), (9, 2.0 / 3.0)):
        x_ref = fo.jacobi_iterations(guess, sweeps, w)
        x = fi.jacobi_iterations(fg, guess, sweeps, w)
        assert rel_inf(x, x_ref) <= 2e-5
    np.testing.assert_array_equal(fi.jacobi_iterations(fg, guess, 0, 0.5), guess)


def test_value_targets_config2_style(oracle, fi):
    """Noisy value constraints + strong smoothness prior (BASELINE config 2, scaled down)."""
    from field_interpolation_amd import synth
    sizes, w, pos, val = synth.config2(side=48, num_points=400, seed=1)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, None, None, val, dtype="f64")
    x_ref = fo.solve_exact_f64()
    x = fi.solve_sparse_linear_exact(fg, tolerance=1e-12)
    assert x is not None and rel_inf(fg.solution_f64(), x_ref) <= 1e-5


def test_upscale_field_bit_exact(oracle, fi):
    rng = np.random.default_rng(9)
    for small, large in (([7], [20]), ([5, 6], [17, 11]), ([4, 3, 5], [9, 8, 11]), ([6, 6], [6, 6]), ([1, 3], [4, 7])):
        f = rng.normal(size=int(np.prod(small))).astype(np.float32)
        got = fi.upscale_field(f, small, large)
        ref = oracle.upscale_field(f, small, large)
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes,ts,kw", [([20, 18], 8, dict()), ([37], 16, dict(model_1=0.3)),
                                        ([12, 9, 10], 4, dict(model_0=0.1, model_1=0.2)),
                                        ([14, 10], 5, dict(model_2=0.3, model_3=0.2, gradient_smoothness=0.3)),
                                        ([9, 8, 7], 16, dict())])
def test_tile_pass_equals_tile_solver(oracle, fi, dtype, sizes, ts, kw):
    """fi_tile_pass against the oracle's tile_solver_square (sparse_linear.cpp:246-390) and against a dense
    float64 re-derivation: every tile solved with its couplings moved to the rhs TWICE (:327-334), 1e-6 on
    the diagonal.  fp64: 1e-7 of the largest entry (CG to 1e-12); fp32: 5e-3 (the reference's float Cholesky
    and CG-to-1e-6 in fp32 both sit at that level on these kappa ~ 1e4 tiles)."""
    rng = np.random.default_rng(len(sizes) + ts)
    pos, nrm, pw, val = random_points(rng, sizes, 150, margin=0.8)
    w = fi.Weights(data_pos=0.8, data_gradient=1.25, **kw)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, None, dtype=dtype)
    n = int(np.prod(sizes))
    AtA, atb, _ = fo.normal_equations()
    M = AtA.toarray()
    g = rng.normal(size=n).astype(np.float32)
    coords = np.stack(np.unravel_index(np.arange(n), sizes[::-1])[::-1], 1)     # x fastest
    tile_of = np.zeros(n, np.int64)
    for d in range(len(sizes) - 1, -1, -1):
        tile_of = tile_of * 64 + coords[:, d] // ts
    expect = g.astype(np.float64).copy()
    for t in np.unique(tile_of):
        mine, other = np.where(tile_of == t)[0], np.where(tile_of != t)[0]
        rhs = atb[mine] - 2.0 * M[np.ix_(mine, other)] @ g[other].astype(np.float64)
        expect[mine] = np.linalg.solve(M[np.ix_(mine, mine)] + 1e-6 * np.eye(len(mine)), rhs)
    x = fg.tile_pass(g, ts)
    tol = 1e-7 if dtype == "f64" else 5e-3
    assert np.abs(fg.solution_f64() - expect).max() <= tol * np.abs(expect).max()
    assert x.dtype == np.float32 and np.abs(x - expect).max() <= max(tol, 1e-6) * np.abs(expect).max()
    # the oracle (float Cholesky per tile) agrees at fp32 level
    o = oracle.SolveOptions(tile=1, tile_size=ts, cg=0)
    xo, _, _ = fo.solve_tiled_with_guess(g, sizes, o)
    assert np.abs(xo - expect).max() <= 5e-3 * np.abs(expect).max()
    # SolveOptions.tile through the mirrored entry point: tile pass, then CG from its result
    so = fi.SolveOptions(tile=True, tile_size=ts, cg=True, error_tolerance=1e-5 if dtype == "f32" else 1e-9)
    xs = fi.solve_tiled_with_guess(fg, np.zeros(n, np.float32), sizes, so)
    assert fg.true_residual() <= so.error_tolerance * 1.01
    with pytest.raises(fi.FiError):
        fg.tile_pass(g, 1)                                     # CHECK_GE_F(tile_size, 2), sparse_linear.cpp:254


def test_tile_pass_with_a_rank_one_cell_of_many_rows(oracle, fi, monkeypatch):
    """a 3-D fp32 cell that holds MORE than 8 data rows whose block has rank one (20 nearest-neighbour value rows
    on one corner) keeps ONE Cholesky factor row; the packed block the tile pre-solver asks for (ensure_cell_blocks on a
    context whose marching kernel owns the cells) must be that row's outer product -- the sum over all 20 rows -- and not
    the outer product of the first data row.  Sparse single-row cells around it keep the context on factor rows (no
    packed blocks).  Compared with the blocks the assembly stores itself (FI_KEEP_BLOCKS) and with the oracle's tile solver."""
    sizes = [24, 20, 16]
    rng = np.random.default_rng(77)
    octant = (np.array([7.0, 9.0, 5.0]) + rng.uniform(0.02, 0.45, size=(20, 3))).astype(np.float32)   # all round to (7, 9, 5)
    singles = np.stack([rng.uniform(0.5, s - 1.5, 300) for s in sizes], 1).astype(np.float32)
    pos = np.concatenate([octant, singles])
    nrm = rng.normal(size=pos.shape).astype(np.float32)
    val = rng.normal(size=len(pos)).astype(np.float32)
    w = fi.Weights(value_kernel=fi.ValueKernel(0), data_gradient=0.0)
    g = rng.normal(size=int(np.prod(sizes))).astype(np.float32)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, None, val, dtype="f32")
    x = fg.tile_pass(g, 8)
    monkeypatch.setenv("FI_KEEP_BLOCKS", "1")
    _, fk = build_pair(oracle, fi, sizes, w, pos, nrm, None, val, dtype="f32")
    xk = fk.tile_pass(g, 8)
    assert np.abs(x - xk).max() <= 2e-5 * np.abs(xk).max()
    xo, _, _ = fo.solve_tiled_with_guess(g, sizes, oracle.SolveOptions(tile=1, tile_size=8, cg=0))
    assert np.abs(x - xo).max() <= 5e-3 * np.abs(xo).max()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("sizes,kw,gk", [([40], dict(), 1), ([17, 12], dict(model_1=0.3), 1),
                                        ([9, 8, 7], dict(model_0=0.2, model_3=0.4, gradient_smoothness=0.3), 0),
                                        ([14, 11], dict(model_4=0.2), 2), ([8, 7, 9], dict(), 2)])
def test_error_map_equals_reference_blame(oracle, fi, dtype, sizes, kw, gk):
    """fi_error_map against the oracle's generate_error_map (field_interpolation.cpp:402-429) on the very same
    rows: model rows (every order), cell rows, and the generic rows of GradientKernel::kLinearInterpolation
    (duplicate (row, col) entries blamed per triplet).  2e-4 of the largest entry (the oracle sums in fp32)."""
    rng = np.random.default_rng(len(sizes) * 5 + gk)
    pos, nrm, pw, val = random_points(rng, sizes, 80, margin=0.8)
    w = fi.Weights(data_pos=0.8, data_gradient=1.25, gradient_kernel=fi.GradientKernel(gk), **kw)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, pw, None, dtype=dtype)
    x = rng.normal(size=int(np.prod(sizes))).astype(np.float32)
    expect = fo.error_map(x)
    heat = fi.generate_error_map(fg, x)
    assert heat.dtype == np.float32 and heat.shape == expect.shape
    assert np.abs(heat - expect).max() <= 2e-4 * np.abs(expect).max()
    # an exact solution of a consistent system blames nobody: model rows only, constant field
    f0 = fi.LatticeField(sizes, dtype=dtype)
    f0.add_field_constraints(fi.Weights(model_1=0.3, model_2=0.5))
    f0.assemble()
    assert np.abs(f0.error_map(np.full(int(np.prod(sizes)), 3.0, np.float32))).max() <= 1e-10
