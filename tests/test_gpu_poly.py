"""CG preconditioned by the Chebyshev polynomial (FI_OPT_POLY_TERMS; fi_solver.hip cg_run_poly, the epilogue of the plain
marching kernel in fi_stencil.hip).  The preconditioner changes the path to the solution, not the solution: the same
stop rule (sparse_linear.cpp:199-206 semantics, ||r|| <= tol ||Atb||), the same answers.

  * fp64, every term count, several model-weight combinations and lattice shapes (rows that are not a multiple of the
    16-byte group, tiles that overhang, lattices smaller than a tile): the solution equals the oracle's float64 direct
    solution to 1e-7 relative (BASELINE tolerance 1e-5) and the verified residual meets the tolerance;
  * the polynomial in the model rows + the data DIAGONAL must stay positive definite on data-heavy problems (many rows
    per cell, large data weights) -- CG would break down otherwise;
  * operator applications: about as many as Jacobi-PCG takes iterations (the point of the method is cheaper ones);
  * slabs (loop-back group): the decomposed solve takes the same outer iterations and gives the same solution;
  * cascade start with the polynomial on every level;
  * 2-D lattices run it through the tile kernel; contexts neither tiled kernel covers (1-D, model_3 / model_4,
    gradient_smoothness, triplet rows) ignore the option.
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


@pytest.mark.parametrize("terms", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("sizes,kw", [([14, 12, 13], dict()),
                                      ([18, 9, 10], dict(model_2=0.0, model_1=0.7)),
                                      ([9, 17, 12], dict(model_0=0.2, model_1=0.4, model_2=0.9)),
                                      ([21, 8, 8], dict(model_0=0.05, model_2=0.5))])
def test_solution_equals_direct_solution(oracle, fi, sizes, kw, terms):
    rng = np.random.default_rng(terms + sizes[0])
    pos, nrm = sphere_points(rng, sizes, 300)
    fo, fg = build_pair(oracle, fi, sizes, fi.Weights(**kw), pos, nrm, None, None, dtype="f64")
    x_ref = fo.solve_exact_f64()
    assert x_ref is not None
    fg.assemble()
    x0, it0, _ = fg.solve_cg(None, 0, 1e-12)
    fg.set_polynomial(terms)
    x, it, rel = fg.solve_cg(None, 0, 1e-12)
    assert x is not None and rel <= 1e-12 and fg.true_residual() <= 1.01e-12
    assert rel_inf(fg.solution_f64(), x_ref) <= 1e-7
    # operator applications: terms per outer iteration; within 2x of the Jacobi-PCG count (usually 1.0-1.5x; long
    # polynomials over the default interval [hi / 10, hi] pay more: 8 terms 2.0-2.5x)
    assert it * terms <= (2.0 if terms <= 5 else 3.0) * it0 + 2 * terms, (it, it0)


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-9)])
def test_value_constraints_odd_shapes(oracle, fi, dtype, tol):
    """Scattered value constraints (config-4 kind) on lattices whose rows are not a multiple of the 16-byte group."""
    for sizes in ([13, 11, 10], [34, 9, 9], [130, 5, 6]):
        rng = np.random.default_rng(sizes[0])
        pos, nrm, pw, val = random_points(rng, sizes, 400, margin=0.5)
        w = fi.Weights(model_2=0.5, data_gradient=0.0)
        fo, fg = build_pair(oracle, fi, sizes, w, pos, None, pw, val, dtype=dtype)
        x_ref = fo.solve_exact_f64()
        fg.assemble()
        fg.set_polynomial(4)
        x, it, rel = fg.solve_cg(None, 0, tol)
        assert x is not None and rel <= tol and fg.true_residual() <= 3 * tol
        if dtype == "f64":
            assert rel_inf(fg.solution_f64(), x_ref) <= 1e-5


def test_data_heavy_problem_keeps_the_preconditioner_positive(oracle, fi):
    """Thousands of rows per cell and large data weights: the data diagonal dominates the model diagonal."""
    sizes = [10, 9, 8]
    rng = np.random.default_rng(5)
    pos, nrm, pw, val = random_points(rng, sizes, 20000, margin=0.2, with_edge_cases=False)
    w = fi.Weights(model_2=0.05, data_pos=7.0, data_gradient=0.0)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, None, None, val, dtype="f64")
    x_ref = fo.solve_exact_f64()
    fg.assemble()
    for terms in (2, 3, 4, 6):
        fg.set_polynomial(terms)
        x, it, rel = fg.solve_cg(None, 0, 1e-11)
        assert x is not None and rel <= 1e-11
        assert rel_inf(fg.solution_f64(), x_ref) <= 1e-7


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("nranks", [2, 3])
def test_slabs_equal_undivided(fi, dtype, nranks):
    sizes = [16, 12, 24]
    rng = np.random.default_rng(nranks)
    pos, nrm = sphere_points(rng, sizes, 250)
    w = fi.Weights()
    one = fi.LatticeField(sizes, dtype=dtype)
    grp = fi.LatticeGroup(sizes, nranks, dtype=dtype)
    for f in (one, grp):
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        f.set_polynomial(4)
    tol = 1e-9 if dtype == "f64" else 1e-4
    x1, it1, rel1 = one.solve_cg(None, 0, tol)
    xg, itg, relg = grp.solve_cg(None, 0, tol)
    assert abs(itg - it1) <= max(2, it1 // 25), (itg, it1)
    assert relg <= tol and rel1 <= tol
    assert grp.true_residual() <= tol * 1.01
    assert rel_inf(grp.solution_f64(), one.solution_f64()) <= (1e-6 if dtype == "f64" else 2e-2)


def test_cascade_start_with_polynomial_levels(fi):
    from field_interpolation_amd import synth
    sizes, w, pos, val = synth.config4(side=64, num_points=15625, seed=3)
    f = fi.LatticeField(sizes, dtype="f32")
    f.add_field_constraints(w)
    f.set_levels(2, 1e-5)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.assemble()
    x0, it0, rel0 = f.solve_cg(None, 0, 1e-5)
    st0 = f.stats()
    f.set_polynomial(4)
    x1, it1, rel1 = f.solve_cg(None, 0, 1e-5)
    st1 = f.stats()
    assert rel1 <= 1e-5 and f.true_residual() <= 1.5e-5
    assert st1["num_levels"] == 3 and st1["coarse_iterations"] > 0
    assert it1 * 4 <= 2.0 * it0 + 8 and st1["coarse_iterations"] < st0["coarse_iterations"]
    assert np.abs(x1 - x0).max() <= 2e-3 * np.abs(x0).max()      # two fp32 solves to a 1e-5 residual


@pytest.mark.parametrize("sizes,kw,n", [([30, 28], dict(), 120), ([70, 41], dict(model_1=0.3, model_2=0.8), 400),
                                        ([131, 33], dict(model_0=0.1, model_2=0.5), 500), ([64, 64], dict(model_2=0.0, model_1=1.0), 300)])
@pytest.mark.parametrize("terms", [2, 3, 4, 6])
def test_two_dimensional_lattices(oracle, fi, sizes, kw, n, terms):
    """2-D lattices run the polynomial through the tile kernel's epilogue (fi_stencil2d.hip, Epi2 mode 2): the same answers
    as the oracle's float64 direct solve, fewer outer iterations than Jacobi-PCG takes steps."""
    rng = np.random.default_rng(terms + sizes[0])
    pos, nrm = sphere_points(rng, sizes, n)
    fo, fg = build_pair(oracle, fi, sizes, fi.Weights(**kw), pos, nrm, None, None, dtype="f64")
    x_ref = fo.solve_exact_f64()
    fg.assemble()
    x0, it0, _ = fg.solve_cg(None, 0, 1e-12)
    fg.set_polynomial(terms)
    x, it, rel = fg.solve_cg(None, 0, 1e-12)
    assert x is not None and rel <= 1e-12 and fg.true_residual() <= 1.01e-12
    assert rel_inf(fg.solution_f64(), x_ref) <= 1e-7
    assert it < it0 and it * terms <= (2.0 if terms <= 4 else 3.0) * it0 + 2 * terms, (it, it0)
    # fp32, and over slabs (loop-back group): the same outer iterations, the same solution
    one = fi.LatticeField(sizes, dtype="f32")
    grp = fi.LatticeGroup(sizes, 3, dtype="f32")
    w = fi.Weights(**kw)
    for f in (one, grp):
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.set_polynomial(terms)
        f.assemble()
    x1, it1, r1 = one.solve_cg(None, 0, 1e-5)
    xg, itg, rg = grp.solve_cg(None, 0, 1e-5)
    assert r1 <= 1e-5 and rg <= 1e-5 and abs(it1 - itg) <= max(2, it1 // 10)
    assert np.abs(xg - x1).max() <= 5e-3 * np.abs(x1).max()


def test_contexts_without_the_tiled_kernels_ignore_the_option(oracle, fi):
    """model_3 rows and 1-D lattices run the Jacobi-preconditioned recurrence whatever the option says."""
    for sizes, kw in (([40], dict()), ([10, 9, 8], dict(model_3=0.3))):
        rng = np.random.default_rng(1)
        pos, nrm = sphere_points(rng, sizes, 120)
        fo, fg = build_pair(oracle, fi, sizes, fi.Weights(**kw), pos, nrm, None, None, dtype="f64")
        fg.assemble()
        x0, it0, _ = fg.solve_cg(None, 0, 1e-10)
        fg.set_polynomial(4)
        x1, it1, _ = fg.solve_cg(None, 0, 1e-10)
        assert it1 == it0
        np.testing.assert_array_equal(x0, x1)


def test_timing_switches_do_nothing_in_the_shipped_library(fi, monkeypatch):
    """FI_DBG (timing modes whose results are wrong by construction) is read by timing builds only."""
    from field_interpolation_amd import synth
    sizes, w, pos, val = synth.config4(side=40, num_points=3815, seed=3)
    x = np.random.default_rng(0).normal(size=40 ** 3)

    def apply():
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        return f.apply_AtA(x)

    y0 = apply()
    monkeypatch.setenv("FI_DBG", "3")
    monkeypatch.setenv("FI_TXT", "16")
    np.testing.assert_array_equal(apply(), y0)


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-10)])
@pytest.mark.parametrize("sizes,kw", [([40, 36, 44], dict(model_2=0.5)), ([33, 30, 21], dict(model_1=0.3, model_2=0.6)),
                                      ([130, 9, 12], dict(model_2=0.0, model_1=0.8))])
def test_first_step_forms_its_operand_on_load(fi, sizes, kw, dtype, tol, monkeypatch):
    """Undivided lattices, 3 terms or more: the first step of the polynomial reads r and the bfloat16 scaling and forms
    z_0 = Dinv r / theta while it loads them (template flag PRO of the marching kernel), k_pcg_resid stores no z_0 and
    the second step recomputes it as its z_prev.  FI_NO_Z0_ON_LOAD keeps the stored z_0 (the form slabs use): the same
    arithmetic, so the same iteration count and -- up to the order of two roundings -- the same iterates."""
    rng = np.random.default_rng(sizes[0])
    pos, nrm, pw, val = random_points(rng, sizes, 3000, margin=0.5)
    w = fi.Weights(data_gradient=0.0, **kw)
    out = []
    for stored in (False, True):
        if stored:
            monkeypatch.setenv("FI_NO_Z0_ON_LOAD", "1")
        else:
            monkeypatch.delenv("FI_NO_Z0_ON_LOAD", raising=False)
        for terms in (3, 4, 6):
            f = fi.LatticeField(sizes, dtype=dtype)
            f.add_field_constraints(w)
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, pw, values=val)
            f.assemble()
            f.set_polynomial(terms)
            x, it, rel = f.solve_cg(None, 0, tol)
            assert f.stats()["converged"] == 1 and f.true_residual() <= 1.5 * tol
            out.append((it, f.solution_f64().copy()))
    monkeypatch.delenv("FI_NO_Z0_ON_LOAD", raising=False)
    for k in range(3):
        assert abs(out[k][0] - out[3 + k][0]) <= 1, (out[k][0], out[3 + k][0])
        assert rel_inf(out[k][1], out[3 + k][1]) <= (1e-3 if dtype == "f32" else 1e-8)


@pytest.mark.parametrize("scale,expect_jacobi", [(0.55, False), (0.2, True)])
def test_too_narrow_an_interval_is_widened_or_given_up(oracle, fi, monkeypatch, scale, expect_jacobi):
    """The polynomial is positive definite only while the spectrum of Dinv A~ stays below the interval's end; the bound is a
    power-method estimate (a lower bound + 10 %).  FI_POLY_LAMBDA_SCALE narrows the interval on purpose: CG then meets
    non-positive curvature, the solve widens the interval (x 1.25, twice, kept for the context's later solves) and goes
    on from its last iterate -- or, when that is not enough, finishes with the Jacobi diagonal.  Same answer either way."""
    sizes = [40, 36, 33]
    rng = np.random.default_rng(11)
    pos, nrm, pw, val = random_points(rng, sizes, 3000, margin=0.5, with_edge_cases=False)
    w = fi.Weights(model_2=0.5, data_gradient=0.0)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, None, None, val, dtype="f64")
    fg.set_polynomial(4, 30.0)
    fg.assemble()
    x0, it0, rel0 = fg.solve_cg(None, 0, 1e-10)
    ref = fg.solution_f64().copy()
    monkeypatch.setenv("FI_POLY_LAMBDA_SCALE", str(scale))
    x1, it1, rel1 = fg.solve_cg(None, 0, 1e-10)
    st = fg.stats()
    assert x1 is not None and st["converged"] == 1 and fg.true_residual() <= 1.01e-10
    assert rel_inf(fg.solution_f64(), ref) <= 1e-7
    if expect_jacobi:
        assert it1 > 3 * it0          # Jacobi-PCG steps, one operator application each
    monkeypatch.delenv("FI_POLY_LAMBDA_SCALE")
    # ... and a wide ratio (a long interval has the least headroom) on a larger lattice
    sizes = [96, 96, 96]
    from field_interpolation_amd import synth
    sz, w4, p4, v4 = synth.config4(side=96, num_points=52734, seed=3)
    f = fi.LatticeField(sz, dtype="f32")
    f.add_field_constraints(w4)
    f.set_polynomial(6, 300.0)
    f.add_points(w4.data_pos, w4.value_kernel, 0.0, w4.gradient_kernel, p4, None, None, values=v4)
    f.assemble()
    res = f.solve_cg(None, 0, 1e-5)
    assert res is not None and f.stats()["converged"] == 1 and f.true_residual() <= 1.5e-5


@pytest.mark.parametrize("dtype,terms,nranks,sizes", [("f32", 4, 4, [40, 36, 64]), ("f64", 4, 3, [24, 20, 40]), ("f32", 3, 2, [33, 30, 26]),
                                                      ("f32", 5, 2, [48, 16, 40])])
def test_deep_exchange_equals_one_exchange_per_step(fi, monkeypatch, dtype, terms, nranks, sizes):
    """Slabs: with the polynomial set before the assemble the vectors carry 2 (d - 1) ghost planes; r's travel once per
    polynomial and every step also computes the ghost planes the next one reads (what the neighbour computes for its own
    planes, bit for bit).  The same iterates as with one exchange per step (FI_NO_DEEP_HALO): equal iteration counts, equal
    bits -- with a cascade level below, whose slabs are too thin for the deep exchange and fall back by themselves."""
    from field_interpolation_amd import synth
    rng = np.random.default_rng(terms)
    pos, nrm, pw, val = random_points(rng, sizes, 3000, margin=0.5, with_edge_cases=False)
    w = fi.Weights(model_2=0.5, data_gradient=0.0)
    out = []
    for deep in (True, False):
        if deep:
            monkeypatch.delenv("FI_NO_DEEP_HALO", raising=False)
        else:
            monkeypatch.setenv("FI_NO_DEEP_HALO", "1")
        g = fi.LatticeGroup(sizes, nranks, dtype=dtype)
        g.add_field_constraints(w)
        g.set_polynomial(terms, 30.0)
        g.set_levels(1, 1e-5)
        g.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        g.assemble()
        x, it, rel = g.solve_cg(None, 0, 1e-5 if dtype == "f32" else 1e-9)
        st = g.stats()
        out.append((x.copy(), it, rel, st["coarse_iterations"], g.true_residual(), st["halo_exchanges"], st["reductions"]))
        del g
    (xd, itd, reld, cd, td, exd, red), (xs, its, rels, cs, ts, exs, res) = out
    assert itd == its and cd == cs, (itd, its, cd, cs)
    # per pass of the recurrence (start, each outer iteration, each restart): 1 exchange for the apply + 1 for the polynomial
    # instead of 1 + (d - 1); two reductions either way
    passes = (exs - exd) // (terms - 2)
    assert exs - exd == passes * (terms - 2) and passes >= itd + 1, (exd, exs, itd)
    assert exd <= 2 * passes + 2 + passes // 8 + 2 and red == res   # (+ the applies that replace s = A p every 16th step)
    np.testing.assert_array_equal(xd, xs)
    assert td <= 1.5 * (1e-5 if dtype == "f32" else 1e-9)
    # and the undivided solve agrees
    one = fi.LatticeField(sizes, dtype=dtype)
    one.add_field_constraints(w)
    one.set_polynomial(terms, 30.0)
    one.set_levels(1, 1e-5)
    one.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    one.assemble()
    x1, it1, _ = one.solve_cg(None, 0, 1e-5 if dtype == "f32" else 1e-9)
    assert abs(it1 - itd) <= max(2, it1 // 10)
    assert np.abs(x1 - xd).max() <= (2e-2 if dtype == "f32" else 1e-6) * np.abs(x1).max()


@pytest.mark.parametrize("dtype,tol,sizes,nranks", [("f64", 1e-9, [16, 12, 24], 2), ("f32", 1e-4, [16, 12, 24], 3),
                                                    ("f32", 1e-5, [40, 36, 64], 4)])
def test_single_reduction_recurrence_over_slabs(fi, monkeypatch, dtype, tol, sizes, nranks):
    """Over slabs the polynomial PCG runs in the Chronopoulos-Gear form (cg_run_poly_sr): r.z, z.Az and r.r in ONE
    all-reduce per outer iteration.  The same iteration counts (+- a few) and the same solution as the two-reduction form
    (FI_NO_SINGLE_REDUCTION) and as the undivided solve -- on an ill-conditioned SDF system of several hundred outer
    iterations too; one reduction per outer iteration (+ start and verification) instead of two."""
    rng = np.random.default_rng(nranks)
    pos, nrm = sphere_points(rng, sizes, 250 if sizes[0] < 20 else 3000)
    w = fi.Weights()

    def solve(single):
        if single:
            monkeypatch.delenv("FI_NO_SINGLE_REDUCTION", raising=False)
        else:
            monkeypatch.setenv("FI_NO_SINGLE_REDUCTION", "1")
        g = fi.LatticeGroup(sizes, nranks, dtype=dtype)
        g.add_field_constraints(w)
        g.set_polynomial(4)
        g.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        g.assemble()
        x, it, rel = g.solve_cg(None, 0, tol)
        st = g.stats()
        assert rel <= tol and g.true_residual() <= 1.01 * tol
        return g.solution_f64().copy(), it, st["reductions"], st["halo_exchanges"]

    xs, its, reds, exs = solve(True)
    xt, itt, redt, ext = solve(False)
    assert abs(its - itt) <= max(3, itt // 12), (its, itt)
    assert rel_inf(xs, xt) <= (1e-6 if dtype == "f64" else 3e-2)
    assert reds <= its + 8 and redt >= 2 * itt, (reds, its, redt, itt)
    one = fi.LatticeField(sizes, dtype=dtype)
    one.add_field_constraints(w)
    one.set_polynomial(4)
    one.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    one.assemble()
    x1, it1, _ = one.solve_cg(None, 0, tol)
    assert abs(its - it1) <= max(3, it1 // 12), (its, it1)


def test_bound_of_the_polynomial_is_remembered_per_lattice(fi, monkeypatch):
    """The polynomial's eigenvalue bound depends on the lattice's extents and the model weights only; a process remembers
    it (a context that lives for one solve would run the power method for a number already known).  Remembered or
    recomputed (FI_NO_LAMBDA_CACHE): the same bits, so the same iterates."""
    sizes = [44, 36, 40]
    rng = np.random.default_rng(23)
    pos = np.stack([rng.uniform(0, s - 1, 4000) for s in sizes], axis=1).astype(np.float32)
    val = rng.normal(size=len(pos)).astype(np.float32)
    w = fi.Weights(model_2=0.45, model_1=0.07, data_gradient=0.0)
    got = []
    for no_cache in (False, False, True):
        if no_cache:
            monkeypatch.setenv("FI_NO_LAMBDA_CACHE", "1")
        f = fi.LatticeField(sizes, dtype="f32")
        f.add_field_constraints(w)
        f.set_polynomial(4, 30.0)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-6)
        got.append((x.copy(), it))
    monkeypatch.delenv("FI_NO_LAMBDA_CACHE", raising=False)
    for x, it in got[1:]:
        assert it == got[0][1] and np.array_equal(x, got[0][0])
