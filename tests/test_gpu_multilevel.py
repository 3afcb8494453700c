"""Multilevel solves on the GPU: coarse-to-fine cascade start (FI_OPT_LEVELS; the reference's recipe of
src/sdf_field.cpp:272-288 generalised) and V-cycle preconditioned CG (FI_OPT_MULTIGRID).  Neither changes the
system being solved: the converged solution must equal the plain Jacobi-PCG / oracle solution."""
import numpy as np
import pytest

from util import build_pair, rel_inf, sphere_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fi():
    import field_interpolation_amd as fi
    from field_interpolation_amd import _capi
    assert _capi.device_count() >= 1
    return fi


def _problem(oracle, fi, sizes, dtype, n=600, kw=None):
    rng = np.random.default_rng(2)
    pos, nrm = sphere_points(rng, sizes, n)
    w = fi.Weights(**(kw or {}))
    return build_pair(oracle, fi, sizes, w, pos, nrm, None, None, dtype=dtype), (w, pos, nrm)


@pytest.mark.parametrize("sizes", [[40, 36], [24, 20, 28], [33, 17, 19]])
@pytest.mark.parametrize("mode", ["cascade", "multigrid"])
def test_multilevel_converges_to_the_same_solution(oracle, fi, sizes, mode):
    (fo, fg), (w, pos, nrm) = _problem(oracle, fi, sizes, "f64")
    x_ref = fo.solve_exact_f64()
    plain_iters = fg.solve_cg(None, 0, 1e-10)[1]
    ml = fi.LatticeField(sizes, dtype="f64")
    ml.add_field_constraints(w)
    ml.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    ml.set_levels(3)
    if mode == "multigrid":
        ml.set_multigrid(True)
    ml.assemble()
    st = ml.stats()
    assert st["num_levels"] >= 2
    x, iters, rel = ml.solve_cg(None, 0, 1e-10)
    assert rel <= 1e-10 and ml.stats()["converged"] == 1
    assert rel_inf(ml.solution_f64(), x_ref) <= 1e-5          # BASELINE.json tolerance
    assert rel_inf(ml.solution_f64(), x_ref) <= 1e-6
    assert iters < plain_iters                                   # the point of having levels
    if mode == "multigrid":
        assert iters <= plain_iters // 3


def test_levels_stop_at_small_lattices_and_ignore_hand_built_rows(fi):
    f = fi.LatticeField([20, 9])          # 9 -> 5 < 8: no coarser level
    f.add_field_constraints(fi.Weights())
    f.set_levels(4)
    f.add_points(1.0, 1, 0.0, 1, np.array([[3.2, 4.1]], np.float32), None, None, values=np.array([1.0], np.float32))
    f.assemble()
    assert f.stats()["num_levels"] == 1
    g = fi.LatticeField([64, 64])
    g.add_field_constraints(fi.Weights())
    g.set_levels(2)
    g.add_points(1.0, 1, 0.0, 1, np.array([[3.2, 4.1]], np.float32), None, None, values=np.array([1.0], np.float32))
    g.add_rows_coo(np.array([0]), np.array([7]), np.array([1.0], np.float32), np.array([2.0], np.float32))
    g.assemble()
    assert g.stats()["num_levels"] == 1     # hand-built rows have no geometry to coarsen
    x, it, rel = g.solve_cg(None, 0, 1e-4)
    assert rel <= 1e-4


def test_gradient_linear_kernel_is_coarsened_too(oracle, fi):
    sizes = [32, 36]
    rng = np.random.default_rng(8)
    pos, nrm = sphere_points(rng, sizes, 300)
    w = fi.Weights(gradient_kernel=fi.GradientKernel.kLinearInterpolation)
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm, None, None, dtype="f64")
    fg.set_levels(2)
    fg.set_multigrid(True)
    fg.assemble()
    assert fg.stats()["num_levels"] == 3
    x, it, rel = fg.solve_cg(None, 0, 1e-10)
    assert rel_inf(fg.solution_f64(), fo.solve_exact_f64()) <= 1e-6


def test_fp32_multigrid_reaches_a_verified_residual(oracle, fi):
    sizes = [48, 40, 44]
    (fo, fg), _ = _problem(oracle, fi, sizes, "f32", n=2500)
    fg.set_levels(3)
    fg.set_multigrid(True)
    fg.assemble()
    x, it, rel = fg.solve_cg(None, 0, 1e-5)
    st = fg.stats()
    assert st["converged"] == 1 and st["restarts"] >= 1 and st["verified_residual"] <= 1e-5
    assert fg.true_residual() <= 1.2e-5
    # independent check of the residual with the oracle's rows: ||A^T(A x - b)|| / ||A^T b||
    rows, cols, vals, rhs = fo.get()
    r = np.zeros(len(rhs))
    np.add.at(r, rows, vals.astype(np.float64) * x.astype(np.float64)[cols])
    r -= rhs
    g = np.zeros(fo.num_unknowns)
    np.add.at(g, cols, vals.astype(np.float64) * r[rows])
    gb = np.zeros(fo.num_unknowns)
    np.add.at(gb, cols, vals.astype(np.float64) * rhs.astype(np.float64)[rows])
    assert np.linalg.norm(g) / np.linalg.norm(gb) <= 2e-5


@pytest.mark.parametrize("sizes,levels,gk", [([96, 80], 3, 1), ([40, 32, 48], 2, 1), ([48, 40, 32], 1, 1), ([80, 64], 2, 2),
                                            ([32, 40, 32], 1, 0)])
def test_mixed_precision_vcycle(oracle, fi, sizes, levels, gk):
    """FI_OPT_MIXED_PRECISION: CG in fp64 with the V-cycle preconditioner on an fp32 replica.  The stop test is the
    fp64 residual, so 1e-10 is reached although fp32 alone stalls near 1e-4 on these SDF systems; iteration counts
    stay those of the pure-fp64 solve (+-10 %: the preconditioner is only perturbed by fp32 rounding); the solution
    equals the pure-fp64 one to 1e-7 and the float64 direct solution of the oracle's rows to 1e-6."""
    rng = np.random.default_rng(5)
    pos, nrm = sphere_points(rng, sizes, 800)
    w = fi.Weights(gradient_kernel=fi.GradientKernel(gk))     # 2: rows 3 points wide -> the generic sparse-row path
    fo, pure = build_pair(oracle, fi, sizes, w, pos, nrm, None, None, dtype="f64")
    mixed = fi.LatticeField(sizes, dtype="f64")
    mixed.add_field_constraints(w)
    mixed.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    for f in (pure, mixed):
        f.set_levels(levels)
        f.set_multigrid(True)
    mixed.set_mixed_precision(True)
    mixed.set_mg_smoother(False)      # like for like: the pure-fp64 V-cycle smooths with the Chebyshev polynomial in A
    for f in (pure, mixed):
        f.assemble()
    assert mixed.stats()["num_levels"] == pure.stats()["num_levels"] == levels + 1
    tol = 1e-10
    xp, itp, rp = pure.solve_cg(None, 0, tol)
    xm, itm, rm = mixed.solve_cg(None, 0, tol)
    st = mixed.stats()
    assert st["converged"] == 1 and rm <= tol
    assert mixed.true_residual() <= tol * 1.01
    assert abs(itm - itp) <= max(2, itp // 10)
    assert rel_inf(mixed.solution_f64(), pure.solution_f64()) <= 1e-7
    if len(sizes) == 2:   # sparse Cholesky of a 3-D lattice this size takes minutes on the CPU
        x64 = fo.solve_exact_f64()
        assert rel_inf(mixed.solution_f64(), x64) <= 1e-6
    # the default smoother of fp32 3-D levels (the polynomial in A_model + f diag(A_data)): another preconditioner, the same
    # answer, about as many iterations
    if len(sizes) == 3:
        mp = fi.LatticeField(sizes, dtype="f64")
        mp.add_field_constraints(w)
        mp.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        mp.set_levels(levels)
        mp.set_multigrid(True)
        mp.set_mixed_precision(True)
        mp.assemble()
        xq, itq, rq = mp.solve_cg(None, 0, tol)
        assert mp.stats()["converged"] == 1 and mp.true_residual() <= tol * 1.01
        assert itq <= itp + max(3, itp // 3), (itq, itp)
        assert rel_inf(mp.solution_f64(), pure.solution_f64()) <= 1e-7
    # a caller's guess is honoured (warm start from the fp32-rounded answer: a handful of iterations)
    xw, itw, rw = mixed.solve_cg(xm, 0, 1e-6)
    assert rw <= 1e-6 and itw <= itm // 2
    # only FI_F64 contexts take the option
    f32 = fi.LatticeField(sizes, dtype="f32")
    with pytest.raises(fi.FiError):
        f32.set_mixed_precision(True)


def test_mixed_precision_reassembly_reuses_the_replica(fi):
    """clear_points + new points + assemble on a mixed-precision field: the fp32 replica and its levels are rebuilt
    in place; the second solve equals the solve of a fresh field with the second point set."""
    sizes = [72, 56]
    rng = np.random.default_rng(9)
    w = fi.Weights()
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w)
    f.set_levels(2)
    f.set_multigrid(True)
    f.set_mixed_precision(True)
    sols = []
    for k in range(3):
        pos, nrm = sphere_points(rng, sizes, 500 + 100 * k)
        f.clear_points()
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-9)
        assert rel <= 1e-9 and f.true_residual() <= 1.01e-9
        sols.append((pos, nrm, f.solution_f64().copy()))
    pos, nrm, x_last = sols[-1]
    fresh = fi.LatticeField(sizes, dtype="f64")
    fresh.add_field_constraints(w)
    fresh.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    fresh.assemble()
    fresh.solve_cg(None, 20000, 1e-9)
    assert rel_inf(x_last, fresh.solution_f64()) <= 1e-5
    # switching the option off drops the replica and the field solves in plain fp64 again
    f.set_mixed_precision(False)
    f.assemble()
    x, it, rel = f.solve_cg(None, 0, 1e-9)
    assert rel <= 1e-9 and rel_inf(f.solution_f64(), x_last) <= 1e-5


@pytest.mark.parametrize("sizes,kw", [([48, 40, 44], None), ([40, 44, 36], {"model_1": 0.3}), ([36, 40, 41], {"model_1": 0.4, "model_2": 0.0})])
def test_fused_smoother_equals_the_unfused_one(fi, sizes, kw, monkeypatch):
    """fp32 V-cycles smooth through the marching kernel's epilogue (one launch per Chebyshev step, three-term form in
    the iterates); FI_NO_FUSED_SMOOTHER keeps apply + vector kernel (the form of the fp64 and 2-D paths).  Same
    polynomial: same iteration count (rounding may move it by one) and the same solution."""
    rng = np.random.default_rng(12)
    pos, nrm = sphere_points(rng, sizes, 2500)
    w = fi.Weights(**(kw or {}))
    out = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("FI_NO_FUSED_SMOOTHER", "1")
        else:
            monkeypatch.delenv("FI_NO_FUSED_SMOOTHER", raising=False)
        f = fi.LatticeField(sizes, dtype="f32")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.set_levels(2)
        f.set_multigrid(True)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-5)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.2e-5
        out.append((x.copy(), it))
    monkeypatch.delenv("FI_NO_FUSED_SMOOTHER", raising=False)
    assert abs(out[0][1] - out[1][1]) <= 1
    assert rel_inf(out[0][0], out[1][0]) <= 2e-4      # two fp32 solves to a 1e-5 residual


@pytest.mark.parametrize("sizes,levels,oriented,mixed", [([256, 256], 5, False, True), ([200, 136], 4, True, True), ([64, 64, 64], 3, False, True),
                                                        ([72, 60, 56], 3, True, True), ([128, 128], 4, True, False), ([40, 36, 44], 2, False, False)])
def test_small_levels_in_one_workgroup_equal_the_tiled_kernels(fi, monkeypatch, sizes, levels, oriented, mixed):
    """The small-level engine (fi_tail.hip): every level of <= 4096 unknowns at the bottom of a hierarchy runs its share of a
    V-cycle in ONE launch of one workgroup, vectors in LDS -- the same smoothers, constants and transfers as the tiled
    kernels (FI_NO_TAIL), so V-cycle PCG takes the same iterations (rounding may move the count by one) to the same
    solution.  2-D and 3-D, value rows (polynomial smoother on 3-D levels) and oriented points (Chebyshev in the full
    operator), fp64 CG around the fp32 cycle and plain fp32."""
    from field_interpolation_amd import _capi
    rng = np.random.default_rng(sum(sizes))
    D = len(sizes)
    n = 400 if D == 2 else 4000
    if oriented:
        pos, nrm = sphere_points(rng, sizes, n)
        val = None
    else:
        pos = np.stack([rng.uniform(0.0, s - 1.0, n) for s in sizes], 1).astype(np.float32)
        nrm = None
        val = rng.normal(size=n).astype(np.float32)
    w = fi.Weights() if oriented else fi.Weights(model_2=0.5)
    out = []
    for no_tail in (False, True):
        if no_tail:
            monkeypatch.setenv("FI_NO_TAIL", "1")
        else:
            monkeypatch.delenv("FI_NO_TAIL", raising=False)
        f = fi.LatticeField(sizes, dtype="f64" if mixed else "f32")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient if oriented else 0.0, w.gradient_kernel, pos, nrm, None, values=val)
        f.set_levels(levels, 1e-4)
        f.set_multigrid(True)
        if mixed:
            f.set_mixed_precision(True)
        f.assemble()
        tol = 1e-8 if mixed else 1e-5
        x, it, rel = f.solve_cg(None, 0, tol)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.2 * tol
        out.append((f.solution_f64() if mixed else x.astype(np.float64), it, f.stats()["coarse_iterations"]))
        del f
    monkeypatch.delenv("FI_NO_TAIL", raising=False)
    assert abs(out[0][1] - out[1][1]) <= 1, (out[0][1:], out[1][1:])
    assert rel_inf(out[0][0], out[1][0]) <= (1e-5 if mixed else 3e-4)


@pytest.mark.parametrize("sizes,kw", [([64, 64, 64], dict(model_2=0.5)), ([72, 60, 56], dict(model_1=0.4, model_2=0.5)),
                                      ([48, 52, 44], dict(model_1=0.6, model_2=0.0)), ([80, 40, 36], dict(model_0=0.1, model_1=0.2, model_2=0.7))])
def test_small_levels_polynomial_steps_without_the_march(fi, monkeypatch, sizes, kw):
    """On fp32 levels of <= 2^19 points the steps of the V-cycle's polynomial smoother run as k_cheb_direct3 -- a thread per
    point, one round of neighbour loads -- instead of the z-marching kernel (FI_NO_DIRECT_STEP): the same step (operand
    formed on load or stored, every combination of model_0 / model_1 / model_2), so V-cycle PCG takes the same iterations
    (rounding may move the count by one) to the same solution.  Value rows: the levels smooth with the polynomial."""
    rng = np.random.default_rng(sum(sizes))
    n = 5000
    pos = np.stack([rng.uniform(0.0, s - 1.0, n) for s in sizes], 1).astype(np.float32)
    val = rng.normal(size=n).astype(np.float32)
    w = fi.Weights(**kw)
    out = []
    for no_direct in (False, True):
        if no_direct:
            monkeypatch.setenv("FI_NO_DIRECT_STEP", "1")
        else:
            monkeypatch.delenv("FI_NO_DIRECT_STEP", raising=False)
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.set_levels(2, 1e-4)
        f.set_multigrid(True)
        f.set_mixed_precision(True)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-8)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.2e-8
        out.append((f.solution_f64(), it))
        del f
    monkeypatch.delenv("FI_NO_DIRECT_STEP", raising=False)
    assert abs(out[0][1] - out[1][1]) <= 1, (out[0][1], out[1][1])
    assert rel_inf(out[0][0], out[1][0]) <= 1e-5


@pytest.mark.parametrize("sizes,kw,dtype,mixed", [([160, 144, 136], dict(model_2=0.5), "f64", True),      # bfloat16 iterates on the finest level
                                                  ([72, 60, 56], dict(model_1=0.4, model_2=0.5), "f64", True),   # small levels: k_cheb_direct3
                                                  ([96, 80, 72], dict(model_0=0.1, model_2=0.7), "f32", False),
                                                  ([64, 56, 48], dict(model_1=0.5, model_2=0.0), "f64", False)])  # the V-cycle in fp64
def test_post_smoothing_added_by_the_polynomials_last_step(fi, monkeypatch, sizes, kw, dtype, mixed):
    """x += M (b - A x): the last step of the post-smoothing's polynomial adds its result onto x itself (ChebEpi::acc) instead
    of leaving it to a sum of its own (FI_NO_STEP_ONTO) -- the same additions in the same order: bit-equal solves.  (Both legs
    with the CG's r . z from its own pass, FI_NO_TWIN_DOT: the partials that last step can deliver round r to fp32 first.)"""
    monkeypatch.setenv("FI_NO_TWIN_DOT", "1")
    rng = np.random.default_rng(sum(sizes) + 5)
    n = 20000
    pos = np.stack([rng.uniform(0.0, s - 1.0, n) for s in sizes], 1).astype(np.float32)
    val = rng.normal(size=n).astype(np.float32)
    w = fi.Weights(**kw)
    out = []
    for separate in (False, True):
        if separate:
            monkeypatch.setenv("FI_NO_STEP_ONTO", "1")
        else:
            monkeypatch.delenv("FI_NO_STEP_ONTO", raising=False)
        f = fi.LatticeField(sizes, dtype=dtype)
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.set_levels(2, 1e-4)
        f.set_multigrid(True)
        if mixed:
            f.set_mixed_precision(True)
        f.assemble()
        tol = 1e-8 if dtype == "f64" else 1e-5
        x, it, rel = f.solve_cg(None, 0, tol)
        assert f.stats()["converged"] == 1
        out.append((np.array(f.solution_f64() if dtype == "f64" else x), it))
        del f
    monkeypatch.delenv("FI_NO_STEP_ONTO", raising=False)
    assert out[0][1] == out[1][1], (out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0])


@pytest.mark.parametrize("sizes,kw,oriented,mixed", [([128, 128, 128], dict(model_2=0.5), False, True),          # config 4's shape: levels 64^3, 32^3
                                                     ([96, 80, 72], dict(model_1=0.4, model_2=0.5), False, True),
                                                     ([88, 72, 80], dict(model_1=0.6, model_2=0.0), False, False),   # fp32 V-cycle PCG
                                                     ([96, 96, 96], dict(), True, True)])                            # an SDF: the full-operator smoother
def test_small_levels_full_operator_as_diagonals(fi, monkeypatch, sizes, kw, oriented, mixed):
    """fp32 levels of <= 2^19 points apply their FULL operator (V-cycle residuals, the full-operator smoother's steps) as the
    model star + 27 diagonals of the data rows, a thread per point (k_full_direct3), instead of walking the z-marching kernel
    through single planes of packed cells (FI_NO_DIRECT_FULL): the same operator, another order of sums -- the same iteration
    counts (+-1) and the same solution."""
    rng = np.random.default_rng(sum(sizes) + 3)
    w = fi.Weights(**kw)
    if oriented:
        n = 6000
        centre = np.array(sizes, np.float32) / 2.0
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pos = (centre + 0.3 * min(sizes) * d).astype(np.float32)
        nrm, val = d, None
    else:
        n = 40000
        pos = np.stack([rng.uniform(0.0, s - 1.0, n) for s in sizes], 1).astype(np.float32)
        nrm, val = None, rng.normal(size=n).astype(np.float32)
    out = []
    for no_direct in (False, True):
        if no_direct:
            monkeypatch.setenv("FI_NO_DIRECT_FULL", "1")
        else:
            monkeypatch.delenv("FI_NO_DIRECT_FULL", raising=False)
        f = fi.LatticeField(sizes, dtype="f64" if mixed else "f32")
        f.add_field_constraints(w)
        if oriented:
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        else:
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.set_levels(2, 1e-4)
        f.set_multigrid(True)
        if mixed:
            f.set_mixed_precision(True)
        f.assemble()
        tol = 1e-8 if mixed else 2e-5
        x, it, rel = f.solve_cg(None, 0, tol)
        assert f.stats()["converged"] == 1
        out.append((np.array(f.solution_f64() if mixed else x, dtype=np.float64), it))
        del f
    monkeypatch.delenv("FI_NO_DIRECT_FULL", raising=False)
    assert abs(out[0][1] - out[1][1]) <= 1, (out[0][1], out[1][1])
    assert rel_inf(out[0][0], out[1][0]) <= (1e-5 if mixed else 2e-3)


@pytest.mark.parametrize("sizes", [[256, 128, 128], [128, 96, 64]])
def test_row_form_transfers_equal_the_block_form(fi, monkeypatch, sizes):
    """On lattices halved cell-centred (even extents) the V-cycle's restriction (x / y pass) and interpolation and the start's
    cubic interpolation run in a ROW form -- a thread per two coarse columns walking rows, whole 16-byte accesses, no LDS
    (k_restrict3_xy_rows, k_prolong3_rows, k_prolong3_cubic_rows; the interpolations from 128 x 64 x 64 coarse points up) --
    with the weights of restrict_taps / prolong_taps / cubic_taps in another order of sums than the tiled and per-block
    kernels (FI_TILED_RESTRICT, FI_BLOCK_PROLONG): the same iteration counts and the same solution."""
    rng = np.random.default_rng(sum(sizes) + 9)
    n = 60000
    pos = np.stack([rng.uniform(0.0, s - 1.0, n) for s in sizes], 1).astype(np.float32)
    val = rng.normal(size=n).astype(np.float32)
    w = fi.Weights(model_2=0.5)
    out = []
    for block in (False, True):
        for k in ("FI_TILED_RESTRICT", "FI_BLOCK_PROLONG"):
            if block:
                monkeypatch.setenv(k, "1")
            else:
                monkeypatch.delenv(k, raising=False)
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.set_levels(2, 1e-4)
        f.set_multigrid(True)
        f.set_mixed_precision(True)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-8)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.2e-8
        out.append((np.array(f.solution_f64()), it))
        del f
    for k in ("FI_TILED_RESTRICT", "FI_BLOCK_PROLONG"):
        monkeypatch.delenv(k, raising=False)
    assert abs(out[0][1] - out[1][1]) <= 1, (out[0][1], out[1][1])
    assert rel_inf(out[0][0], out[1][0]) <= 1e-5


@pytest.mark.parametrize("sizes,kw", [([160, 144, 136], dict(model_2=0.5)), ([136, 128, 132], dict(model_1=0.3, model_2=0.6))])
def test_r_dot_z_from_the_cycles_last_launch(fi, monkeypatch, sizes, kw):
    """Mixed precision on one undivided context: the fp64 CG's r . z is t^2 (b . x) of the fp32 replica's cycle, and the cycle's
    last launch -- the post-smoothing polynomial's last step -- sums b . x on the way (ChebEpi::dotv) instead of a pass of its
    own over r and z (k_dot_mixed, FI_NO_TWIN_DOT).  r enters rounded to fp32 there: the same iteration counts, the same
    solution to the solve's tolerance, the same verified residual."""
    rng = np.random.default_rng(sum(sizes) + 11)
    n = 30000
    pos = np.stack([rng.uniform(0.0, s - 1.0, n) for s in sizes], 1).astype(np.float32)
    val = rng.normal(size=n).astype(np.float32)
    w = fi.Weights(**kw)
    out = []
    for own_pass in (False, True):
        if own_pass:
            monkeypatch.setenv("FI_NO_TWIN_DOT", "1")
        else:
            monkeypatch.delenv("FI_NO_TWIN_DOT", raising=False)
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.set_levels(2, 1e-4)
        f.set_multigrid(True)
        f.set_mixed_precision(True)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-9)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.2e-9
        out.append((np.array(f.solution_f64()), it))
        del f
    monkeypatch.delenv("FI_NO_TWIN_DOT", raising=False)
    assert abs(out[0][1] - out[1][1]) <= 1, (out[0][1], out[1][1])
    assert rel_inf(out[0][0], out[1][0]) <= 1e-6


def test_levels_on_diagonals_serve_every_solver_of_the_context(fi):
    """Levels of a V-cycle hierarchy whose full operator runs on diagonals keep no cell lists for the marching kernel
    (MarchState::no_lists).  The solver may change after the assembly -- V-cycle PCG off: coarse-to-fine start + Jacobi-PCG on the
    same levels -- and the levels must serve it all the same (their operator stays the diagonals')."""
    sizes = [96, 80, 72]
    rng = np.random.default_rng(77)
    n = 30000
    pos = np.stack([rng.uniform(0.0, s - 1.0, n) for s in sizes], 1).astype(np.float32)
    val = rng.normal(size=n).astype(np.float32)
    w = fi.Weights(model_2=0.5)
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.set_levels(2, 1e-4)
    f.set_multigrid(True)
    f.assemble()
    x1, it1, rel1 = f.solve_cg(None, 0, 1e-8)
    assert f.stats()["converged"] == 1
    a = np.array(f.solution_f64())
    f.set_multigrid(False)                      # no re-assembly: the same levels, now under the start + Jacobi-PCG
    x2, it2, rel2 = f.solve_cg(None, 0, 1e-8)
    assert f.stats()["converged"] == 1 and f.stats()["coarse_iterations"] > 0 and it2 > it1
    assert rel_inf(a, np.array(f.solution_f64())) <= 1e-5


def test_levels_built_beside_the_finest_level_are_the_same_levels(fi, monkeypatch):
    """fi_assemble builds the coarser levels on a helper thread and a second stream while the calling thread assembles
    the finest level; FI_SERIAL_LEVELS builds them afterwards on the solver stream.  Same kernels on the same data: the
    solves are bitwise equal, assemble after assemble (the levels' buffers are reused)."""
    sizes = [48, 44, 52]
    rng = np.random.default_rng(21)
    w = fi.Weights(model_2=0.5, data_gradient=0.0)
    sols = {}
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("FI_SERIAL_LEVELS", "1")
        else:
            monkeypatch.delenv("FI_SERIAL_LEVELS", raising=False)
        f = fi.LatticeField(sizes, dtype="f32")
        f.add_field_constraints(w)
        f.set_levels(2)
        f.set_polynomial(4)
        rng = np.random.default_rng(21)
        out = []
        for k in range(3):
            pos = np.stack([rng.uniform(0, s - 1, 4000 + 500 * k) for s in sizes], axis=1).astype(np.float32)
            val = rng.normal(size=len(pos)).astype(np.float32)
            f.clear_points()
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
            f.assemble()
            assert f.stats()["num_levels"] == 3
            x, it, rel = f.solve_cg(None, 0, 1e-5)
            out.append((x.copy(), it))
        sols[serial] = out
    monkeypatch.delenv("FI_SERIAL_LEVELS", raising=False)
    for a, b in zip(sols[False], sols[True]):
        assert a[1] == b[1] and np.array_equal(a[0], b[0])


def test_cubic_start_changes_the_path_not_the_solution(oracle, fi, monkeypatch):
    """The coarse-to-fine start interpolates cubically on undivided 3-D lattices (FI_LINEAR_START: trilinear, the form
    slabs and the V-cycle's P use).  A start is a start: both solves end at the oracle's solution."""
    sizes = [24, 20, 28]          # the oracle's sparse Cholesky of a larger 3-D lattice takes minutes
    (fo, _), (w, pos, nrm) = _problem(oracle, fi, sizes, "f64", n=600)
    x_ref = fo.solve_exact_f64()
    its = []
    for linear in (False, True):
        if linear:
            monkeypatch.setenv("FI_LINEAR_START", "1")
        else:
            monkeypatch.delenv("FI_LINEAR_START", raising=False)
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.set_levels(1)
        f.assemble()
        assert f.stats()["num_levels"] == 2
        x, it, rel = f.solve_cg(None, 0, 1e-10)
        assert rel <= 1e-10 and rel_inf(f.solution_f64(), x_ref) <= 1e-6
        its.append(it)
    monkeypatch.delenv("FI_LINEAR_START", raising=False)
    assert its[0] <= its[1] + 2


@pytest.mark.parametrize("sizes", [[24, 20, 28], [25, 21, 29], [26, 21, 24], [17, 16, 15]])
@pytest.mark.parametrize("levels", [1, 2])
def test_start_guess_reproduces_a_linear_field(fi, monkeypatch, sizes, levels):
    """The interpolation of the coarse-to-fine start on its own (FI_START_ONLY: no iteration on the finest level).  Data
    sampled from a linear function under model_2 alone: every level's solution is that function at the level's own
    points, and both interpolations -- trilinear and cubic, vertex- and cell-centred axes, the extrapolating end points --
    reproduce a linear function exactly.  A wrong index or weight anywhere shows as an O(1) error."""
    if min(sizes) // (2 ** levels) < 3:
        levels = 1
    rng = np.random.default_rng(5)
    n = 4000
    # (points two cells inside: a coarser cell-centred lattice ends half a fine cell inside the fine one, and a row whose
    # cell sticks out of ITS lattice loses corners -- field_interpolation.cpp:19-44 -- and with them the linear function)
    pos = np.stack([rng.uniform(2, s - 3, n) for s in sizes], axis=1).astype(np.float32)
    coef = np.array([0.3, -0.2, 0.15])
    val = (pos.astype(np.float64) @ coef + 1.5).astype(np.float32)
    w = fi.Weights(model_2=0.5, data_gradient=0.0)
    grid = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in sizes[::-1]], indexing="ij")   # z, y, x
    exact = (coef[0] * grid[2] + coef[1] * grid[1] + coef[2] * grid[0] + 1.5).ravel()
    monkeypatch.setenv("FI_START_ONLY", "1")
    try:
        for linear in (False, True):
            if linear:
                monkeypatch.setenv("FI_LINEAR_START", "1")
            else:
                monkeypatch.delenv("FI_LINEAR_START", raising=False)
            f = fi.LatticeField(sizes, dtype="f64")
            f.add_field_constraints(w)
            f.set_levels(levels, 1e-11)
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
            f.assemble()
            assert f.stats()["num_levels"] >= 2
            x, it, rel = f.solve_cg(None, 0, 1e-10)
            assert it == 0
            err = np.abs(f.solution_f64() - exact).max() / np.abs(exact).max()
            assert err <= 2e-5, (sizes, levels, linear, err)     # (fp32 positions and values: 1e-7 relative each)
    finally:
        monkeypatch.delenv("FI_START_ONLY", raising=False)
        monkeypatch.delenv("FI_LINEAR_START", raising=False)


@pytest.mark.parametrize("mixed", [False, True])
def test_level_chains_in_parallel_same_bits(fi, monkeypatch, mixed):
    """The coarser levels beyond the first are assembled on a thread and a stream each (build_levels on a helper's
    stream); FI_SERIAL_LEVEL_CHAINS builds them one after the other.  Same kernels on the same data: the solves are
    bitwise equal, assemble after assemble (the levels' buffers and streams are reused), in fp32 and in the
    mixed-precision form whose replica carries the levels."""
    sizes = [96, 80, 88]
    w = fi.Weights(model_2=0.5, data_gradient=0.0)
    sols = {}
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("FI_SERIAL_LEVEL_CHAINS", "1")
        else:
            monkeypatch.delenv("FI_SERIAL_LEVEL_CHAINS", raising=False)
        f = fi.LatticeField(sizes, dtype="f64" if mixed else "f32")
        f.add_field_constraints(w)
        f.set_levels(3, 1e-6 if mixed else 1e-5)
        f.set_multigrid(True)
        if mixed:
            f.set_mixed_precision(True)
        rng = np.random.default_rng(31)
        out = []
        for k in range(3):
            pos = np.stack([rng.uniform(0, s - 1, 6000 + 700 * k) for s in sizes], axis=1).astype(np.float32)
            val = rng.normal(size=len(pos)).astype(np.float32)
            f.clear_points()
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
            f.assemble()
            assert f.stats()["num_levels"] == 4
            x, it, rel = f.solve_cg(None, 0, 1e-7 if mixed else 1e-5)
            out.append((x.copy(), it))
        sols[serial] = out
    monkeypatch.delenv("FI_SERIAL_LEVEL_CHAINS", raising=False)
    for a, b in zip(sols[False], sols[True]):
        assert a[1] == b[1] and np.array_equal(a[0], b[0])


@pytest.mark.parametrize("sizes,n", [([40, 36, 44], 3000), ([33, 30, 35], 20000)])
def test_lumped_replica_of_value_row_data(oracle, fi, monkeypatch, sizes, n):
    """Mixed precision on value rows: the fp32 replica's finest level runs on the LUMPED operator A_model + diag(row sums of
    the data term) -- no rows, cells or sort for it (fi_solver.hip, twin_assemble_lumped) -- while CG itself iterates on the
    exact fp64 operator.  Same solution as the oracle's, about the iterations of the replica that assembles its own cells
    (FI_NO_LUMPED_TWIN), sparse data and dense (several rows per cell); a re-assemble with other points follows."""
    rng = np.random.default_rng(31)
    w = fi.Weights(model_2=0.5)
    pos = np.stack([rng.uniform(-0.5, s - 0.5, n) for s in sizes], axis=1).astype(np.float32)
    val = (np.sin(pos[:, 0] * 0.3) + 0.1 * rng.normal(size=n)).astype(np.float32)
    fo = oracle.LatticeField(sizes)
    fo.add_field_constraints(oracle.Weights(model_2=0.5))
    fo.add_value_constraints(pos, val, w.data_pos)
    x_ref = fo.solve_exact_f64() if int(np.prod(sizes)) <= 40000 else None
    its = {}
    for lumped in (True, False):
        if lumped:
            monkeypatch.delenv("FI_NO_LUMPED_TWIN", raising=False)
        else:
            monkeypatch.setenv("FI_NO_LUMPED_TWIN", "1")
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.set_levels(2, 1e-5)
        f.set_multigrid(True)
        f.set_mixed_precision(True)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-9)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.01e-9
        its[lumped] = it
        sol = f.solution_f64()
        if x_ref is not None:
            assert rel_inf(sol, x_ref) <= 1e-6
        else:
            AtA, atb, _ = fo.normal_equations()
            assert np.linalg.norm(atb - AtA @ sol) <= 2e-9 * np.linalg.norm(atb)
        if lumped:      # other points on the same context: the replica follows
            pos2 = pos[: n // 2] + np.float32(0.25)
            f.clear_points()
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos2, None, None, values=val[: n // 2])
            f.assemble()
            x2, it2, rel2 = f.solve_cg(None, 0, 1e-9)
            assert f.stats()["converged"] == 1 and f.true_residual() <= 1.01e-9 and it2 <= 2 * it + 4
    monkeypatch.delenv("FI_NO_LUMPED_TWIN", raising=False)
    assert its[True] <= its[False] + max(3, its[False] // 3), its


def test_bfloat16_iterates_of_the_smoother(fi, monkeypatch):
    """On undivided fp32 levels of 2^21 points and more the polynomial smoother keeps its iterates between the first and the
    last step as bfloat16 (fi_multigrid.hip poly_chain; a step moves 8-14 bytes per point instead of 10-18).  They are operands
    of a preconditioner: the fp64 CG around it must reach the same true residual in the same number of iterations (+-1) as with
    fp32 iterates (FI_NO_Z16), and the two fields must agree to the stop tolerance's accuracy.  128^3 = 2^21 points, value rows
    (the lumped replica), 4 and 5 terms."""
    from field_interpolation_amd import synth
    sizes, w, pos, val = synth.config4(side=128, num_points=125000, seed=5)
    for terms in (5, 4):
        out = {}
        for z16 in (True, False):
            if z16:
                monkeypatch.delenv("FI_NO_Z16", raising=False)
            else:
                monkeypatch.setenv("FI_NO_Z16", "1")
            f = fi.LatticeField(sizes, dtype="f64")
            f.add_field_constraints(w)
            f.set_levels(2, 3e-4)
            f.set_multigrid(True)
            f.set_mixed_precision(True)
            f.set_mg_smoother(True, 4.0, terms, 30.0)
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
            f.assemble()
            x, it, rel = f.solve_cg(None, 0, 1e-8)
            assert f.stats()["converged"] == 1 and f.true_residual() <= 1.01e-8
            out[z16] = (it, f.solution_f64().copy())
        monkeypatch.delenv("FI_NO_Z16", raising=False)
        assert abs(out[True][0] - out[False][0]) <= 1, (terms, out[True][0], out[False][0])
        scale = np.abs(out[False][1]).max()
        assert np.abs(out[True][1] - out[False][1]).max() <= 2e-6 * scale


def test_solves_without_looks_at_the_stop_flag(fi, monkeypatch):
    """From the second solve of a context on, V-cycle PCG looks at its stop flag only from the previous solve's iteration count
    on, and the Jacobi-PCG levels of the coarse-to-fine start get that many iterations and no look at all (the flag's copy is
    read at the level's next solve).  The device runs the same kernels either way: three solves in a row must give the
    iteration counts and the very bits of FI_LOOK_ALWAYS -- also after the data have changed under the predictions (other
    points, then fewer points: the levels need other iteration counts than predicted and must fall back to watching)."""
    from field_interpolation_amd import synth
    sizes, w, pos, val = synth.config4(side=64, num_points=20000, seed=8)
    rng = np.random.default_rng(8)
    pos2 = (pos[:5000] * np.float32(0.5) + np.float32(7.0)).astype(np.float32)
    val2 = rng.normal(size=len(pos2)).astype(np.float32)
    runs = {}
    for always in (False, True):
        if always:
            monkeypatch.setenv("FI_LOOK_ALWAYS", "1")
        else:
            monkeypatch.delenv("FI_LOOK_ALWAYS", raising=False)
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.set_levels(2, 3e-4)
        f.set_multigrid(True)
        f.set_mixed_precision(True)
        out = []
        for p_, v_ in ((pos, val), (pos, val), (pos, val), (pos2, val2), (pos2, val2), (pos, val)):
            f.clear_points()
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, p_, None, None, values=v_)
            f.assemble()
            x, it, rel = f.solve_cg(None, 0, 1e-8)
            assert f.stats()["converged"] == 1 and f.true_residual() <= 1.01e-8
            out.append((it, f.stats()["coarse_iterations"], f.solution_f64().copy()))
        runs[always] = out
    monkeypatch.delenv("FI_LOOK_ALWAYS", raising=False)
    # the watched and the unwatched runs agree bit for bit wherever the levels took the predicted counts (solves 2, 3 and 5)
    for k in (0, 1, 2, 4):
        assert runs[False][k][0] == runs[True][k][0]
        np.testing.assert_array_equal(runs[False][k][2], runs[True][k][2])
    # Right behind a change of the data a level gets the iterations its PREVIOUS data needed (solve 4: too few, a poorer
    # start guess; solve 6: more than needed, harmless) -- the fine solve still meets its tolerance, within a quarter of the
    # watched run's iterations, and the level watches its flag again from the next solve on (solve 5: the same bits).
    for k in (3, 5):
        assert abs(runs[False][k][0] - runs[True][k][0]) <= max(2, runs[True][k][0] // 4), (k, runs[False][k][0], runs[True][k][0])
        scale = np.abs(runs[True][k][2]).max()
        assert np.abs(runs[False][k][2] - runs[True][k][2]).max() <= 3e-6 * scale   # (two solves to a residual of 1e-8 each: ~1e-6 apiece)



@pytest.mark.parametrize("sizes,levels,kc", [([72, 64, 80], 3, 2), ([320, 272], 4, 2), ([96, 80, 64], 2, 1)])
def test_kcycle_preconditioner(fi, monkeypatch, sizes, levels, kc):
    """FI_OPT_MG_KCYCLE on oriented-point data: the first `kc` coarse levels corrected by two flexible-CG steps each, the
    outer CG with the flexible beta.  Converges to the V-cycle's solution in fewer iterations; FI_NO_KCYCLE takes the option
    back to the V-cycle's very iteration count; a loop-back group (the K-cycle needs an undivided lattice) runs the V-cycle."""
    rng = np.random.default_rng(3 * sum(sizes))
    pos, nrm = sphere_points(rng, sizes, 1500, noise=0.3)
    w = fi.Weights()

    def build(kcycle, group=False):
        f = fi.LatticeGroup(sizes, 2, dtype="f64") if group else fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.set_levels(levels, 1e-2)
        f.set_multigrid(True)
        f.set_mixed_precision(True)
        if kcycle and not group:
            f.set_kcycle(kcycle)
        f.assemble()
        return f

    tol = 1e-9
    v = build(0)
    xv, itv, relv = v.solve_cg(None, 0, tol)
    k = build(kc)
    xk, itk, relk = k.solve_cg(None, 0, tol)
    assert k.stats()["converged"] == 1 and k.true_residual() <= 1.5 * tol
    assert itk < itv, (itk, itv)
    assert rel_inf(k.solution_f64(), v.solution_f64()) <= 2e-6
    monkeypatch.setenv("FI_NO_KCYCLE", "1")
    xo, ito, relo = k.solve_cg(None, 0, tol)
    monkeypatch.delenv("FI_NO_KCYCLE")
    assert ito == itv
    if len(sizes) == 3:
        g = build(kc, group=True)
        xg, itg, relg = g.solve_cg(None, 0, tol)
        assert abs(itg - itv) <= max(2, itv // 10)
