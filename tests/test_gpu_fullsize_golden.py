"""The BENCHMARK's own sizes against the ORACLE (SURVEY.md 8(d) P4 "at benchmark sizes").

tests/golden/config4_256_oracle_f64.npz: config 4 at 256^3 (bench.py's default workload, synth.config4 seed 3) solved by
the oracle -- the reference's rows, explicit AtA in fp64 (sparse_linear.cpp:105-113), fp64 Jacobi-PCG to a true residual
of 9.8e-11 (474 iterations, re-checked through A^T(A x) from the rows) -- in the build container by
tests/golden/make_golden_fullsize.py; every 8th point per axis is stored (32^3 values) with the whole field's sum, sum of
squares and max |x|.  tests/golden/config5_*_oracle_f64.npz: config 5's shape at the largest side the oracle's PCG
finishes.  The oracle is not called here: committed vectors only.

North-star tolerance: field values within 1e-5 relative (max-norm, relative to max |x|) of the CPU reference.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELD_TOL = 1e-5          # BASELINE.json north_star: "field values within 1e-5 relative of the CPU reference"


@pytest.fixture(scope="module")
def fi():
    import field_interpolation_amd as fi
    from field_interpolation_amd import _capi
    assert _capi.device_count() >= 1
    return fi


def _against_sample(x, g):
    sizes = [int(s) for s in g["sizes"]]
    stride = int(g["stride"])
    grid = np.asarray(x, np.float64).reshape(sizes[::-1])
    got = grid[tuple(slice(0, None, stride) for _ in sizes)]
    return float(np.abs(got - g["sample"]).max() / float(g["field_maxabs"]))


@pytest.mark.parametrize("seed", [3, 11, 12])
def test_config4_at_256_cubed_against_the_oracle(fi, capsys, seed):
    """bench.py's headline solver with bench.py's EXACT settings (field_interpolation_amd/bench_settings.py: levels, the
    start's tolerance, the stop rule -- by the field, FI_OPT_FIELD_TOLERANCE, no workload-specific constant since round 6) on
    three seeds of the metric's workload, each against the oracle's fp64 solution of that seed: the field stays within the
    north-star's 1e-5 on every one, and the solver's own estimate is no smaller than a fifth of the error the oracle shows."""
    from field_interpolation_amd import bench_settings as bs
    from field_interpolation_amd import synth
    name = "config4_256_oracle_f64.npz" if seed == 3 else "config4_256_seed%d_oracle_f64.npz" % seed
    g = np.load(os.path.join(GOLDEN, name))
    assert float(g["true_rel_residual"]) <= 2e-10 and int(g["seed"]) == seed
    sizes, w, pos, val = synth.config4(seed=seed)
    assert sizes == [int(s) for s in g["sizes"]] and len(pos) == int(g["num_points"])
    assert seed in bs.CONFIG4_SEEDS

    a = bs.headline_field(fi, 4, sizes, w, by_field=True)
    a.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    a.assemble()
    assert bs.SETTINGS[4]["levels"] == 3
    xa, ita, rela = a.solve_cg(None, 0, bs.SETTINGS[4]["tol"])
    st = a.stats()
    tol = st["stop_residual"]
    assert st["converged"] == 1 and st["field_rounds"] == 1 and 0 <= st["field_estimate"] <= FIELD_TOL and 4 <= ita <= 8
    x64 = a.solution_f64()
    err = _against_sample(x64, g)
    assert st["field_estimate"] >= 0.2 * err
    # whole-field checksums of the oracle's solution (what the strided sample cannot see)
    sum_err = abs(x64.sum() - float(g["field_sum"])) / (float(g["field_maxabs"]) * x64.size)
    sq_err = abs((x64 * x64).sum() / float(g["field_sumsq"]) - 1.0)
    with capsys.disabled():
        print("\n[P4 config 4 at 256^3, seed %d, against the oracle] bench settings (fp64 CG + fp32 V-cycle, %d levels to %g, stopped by the "
              "field at residual %.2e, estimate %.2e): %d iterations, field error %.2e (sample of %d values), mean error %.1e, energy error %.1e"
              % (seed, bs.SETTINGS[4]["levels"], bs.SETTINGS[4]["coarse_tol"], tol, st["field_estimate"], ita, err, g["sample"].size, sum_err, sq_err))
    assert err <= FIELD_TOL
    assert sum_err <= FIELD_TOL and sq_err <= 10 * FIELD_TOL
    if seed != 3:
        return
    # ... and driven to the oracle's own tolerance: what fp64 delivers
    a.set_field_tolerance(0.0)
    xb, itb, relb = a.solve_cg(None, 0, 1e-10)
    assert a.true_residual() <= 1.01e-10
    err10 = _against_sample(a.solution_f64(), g)
    with capsys.disabled():
        print("[P4 config 4 at 256^3 against the oracle] to 1e-10: %d iterations, field error %.2e" % (itb, err10))
    assert err10 <= 1e-7
    del a

    # the fast fp32 mode (residual 1e-5): its error is what kappa allows -- reported, bounded, NOT within the tolerance
    f = fi.LatticeField(sizes, dtype="f32")
    f.add_field_constraints(w)
    f.set_levels(1, 1e-5)
    f.set_polynomial(4, 30.0)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.assemble()
    x, it, rel = f.solve_cg(None, 0, 1e-5)
    err32 = _against_sample(x, g)
    with capsys.disabled():
        print("[P4 config 4 at 256^3 against the oracle] fp32 polynomial PCG to 1e-5: field error %.2e" % err32)
    assert err32 <= 5e-3


def test_config5_shape_against_the_oracle(fi, capsys):
    from field_interpolation_amd import synth
    paths = sorted(glob.glob(os.path.join(GOLDEN, "config5_*_oracle_f64.npz")))
    assert paths, "tests/golden/config5_*_oracle_f64.npz is missing"
    for path in paths:
        g = np.load(path)
        sizes = [int(s) for s in g["sizes"]]
        side = sizes[0]
        sz, w, pos, nrm = synth.config5(side=side, num_points=int(g["num_points"]), seed=int(g["seed"]))
        assert sz == sizes
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.set_levels(4, 1e-4)
        f.set_multigrid(True)
        f.set_mixed_precision(True)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-10)
        assert x is not None and f.true_residual() <= 1.01e-10
        err = _against_sample(f.solution_f64(), g)
        with capsys.disabled():
            print("\n[P4 config-5 shape at %d^3 against the oracle (%d PCG iterations there)] mixed V-cycle PCG %d iterations, "
                  "field error %.2e" % (side, int(g["iterations"]), it, err))
        assert err <= FIELD_TOL


def test_config2_full_size_against_the_oracles_exact_solve(fi, capsys):
    """BASELINE config 2 at its FULL size (1024^2, 10 k noisy value constraints, model_2 = 10) against the reference's exact
    route -- the oracle's explicit fp64 AtA, banded Cholesky, one refinement step through the rows
    (tests/golden/make_golden_2d.py) -- with bench.py --config 2's settings.  At the configuration's residual (1e-5) the field
    is what kappa allows (reported); driven to the residual that buys it the field is within 1e-5."""
    from field_interpolation_amd import bench_settings as bs
    from field_interpolation_amd import synth
    g = np.load(os.path.join(GOLDEN, "config2_1024_oracle_f64.npz"))
    assert float(g["true_rel_residual"]) <= 1e-10
    sizes, w, pos, val = synth.config2()
    assert sizes == [int(s) for s in g["sizes"]] and len(pos) == int(g["num_points"])
    f = bs.headline_field(fi, 2, sizes, w)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.assemble()
    rows = []
    for tol in (bs.SETTINGS[2]["tol"], 1e-7, 1e-9, 1e-11):
        x, it, rel = f.solve_cg(None, 0, tol)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.01 * tol
        rows.append((tol, it, _against_sample(f.solution_f64(), g)))
        if rows[-1][2] <= FIELD_TOL:
            break
    with capsys.disabled():
        print("\n[config 2 at 1024^2 against the oracle's exact solve] " + "; ".join("residual %g: %d iterations, field error %.2e" % r for r in rows))
    assert rows[0][2] <= 5e-2          # the configuration's own residual: reported, bounded
    assert rows[-1][2] <= FIELD_TOL    # the solver reaches the reference's solution


def test_config3_shape_against_the_oracles_exact_solve(fi, capsys):
    """BASELINE config 3's shape (SDF from oriented points: triangle + inverted circle, default Weights) at 1024^2 against
    the reference's exact route (as above), with bench.py --config 3's settings on the same coarsest lattice."""
    from field_interpolation_amd import bench_settings as bs
    from field_interpolation_amd import synth
    g = np.load(os.path.join(GOLDEN, "config3_1024_oracle_f64.npz"))
    assert float(g["true_rel_residual"]) <= 1e-10
    sizes, w, pos, nrm = synth.config3(side=1024, points_per_shape=int(g["num_points"]) // 2, seed=2)
    assert sizes == [int(s) for s in g["sizes"]]
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w)
    bs.configure(f, bs.SETTINGS[3]["levels"] - 2, bs.SETTINGS[3]["coarse_tol"], kcycle=bs.SETTINGS[3].get("kcycle", 0), cheb=bs.SETTINGS[3].get("cheb"))     # 4096 -> 1024: two levels less, the same coarsest lattice
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    f.assemble()
    rows = []
    for tol in (bs.SETTINGS[3]["tol"], 1e-7, 1e-9, 1e-11):
        x, it, rel = f.solve_cg(None, 0, tol)
        assert f.stats()["converged"] == 1 and f.true_residual() <= 1.01 * tol
        rows.append((tol, it, _against_sample(f.solution_f64(), g)))
        if rows[-1][2] <= FIELD_TOL:
            break
    with capsys.disabled():
        print("\n[config 3's shape at 1024^2 against the oracle's exact solve] " + "; ".join("residual %g: %d iterations, field error %.2e" % r for r in rows))
    assert rows[-1][2] <= FIELD_TOL


def test_the_field_rule_on_the_other_configurations(fi, capsys):
    """bench.py's stop rule (FI_OPT_FIELD_TOLERANCE = 1e-5, no constant that depends on the workload) with bench.py's settings
    on every other configuration the oracle has solved: config 2 at its full size, config 3's shape at 1024^2, config 5's shape
    at 128^3 -- value data with a stiff prior, and oriented points, whose field costs 300 times more residual than config 2's.
    The field is within 1e-5 of the oracle's on each (only config 4 met it, at a residual tuned for it)."""
    from field_interpolation_amd import bench_settings as bs
    from field_interpolation_amd import synth
    rows = []
    g = np.load(os.path.join(GOLDEN, "config2_1024_oracle_f64.npz"))
    sizes, w, pos, val = synth.config2()
    f = bs.headline_field(fi, 2, sizes, w, by_field=True)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    rows.append(("config 2 at 1024^2", f, g, bs.SETTINGS[2]["tol"]))
    g = np.load(os.path.join(GOLDEN, "config3_1024_oracle_f64.npz"))
    sizes, w, pos, nrm = synth.config3(side=1024, points_per_shape=int(g["num_points"]) // 2, seed=2)
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w)
    bs.configure(f, bs.SETTINGS[3]["levels"] - 2, bs.SETTINGS[3]["coarse_tol"], by_field=True, kcycle=bs.SETTINGS[3].get("kcycle", 0), cheb=bs.SETTINGS[3].get("cheb"))
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    rows.append(("config 3's shape at 1024^2", f, g, bs.SETTINGS[3]["tol"]))
    g = np.load(os.path.join(GOLDEN, "config5_128_oracle_f64.npz"))
    sizes, w, pos, nrm = synth.config5(side=128, num_points=int(g["num_points"]), seed=int(g["seed"]))
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w)
    bs.configure(f, bs.SETTINGS[5]["levels"] - 2, bs.SETTINGS[5]["coarse_tol"], by_field=True, kcycle=bs.SETTINGS[5].get("kcycle", 0), cheb=bs.SETTINGS[5].get("cheb"))
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    rows.append(("config 5's shape at 128^3", f, g, bs.SETTINGS[5]["tol"]))
    for name, f, g, tol in rows:
        f.assemble()
        res = f.solve_cg(None, 0, tol)
        st = f.stats()
        err = _against_sample(f.solution_f64(), g)
        with capsys.disabled():
            print("\n[the field rule, %s] %d iterations, stopped at residual %.2e, estimate %.2e, field error against the oracle %.2e "
                  "(%.0f per unit of residual)" % (name, res[1], st["stop_residual"], st["field_estimate"], err, err / st["stop_residual"]))
        assert st["converged"] == 1 and st["field_rounds"] == 1 and 0 <= st["field_estimate"] <= FIELD_TOL
        assert err <= FIELD_TOL


def test_the_field_rule_on_random_problems(fi, capsys):
    """The stop rule away from the configurations it was built on: sixteen random 3-D problems of tests/stress_field_rule.py
    (value data or oriented points, one to three levels, fp64 or mixed V-cycles, tolerances 1e-4 .. 1e-6, solves of 6 to 400
    iterations) against the same context's solve to the fp64 floor.  The rule is an estimate: the sweep's bar is twice the
    tolerance (200 cases in 3-D on the GPU box: 3 above the tolerance, none above twice; 150 in 2-D: 7 and 1, worst 2.4 x; 300 with the K-cycle: 12 and 2;
    the rule of the round's first builds -- one step's residual ratio alone -- failed 12 of the first 61, by up to 16 x)."""
    import stress_field_rule as sweep
    worst = 0.0
    for seed in range(71000, 71016):
        desc, errs, ratio = sweep.one_case(seed)
        assert not errs, desc + " -> " + "; ".join(errs)
        worst = max(worst, ratio)
    with capsys.disabled():
        print("\n[the field rule on 16 random problems] largest true error / tolerance %.2f" % worst)


def test_config4_over_eight_slabs_against_the_oracle(fi, capsys):
    """The lattice of `bench.py --gpus 8` (256^3 cut into eight slabs of 32 planes; the loop-back group: the RCCL path's
    kernels, geometry, ownership rules and reductions in one process) with bench.py's settings and stop rule -- over slabs the
    slabs' maxima ride on the r . r sum -- against the oracle's fp64 solution: within the north-star's 1e-5, in the
    undivided solve's iterations give or take one."""
    from field_interpolation_amd import bench_settings as bs
    from field_interpolation_amd import synth
    g = np.load(os.path.join(GOLDEN, "config4_256_oracle_f64.npz"))
    sizes, w, pos, val = synth.config4(seed=3)
    grp = fi.LatticeGroup(sizes, 8, dtype="f64")
    grp.add_field_constraints(w)
    bs.configure(grp, bs.SETTINGS[4]["levels"], bs.SETTINGS[4]["coarse_tol"], by_field=True)
    grp.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    grp.assemble()
    x, it, rel = grp.solve_cg(None, 0, bs.SETTINGS[4]["tol"])
    st = grp.stats()
    err = _against_sample(grp.solution_f64(), g)
    with capsys.disabled():
        print("\n[config 4 at 256^3 over eight slabs, against the oracle] %d iterations, stopped by the field at residual %.2e "
              "(estimate %.2e), field error %.2e" % (it, st["stop_residual"], st["field_estimate"], err))
    assert st["converged"] == 1 and 0 < st["field_estimate"] <= FIELD_TOL and 6 <= it <= 8
    assert err <= FIELD_TOL
