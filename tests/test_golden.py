"""tests/golden/: the known answers that come from the reference's documents (known_answers.json) and the
regression vectors made by this repository's oracle (oracle_cases.npz; tests/golden/make_golden.py).
CPU: the oracle still reproduces every stored vector.  GPU: the HIP path against the stored vectors alone -- the
oracle is not called -- fp64 contexts to 1e-12 (operator pieces) / 1e-6 (solutions), fp32 to 3e-6."""
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load_cases():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CASES, mod.known_answers()


CASES, KNOWN = _load_cases()
DATA = np.load(os.path.join(HERE, "oracle_cases.npz"))
IDS = [c[0] for c in CASES]


def test_known_answers_file_is_what_the_script_writes():
    with open(os.path.join(HERE, "known_answers.json")) as f:
        assert json.load(f) == json.loads(json.dumps(KNOWN))


def test_known_answers_are_consistent():
    """The README system's least-squares solution and the survey probe are fixed points of plain linear algebra."""
    A, b = np.array(KNOWN["readme_example"]["A"], float), np.array(KNOWN["readme_example"]["b"], float)
    x = np.linalg.lstsq(A, b, rcond=None)[0]
    assert np.abs(A.T @ (A @ x - b)).max() < 1e-12
    s = np.array(KNOWN["field_1d_resolution_12"]["solution"])
    np.testing.assert_allclose(s, s[::-1], atol=1e-7)          # the default input is mirror symmetric


def _build_oracle(oracle, case):
    name, sizes, kw, vk, gk, npts, with_nrm, with_val = case
    pos, nrm, pw, val = (DATA[name + "/" + k] for k in ("pos", "nrm", "pw", "val"))
    w = oracle.Weights(data_pos=0.8, data_gradient=1.25, value_kernel=vk, gradient_kernel=gk, **kw)
    f = oracle.LatticeField(sizes)
    f.add_field_constraints(w)
    if with_val:
        for i in range(npts):
            wi = float(np.float32(pw[i]) * np.float32(w.data_pos))
            if vk == 0:
                f.add_value_constraint_nearest_neighbor(pos[i], nrm[i], float(val[i]), wi)
            else:
                f.add_value_constraint(pos[i], float(val[i]), wi)
            if with_nrm:
                f.add_gradient_constraint(pos[i], nrm[i], float(np.float32(pw[i]) * np.float32(w.data_gradient)), gk)
    else:
        f.add_points(w.data_pos, vk, w.data_gradient, gk, pos, nrm if with_nrm else None, pw)
    return f


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_reproduces_the_fixtures(oracle, case):
    name = case[0]
    f = _build_oracle(oracle, case)
    assert [f.num_rows, f.num_triplets] == list(DATA[name + "/counts"])
    rows, cols, vals, rhs = f.get()
    sums = np.array([np.abs(vals.astype(np.float64)).sum(), rhs.astype(np.float64).sum(),
                     (vals.astype(np.float64) * (cols + 1)).sum()])
    np.testing.assert_allclose(sums, DATA[name + "/triplet_sums"], rtol=1e-12, atol=1e-12)
    AtA, atb, diag = f.normal_equations()
    np.testing.assert_allclose(atb, DATA[name + "/atb"], rtol=0, atol=1e-12 * max(np.abs(atb).max(), 1))
    np.testing.assert_allclose(diag, DATA[name + "/diag"], rtol=1e-12)
    np.testing.assert_allclose(AtA @ DATA[name + "/x"], DATA[name + "/AtAx"], rtol=0,
                               atol=1e-12 * np.abs(DATA[name + "/AtAx"]).max())
    np.testing.assert_allclose(f.solve_exact_f64(), DATA[name + "/solution"], rtol=0,
                               atol=1e-9 * np.abs(DATA[name + "/solution"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_gpu_against_the_fixtures(case, dtype):
    import field_interpolation_amd as fi
    name, sizes, kw, vk, gk, npts, with_nrm, with_val = case
    pos, nrm, pw, val = (DATA[name + "/" + k] for k in ("pos", "nrm", "pw", "val"))
    w = fi.Weights(data_pos=0.8, data_gradient=1.25, value_kernel=fi.ValueKernel(vk), gradient_kernel=fi.GradientKernel(gk), **kw)
    f = fi.LatticeField(sizes, dtype=dtype)
    f.add_field_constraints(w)
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient if with_nrm else 0.0, w.gradient_kernel, pos,
                 nrm if (with_nrm or vk == 0) else None, pw, values=val if with_val else None)
    f.assemble()
    tol = 1e-12 if dtype == "f64" else 3e-6
    atb, diag, x, y = (DATA[name + "/" + k] for k in ("atb", "diag", "x", "AtAx"))
    assert np.abs(f.Atb() - atb).max() <= tol * max(np.abs(atb).max(), 1e-300)
    assert np.abs(f.diag() - diag).max() <= tol * np.abs(diag).max()
    assert np.abs(f.apply_AtA(x) - y).max() <= 10 * tol * np.abs(y).max()
    if dtype == "f64":
        res = f.solve_cg(None, 200000, 1e-13)
        assert res is not None
        sol = DATA[name + "/solution"]
        assert np.abs(f.solution_f64() - sol).max() <= 1e-6 * np.abs(sol).max()
