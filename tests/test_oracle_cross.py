"""The C++ oracle (oracle/fi_oracle.cpp) against the independent numpy restatement
(oracle/fi_oracle_py.py): identical triplets (row, col, fp32 bits) and fp32 rhs, in order."""
import numpy as np
import pytest

from oracle import fi_oracle_py as py


def _points(rng, sizes, n, margin=1.5):
    D = len(sizes)
    pos = np.stack([rng.uniform(-margin, s - 1 + margin, n) for s in sizes], axis=1).astype(np.float32)
    # sprinkle exact lattice hits, cell borders and half positions (t == 0 / 0.5 edge cases)
    pos[: n // 6] = np.round(pos[: n // 6])
    pos[n // 6: n // 3] = np.floor(pos[n // 6: n // 3]) + 0.5
    nrm = rng.normal(size=(n, D)).astype(np.float32)
    pw = rng.uniform(0.0, 2.0, n).astype(np.float32)
    pw[::7] = 0.0          # zero-weight points are dropped (field_interpolation.cpp:63,130)
    return pos, nrm, pw


def _same(f_cpp, f_py):
    r1, c1, v1, b1 = f_cpp.get()
    r2, c2, v2, b2 = f_py.arrays()
    assert len(b1) == len(b2) and len(v1) == len(v2)
    np.testing.assert_array_equal(r1, r2)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(v1.view(np.uint32), v2.view(np.uint32))
    np.testing.assert_array_equal(b1.view(np.uint32), b2.view(np.uint32))


@pytest.mark.parametrize("sizes", [[12], [9, 7], [5, 6, 4]])
@pytest.mark.parametrize("vk", [0, 1])
@pytest.mark.parametrize("gk", [0, 1, 2])
def test_point_rows_identical(oracle, sizes, vk, gk):
    rng = np.random.default_rng(hash((tuple(sizes), vk, gk)) % 2**32)
    pos, nrm, pw = _points(rng, sizes, 60)
    w = oracle.Weights(data_pos=0.7, data_gradient=1.3, value_kernel=vk, gradient_kernel=gk)
    f1 = oracle.LatticeField(sizes)
    f1.add_points(w.data_pos, vk, w.data_gradient, gk, pos, nrm, pw)
    f2 = py.PyField(sizes)
    f2.add_points(w.data_pos, vk, w.data_gradient, gk, pos, nrm, pw)
    _same(f1, f2)


@pytest.mark.parametrize("sizes", [[11], [7, 6], [5, 4, 6]])
@pytest.mark.parametrize("kw", [
    dict(),                                                        # default: model_2 = 0.5
    dict(model_0=0.1, model_1=0.3, model_2=0.5, model_3=0.7, model_4=0.9),
    dict(model_2=0.0, gradient_smoothness=0.25),
    dict(model_2=10.0, model_1=1.0, gradient_smoothness=0.5),
])
def test_model_rows_identical(oracle, sizes, kw):
    w = oracle.Weights(**kw)
    f1 = oracle.LatticeField(sizes)
    f1.add_field_constraints(w)
    f2 = py.PyField(sizes)
    f2.add_field_constraints(w)
    _same(f1, f2)


@pytest.mark.parametrize("sizes", [[16], [8, 9], [4, 5, 6]])
def test_sdf_from_points_identical_and_solution(oracle, sizes):
    rng = np.random.default_rng(len(sizes))
    pos, nrm, pw = _points(rng, sizes, 40, margin=0.5)
    w = oracle.Weights()
    f1 = oracle.sdf_from_points(sizes, w, pos, nrm, pw)
    f2 = py.sdf_from_points(sizes, w, pos, nrm, pw)
    _same(f1, f2)
    # float64 least squares from the python side vs the oracle's exact solve
    A, b = f2.dense()
    x_ref = np.linalg.solve(A.T @ A, A.T @ b)
    x = f1.solve_exact_f64()
    np.testing.assert_allclose(x, x_ref, rtol=1e-9, atol=1e-11)
    # and the normal equations themselves
    AtA, atb, diag = f1.normal_equations()
    np.testing.assert_allclose(AtA.toarray(), A.T @ A, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(atb, A.T @ b, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(diag, np.diag(A.T @ A), rtol=1e-12, atol=1e-14)


def test_single_constraint_return_values(oracle):
    f = oracle.LatticeField([4, 4])
    p = py.PyField([4, 4])
    cases = [([-1.5, 1.0], False), ([-0.5, 1.0], True), ([3.0, 3.0], True), ([3.5, 3.5], True), ([4.0, 1.0], False)]
    for pos, expect in cases:
        assert f.add_value_constraint(pos, 1.0, 1.0) == expect == p.add_value_constraint(pos, 1.0, 1.0)
    assert not f.add_value_constraint([1.0, 1.0], 1.0, 0.0)
    for k in (0, 1):
        assert f.add_gradient_constraint([2.5, 2.5], [1, 0], 1.0, k)
        assert not f.add_gradient_constraint([3.0, 2.5], [1, 0], 1.0, k)      # cell_index: p+1 < size
        assert not f.add_gradient_constraint([-0.1, 2.5], [1, 0], 1.0, k)
    assert f.add_gradient_constraint([3.4, 2.5], [1, 0], 1.0, 2)              # 3.4-0.5 -> cell 2, idx+1 = 3 ok
    assert not f.add_gradient_constraint([9.0, 2.5], [1, 0], 1.0, 2)
    with pytest.raises(ValueError):
        f.add_gradient_constraint([1.0, 1.0], [1, 0], 1.0, 7)
    assert f.add_value_constraint_nearest_neighbor([3.4, -0.4], [1, 1], 2.0, 1.0)
    assert not f.add_value_constraint_nearest_neighbor([3.5, 0.0], [1, 1], 2.0, 1.0)   # round(3.5) = 4
    assert not f.add_value_constraint_nearest_neighbor([0.0, -0.5], [1, 1], 2.0, 1.0)  # round(-0.5) = -1
