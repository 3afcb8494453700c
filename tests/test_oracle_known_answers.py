"""Pins the CPU oracle against every known answer that exists for the path.

The reference ships no tests and cannot be built here, so these are all the
anchors there are:
  * README.md:11-40: the 6-point worked example (8x6 matrix A and rhs b); README.md "Data interpolation",
    "Extending it to multiple dimensions", "Making it work without point normals": the rows of f(3.4) = 10 and
    f'(3.1) = -12, the 3-D smoothness rows, and the straight line through f(0) = 10, f(10) = 0;
  * survey-time output of the reference assembly for the field_1d.cpp:20-29
    default input (resolution 12): 14 rows / 38 triplets and the float64 least-squares solution
    (cond(AtA) = 165.7);
  * closed-form row / triplet counts of configs C1..C3 (default Weights).
"""
import json
import os

import numpy as np
import pytest


with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answers.json")) as _f:
    KNOWN = json.load(_f)          # tests/golden/known_answers.json: the numbers typed in from README.md / SURVEY.md
README_A = np.array(KNOWN["readme_example"]["A"], dtype=np.float64)
README_B = np.array(KNOWN["readme_example"]["b"], dtype=np.float64)
SURVEY_FIELD_1D_X = np.array(KNOWN["field_1d_resolution_12"]["solution"])


def _dense(f, n):
    rows, cols, vals, rhs = f.get()
    A = np.zeros((len(rhs), n))
    np.add.at(A, (rows, cols), vals.astype(np.float64))
    return A, rhs.astype(np.float64)


def test_readme_worked_example(oracle):
    """README.md:11-40.  Value rows via add_value_constraint (f(0)=4, f(5)=2), the gradient at
    x=0 via add_gradient_constraint (kNearestNeighbor: f(1)-f(0)=1), smoothness via
    add_field_constraints(model_2=1).  The README's last gradient row f(5)-f(4)=-1 sits at x=5,
    where cell_index (field_interpolation.cpp:116-117) rejects the point -- so that row is entered
    with add_equation, exactly as SURVEY.md section 4 prescribes for this example."""
    f = oracle.LatticeField([6])
    assert f.add_value_constraint([0.0], 4.0, 1.0)
    assert f.add_value_constraint([5.0], 2.0, 1.0)
    assert f.add_gradient_constraint([0.0], [1.0], 1.0, oracle.GRAD_NEAREST)
    assert not f.add_gradient_constraint([5.0], [-1.0], 1.0, oracle.GRAD_NEAREST)   # dropped at the border
    f.add_equation(1.0, -1.0, [(4, -1.0), (5, 1.0)])
    f.add_field_constraints(oracle.Weights(model_2=1.0))
    A, b = _dense(f, 6)
    # add_value_constraint also stores the zero-weight neighbour (x1 with weight 0 for pos 0.0);
    # the dense matrix is what the README prints.
    assert A.shape == (8, 6)
    np.testing.assert_array_equal(A, README_A)
    np.testing.assert_array_equal(b, README_B)
    x = f.solve_exact()
    x_ref = np.linalg.lstsq(README_A, README_B, rcond=None)[0]
    np.testing.assert_allclose(x, x_ref, rtol=0, atol=2e-6)


def _field_1d(oracle, resolution):
    """src/field_1d.cpp:98-110 with the default input of :20-29."""
    w = oracle.Weights()
    f = oracle.LatticeField([resolution])
    for pos, value, grad in [(0.2, 0.0, +1.0), (0.8, 0.0, -1.0)]:
        pos_lattice = np.float32(pos) * np.float32(resolution - 1)
        grad_lattice = np.float32(grad) / np.float32(resolution - 1)
        f.add_value_constraint([pos_lattice], value, w.data_pos)
        f.add_gradient_constraint([pos_lattice], [grad_lattice], w.data_gradient, w.gradient_kernel)
    f.add_field_constraints(w)
    return f


def test_field_1d_default_matches_survey_probe(oracle):
    f = _field_1d(oracle, 12)
    probe = KNOWN["field_1d_resolution_12"]
    assert f.num_rows == probe["rows"] and f.num_triplets == probe["triplets"]          # SURVEY.md 8(c) [probe]
    x = f.solve_exact()
    np.testing.assert_allclose(x, SURVEY_FIELD_1D_X, rtol=0, atol=1e-7)
    AtA, _, _ = f.normal_equations()
    assert abs(np.linalg.cond(AtA.toarray()) - 165.7) < 0.1   # cond(AtA)=165.7


def test_config_c1_counts(oracle):
    """1022 model rows / 3066 triplets + (2+2) data rows / 8 triplets."""
    f = _field_1d(oracle, 1024)
    assert f.num_rows == 1022 + 4
    assert f.num_triplets == 3066 + 8


@pytest.mark.parametrize("sizes,rows,trip", [
    ([1024, 1024], 2093056, 6279168),       # C2 model part
    ([64, 64, 64], 3 * 64 * 64 * 62, 3 * 3 * 64 * 64 * 62),
])
def test_model_row_counts_closed_form(oracle, sizes, rows, trip):
    """model rows = D*N*(1-2/side), x3 triplets (default Weights)."""
    f = oracle.LatticeField(sizes)
    f.add_field_constraints(oracle.Weights())
    assert f.num_rows == rows and f.num_triplets == trip


def test_ata_interior_stencil_is_1_m4_6_m4_1(oracle):
    """interior row of model_2=w in 1-D is w^2*[1 -4 6 -4 1]; the diagonal
    runs 1,5,6,...,6,5,1 (section 7 hard part 5)."""
    f = oracle.LatticeField([16])
    f.add_field_constraints(oracle.Weights(model_2=0.5))
    AtA, atb, diag = f.normal_equations()
    M = AtA.toarray() / 0.25
    np.testing.assert_allclose(M[8, 6:11], [1, -4, 6, -4, 1])
    np.testing.assert_allclose(np.diag(M), [1, 5] + [6] * 12 + [5, 1])
    assert not atb.any()


def test_model_0_is_emitted_once_per_axis(oracle):
    """field_interpolation.cpp:257-263 sits inside the per-axis loop: diag = D*model_0^2."""
    f = oracle.LatticeField([5, 4, 3])
    f.add_field_constraints(oracle.Weights(model_0=2.0, model_2=0.0))
    _, _, diag = f.normal_equations()
    np.testing.assert_allclose(diag, 3 * 4.0)


def test_readme_data_interpolation_rows(oracle):
    """README.md "Data interpolation": `f(3.4) = 10` on an integer lattice becomes `0.6 f(3) + 0.4 f(4) = 10`
    (linear-interpolation value kernel), and `f'(3.1) = -12` becomes `f(4) - f(3) = -12` (nearest-neighbour gradient
    kernel).  fp32 weights: 3.4f - 3 = 0.4000001."""
    f = oracle.LatticeField([8])
    assert f.add_value_constraint([3.4], 10.0, 1.0)
    assert f.add_gradient_constraint([3.1], [-12.0], 1.0, oracle.GRAD_NEAREST)
    A, b = _dense(f, 8)
    assert A.shape == (2, 8)
    expect = np.zeros((2, 8))
    expect[0, 3], expect[0, 4] = 0.6, 0.4
    expect[1, 3], expect[1, 4] = -1.0, 1.0
    np.testing.assert_allclose(A, expect, rtol=0, atol=1e-6)
    np.testing.assert_allclose(b, [10.0, -12.0], rtol=0, atol=1e-5)


def test_readme_smoothness_rows_3d(oracle):
    """README.md "Extending it to multiple dimensions": f(x,y,z) - 2 f(x+1,y,z) + f(x+2,y,z) = 0 and the same along
    y and z: with model_2 = 1 every model row is [1, -2, 1] on three points one stride apart along one axis, rhs 0."""
    sizes = [5, 4, 6]
    f = oracle.LatticeField(sizes)
    f.add_field_constraints(oracle.Weights(model_2=1.0))
    rows, cols, vals, rhs = f.get()
    assert not rhs.any()
    strides = [1, sizes[0], sizes[0] * sizes[1]]
    n_expect = sum((sizes[d] - 2) * int(np.prod(sizes)) // sizes[d] for d in range(3))
    assert len(rhs) == n_expect
    order = np.lexsort((cols, rows))
    r, c, v = rows[order].reshape(-1, 3), cols[order].reshape(-1, 3), vals[order].reshape(-1, 3)
    assert (r[:, 0] == r[:, 2]).all()
    np.testing.assert_array_equal(v, np.tile(np.float32([1, -2, 1]), (len(r), 1)))
    step = c[:, 1] - c[:, 0]
    assert (c[:, 2] - c[:, 1] == step).all() and set(np.unique(step)) == set(strides)


def test_readme_pivot_extrapolation(oracle):
    """README.md "Making it work without point normals": with a smoothness constraint and the data f(0) = 10,
    f(10) = 0 the solver "will be able to figure out that f(20) = -10": the least-squares solution is the straight
    line 10 - x, whatever the weights."""
    f = oracle.LatticeField([21])
    assert f.add_value_constraint([0.0], 10.0, 1.0)
    assert f.add_value_constraint([10.0], 0.0, 1.0)
    f.add_field_constraints(oracle.Weights(model_2=0.5))
    x = f.solve_exact()
    np.testing.assert_allclose(x, 10.0 - np.arange(21), rtol=0, atol=2e-4)
    assert abs(x[20] + 10.0) <= 2e-4


@pytest.mark.parametrize("term,degree,weight", [("model_1", 0, 1e3), ("model_2", 1, 1e3), ("model_3", 2, 1e4),
                                                ("model_4", 3, 1e5)])
def test_header_limit_behaviour_of_the_model_weights(oracle, term, degree, weight):
    """field_interpolation.hpp:79-85: a large model_1 "take[s] the average of the data", a large model_2 fits a line,
    model_3 a quadratic, model_4 a cubic: with weight 1e3..1e5 against data weight 1 the solution is the least-squares
    polynomial of that degree through the data (values on lattice points) to ~1e-3 of the data's range."""
    rng = np.random.default_rng(degree)
    n = 40
    at = np.sort(rng.choice(n, 12, replace=False))
    val = rng.normal(size=12)
    f = oracle.LatticeField([n])
    for p, v in zip(at, val):
        assert f.add_value_constraint([float(p)], float(v), 1.0)
    f.add_field_constraints(oracle.Weights(**{"model_2": 0.0, term: weight}))
    x = f.solve_exact_f64()
    fit = np.polyval(np.polyfit(at, val, degree), np.arange(n))
    assert np.abs(x - fit).max() <= 2e-3 * (np.abs(val).max() + np.abs(fit).max())


def test_header_model_0_pulls_the_field_to_zero(oracle):
    """field_interpolation.hpp:78: "If this is large everything will be zero"."""
    f = oracle.LatticeField([12, 9])
    rng = np.random.default_rng(4)
    for _ in range(20):
        f.add_value_constraint([rng.uniform(0, 11), rng.uniform(0, 8)], 5.0, 1.0)
    f.add_field_constraints(oracle.Weights(model_0=1e3, model_2=0.0))
    x = f.solve_exact_f64()
    assert np.abs(x).max() <= 5.0 * 1e-5
