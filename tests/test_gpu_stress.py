"""A fixed-seed slice of the randomised parity sweep (tests/stress_parity.py): random shapes (1-3 D), weights, value /
gradient kernels, point clouds with edge cases, fp32 / fp64, random slab counts -- operator pieces against the
oracle's float64 normal equations, slabs against the undivided operator, error map and tile pre-solver against the
oracle.  3400 further seeds were run by hand on the MI355X box without a failure."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sweep():
    from field_interpolation_amd import _capi
    assert _capi.device_count() >= 1
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_parity.py")
    spec = importlib.util.spec_from_file_location("stress_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("first", [100, 140, 180, 220])
def test_random_cases(sweep, first):
    failures = []
    for seed in range(first, first + 40):
        desc, errs = sweep.one_case(seed)
        if errs:
            failures.append((desc, errs))
    assert not failures, failures[:3]


@pytest.fixture(scope="module")
def solve_sweep():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_solve.py")
    spec = importlib.util.spec_from_file_location("stress_solve", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_solver_cases(solve_sweep, monkeypatch):
    """A fixed-seed slice of tests/stress_solve.py (every solver mode, 1-4 slabs, against the oracle's float64 direct solve).
    Seed 5039 -- nearest-neighbour gradient rows, mixed precision, one coarser level -- stagnated at 2e-3 for 60 000
    iterations while the finest level's smoother took its bound from the level below it (rounds 1-2); 700 seeds of round 3
    without a failure."""
    monkeypatch.setenv("FI_SOLVE_TIMEOUT_S", "30")
    failures = []
    for seed in [5039] + list(range(5000, 5012)):
        desc, errs = solve_sweep.one_case(seed)
        if errs:
            failures.append((desc, errs))
    assert not failures, failures[:3]
