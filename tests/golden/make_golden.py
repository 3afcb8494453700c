#!/usr/bin/env python3
"""Writes the fixtures of tests/golden/.

known_answers.json  the only numbers that come from the REFERENCE: the README's worked example
                    (README.md:11-40) and the survey-time probe of the reference assembly for the field_1d.cpp
                    default input, plus the closed-form row counts of SURVEY.md section 8.
                    Typed in from those documents; this script only re-writes them in one place.
oracle_cases.npz    regression vectors made by THIS repository's oracle (the reference cannot be built in this
                    image: it needs loguru and Eigen -- DESIGN.md section 2), for a matrix of small cases: the
                    inputs themselves, row / triplet counts, A^T b, diag(A^T A), (A^T A) x for a stored x, and
                    the float64 least-squares solution.  They pin the oracle against drift and give the GPU tests
                    vectors to compare with that do not need the oracle at run time.

usage (repository root):  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [
    # name, sizes, weights, value kernel, gradient kernel, points, with normals, with values
    ("1d_default", [64], dict(), 1, 1, 24, True, False),
    ("1d_all_orders", [40], dict(model_0=0.3, model_1=0.7, model_2=0.5, model_3=0.9, model_4=1.1), 1, 0, 30, True, True),
    ("2d_default", [24, 20], dict(), 1, 1, 160, True, False),
    ("2d_values_only", [32, 17], dict(model_2=10.0), 1, 1, 120, False, True),
    ("2d_nearest_kernels", [16, 12], dict(model_1=0.4, gradient_smoothness=0.3), 0, 0, 90, True, True),
    ("2d_linear_gradient", [20, 14], dict(model_2=0.5), 1, 2, 110, True, False),
    ("3d_default", [12, 10, 9], dict(), 1, 1, 200, True, False),
    ("3d_values_model_1", [16, 8, 11], dict(model_2=0.0, model_1=0.8, model_0=0.1), 1, 1, 150, False, True),
    ("3d_linear_gradient", [9, 8, 7], dict(model_2=0.6, gradient_smoothness=0.2), 1, 2, 140, True, False),
]


def known_answers():
    return {
        "readme_example": {
            "source": "README.md:11-40",
            "A": [[1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], [-1, 1, 0, 0, 0, 0], [0, 0, 0, 0, -1, 1],
                  [1, -2, 1, 0, 0, 0], [0, 1, -2, 1, 0, 0], [0, 0, 1, -2, 1, 0], [0, 0, 0, 1, -2, 1]],
            "b": [4, 2, 1, -1, 0, 0, 0, 0],
        },
        "field_1d_resolution_12": {
            "source": "SURVEY.md 8(c): reference assembly run at survey time on the field_1d.cpp:20-29 default input",
            "rows": 14, "triplets": 38, "cond_AtA": 165.7,
            "solution": [-0.1846154, -0.1006993, -0.0167832, 0.0671329, 0.1230769, 0.1510490,
                         0.1510490, 0.1230769, 0.0671329, -0.0167832, -0.1006993, -0.1846154],
        },
        "row_counts": {
            "source": "SURVEY.md section 8 table (default Weights: model_2 only)",
            "C1_1024": {"model_rows": 1022, "model_triplets": 3066, "data_rows": 4, "data_triplets": 8},
            "C2_1024x1024": {"model_rows": 2093056, "model_triplets": 6279168},
        },
    }


def main():
    from oracle import fi_oracle as oracle
    from util import random_points
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(known_answers(), f, indent=1)
    out = {}
    for k, (name, sizes, kw, vk, gk, npts, with_nrm, with_val) in enumerate(CASES):
        rng = np.random.default_rng(1000 + k)
        pos, nrm, pw, val = random_points(rng, sizes, npts, margin=1.0)
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
        n = int(np.prod(sizes))
        AtA, atb, diag = f.normal_equations()
        x = np.random.default_rng(2000 + k).normal(size=n)
        rows, cols, vals, rhs = f.get()
        out[name + "/pos"], out[name + "/nrm"], out[name + "/pw"], out[name + "/val"] = pos, nrm, pw, val
        out[name + "/counts"] = np.array([f.num_rows, f.num_triplets], np.int64)
        out[name + "/triplet_sums"] = np.array([np.abs(vals.astype(np.float64)).sum(), rhs.astype(np.float64).sum(),
                                                (vals.astype(np.float64) * (cols + 1)).sum()])
        out[name + "/atb"], out[name + "/diag"] = atb, diag
        out[name + "/x"], out[name + "/AtAx"] = x, AtA @ x
        out[name + "/solution"] = f.solve_exact_f64()
    np.savez_compressed(os.path.join(HERE, "oracle_cases.npz"), **out)
    print("wrote known_answers.json and oracle_cases.npz (%d cases)" % len(CASES))


if __name__ == "__main__":
    main()
