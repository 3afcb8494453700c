#!/usr/bin/env python3
"""Single-reduction recurrence over a loop-back group vs the two-reduction form: iterations on the SDF test problem."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import field_interpolation_amd as fi
from util import sphere_points
for dtype, tol in (("f64", 1e-9), ("f32", 1e-4)):
    sizes = [16, 12, 24]
    rng = np.random.default_rng(2)
    pos, nrm = sphere_points(rng, sizes, 250)
    w = fi.Weights()
    g = fi.LatticeGroup(sizes, 2, dtype=dtype) if not os.environ.get("ONE") else fi.LatticeField(sizes, dtype=dtype)
    g.add_field_constraints(w); g.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None); g.assemble(); g.set_polynomial(4)
    x, it, rel = g.solve_cg(None, 0, tol)
    print(os.environ.get("TAG"), dtype, "iterations", it, "rel", rel, "true", g.true_residual(), flush=True)
