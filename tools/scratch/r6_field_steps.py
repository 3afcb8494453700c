#!/usr/bin/env python3
"""Check of the step sizes the field rule works with: ||x_k - x_(k-1)||_inf / ||x_k||_inf from iterates recomputed with
max_iterations = k against the solver's own |alpha| max |p| / max |x| (FI_FIELD_TRACE: a timing build, tools/build_variant.sh).  usage: r6_field_steps.py <seed> <k0> <k1>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import field_interpolation_amd as fi
from util import rel_inf, sphere_points
seed = int(sys.argv[1]); k0 = int(sys.argv[2]); k1 = int(sys.argv[3])
rng = np.random.default_rng(seed)
big = rng.random() < 0.25
sizes = [int(rng.integers(40, 161 if big else 73)) for _ in range(3)]
if seed >= 80000:
    sizes = [int(rng.integers(96, 1025 if big else 385)) for _ in range(2)]
sizes[0] = max(8, (sizes[0] // 4) * 4)
kw = dict(model_2=float(rng.uniform(0.2, 1.0)))
if rng.random() < 0.4: kw["model_1"] = float(rng.uniform(0.02, 0.5))
if rng.random() < 0.15: kw["model_0"] = float(rng.uniform(0.001, 0.02))
sdf = rng.random() < 0.5
gk = int(rng.integers(0, 3)) if sdf else 1
w = fi.Weights(gradient_kernel=fi.GradientKernel(gk), **kw)
n = int(np.prod(sizes))
npts = int(rng.integers(200, max(400, n // 20)))
pos, nrm = sphere_points(rng, sizes, npts, noise=float(rng.uniform(0.1, 1.0)))
val = None if sdf else rng.normal(size=npts).astype(np.float32)
mixed = rng.random() < 0.7
levels = int(rng.integers(1, 4)) if len(sizes) == 3 else int(rng.integers(1, 6))
f = fi.LatticeField(sizes, dtype="f64")
f.add_field_constraints(w)
f.add_points(w.data_pos, w.value_kernel, w.data_gradient if sdf else 0.0, w.gradient_kernel, pos, nrm if sdf else None, None, values=val)
f.set_levels(levels, 1e-3); f.set_multigrid(True)
if mixed: f.set_mixed_precision(True)
f.assemble()
f.solve_cg(None, 4000, 1e-13)
ref = f.solution_f64().copy()
prev = None
for k in range(k0 - 1, k1 + 1):
    try:
        f.solve_cg(None, k, 1e-13)
    except Exception:
        pass
    x = f.solution_f64().copy()
    if prev is not None:
        print("k %d: ||x_k - x_(k-1)|| / ||x_k|| = %.6e   true error %.6e  iterations reported %d" % (k, np.abs(x - prev).max() / np.abs(x).max(), rel_inf(x, ref), f.stats()["iterations"]))
    prev = x
