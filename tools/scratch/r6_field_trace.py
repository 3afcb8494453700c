#!/usr/bin/env python3
"""Per-iteration history of a stress_field_rule case: residual, step, A-norm step, the rule's estimate -- and the TRUE error of
every iterate (the iterates are recomputed with max_iterations = k).  usage: r6_field_trace.py <seed> [every]"""
import os, sys, re, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import field_interpolation_amd as fi
from util import rel_inf, sphere_points
seed = int(sys.argv[1]); every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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
tol = float(rng.choice([1e-4, 1e-5, 1e-6]))
f = fi.LatticeField(sizes, dtype="f64")
f.add_field_constraints(w)
f.add_points(w.data_pos, w.value_kernel, w.data_gradient if sdf else 0.0, w.gradient_kernel, pos, nrm if sdf else None, None, values=val)
f.set_levels(levels, 1e-3); f.set_multigrid(True)
if mixed: f.set_mixed_precision(True)
f.assemble()
res = f.solve_cg(None, 4000, 1e-13)
ref = f.solution_f64().copy(); itref = res[1]
print("seed", seed, "sizes", sizes, "tol", tol, "reference iterations", itref, flush=True)
# every truncated solve prints its own history (FI_FIELD_TRACE, stderr) and then its true error: one consistent record per k
os.environ["FI_FIELD_TRACE"] = "1"
f.set_field_tolerance(1e-30)     # never met: the whole history
for k in range(2, itref + 1, every):
    try:
        f.solve_cg(None, k, 1e-5)
    except Exception:
        pass
    sys.stderr.flush()
    print("RUN %d true %.6e" % (k, rel_inf(f.solution_f64(), ref)), flush=True)
