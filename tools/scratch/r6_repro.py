#!/usr/bin/env python3
"""Is a solve with max_iterations = k reproducible, and does it depend on the solve before it?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import field_interpolation_amd as fi
from util import rel_inf, sphere_points
rng = np.random.default_rng(5)
sizes = [344, 141]
w = fi.Weights(model_2=0.658, model_1=0.225)
pos, nrm = sphere_points(rng, sizes, 962, noise=0.5)
val = rng.normal(size=962).astype(np.float32)
def make():
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.set_levels(3, 1e-3); f.set_multigrid(True); f.set_mixed_precision(True)
    f.assemble()
    return f
f = make()
f.solve_cg(None, 4000, 1e-13); ref = f.solution_f64().copy()
def run(k):
    try: f.solve_cg(None, k, 1e-13)
    except Exception: pass
    st = f.stats()
    return f.solution_f64().copy(), st["iterations"], st["coarse_iterations"]
for seq in ([7, 7, 7], [3, 7], [12, 7], [4000, 7], [7]):
    outs = [run(k) for k in seq]
    x, it, cit = outs[-1]
    print("sequence %s: last solve %d iterations (coarse %d), error %.6e, max |x| %.6e, checksum %.17g" % (seq, it, cit, rel_inf(x, ref), np.abs(x).max(), float(x.sum())))
g = make()
try: g.solve_cg(None, 7, 1e-13)
except Exception: pass
x = g.solution_f64(); print("fresh context: error %.6e checksum %.17g coarse %d" % (rel_inf(x, ref), float(x.sum()), g.stats()["coarse_iterations"]))
