import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import field_interpolation_amd as fi
from util import sphere_points
for sizes in ([32, 32, 96], [32,32,64]):
    rng = np.random.default_rng(6)
    pos, nrm = sphere_points(rng, sizes, 500)
    w = fi.Weights(model_1=0.05)
    for lev in (0, 1, 2):
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.set_levels(lev)
        f.assemble()
        for tol in (1e-5, 1e-9):
            x, it, r = f.solve_cg(None, 0, tol)
            print(sizes, "levels", lev, "tol", tol, "iterations", it, "coarse", f.stats()["coarse_iterations"], flush=True)
