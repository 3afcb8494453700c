"""Start guess (FI_START_ONLY) of a linear data field: where does it differ from the field?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi

os.environ["FI_START_ONLY"] = "1"
for sizes in ([24, 20, 28], [25, 21, 29], [9, 8, 7]):
    rng = np.random.default_rng(5)
    n = 4000
    pos = np.stack([rng.uniform(2, s - 3, n) for s in sizes], axis=1).astype(np.float32)
    coef = np.array([0.3, -0.2, 0.15])
    val = (pos.astype(np.float64) @ coef + 1.5).astype(np.float32)
    w = fi.Weights(model_2=0.5, data_gradient=0.0)
    grid = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in sizes[::-1]], indexing="ij")
    exact = (coef[0] * grid[2] + coef[1] * grid[1] + coef[2] * grid[0] + 1.5)
    for linear in (True, False):
        if linear:
            os.environ["FI_LINEAR_START"] = "1"
        else:
            os.environ.pop("FI_LINEAR_START", None)
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.set_levels(1, 1e-11)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        x, it, rel = f.solve_cg(None, 0, 1e-10)
        d = np.abs(f.solution_f64().reshape(exact.shape) - exact)
        k = np.unravel_index(np.argmax(d), d.shape)
        print(sizes, "linear" if linear else "cubic", "it", it, "coarse", f.stats()["coarse_iterations"], "max err", d.max(), "at (z,y,x)", k,
              "interior max", d[3:-3, 3:-3, 3:-3].max(), flush=True)
