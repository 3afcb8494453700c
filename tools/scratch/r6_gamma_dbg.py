import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth, bench_settings as bs
side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sizes, w, pos, nrm = synth.config5(side=side, num_points=int(5e6 * (side / 512.0) ** 2), seed=4)
f = fi.LatticeField(sizes, dtype="f64")
f.add_field_constraints(w)
bs.configure(f, int(sys.argv[2]) if len(sys.argv) > 2 else 4, 1e-2, mixed=(os.environ.get("NOMIX") is None), by_field=True)
if os.environ.get("KC"):
    f.set_kcycle(int(os.environ["KC"]))
f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
f.assemble()
f.solve_cg(None, 0, 1e-13)
ref = f.solution_f64().copy()
f.set_field_tolerance(1e-5)
for k in range(3):
    try:
        f.clear_points(); f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None); f.assemble()
        res = f.solve_cg(None, 0, 1e-6)
        st = f.stats()
        err = float(np.abs(f.solution_f64() - ref).max() / np.abs(ref).max())
        print("solve", k, "iterations", st["iterations"], "converged", st["converged"], "estimate %.2e" % st["field_estimate"], "solve ms %.1f" % st["solve_ms"], "error %.2e" % err, flush=True)
    except Exception as e:
        print("EXC", e)
