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
f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
f.assemble()
try:
    res = f.solve_cg(None, 0, 1e-6)
    st = f.stats()
    print("iterations", st["iterations"], "converged", st["converged"], "estimate", st["field_estimate"], "solve ms", st["solve_ms"])
except Exception as e:
    print("EXC", e)
