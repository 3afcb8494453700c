#!/usr/bin/env python3
"""Isolated time of the operator apply (fi_time_apply) at SIDE (512): no data (the plain variant: the stencil's own ceiling),
config 4's value data, config 5's oriented points; DTYPES (f64,f32).  FI_HIP_LIB selects a variant build."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth

side = int(os.environ.get("SIDE", "512"))
name = os.environ.get("NAME", "apply")
for what in os.environ.get("DATA", "none,c4,c5").split(","):
    nrm = val = None
    if what == "c5":
        sizes, w, pos, nrm = synth.config5(side=side, num_points=int(round(5e6 * (side / 512.0) ** 2)), seed=4)
    else:
        # (none: one point -- the context needs data rows to assemble; a single cell does not change the time)
        sizes, w, pos, val = synth.config4(side=side, num_points=1 if what == "none" else int(round(1e6 * (side / 256.0) ** 3)), seed=3)
    for dt in os.environ.get("DTYPES", "f64,f32").split(","):
        f = fi.LatticeField(sizes, dtype=dt)
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient if nrm is not None else 0.0, w.gradient_kernel, pos, nrm, None,
                     values=val)
        f.assemble()
        f.time_apply(5)
        ms = min(f.time_apply(30) for _ in range(3))
        st = f.stats()
        print("%s side %d data %-4s %s: %.1f us, %.0f MB = %.3f of 8 TB/s (cells %d)" % (
            name, side, what, dt, ms * 1e3, st["spmv_bytes"] / 1e6, st["spmv_bytes"] / (ms * 1e-3) / 8e12, st["num_cells"]), flush=True)
        del f
