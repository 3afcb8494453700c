#!/usr/bin/env python3
"""Scratch: where the 2-D fp64 apply with config-3 data spends its time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
sizes, w, pos, nrm = synth.config3()
for dtype in ("f64", "f32"):
    for frac in (0.0, 0.001, 0.01, 0.1, 1.0):
        n = int(len(pos) * frac)
        f = fi.LatticeField(sizes, dtype=dtype)
        f.add_field_constraints(w)
        if n:
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos[:n], nrm[:n], None)
        f.assemble()
        ms = f.time_apply(30)
        print("%s points %7d cells %7d: apply %.1f us" % (dtype, n, f.stats()["num_cells"], ms * 1e3), flush=True)
