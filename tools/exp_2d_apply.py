#!/usr/bin/env python3
"""Scratch: 2-D AtA apply at 4096^2 with config-3 data vs without."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth
sizes, w, pos, nrm = synth.config3()
for dtype in ("f32", "f64"):
    for with_data in (False, True):
        f = fi.LatticeField(sizes, dtype=dtype)
        f.add_field_constraints(w)
        if with_data:
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        ms = f.time_apply(30)
        print("4096^2 %s data %d: apply %.1f us (%.0f GB/s)" % (dtype, with_data, ms * 1e3, f.stats()["spmv_bytes"] / ms / 1e6), flush=True)
