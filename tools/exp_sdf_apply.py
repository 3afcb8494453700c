#!/usr/bin/env python3
"""Scratch: AtA apply on SDF-type data (oriented points on a sphere: few, dense layers) vs no data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth
for side, n in ((256, 1250000), (512, 5000000)):
    for dtype in ("f32", "f64"):
        sizes, w, pos, nrm = synth.config5(side=side, num_points=n)
        for with_data in (False, True):
            f = fi.LatticeField(sizes, dtype=dtype)
            f.add_field_constraints(w)
            if with_data:
                f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
            f.assemble()
            ms = f.time_apply(20)
            st = f.stats()
            print("%d^3 %s data %d: cells %d apply %.1f us (%.0f GB/s algorithmic)" % (side, dtype, with_data, st["num_cells"], ms * 1e3,
                  st["spmv_bytes"] / ms / 1e6), flush=True)
            del f
