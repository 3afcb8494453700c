#!/usr/bin/env python3
"""Scratch: AtA apply on SDF-type data (oriented points on a sphere: few, dense layers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth
for side, n in ((256, 1250000), (512, 5000000)):
    for dtype in os.environ.get("DTYPES", "f32,f64").split(","):
        sizes, w, pos, nrm = synth.config5(side=side, num_points=n)
        f = fi.LatticeField(sizes, dtype=dtype)
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        f.time_apply(5)
        ms = min(f.time_apply(20) for _ in range(2))
        st = f.stats()
        print("%d^3 %s: cells %d apply %.1f us (%.0f GB/s algorithmic)" % (side, dtype, st["num_cells"], ms * 1e3,
              st["spmv_bytes"] / ms / 1e6), flush=True)
        del f
