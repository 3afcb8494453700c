#!/usr/bin/env python3
"""Scratch experiment: AtA-apply launch time for several data densities (run on the GPU box)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth

side = int(os.environ.get("SIDE", "256"))
dtype = os.environ.get("DTYPE", "f32")
sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
for label, n in (("no points", 0), ("1 point", 1), ("1% points", len(pos) // 100), ("10% points", len(pos) // 10),
                 ("all points", len(pos))):
    f = fi.LatticeField(sizes, dtype=dtype)
    f.add_field_constraints(w)
    if n:
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos[:n], None, None, values=val[:n])
    f.assemble()
    ms = f.time_apply(50)
    st = f.stats()
    print("%-12s cells %8d  apply %.1f us  (%.0f GB/s algorithmic)" % (label, st["num_cells"], ms * 1e3,
                                                                   st["spmv_bytes"] / ms / 1e6), flush=True)
