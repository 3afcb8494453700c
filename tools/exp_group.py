#!/usr/bin/env python3
"""Scratch: Jacobi-PCG over the loop-back group (slabs in one process, the RCCL path's kernels and launch structure
without the wire) vs the single-context solve: what the separate reductions and halo copies cost per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
side = int(os.environ.get("SIDE", "256"))
sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
for nr in (1, 2, 4):
    if nr == 1:
        f = fi.LatticeField(sizes, dtype="f32")
    else:
        f = fi.LatticeGroup(sizes, nr, dtype="f32")
    f.add_field_constraints(w)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.assemble()
    for rep in range(2):
        t0 = time.perf_counter()
        res = f.solve_cg(None, 0, 1e-5)
        t1 = time.perf_counter()
    it = res[1]
    print("slabs %d: %d iterations, %.2f ms wall -> %.1f us per iteration" % (nr, it, (t1 - t0) * 1e3, (t1 - t0) * 1e6 / it), flush=True)
    del f
