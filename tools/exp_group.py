#!/usr/bin/env python3
"""Scratch: the solvers over the loop-back group (slabs in one process on one GPU: the RCCL path's kernels and launch
structure without the wire) vs the single-context solve: what the separate reductions and halo copies cost per operator
application.  POLY=0 runs Jacobi-PCG (one application per iteration), POLY=4 the polynomial preconditioner."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
side = int(os.environ.get("SIDE", "256"))
sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
for poly in [int(v) for v in os.environ.get("POLY", "0,4").split(",")]:
    for nr in (1, 2, 4, 8):
        if nr == 1:
            f = fi.LatticeField(sizes, dtype="f32")
        else:
            f = fi.LatticeGroup(sizes, nr, dtype="f32")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        if poly:
            f.set_polynomial(poly, 30.0)
        for rep in range(2):
            t0 = time.perf_counter()
            res = f.solve_cg(None, 0, 1e-5)
            t1 = time.perf_counter()
        it = res[1]
        st = f.stats()
        print("poly %d slabs %d: %d iterations (%d operator applications), solve %.2f ms -> %.1f us per application" % (
            poly, nr, it, st["operator_applies"], st["solve_ms"], st["solve_ms"] * 1e3 / max(st["operator_applies"], 1)), flush=True)
        del f
