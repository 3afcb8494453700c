#!/usr/bin/env python3
"""Scratch experiment: cascade levels vs iterations / solve time (run on the GPU box)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth

side = int(os.environ.get("SIDE", "256"))
cfg = os.environ.get("CFG", "4")
if cfg == "4":
    sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
    nrm = None
else:
    sizes, w, pos, nrm = synth.config5(side=side, num_points=int(5e6 * (side / 512) ** 2), seed=4)
    val = None
tol = float(os.environ.get("TOL", "1e-5"))
for levels in [int(v) for v in os.environ.get("LEVELS", "0,1,2,3,4,5").split(",")]:
    for ctol in [float(v) for v in os.environ.get("CTOLS", "1e-3").split(",")]:
        f = fi.LatticeField(sizes, dtype="f32")
        f.add_field_constraints(w)
        f.set_levels(levels, ctol)
        if os.environ.get("MG"):
            f.set_multigrid(True)
        if nrm is None:
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        else:
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        t0 = time.perf_counter()
        out = f.solve_cg(None, int(os.environ.get("MAXIT", "20000")), tol)
        dt = time.perf_counter() - t0
        st = f.stats()
        print("levels %d ctol %.0e: fine iters %5d coarse iters %6d  solve %.1f ms (gpu %.1f) assemble %.1f ms  rel %.2e true %.2e conv %d"
              % (st["num_levels"] - 1, ctol, st["iterations"], st["coarse_iterations"], dt * 1e3, st["solve_ms"],
                 st["assemble_ms"], st["rel_residual"], f.true_residual(), st["converged"]), flush=True)
        del f
