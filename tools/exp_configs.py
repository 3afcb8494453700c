#!/usr/bin/env python3
"""Scratch experiment: BASELINE configs 2, 3, 5 at full size on one GPU (run on the GPU box)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth

which = os.environ.get("CFGS", "2,3,5").split(",")
for cfg in which:
    if cfg == "2":
        sizes, w, pos, val = synth.config2()
        nrm = None
        tol = 1e-5
    elif cfg == "3":
        sizes, w, pos, nrm = synth.config3()
        val = None
        tol = 1e-5
    else:
        sizes, w, pos, nrm = synth.config5()
        val = None
        tol = 1e-6
    for mode in os.environ.get("MODES", "mg").split(","):
        f = fi.LatticeField(sizes, dtype=os.environ.get("DTYPE", "f32"))
        f.add_field_constraints(w)
        f.set_levels(int(os.environ.get("NLEV", "6")), float(os.environ.get("CTOL", "1e-4")))
        f.set_multigrid(mode in ("mg", "mixed"))
        if mode == "mixed":
            f.set_mixed_precision(True)
        t0 = time.perf_counter()
        if nrm is None:
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        else:
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        t1 = time.perf_counter()
        guess = np.zeros(f.num_unknowns, np.float32) if os.environ.get("ZERO_GUESS") else None   # a guess skips the cascade start
        res = f.solve_cg(guess, int(os.environ.get("MAXIT", "3000")), tol)
        t2 = time.perf_counter()
        st = f.stats()
        print("config %s %s %s: levels %d iters %d (coarse %d) assemble %.1f ms (wall %.0f) solve %.1f ms (wall %.0f) rel %.2e true %.2e conv %d cells %d"
              % (cfg, sizes, mode, st["num_levels"], st["iterations"], st["coarse_iterations"], st["assemble_ms"],
                 (t1 - t0) * 1e3, st["solve_ms"], (t2 - t1) * 1e3, st["rel_residual"], f.true_residual(), st["converged"],
                 st["num_cells"]), flush=True)
        if os.environ.get("TIME_APPLY"):
            ms = f.time_apply(50)
            print("   apply %.1f us, algorithmic %.1f MB -> %.0f GB/s" % (ms * 1e3, f.stats()["spmv_bytes"] / 1e6,
                  f.stats()["spmv_bytes"] / ms / 1e6), flush=True)
        del f
