#!/usr/bin/env python3
"""Scratch: V-cycle CG from the cascade start (guess None) vs from zero, configs 3 and 5, wall time per solve."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth
for cfg in os.environ.get("CFGS", "3,5").split(","):
    sizes, w, pos, nrm = synth.config3() if cfg == "3" else synth.config5()
    tol = 1e-5 if cfg == "3" else 1e-6
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w); f.set_levels(7, 1e-4); f.set_multigrid(True); f.set_mixed_precision(True)
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
    f.assemble()
    zero = np.zeros(int(np.prod(sizes)), np.float32)
    for name, guess in (("cascade", None), ("zero", zero), ("cascade", None), ("zero", zero)):
        t0 = time.perf_counter()
        x, it, rel = f.solve_cg(guess, 3000, tol)
        t1 = time.perf_counter()
        print("config %s %s: wall %.1f ms (cg %.1f), %d it, coarse %d, true %.2e" % (cfg, name, (t1 - t0) * 1e3, f.stats()["solve_ms"], it,
              f.stats()["coarse_iterations"], f.true_residual()), flush=True)
