#!/usr/bin/env python3
"""Scratch: cold vs warm assemble + solve of configs 2 and 3 (levels, V-cycle, mixed precision)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth
for cfg in ("2", "3"):
    if cfg == "2":
        sizes, w, pos, val = synth.config2(); nrm = None
    else:
        sizes, w, pos, nrm = synth.config3(); val = None
    f = fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w); f.set_levels(7, 1e-4); f.set_multigrid(True); f.set_mixed_precision(True)
    for rep in range(3):
        f.clear_points()
        t0 = time.perf_counter()
        if nrm is None:
            f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        else:
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, None)
        f.assemble()
        t1 = time.perf_counter()
        x, it, rel = f.solve_cg(None, 3000, 1e-5)
        t2 = time.perf_counter()
        print("config %s rep %d: add+assemble wall %.1f ms (gpu %.1f), solve wall %.1f ms (gpu %.1f), %d it" % (
            cfg, rep, (t1 - t0) * 1e3, f.stats()["assemble_ms"], (t2 - t1) * 1e3, f.stats()["solve_ms"], it), flush=True)
