#!/usr/bin/env python3
"""Scratch: config 3, how far from the converged field is the iterate at a given relative residual?"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth

sizes, w, pos, nrm = synth.config3()
f = fi.sdf_from_points(sizes, w, pos, nrm, dtype="f64")
f.set_levels(7, 1e-4)
f.set_multigrid(True)
f.set_mixed_precision(os.environ.get("MIXED", "1") == "1")
f.assemble()
guess = None
for tol in (1e-5, 1e-6, 1e-7, 1e-8, 1e-9):
    x, it, rel = f.solve_cg(guess, 5000, tol)
    field = x.reshape(sizes[1], sizes[0])
    out = []
    for sgn in (+1.0, -1.0):
        q = pos[::50] + sgn * 30.0 * nrm[::50]
        ok = (q >= 0).all(1) & (q <= sizes[0] - 2).all(1)
        v = field[np.round(q[ok, 1]).astype(int), np.round(q[ok, 0]).astype(int)]
        out.append((float(np.mean(np.sign(v) == sgn)), float(np.median(v))))
    print("tol %.0e: %d iterations (%.0f ms), +30: sign ok %.3f median %.2f; -30: sign ok %.3f median %.2f; corner %.1f"
          % (tol, it, f.stats()["solve_ms"], out[0][0], out[0][1], out[1][0], out[1][1], field[10, 10]), flush=True)
