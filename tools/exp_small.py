#!/usr/bin/env python3
"""Scratch: the apply of a small, saturated lattice (the 64^3 level of config 4's cascade: 1 M points in 2.6e5 cells)
per chunk length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
sizes, w, pos, val = synth.config4(side=256, num_points=1000000, seed=3)
S = int(os.environ.get("SIDE", "64"))
lev = {64: 2, 128: 1}[S]
pos = (pos / float(256 // S)).astype(np.float32)
for zc in os.environ.get("ZCS", "auto,2,3,4,6,8").split(","):
    os.environ.pop("FI_ZC", None)
    if zc != "auto":
        os.environ["FI_ZC"] = zc
    f = fi.LatticeField([S, S, S], dtype="f32")
    f.add_field_constraints(fi.Weights(model_2=0.5 * (8.0 / 16.0) ** (0.5 * lev)))
    f.add_points(1.0, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.assemble()
    f.time_apply(10)
    ms = min(f.time_apply(50) for _ in range(3))
    print("%d^3 zc %s: apply %.1f us, cells %d" % (S, zc, ms * 1e3, f.stats()["num_cells"]), flush=True)
    del f
