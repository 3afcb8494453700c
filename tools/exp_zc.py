#!/usr/bin/env python3
"""Scratch experiment: AtA-apply launch time vs z-chunk length (run on the GPU box)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth

for side in [int(v) for v in os.environ.get("SIDES", "256,512").split(",")]:
    for dtype in os.environ.get("DTYPES", "f32").split(","):
        for zc in os.environ.get("ZCS", "8,16,32,64").split(","):
            os.environ["FI_ZC"] = zc
            f = fi.LatticeField([side, side, side], dtype=dtype)
            f.add_field_constraints(fi.Weights())
            f.assemble()
            ms = f.time_apply(30)
            st = f.stats()
            print("side %d %s zc %s: apply %.1f us  (%.0f GB/s algorithmic)" % (side, dtype, zc, ms * 1e3,
                                                                            st["spmv_bytes"] / ms / 1e6), flush=True)
            del f
