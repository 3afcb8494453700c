#!/usr/bin/env python3
"""Scratch experiment: no-data AtA apply for lattice shapes with and without power-of-two plane strides."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi

for sizes in ([512, 512, 512], [512, 504, 512], [528, 512, 512], [520, 520, 496], [512, 512, 256], [640, 640, 320], [384, 384, 384]):
    f = fi.LatticeField(sizes, dtype="f32")
    f.add_field_constraints(fi.Weights())
    f.assemble()
    ms = f.time_apply(30)
    print("%s: apply %.1f us (%.0f GB/s)" % (sizes, ms * 1e3, f.stats()["spmv_bytes"] / ms / 1e6), flush=True)
    del f
