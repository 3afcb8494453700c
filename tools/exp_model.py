#!/usr/bin/env python3
"""Scratch experiment: no-data AtA apply at 512^3 for different model terms (instruction count vs HBM)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi

side = int(os.environ.get("SIDE", "512"))
for name, kw in (("model_2", dict()), ("model_1", dict(model_2=0.0, model_1=1.0)), ("both", dict(model_1=0.5))):
    for dtype in ("f32", "f64"):
        f = fi.LatticeField([side, side, side], dtype=dtype)
        f.add_field_constraints(fi.Weights(**kw))
        f.assemble()
        ms = f.time_apply(30)
        print("%s %s %d^3: apply %.1f us (%.0f GB/s)" % (name, dtype, side, ms * 1e3, f.stats()["spmv_bytes"] / ms / 1e6), flush=True)
        del f
