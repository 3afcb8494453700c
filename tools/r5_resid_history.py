#!/usr/bin/env python3
"""Residual after k iterations of the headline solver on config 4's three data sets (how far is a 4-iteration solve?)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth, bench_settings as bs

for seed in bs.CONFIG4_SEEDS:
    sizes, w, pos, val = synth.config4(seed=seed)
    f = bs.headline_field(fi, 4, sizes, w)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.assemble()
    row = []
    for k in range(1, 8):
        x, it, rel = f.solve_cg(None, k, 1e-12)
        row.append("%d: %.2e" % (it, f.true_residual()))
    print("seed %d:" % seed, "  ".join(row), flush=True)
