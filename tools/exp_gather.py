#!/usr/bin/env python3
"""Scratch: Atb / diag by the gather kernel vs the 2^D scatter launches (FI_NO_GATHER): must agree bit for bit."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, hashlib, time
sys.path.insert(0, %r)
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
for side, dtype, grad in ((96, "f32", False), (64, "f64", True), (256, "f32", False)):
    sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
    f = fi.LatticeField(sizes, dtype=dtype)
    f.add_field_constraints(w)
    nrm = None
    if grad:
        nrm = np.random.default_rng(1).normal(size=pos.shape).astype(np.float32)
    f.add_points(w.data_pos, w.value_kernel, 1.0 if grad else 0.0, w.gradient_kernel, pos, nrm, None, values=val)
    t0 = time.perf_counter(); f.assemble(); t1 = time.perf_counter()
    f.clear_points() if hasattr(f, "clear_points") else None
    print(side, dtype, hashlib.md5(np.ascontiguousarray(f.Atb()).tobytes()).hexdigest(), hashlib.md5(np.ascontiguousarray(f.diag()).tobytes()).hexdigest(), "assemble %%.2f ms" %% ((t1 - t0) * 1e3), flush=True)
''' % ROOT
for flag in ("", "1"):
    env = dict(os.environ)
    if flag:
        env["FI_NO_GATHER"] = "1"
    print("FI_NO_GATHER=%s" % flag, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
