#!/usr/bin/env python3
"""Scratch: config 3 to 1e-8, pure fp64 and mixed, fused and unfused smoother: iteration counts."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
sizes, w, pos, nrm = synth.config3()
for mixed in (False, True):
    f = fi.sdf_from_points(sizes, w, pos, nrm, dtype="f64")
    f.set_levels(7, 1e-4)
    f.set_multigrid(True)
    f.set_mixed_precision(mixed)
    f.assemble()
    x, it, rel = f.solve_cg(None, 3000, 1e-8)
    print(os.environ.get("FI_NO_FUSED_SMOOTHER", "fused"), "mixed" if mixed else "fp64", it, rel, f.true_residual(), f.stats()["solve_ms"], flush=True)
    del f
''' % ROOT
for env in ({}, {"FI_NO_FUSED_SMOOTHER": "1"}):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), check=False)
