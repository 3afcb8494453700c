#!/usr/bin/env python3
"""Scratch: per-kernel time of a fine-level CG iteration (config 4, 256^3, plain Jacobi-PCG, 217 iterations) from the
solve's wall time and the apply samples, per library variant."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import field_interpolation_amd as fi
from field_interpolation_amd import synth
sizes, w, pos, val = synth.config4(side=256, num_points=1000000, seed=3)
f = fi.LatticeField(sizes, dtype="f32")
f.add_field_constraints(w)
f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
f.assemble()
f.solve_cg(None, 0, 1e-5)
x, it, rel = f.solve_cg(None, 0, 1e-5)
st = f.stats()
print("%%-6s CG %%d it rel %%.2e: solve %%.2f ms = %%.1f us per iteration, apply in CG %%.1f us -> vector kernels %%.1f us" %% (
    os.environ.get("VARIANT"), it, rel, st["solve_ms"], st["solve_ms"] * 1e3 / it, st["spmv_ms_avg"] * 1e3,
    st["solve_ms"] * 1e3 / it - st["spmv_ms_avg"] * 1e3), flush=True)
''' % ROOT
for variant in os.environ.get("VARIANTS", "base").split(","):
    env = dict(os.environ, VARIANT=variant)
    if variant != "base":
        env["FI_HIP_LIB"] = os.path.join(ROOT, "exp_libs", "libfi_%s.so" % variant)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
