#!/usr/bin/env python3
"""Scratch experiment: config-4 fused apply (isolated) and a CG solve per library variant."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import field_interpolation_amd as fi
from field_interpolation_amd import synth
side = int(os.environ.get("SIDE", "256")); dtype = os.environ.get("DTYPE", "f32")
sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
f = fi.LatticeField(sizes, dtype=dtype)
f.add_field_constraints(w)
f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
f.assemble()
f.time_apply(10)
ms = min(f.time_apply(50) for _ in range(3))
if os.environ.get("NOSOLVE"):
    it, rel = 0, 0.0
else:
    x, it, rel = f.solve_cg(None, 500, 1e-5)
st = f.stats()
print("%%-6s side %%d %%s: apply isolated %%.1f us; CG %%d it rel %%.2e, apply in CG %%.1f us, solve %%.2f ms" %% (
    os.environ.get("VARIANT"), side, dtype, ms * 1e3, it, rel, st["spmv_ms_avg"] * 1e3, st.get("solve_ms", 0.0)), flush=True)
''' % ROOT
for variant in os.environ.get("VARIANTS", "base").split(","):
    env = dict(os.environ, VARIANT=variant)
    if variant != "base":
        env["FI_HIP_LIB"] = os.path.join(ROOT, "exp_libs", "libfi_%s.so" % variant)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
