#!/usr/bin/env python3
"""Scratch experiment: plain (no data) AtA apply at 512^3 / 256^3 per library variant and chunk length."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import field_interpolation_amd as fi
side = int(os.environ["SIDE"]); dtype = os.environ["DTYPE"]
f = fi.LatticeField([side, side, side], dtype=dtype)
f.add_field_constraints(fi.Weights())
f.assemble()
f.time_apply(10)
ms = min(f.time_apply(30) for _ in range(3))
st = f.stats()
print("%%-6s side %%d %%s zc %%-4s: apply %%.1f us  (%%.0f GB/s algorithmic)" %% (os.environ.get("VARIANT"), side, dtype, os.environ.get("FI_ZC", "auto"), ms * 1e3, st["spmv_bytes"] / ms / 1e6), flush=True)
''' % ROOT
for variant in os.environ.get("VARIANTS", "base,nt").split(","):
    for side, dtype in [(512, "f32"), (512, "f64"), (256, "f32")]:
        for zc in os.environ.get("ZCS", "auto,64,128").split(","):
            env = dict(os.environ, SIDE=str(side), DTYPE=dtype, VARIANT=variant)
            env.pop("FI_ZC", None)
            if zc != "auto":
                env["FI_ZC"] = zc
            if variant != "base":
                env["FI_HIP_LIB"] = os.path.join(ROOT, "exp_libs", "libfi_%s.so" % variant)
            subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
