#!/usr/bin/env python3
"""Scratch: isolated launches of the polynomial's steps (timing build, FI_TIME_STEP) under FI_DBG load-skipping modes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import field_interpolation_amd as fi
from field_interpolation_amd import synth
side = int(os.environ.get("SIDE", "256"))
sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
f = fi.LatticeField(sizes, dtype="f32")
f.add_field_constraints(w)
f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
f.assemble()
f.time_apply(10)
ms = min(f.time_apply(50) for _ in range(3))
print("step %%s dbg %%s: %%.1f us" %% (os.environ.get("FI_TIME_STEP", "apply"), os.environ.get("FI_DBG", "0"), ms * 1e3), flush=True)
''' % ROOT
for step in ("0", "1", "2", "3"):
    for dbg in (("0", "64", "128", "192", "1", "193") if step == "1" else ("0", "1")):
        env = dict(os.environ, FI_TIME_STEP=step, FI_DBG=dbg, FI_HIP_LIB=os.path.join(ROOT, "exp_libs", "libfi_tbs.so"))
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
