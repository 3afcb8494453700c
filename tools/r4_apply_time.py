#!/usr/bin/env python3
"""Isolated time of the full operator apply with fused data cells (fi_time_apply, HIP events around `reps` launches), config 4:
SIDE (256), DTYPES (f32,f64).  FI_HIP_LIB selects a variant build (tools/build_variant.sh)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
side = int(os.environ.get("SIDE", "256"))
nrm = None
if os.environ.get("CFG") == "5":
    sizes, w, pos, nrm = synth.config5(side=side, num_points=int(round(5e6 * (side / 512.0) ** 2)), seed=4)
    val = None
else:
    sizes, w, pos, val = synth.config4(side=side, num_points=int(round(1e6 * (side / 256.0) ** 3)), seed=3)
out = []
for dt in os.environ.get("DTYPES", "f32,f64").split(","):
    f = fi.LatticeField(sizes, dtype=dt)
    f.add_field_constraints(w)
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient if nrm is not None else 0.0, w.gradient_kernel, pos, nrm, None, values=val)
    f.assemble()
    f.time_apply(5)
    ms = f.time_apply(40)
    st = f.stats()
    out.append("%s %.1f us = %.3f of 8 TB/s" % (dt, ms * 1e3, st["spmv_bytes"] / (ms * 1e-3) / 8e12))
    del f
print("%s side %d: %s" % (os.environ.get("NAME", "apply"), side, "; ".join(out)), flush=True)
