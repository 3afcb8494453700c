#!/usr/bin/env python3
"""Scratch: 20 isolated applies with config-4 data and 20 without (256^3 fp32) -- the target of a rocprofv3 --pmc pass."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth
side = int(os.environ.get("SIDE", "256"))
sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
f = fi.LatticeField(sizes, dtype="f32")
f.add_field_constraints(w)
f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
f.assemble()
print("fused %.1f us" % (f.time_apply(20) * 1e3))
g = fi.LatticeField(sizes, dtype="f32")
g.add_field_constraints(w)
g.assemble()
print("plain %.1f us" % (g.time_apply(20) * 1e3))
