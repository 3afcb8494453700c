#!/usr/bin/env python3
"""Scratch: in-kernel time stamps of one workgroup of the fused apply (FI_STAMPS build): where a plane step's cycles go."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import field_interpolation_amd as fi
from field_interpolation_amd import synth, _capi

side = int(os.environ.get("SIDE", "256"))
wgsel = int(os.environ.get("WG", "300"))
os.environ["FI_DBG"] = str(wgsel << 8)
nodata = os.environ.get("NODATA")
sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
f = fi.LatticeField(sizes, dtype=os.environ.get("DT", "f32"))
f.add_field_constraints(w)
if not nodata:
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
f.assemble()
f.time_apply(5)
ms = f.time_apply(20)
buf = (C.c_ulonglong * (64 * 8 * 4))()
L = _capi.lib()
assert L.fi_debug_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(4, 64, 8).astype(np.int64)
print("apply %.1f us" % (ms * 1e3))
names = ["entry", "wrote plane", "barrier out", "epilogue done", "scatter done", "prefetch issued", "stencil done"]
for wave in range(4):
    t = a[wave]
    steps = [s for s in range(63) if t[s, 0] > 0 and t[s, 6] > 0 and t[s + 1, 0] > 0]
    if len(steps) < 3:
        continue
    steps = steps[1:]
    base = t[steps, 0]
    seg = np.diff(np.concatenate([t[steps, :7], t[[s + 1 for s in steps], 0:1]], axis=1), axis=1)
    print("wave %d: %d steps, mean cycles (s_memtime ticks) per segment:" % (wave, len(steps)))
    for k in range(7):
        nm = names[k] + " -> " + (names[k + 1] if k < 6 else "next entry")
        print("   %-36s mean %8.1f  min %6d max %6d" % (nm, seg[:, k].mean(), seg[:, k].min(), seg[:, k].max()))
    print("   step total mean %.1f (clock: s_memtime)" % (np.diff(t[steps + [steps[-1] + 1], 0]).mean()))
