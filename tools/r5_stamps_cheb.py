#!/usr/bin/env python3
"""In-kernel time stamps (FI_STAMPS build, tools/build_variant.sh) of one workgroup of the polynomial's step on config 4's
finest level: where a plane step's cycles go.  The build selects the launch that reports (FI_STAMPS_SEL)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wgsel = int(os.environ.get("WG", "300"))
os.environ["FI_DBG"] = str(wgsel << 8)
import field_interpolation_amd as fi
from field_interpolation_amd import synth, _capi, bench_settings as bs

sizes, w, pos, val = synth.config4(seed=3)
f = bs.headline_field(fi, 4, sizes, w)
f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
f.assemble()
x, it, rel = f.solve_cg(None, 0, 3e-7)
buf = (C.c_ulonglong * (64 * 8 * 4))()
L = _capi.lib()
assert L.fi_debug_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(4, 64, 8).astype(np.int64)
print("iterations %d" % it)
names = {0: "entry", 1: "wrote plane (LDS), halo + own loads issued next", 2: "barrier out", 5: "x/y/z stencil starts", 6: "stencil done, epilogue starts"}
keys = [0, 1, 2, 5, 6]
for wave in range(4):
    t = a[wave]
    steps = [s for s in range(62) if t[s, 0] > 0 and t[s, 6] > 0 and t[s + 1, 0] > 0]
    if len(steps) < 3:
        continue
    steps = steps[1:]
    cols = np.stack([t[steps, k] for k in keys] + [t[[s + 1 for s in steps], 0]], axis=1)
    seg = np.diff(cols, axis=1)
    print("wave %d: %d steps, s_memtime ticks (100 MHz: 10 ns each) per segment:" % (wave, len(steps)))
    for i, k in enumerate(keys):
        print("   after '%-50s' mean %7.1f  min %5d max %5d" % (names[k] + "'", seg[:, i].mean(), seg[:, i].min(), seg[:, i].max()))
    print("   step total mean %.1f ticks" % (cols[:, -1] - cols[:, 0]).mean())
