#!/usr/bin/env python3
"""fi_assemble of ONE level alone (no hierarchy, no solve): wall time per call, for kernel traces without other levels'
kernels sharing the GPU.  env: SIDE (256)  DT (f64|f32)  POINTS (1e6 scaled with the side)  REPS (5)"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth

side = int(os.environ.get("SIDE", "256"))
dt = os.environ.get("DT", "f64")
reps = int(os.environ.get("REPS", "5"))
npts = int(float(os.environ.get("POINTS", str(1e6 * (side / 256.0) ** 3))))
sizes, w, pos, val = synth.config4(side=side, num_points=npts, seed=3)
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev)
d_val = torch.from_numpy(val).to(dev)
f = fi.LatticeField(sizes, dtype=dt)
f.add_field_constraints(w)


def step():
    f.clear_points()
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val)
    f.assemble()


for _ in range(3):
    step()
    torch.cuda.synchronize()
    time.sleep(0.01)
t = 0.0
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t += time.perf_counter() - t0
    time.sleep(0.01)
print("assemble alone: side %d %s %d points: %.3f ms per call (cells %d)" % (side, dt, npts, t * 1e3 / reps, f.stats()["num_cells"]), flush=True)
