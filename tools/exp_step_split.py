#!/usr/bin/env python3
"""Scratch: wall time of the four calls of a bench step (config 4, 256^3), synchronised after each."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth
dev = torch.device("cuda", 0)
sizes, w, pos, val = synth.config4(side=256, num_points=1_000_000, seed=3)
f = fi.LatticeField(sizes, dtype="f32")
d_pos = torch.from_numpy(pos).to(dev); d_val = torch.from_numpy(val).to(dev)
d_out = torch.empty(f.num_owned, dtype=torch.float32, device=dev)
f.add_field_constraints(w)
f.set_levels(int(os.environ.get("LEVELS", "2")), 1e-5)
def sync():
    torch.cuda.synchronize()
acc = np.zeros(5)
for it in range(7):
    sync(); t = [time.perf_counter()]
    f.clear_points(); sync(); t.append(time.perf_counter())
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val); sync(); t.append(time.perf_counter())
    f.assemble(); sync(); t.append(time.perf_counter())
    out = f.solve_cg(None, 0, 1e-5, out=d_out); sync(); t.append(time.perf_counter())
    st = f.stats(); t.append(time.perf_counter())
    if it >= 2:
        acc += np.diff(t) * 1e3
    last = st
print("clear %.3f add_points %.3f assemble %.3f (stat %.3f) solve %.3f (stat %.3f) stats %.3f ms" % (
    acc[0] / 5, acc[1] / 5, acc[2] / 5, last["assemble_ms"], acc[3] / 5, last["solve_ms"], acc[4] / 5))
print({k: v for k, v in last.items() if "ms" in k or "iter" in k})
