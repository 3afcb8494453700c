#!/usr/bin/env python3
"""Does the timed region of a bench run start on idle clocks?  The config-4 step timed one by one after the GPU has idled."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth, bench_settings as bs
sizes, w, pos, val = synth.config4(seed=3)
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev); d_val = torch.from_numpy(val).to(dev)
d_out = torch.empty(int(np.prod(sizes)), dtype=torch.float32, device=dev)
f = bs.headline_field(fi, 4, sizes, w, by_field=True)
def step():
    f.clear_points()
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val)
    f.assemble()
    f.solve_cg(None, 0, 1e-5, out=d_out)
for _ in range(4): step()
torch.cuda.synchronize()
for idle in (0.0, 2.0, 10.0, 30.0):
    time.sleep(idle)
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("after %4.1f s idle: steps " % idle + " ".join("%.2f" % t for t in ts), flush=True)
