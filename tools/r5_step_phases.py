#!/usr/bin/env python3
"""Wall time of each API call of a config-4 step (synchronised after every call), against the un-synchronised step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth, bench_settings as bs
sizes, w, pos, val = synth.config4(seed=3)
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev); d_val = torch.from_numpy(val).to(dev)
d_out = torch.empty(int(np.prod(sizes)), dtype=torch.float32, device=dev)
f = bs.headline_field(fi, 4, sizes, w)
def calls():
    return [("clear_points", lambda: f.clear_points()),
            ("add_points", lambda: f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val)),
            ("assemble", lambda: f.assemble()),
            ("solve_cg", lambda: f.solve_cg(None, 0, 3e-7, out=d_out))]
for _ in range(5):
    for n, c in calls(): c()
torch.cuda.synchronize()
acc = {}
for _ in range(20):
    for n, c in calls():
        torch.cuda.synchronize(); t0 = time.perf_counter(); c(); th = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
        a = acc.setdefault(n, [0.0, 0.0]); a[0] += th - t0; a[1] += t1 - t0
for n, a in acc.items():
    print("%-13s host call %.3f ms, until the GPU is idle %.3f ms" % (n, a[0] * 50, a[1] * 50))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    for n, c in calls(): c()
torch.cuda.synchronize()
print("un-synchronised step: %.3f ms" % ((time.perf_counter() - t0) * 50))
st = f.stats()
print({k: st[k] for k in st if k.endswith("_ms")})
