#!/usr/bin/env python3
"""Terms x interval ratio of the V-cycle's polynomial smoother on config 4 with the bench's settings and stop residual: iterations,
ms per assemble + solve (seed 3), field error against the oracle golden."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth, bench_settings as bs

g = np.load(os.path.join(ROOT, "tests", "golden", "config4_256_oracle_f64.npz"))
sizes, w, pos, val = synth.config4(seed=3)
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev); d_val = torch.from_numpy(val).to(dev)
n = int(np.prod(sizes))
d_out = torch.empty(n, dtype=torch.float32, device=dev)
for terms, ratio in [(5, 30), (6, 30), (6, 40), (6, 60), (7, 40), (7, 60), (8, 60), (4, 20)]:
    f = bs.headline_field(fi, 4, sizes, w)
    f.set_mg_smoother(True, None, terms, float(ratio))
    def step():
        f.clear_points()
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val)
        f.assemble()
        return f.solve_cg(None, 0, 3e-7, out=d_out)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): x, it, rel = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
    x64 = f.solution_f64()
    grid = np.asarray(x64).reshape(sizes[::-1]); s = int(g["stride"])
    err = float(np.abs(grid[::s, ::s, ::s] - g["sample"]).max() / float(g["field_maxabs"]))
    print("terms %d ratio %g: %d iterations, %.2f ms, true residual %.2e, field error %.2e" % (terms, ratio, it, ms, f.true_residual(), err), flush=True)
    del f
