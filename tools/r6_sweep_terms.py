#!/usr/bin/env python3
"""Terms x interval ratio of the V-cycle's polynomial smoother on config 4 under the FIELD stop rule (bench settings), over the three
data sets in turn like bench.py: iterations per seed, ms per assemble + solve, field error against each seed's oracle golden."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth, bench_settings as bs

G = os.path.join(ROOT, "tests", "golden")
sets = []
dev = torch.device("cuda", 0)
for sd in bs.CONFIG4_SEEDS:
    sizes, w, pos, val = synth.config4(seed=sd)
    g = np.load(os.path.join(G, "config4_256_oracle_f64.npz" if sd == 3 else "config4_256_seed%d_oracle_f64.npz" % sd))
    sets.append((torch.from_numpy(pos).to(dev), torch.from_numpy(val).to(dev), g))
n = int(np.prod(sizes))
d_out = torch.empty(n, dtype=torch.float32, device=dev)
for terms, ratio in [(5, 30), (5, 40), (5, 60), (6, 30), (6, 40), (6, 60), (6, 90), (7, 60), (7, 90), (4, 20), (4, 30)]:
    f = bs.headline_field(fi, 4, sizes, w, by_field=True)
    f.set_mg_smoother(True, None, terms, float(ratio))
    def step(k):
        p, v, _ = sets[k % 3]
        f.clear_points()
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, p, None, None, values=v)
        f.assemble()
        return f.solve_cg(None, 0, 1e-5, out=d_out)
    for k in range(6): step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(12): step(k)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 12
    its, errs = [], []
    for k in range(3):
        x, it, rel = step(k)
        x64 = f.solution_f64()
        g = sets[k][2]
        grid = np.asarray(x64).reshape(sizes[::-1]); s = int(g["stride"])
        errs.append(float(np.abs(grid[::s, ::s, ::s] - g["sample"]).max() / float(g["field_maxabs"])))
        its.append(it)
    print("terms %d ratio %g: iterations %s, %.2f ms per step, field errors %s" % (terms, ratio, "/".join(str(i) for i in its), ms,
                                                                               " ".join("%.1e" % e for e in errs)), flush=True)
    del f
