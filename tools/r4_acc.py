#!/usr/bin/env python3
"""Round-4 experiment driver for the headline leg (fp64 CG + fp32 V-cycle, config 4): per-step time, iterations, field
error against the ORACLE's golden sample (tests/golden/config4_256_oracle_f64.npz at 256^3).
  env: SIDE (256)  TOL (1e-7)  LEVELS (3)  CTOL (1e-5)  ZERO_GUESS=1 (no coarse-to-fine start)  REPS (5)  CFG (4|5)
Library switches pass through the environment."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth

cfg = int(os.environ.get("CFG", "4"))
side = int(os.environ.get("SIDE", "256" if cfg == 4 else "512"))
tol = float(os.environ.get("TOL", "1e-7" if cfg == 4 else "1e-6"))
levels = int(os.environ.get("LEVELS", "3" if cfg == 4 else "6"))
ctol = float(os.environ.get("CTOL", "1e-5" if cfg == 4 else "1e-4"))
reps = int(os.environ.get("REPS", "5"))
zero = os.environ.get("ZERO_GUESS") == "1"
if cfg == 4:
    sizes, w, pos, val = synth.config4(side=side, num_points=int(round(1e6 * (side / 256.0) ** 3)), seed=3)
    nrm = None
else:
    sizes, w, pos, nrm = synth.config5(side=side, num_points=int(round(5e6 * (side / 512.0) ** 2)), seed=4)
    val = None
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev)
d_val = torch.from_numpy(val).to(dev) if val is not None else None
d_nrm = torch.from_numpy(nrm).to(dev) if nrm is not None else None
n = int(np.prod(sizes))
d_out = torch.empty(n, dtype=torch.float32, device=dev)
d_zero = torch.zeros(n, dtype=torch.float32, device=dev)
f = fi.LatticeField(sizes, dtype="f64")
f.add_field_constraints(w)
f.set_levels(levels, ctol)
f.set_multigrid(True)
f.set_mixed_precision(True)
if os.environ.get("POLY"):
    f.set_polynomial(int(os.environ["POLY"]), float(os.environ.get("POLY_RATIO", "30")))


def step():
    f.clear_points()
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient if d_nrm is not None else 0.0, w.gradient_kernel, d_pos, d_nrm, None,
                 values=d_val)
    f.assemble()
    return f.solve_cg(d_zero if zero else None, 0, tol, out=d_out)


step()
step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    x, it, rel = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / reps
st = f.stats()
err = float("nan")
gpath = os.path.join(ROOT, "tests", "golden", "config4_%d_oracle_f64.npz" % side)
if cfg == 4 and os.path.exists(gpath):
    g = np.load(gpath)
    s = int(g["stride"])
    got = f.solution_f64().reshape(sizes[::-1])[::s, ::s, ::s]
    err = float(np.abs(got - g["sample"]).max() / float(g["field_maxabs"]))
if os.environ.get("PHASES") == "1":   # host-side wall time of each call of a step (synchronised in between)
    acc = [0.0, 0.0, 0.0, 0.0]
    for _ in range(reps):
        torch.cuda.synchronize(); t = [time.perf_counter()]
        f.clear_points(); torch.cuda.synchronize(); t.append(time.perf_counter())
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient if d_nrm is not None else 0.0, w.gradient_kernel, d_pos, d_nrm, None,
                     values=d_val); torch.cuda.synchronize(); t.append(time.perf_counter())
        f.assemble(); torch.cuda.synchronize(); t.append(time.perf_counter())
        f.solve_cg(d_zero if zero else None, 0, tol, out=d_out); torch.cuda.synchronize(); t.append(time.perf_counter())
        for k in range(4):
            acc[k] += (t[k + 1] - t[k]) * 1e3 / reps
    print("phases (host wall, ms): clear %.3f  add_points %.3f  assemble %.3f  solve %.3f  sum %.3f" % (*acc, sum(acc)), flush=True)
print("%s: %.2f ms/step = %.3g pts/s  iters %d (coarse %d)  asm %.2f solve %.2f ms  true_rel %.2e  field_err(oracle) %.2e"
      % (os.environ.get("NAME", "acc"), ms, n / ms * 1e3, st["iterations"], st["coarse_iterations"], st["assemble_ms"],
         st["solve_ms"], f.true_residual(), err), flush=True)
