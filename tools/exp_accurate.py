#!/usr/bin/env python3
"""Experiment: which solver mode reaches a 1e-5 FIELD error on config 4 fastest (run on the GPU box).
MODES="name:dtype:levels:mg:mixed:poly:ratio:tol,..."  e.g. "p64:f64:1:0:0:4:30:3e-9"."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth

side = int(os.environ.get("SIDE", "256"))
npts = int(1e6 * (side / 256) ** 3)
sizes, w, pos, val = synth.config4(side=side, num_points=npts, seed=3)
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev)
d_val = torch.from_numpy(val).to(dev)
d_out = torch.empty(int(np.prod(sizes)), dtype=torch.float32, device=dev)


def make(dtype, levels, mg, mixed, poly, ratio, ctol):
    f = fi.LatticeField(sizes, dtype=dtype)
    f.add_field_constraints(w)
    if levels > 0:
        f.set_levels(levels, ctol)
        if mg:
            f.set_multigrid(True)
        if mixed:
            f.set_mixed_precision(True)
    if poly > 1:
        f.set_polynomial(poly, ratio)
    return f


def step(f, tol):
    f.clear_points()
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val)
    f.assemble()
    return f.solve_cg(None, 0, tol, out=d_out)


if os.environ.get("NOREF"):
    x64, xmax = np.zeros(int(np.prod(sizes))), 1.0
else:
    ref = make("f64", 1, False, False, 4, 30.0, 1e-6)
    step(ref, 1e-11)
    x64 = ref.solution_f64()
    print("reference: true residual %.2e" % ref.true_residual(), flush=True)
    del ref
    xmax = np.abs(x64).max()

for mode in os.environ.get("MODES", "p64:f64:1:0:0:4:30:3e-9").split(","):
    name, dtype, levels, mg, mixed, poly, ratio, tol = mode.split(":")
    ctol = float(os.environ.get("CTOL", "1e-5"))
    try:
        f = make(dtype, int(levels), int(mg), int(mixed), int(poly), float(ratio), ctol)
        step(f, float(tol))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out = step(f, float(tol))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / reps
        st = f.stats()
        x = f.solution_f64() if dtype == "f64" else d_out.cpu().numpy().astype(np.float64)
        err = np.abs(x - x64).max() / xmax
        print("%-10s %s tol %-7s: %7.2f ms/step = %.3g pts/s  iters %4d (coarse %5d) applies %4d  asm %.2f solve %.2f ms  "
              "true_rel %.2e  field_err %.2e" % (name, mode, tol, ms, np.prod(sizes) / ms * 1e3, st["iterations"],
                                                 st["coarse_iterations"], st["operator_applies"], st["assemble_ms"],
                                                 st["solve_ms"], f.true_residual(), err), flush=True)
        del f
    except Exception as e:  # noqa: BLE001
        print("%-10s %s FAILED: %s" % (name, mode, e), flush=True)
