#!/usr/bin/env python3
"""The one experiment driver of the repository (run on the GPU box): solver modes on a BASELINE configuration -- time per
step (clear + add + assemble + solve, inputs in HBM), iterations, verified residual and, for config 4, the FIELD error
against an fp64 solve to 1e-11.
  MODES="name:dtype:levels:mg:mixed:poly:ratio:tol,..."   e.g. "p64:f64:1:0:0:4:30:3e-9,mgmix:f64:3:1:1:0:0:1e-7"
  CFG=4|5 (default 4)  SIDE=<lattice side>  CTOL=<coarse tolerance>  NOREF=1 (skip the reference solve)
Library switches pass through the environment (FI_MG_FULL_SMOOTHER, FI_VERTEX_LEVELS, FI_SERIAL_LEVELS, ...).
(The one-off scripts of rounds 1 and 2 -- tools/exp_*.py, tools/r2_*.sh, cited in profiles/r1_ablation.md and
r2_ablation.md -- are in the history up to commit d1c1349.)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import synth

cfg = os.environ.get("CFG", "4")
if cfg == "5":
    side = int(os.environ.get("SIDE", "512"))
    sizes, w, pos, nrm = synth.config5(side=side, num_points=int(5e6 * (side / 512) ** 2), seed=4)
    val = None
else:
    side = int(os.environ.get("SIDE", "256"))
    sizes, w, pos, val = synth.config4(side=side, num_points=int(1e6 * (side / 256) ** 3), seed=3)
    nrm = None
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev)
d_val = torch.from_numpy(val).to(dev) if val is not None else None
d_nrm = torch.from_numpy(nrm).to(dev) if nrm is not None else None
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
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient if d_nrm is not None else 0.0, w.gradient_kernel, d_pos, d_nrm, None,
                 values=d_val)
    f.assemble()
    return f.solve_cg(None, 0, tol, out=d_out)


if os.environ.get("NOREF") or cfg != "4":
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
