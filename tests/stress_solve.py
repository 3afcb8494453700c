#!/usr/bin/env python3
"""Randomised solver sweep (run on the GPU box): small well-posed problems, every solver mode (plain Jacobi-PCG,
coarse-to-fine cascade, V-cycle preconditioned CG, mixed precision, polynomial preconditioner, slabs through the loop-back group), the
solution against the oracle's float64 direct solve of the same rows.  usage: stress_solve.py [cases] [first seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import field_interpolation_amd as fi  # noqa: E402
from oracle import fi_oracle as oracle  # noqa: E402  (test infrastructure: this file lives under tests/)
from util import build_pair, rel_inf, sphere_points  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    D = int(rng.integers(2, 4))
    sizes = [int(rng.integers(16, 57)) for _ in range(2)] if D == 2 else [int(rng.integers(16, 29)) for _ in range(3)]
    if rng.random() < 0.7:
        sizes[0] = (sizes[0] // 4) * 4
    kw = dict(model_2=float(rng.uniform(0.2, 1.0)))
    if rng.random() < 0.4:
        kw["model_1"] = float(rng.uniform(0.02, 0.5))
    if rng.random() < 0.2:
        kw["model_0"] = float(rng.uniform(0.01, 0.1))
    sdf = rng.random() < 0.6
    gk = int(rng.integers(0, 3)) if sdf else 1
    w = fi.Weights(gradient_kernel=fi.GradientKernel(gk), **kw)
    n = int(np.prod(sizes))
    npts = int(rng.integers(60, 900))
    pos, nrm = sphere_points(rng, sizes, npts, noise=float(rng.uniform(0.1, 1.0)))
    val = None if sdf else rng.normal(size=npts).astype(np.float32)
    fo, _ = build_pair(oracle, fi, sizes, w, pos, nrm if sdf else None, None, val, dtype="f64")
    x64 = fo.solve_exact_f64()
    mode = ["plain", "cascade", "mg", "mixed"][int(rng.integers(0, 4))]
    nranks = int(rng.integers(1, 5))
    levels = 0 if mode == "plain" else int(rng.integers(1, 3))
    desc = "seed %d: sizes %s pts %d sdf %d gk %d %s levels %d ranks %d %s" % (
        seed, sizes, npts, sdf, gk, mode, levels, nranks, {k: round(v, 2) for k, v in kw.items()})
    f = fi.LatticeGroup(sizes, nranks, dtype="f64") if nranks > 1 else fi.LatticeField(sizes, dtype="f64")
    f.add_field_constraints(w)
    f.add_points(w.data_pos, w.value_kernel, w.data_gradient if sdf else 0.0, w.gradient_kernel, pos, nrm if sdf else None, None,
                 values=val)
    poly = int(rng.choice([0, 0, 2, 3, 4, 6]))      # Chebyshev polynomial preconditioner (3-D lattices; ignored elsewhere)
    if poly and mode in ("plain", "cascade"):
        f.set_polynomial(poly)
        desc += " poly %d" % poly
    if levels:
        f.set_levels(levels, 1e-4)
        f.set_multigrid(mode in ("mg", "mixed"))
        if mode == "mixed":
            f.set_mixed_precision(True)
    f.assemble()
    tol = 1e-12
    res = f.solve_cg(None, 60000, tol)
    errs = []
    if res is None:
        return desc, ["breakdown"]
    x, it, rel = res
    true = f.true_residual()
    if not (rel <= tol and true <= 1.01 * tol):
        errs.append("residual %.2e (true %.2e) after %d it" % (rel, true, it))
    e = rel_inf(f.solution_f64(), x64)
    if e > 1e-5:
        errs.append("solution error %.2e" % e)
    return desc + " it %d" % it, errs


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad, t0 = 0, time.time()
    for s in range(first, first + cases):
        try:
            desc, errs = one_case(s)
        except Exception as e:      # noqa: BLE001
            desc, errs = "seed %d" % s, ["EXCEPTION %s: %s" % (type(e).__name__, str(e)[:300])]
        if s < first + 5 or (s - first) % 50 == 49:   # (a line every 50 cases: a silent run looks hung)
            print(desc, flush=True)
        if errs:
            bad += 1
            print("FAIL", desc, "->", "; ".join(errs), flush=True)
            if bad >= 12:
                break
    print("%d cases, %d failures, %.0f s" % (s - first + 1, bad, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
