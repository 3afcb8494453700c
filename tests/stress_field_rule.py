#!/usr/bin/env python3
"""Randomised sweep of the field stop rule (FI_OPT_FIELD_TOLERANCE; run on the GPU box): 3-D (seeds from 80000 on: 2-D; 90000-99999: fp32
contexts in 2-D; from 100000 on: the K-cycle, 3-D, from 110000 on 2-D) lattices of random shape, value
data or oriented points, random weights, levels and tolerance; the field the rule stops at against the same context's
solve to the fp64 floor.  The rule is an estimate (twice the extrapolated difference of consecutive iterates): a case
FAILS when the true error exceeds 2 x the tolerance, and the sweep prints the distribution of error / tolerance.
usage: stress_field_rule.py [cases] [first seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import field_interpolation_amd as fi  # noqa: E402
from util import rel_inf, sphere_points  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    big = rng.random() < 0.25
    sizes = [int(rng.integers(40, 161 if big else 73)) for _ in range(3)]
    if 80000 <= seed < 100000 or seed >= 110000:   # (seeds 80000-99999 and from 110000 on: 2-D lattices, the shapes of configs 2 and 3)
        sizes = [int(rng.integers(96, 1025 if big else 385)) for _ in range(2)]
    sizes[0] = max(8, (sizes[0] // 4) * 4)
    kw = dict(model_2=float(rng.uniform(0.2, 1.0)))
    if rng.random() < 0.4:
        kw["model_1"] = float(rng.uniform(0.02, 0.5))
    if rng.random() < 0.15:
        kw["model_0"] = float(rng.uniform(0.001, 0.02))
    sdf = rng.random() < 0.5
    gk = int(rng.integers(0, 3)) if sdf else 1
    w = fi.Weights(gradient_kernel=fi.GradientKernel(gk), **kw)
    n = int(np.prod(sizes))
    npts = int(rng.integers(200, max(400, n // 20)))
    pos, nrm = sphere_points(rng, sizes, npts, noise=float(rng.uniform(0.1, 1.0)))
    val = None if sdf else rng.normal(size=npts).astype(np.float32)
    mixed = rng.random() < 0.7
    levels = int(rng.integers(1, 4)) if len(sizes) == 3 else int(rng.integers(1, 6))
    tol = float(rng.choice([1e-4, 1e-5, 1e-6]))
    f32 = 90000 <= seed < 100000     # (seeds 90000-99999: fp32 contexts -- value data, the tolerances fp32 can meet -- against an fp64 twin)
    kc = int(rng.integers(1, levels + 1)) if seed >= 100000 else 0   # (seeds from 100000 on: the K-cycle on 1 .. levels coarse levels; 3-D, from 110000: 2-D)
    if kc:
        mixed = True
    if f32:
        sdf, val, mixed = False, (val if val is not None else rng.normal(size=npts).astype(np.float32)), False
        tol = float(rng.choice([1e-3, 1e-4]))
    desc = "seed %d: sizes %s pts %d sdf %d gk %d %s levels %d tol %.0e %s" % (
        seed, sizes, npts, sdf, gk, "fp32" if f32 else (("K-cycle %d" % kc) if kc else ("mixed" if mixed else "fp64 V-cycle")), levels, tol, {k: round(v, 3) for k, v in kw.items()})

    def build(dtype):
        g = fi.LatticeField(sizes, dtype=dtype)
        g.add_field_constraints(w)
        g.add_points(w.data_pos, w.value_kernel, w.data_gradient if sdf else 0.0, w.gradient_kernel, pos, nrm if sdf else None, None,
                     values=val)
        g.set_levels(levels, 1e-3)
        g.set_multigrid(True)
        if mixed:
            g.set_mixed_precision(True)
        if kc:
            g.set_kcycle(kc)
        g.assemble()
        return g

    f = build("f64")
    res = f.solve_cg(None, 4000, 1e-13)
    if res is None:
        return desc, ["breakdown of the reference solve"], None
    ref = f.solution_f64().copy()
    it_ref = res[1]
    if f32:
        f = build("f32")
    f.set_field_tolerance(tol)
    res = f.solve_cg(None, 4000, 1e-5)
    if res is None:
        return desc, ["breakdown"], None
    st = f.stats()
    err = rel_inf(f.solution_f64(), ref)
    errs = []
    if st["converged"] == 0:    # an fp32 solve at its residual floor with the estimate above the tolerance: the library says so
        if f32 and not (0 <= st["field_estimate"] <= tol):
            return desc + " it %d / %d est %.1e err %.1e NOT CERTIFIED (fp32 floor)" % (st["iterations"], it_ref, st["field_estimate"], err), [], None
        if st["iterations"] >= 4000:   # (the iteration cap, like the reference solve of the same case: not the rule's doing)
            return desc + " it %d / %d: the iteration cap" % (st["iterations"], it_ref), [], None
        errs.append("converged = 0 (estimate %.2e)" % st["field_estimate"])
    if err > 2.0 * tol:
        errs.append("field error %.2e for a tolerance of %.0e (estimate %.2e, %d iterations, reference %d)" % (
            err, tol, st["field_estimate"], st["iterations"], it_ref))
    if seed % 3 == 0 and not errs:
        # a warm start: the same system again from a start that is off by a smooth bump of 50 x the tolerance (what a caller
        # re-solving after a small change of the data hands over) -- the rule must not trust the first small steps
        x0 = f.solution_f64().astype(np.float32)
        grid = np.meshgrid(*[np.linspace(0.0, np.pi, n_) for n_ in sizes[::-1]], indexing="ij")
        bump = np.ones_like(grid[0])
        for gcoord in grid:
            bump = bump * np.sin(gcoord)
        x0 = x0 + (50.0 * tol * float(np.abs(ref).max())) * bump.reshape(-1).astype(np.float32)
        res = f.solve_cg(x0, 4000, 1e-5)
        st2 = f.stats()
        err2 = rel_inf(f.solution_f64(), ref)
        if res is None or (st2["converged"] == 1 and err2 > 2.0 * tol):
            errs.append("warm start: field error %.2e for a tolerance of %.0e (estimate %.2e, %d iterations)" % (
                err2, tol, st2["field_estimate"], st2["iterations"]))
        desc += " warm %d it err %.1e" % (st2["iterations"], err2)
        err = max(err, err2 if st2["converged"] == 1 else 0.0)
    return desc + " it %d / %d est %.1e err %.1e" % (st["iterations"], it_ref, st["field_estimate"], err), errs, err / tol


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad, t0, ratios = 0, time.time(), []
    for s in range(first, first + cases):
        try:
            desc, errs, ratio = one_case(s)
        except Exception as e:      # noqa: BLE001
            desc, errs, ratio = "seed %d" % s, ["EXCEPTION %s: %s" % (type(e).__name__, str(e)[:300])], None
        if ratio is not None:
            ratios.append(ratio)
        if s < first + 8 or (s - first) % 10 == 9:   # (a line every 10 cases: a silent run looks hung)
            print(desc, flush=True)
        if errs:
            bad += 1
            print("FAIL", desc, "->", "; ".join(errs), flush=True)
            if bad >= 12:
                break
    r = np.sort(np.asarray(ratios)) if ratios else np.zeros(1)
    print("%d cases, %d failures, %.0f s; error / tolerance: median %.3f, 90 %% %.3f, max %.3f; above 1: %d" % (
        s - first + 1, bad, time.time() - t0, float(np.median(r)), float(r[int(0.9 * (len(r) - 1))]), float(r[-1]), int((r > 1).sum())),
        flush=True)


if __name__ == "__main__":
    main()
