#!/usr/bin/env python3
"""Randomised parity sweep (run on the GPU box): random lattice shapes, weights, kernels, point clouds and slab
counts; the GPU operator pieces against the oracle's explicit float64 normal equations, the decomposed operator
against the undivided one, error map and tile pre-solver against the oracle.  Prints the first failures with
their seeds.  usage: stress_parity.py [cases] [first seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import field_interpolation_amd as fi  # noqa: E402
from oracle import fi_oracle as oracle  # noqa: E402  (test infrastructure: this file lives under tests/)
from util import build_pair, random_points, rel_inf  # noqa: E402

TOL = {"f64": 1e-12, "f32": 3e-6}


def one_case(seed):
    rng = np.random.default_rng(seed)
    D = int(rng.integers(1, 4))
    big = rng.random() < 0.3
    hi = {1: 300, 2: 150 if big else 40, 3: 72 if big else 18}[D]
    sizes = [int(rng.integers(1, hi + 1)) for _ in range(D)]
    if rng.random() < 0.5:
        sizes[0] = max(4, (sizes[0] // 4) * 4)          # the LDS-tiled kernels need x % 4 == 0 (fp32) / % 2 (fp64)
    kw = {}
    for name, p in (("model_0", 0.3), ("model_1", 0.5), ("model_2", 0.8), ("model_3", 0.15), ("model_4", 0.15),
                    ("gradient_smoothness", 0.2)):
        kw[name] = float(rng.uniform(0.05, 1.5)) if rng.random() < p else 0.0
    vk, gk = int(rng.integers(0, 2)), int(rng.integers(0, 3))
    w = fi.Weights(data_pos=float(rng.uniform(0.2, 2)), data_gradient=float(rng.uniform(0.2, 2)),
                   value_kernel=fi.ValueKernel(vk), gradient_kernel=fi.GradientKernel(gk), **kw)
    n = int(np.prod(sizes))
    npts = int(rng.integers(0, 4 * n + 20)) if n < 3000 else int(rng.integers(0, n // 2))
    npts = min(npts, 6000)
    pos, nrm, pw, val = random_points(rng, sizes, npts, margin=float(rng.uniform(0.0, 2.0)))
    use_nrm = rng.random() < 0.8 or vk == 0
    use_val = rng.random() < 0.5 and npts <= 600       # the oracle adds valued points one by one
    dtype = "f64" if rng.random() < 0.5 else "f32"
    desc = "seed %d: sizes %s %s pts %d vk %d gk %d nrm %d val %d %s" % (
        seed, sizes, dtype, npts, vk, gk, use_nrm, use_val, {k: round(v, 2) for k, v in kw.items() if v})
    if npts == 0:
        pos, nrm, pw, val = pos[:0], nrm[:0], pw[:0], val[:0]
    fo, fg = build_pair(oracle, fi, sizes, w, pos, nrm if use_nrm else None, pw, val if use_val else None, dtype=dtype)
    fg.assemble()
    AtA, atb, diag = fo.normal_equations()
    absA = abs(AtA)
    errs = []
    if np.abs(atb).max() > 0:
        if rel_inf(fg.Atb(), atb) > TOL[dtype]:
            errs.append("Atb %.2e" % rel_inf(fg.Atb(), atb))
    elif fg.Atb().any():
        errs.append("Atb nonzero")
    if np.abs(diag).max() > 0 and rel_inf(fg.diag(), diag) > TOL[dtype]:
        errs.append("diag %.2e" % rel_inf(fg.diag(), diag))
    x = rng.normal(size=n)
    y = fg.apply_AtA(x)
    scale = max((absA @ np.abs(x)).max(), 1e-300)
    if np.abs(y - AtA @ x).max() > TOL[dtype] * scale:
        errs.append("apply %.2e" % (np.abs(y - AtA @ x).max() / scale))
    # error map
    xs = rng.normal(size=n).astype(np.float32)
    em_o, em_g = fo.error_map(xs), fg.error_map(xs)
    if np.abs(em_o).max() > 0 and rel_inf(em_g, em_o) > 3e-4:
        errs.append("error_map %.2e" % rel_inf(em_g, em_o))
    # slabs
    if sizes[-1] >= 4 and D >= 1:
        reach = max([k for k, v in ((1, w.model_1), (2, w.model_2), (3, w.model_3), (4, w.model_4)) if v > 0] + [1])
        maxr = sizes[-1] // max(reach, 1)
        if maxr >= 2 and (gk != 2 or (reach >= 2 and use_nrm)):   # kLinearInterpolation rows over slabs need two ghost planes
            nr = int(rng.integers(2, min(maxr, 6) + 1))
            grp = fi.LatticeGroup(sizes, nr, dtype=dtype)
            grp.add_field_constraints(w)
            grp.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm if use_nrm else None, pw,
                           values=val if use_val else None)
            grp.assemble()
            yg = grp.apply_AtA(x)
            if np.abs(yg - y).max() > (1e-12 if dtype == "f64" else 3e-6) * max(np.abs(y).max(), 1e-300):
                errs.append("slabs(%d) apply %.2e" % (nr, np.abs(yg - y).max() / max(np.abs(y).max(), 1e-300)))
            eg = grp.error_map(xs)
            if np.abs(em_g).max() > 0 and rel_inf(eg, em_g) > 1e-5:
                errs.append("slabs(%d) error_map %.2e" % (nr, rel_inf(eg, em_g)))
            desc += " slabs %d" % nr
    # tile pre-solver on small problems (dense float64 re-derivation)
    if n <= 1500 and gk != 2 and np.abs(diag).min() > 0 and rng.random() < 0.5:
        ts = int(rng.integers(2, 9))
        M = AtA.toarray()
        g = rng.normal(size=n).astype(np.float32)
        coords = np.stack(np.unravel_index(np.arange(n), sizes[::-1])[::-1], 1)
        tile_of = np.zeros(n, np.int64)
        for d in range(D - 1, -1, -1):
            tile_of = tile_of * 1024 + coords[:, d] // ts
        expect = g.astype(np.float64).copy()
        for t in np.unique(tile_of):
            mine, other = np.where(tile_of == t)[0], np.where(tile_of != t)[0]
            rhs = atb[mine] - 2.0 * M[np.ix_(mine, other)] @ g[other].astype(np.float64)
            expect[mine] = np.linalg.solve(M[np.ix_(mine, mine)] + 1e-6 * np.eye(len(mine)), rhs)
        fg.tile_pass(g, ts)
        tol = 1e-6 if dtype == "f64" else 2e-2
        scale_t = max(np.abs(expect).max(), 1e-3 * np.abs(g).max())     # b == 0 (targets 0, no normals) gives expect == 0
        if np.abs(fg.solution_f64() - expect).max() > tol * scale_t:
            errs.append("tile(%d) %.2e" % (ts, np.abs(fg.solution_f64() - expect).max() / scale_t))
        desc += " tile %d" % ts
    return desc, errs


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    t0 = time.time()
    seen = {"1-D": 0, "2-D": 0, "3-D": 0, "slabs": 0, "tile": 0, "f32": 0, "val 1": 0, "gk 2": 0}
    for s in range(first, first + cases):
        try:
            desc, errs = one_case(s)
        except Exception as e:      # noqa: BLE001
            desc, errs = "seed %d" % s, ["EXCEPTION %s: %s" % (type(e).__name__, str(e)[:200])]
        for k in seen:
            if k in desc or (k.endswith("-D") and desc.count(",") + 1 == 0):
                seen[k] += 1
        if desc.count("sizes [") and desc.split("sizes [")[1].split("]")[0].count(",") + 1 in (1, 2, 3):
            seen["%d-D" % (desc.split("sizes [")[1].split("]")[0].count(",") + 1)] += 1
        if s < first + 5 or (s - first) % 100 == 99:   # (a line every 100 cases: a silent run looks hung)
            print(desc, flush=True)
        if errs:
            bad += 1
            print("FAIL", desc, "->", "; ".join(errs), flush=True)
            if bad >= 15:
                break
    print("%d cases, %d failures, %.0f s; coverage %s" % (s - first + 1, bad, time.time() - t0, seen), flush=True)


if __name__ == "__main__":
    main()
