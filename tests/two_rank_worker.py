#!/usr/bin/env python3
"""Worker of tests/test_gpu_two_ranks.py, started by torch.distributed.run with two ranks on ONE GPU: the real
one-process-per-slab orchestration (rank bootstrap, fi_slab_point_range filtering of the points, per-rank assembly with
coarser levels, rank-set solvers) with the halo planes and dot products carried by the host-staged test transport
(fi_comm_init_host) instead of RCCL.  Rank 0 also solves the undivided problem and compares."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.distributed as dist

dist.init_process_group("gloo")     # before any GPU call
rank, world = dist.get_rank(), dist.get_world_size()

import field_interpolation_amd as fi                      # noqa: E402
from field_interpolation_amd import dist as fdist         # noqa: E402
from field_interpolation_amd import synth                 # noqa: E402
from util import sphere_points                            # noqa: E402

torch.cuda.set_device(0)


def run(name, sizes, w, pos, nrm, val, dtype, tol, levels=0, poly=0, multigrid=False, mixed=False):
    def configure(f):
        f.add_field_constraints(w)
        if levels:
            f.set_levels(levels, 1e-6 if dtype == "f64" else 1e-5)
            if multigrid:
                f.set_multigrid(True)
                if mixed:
                    f.set_mixed_precision(True)
        if poly:
            f.set_polynomial(poly)

    f = fi.LatticeField(sizes, dtype=dtype, rank=rank, nranks=world)
    fdist.init_comm(f, None, host_staged=True)
    configure(f)
    zlo, zhi = f.point_range()
    D = len(sizes)
    z = pos.reshape(-1, D)[:, D - 1]
    keep = (z >= zlo) & (z < zhi)
    gw = w.data_gradient if nrm is not None else 0.0
    f.add_points(w.data_pos, w.value_kernel, gw, w.gradient_kernel, pos[keep], None if nrm is None else nrm[keep], None,
                 values=None if val is None else val[keep])
    f.assemble()
    x, it, rel = f.solve_cg(None, 0, tol)
    true_rel = f.true_residual()           # a collective over the slabs as well
    st = f.stats()
    parts = [None] * world
    dist.gather_object((x, it, rel, true_rel, st["coarse_iterations"], int(keep.sum())), parts if rank == 0 else None, dst=0)
    out = None
    if rank == 0:
        xs = np.concatenate([p[0] for p in parts])
        one = fi.LatticeField(sizes, dtype=dtype)
        configure(one)
        one.add_points(w.data_pos, w.value_kernel, gw, w.gradient_kernel, pos, nrm, None, values=val)
        one.assemble()
        x1, it1, rel1 = one.solve_cg(None, 0, tol)
        st1 = one.stats()
        out = {"case": name, "iterations": [p[1] for p in parts], "iterations_one": it1, "rel": [p[2] for p in parts],
               "true_rel": [p[3] for p in parts], "coarse_iterations": [p[4] for p in parts],
               "coarse_iterations_one": st1["coarse_iterations"], "points_kept": [p[5] for p in parts], "points": len(pos),
               "max_diff": float(np.abs(xs - x1).max() / np.abs(x1).max()), "tol": tol,
               "checksum": [float(np.sum(xs.astype(np.float64))), float(np.sum(np.abs(xs.astype(np.float64))))]}
        del one
    del f
    dist.barrier()
    return out


results = []
if os.environ.get("FI_WORKER_CASES") == "tail":
    # four slabs of 16 planes: levels 32^3 (8 planes per slab) and 16^3 (4) are slab decompositions, 8^3 is the replicated
    # tail -- whole on every rank, reached through the vector all-reduce of the transport
    rng = np.random.default_rng(9)
    sizes = [64, 64, 64]
    spos, snrm = sphere_points(rng, sizes, 4000)
    results.append(run("SDF 64^3, 4 slabs, V-cycle PCG f64 mixed, 3 levels (8^3 replicated)", sizes, fi.Weights(), spos, snrm, None,
                       "f64", 1e-8, levels=3, multigrid=True, mixed=True))
    results.append(run("SDF 64^3, 4 slabs, V-cycle PCG f32, 3 levels (8^3 replicated)", sizes, fi.Weights(), spos, snrm, None,
                       "f32", 1e-5, levels=3, multigrid=True))
    if rank == 0:
        print("RESULTS " + json.dumps(results), flush=True)
    dist.destroy_process_group()
    sys.exit(0)
if os.environ.get("FI_WORKER_CASES") == "lopsided":
    # ALL the data sit in rank 0's half: rank 1 is handed zero points.  What a rank holds must not decide which collectives
    # it runs (fi_assemble agrees on the data facts first): oriented points (gradient rows -> the Chebyshev smoother in A
    # on every rank), the kLinearInterpolation gradient kernel (triplet rows -> no polynomial, no deep halo on any rank),
    # and value rows through the lumped replica of the mixed-precision solve.
    rng = np.random.default_rng(11)
    sizes = [32, 32, 64]
    d = rng.normal(size=(1500, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    lpos = (np.array([15.5, 15.5, 10.0]) + 6.0 * d + rng.normal(scale=0.2, size=d.shape)).astype(np.float32)
    lnrm = d.astype(np.float32)
    lval = rng.normal(size=len(lpos)).astype(np.float32)
    results.append(run("lopsided SDF, V-cycle PCG f64 mixed (2 levels)", sizes, fi.Weights(), lpos, lnrm, None, "f64", 1e-8, levels=2,
                       multigrid=True, mixed=True))
    wl = fi.Weights(gradient_kernel=fi.GradientKernel.kLinearInterpolation)
    results.append(run("lopsided, gradient kLinearInterpolation, polynomial asked for, f32", sizes, wl, lpos, lnrm, None, "f32", 1e-5,
                       levels=1, poly=4))
    results.append(run("lopsided value rows, V-cycle PCG f64 mixed (lumped replica)", sizes, fi.Weights(), lpos, None, lval, "f64", 1e-8,
                       levels=2, multigrid=True, mixed=True))
    results.append(run("lopsided value rows, cascade + polynomial PCG f32", sizes, fi.Weights(), lpos, None, lval, "f32", 1e-5, levels=1,
                       poly=4))
    if rank == 0:
        print("RESULTS " + json.dumps(results), flush=True)
    dist.destroy_process_group()
    sys.exit(0)
sizes, w, pos, val = synth.config4(side=48, num_points=6591, seed=3)
results.append(run("config4 48^3 cascade(2 levels) + polynomial PCG f32", sizes, w, pos, None, val, "f32", 1e-5, levels=2, poly=4))
results.append(run("config4 48^3 Jacobi-PCG f64", sizes, w, pos, None, val, "f64", 1e-9))
results.append(run("config4 48^3 cascade(2 levels) + Jacobi-PCG f64", sizes, w, pos, None, val, "f64", 1e-9, levels=2))
rng = np.random.default_rng(7)
sizes = [40, 36, 48]
spos, snrm = sphere_points(rng, sizes, 3000)
results.append(run("SDF 40x36x48 V-cycle PCG f64 mixed (3 levels)", sizes, fi.Weights(), spos, snrm, None, "f64", 1e-8, levels=2,
                   multigrid=True, mixed=True))
if rank == 0:
    print("RESULTS " + json.dumps(results), flush=True)
dist.destroy_process_group()
