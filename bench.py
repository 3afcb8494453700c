#!/usr/bin/env python3
"""bench.py -- solved lattice points / second on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path on a synthetic lattice problem whose inputs already sit in
HBM: data-constraint assembly (fi_add_points + fi_assemble) + Jacobi-PCG to ||r|| <= tol*||Atb||
(fi_solve_cg).  Workload (config.workload): BASELINE.json config 4 -- a 256^3 lattice, 1M scattered noisy
value constraints, model_2 = 0.5 -- on one GPU; with --gpus N the lattice is 256 x 256 x (256*N) with
N*1M points of the same density, one 256^3 slab per GPU (weak scaling), halo planes and dot products
over RCCL.

Prints ONE JSON line on rank 0 (see the task contract), including
  "roofline"     achieved HBM GB/s of the AtA-apply kernel: algorithmic bytes / mean
                 launch duration measured with HIP events inside the timed region (fi_stats.spmv_ms_avg);
  "cpu_baseline" the C++ oracle (restatement of the reference's triplets -> AtA -> BiCGSTAB path, fp32,
                 one thread) timed on a bounded sample of the same workload (rank 0, N = 1 only);
  "cpu_best_effort" the same rows solved by a matrix-free Jacobi-PCG with OpenMP on all host cores
                 (not the reference's algorithm, a second CPU figure beside the port).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic(side, points, dtype):
    """HBM bytes per launch of the AtA-apply kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 x2 read correction; tools/pmc_traffic.py).  Counters cannot be
    read from inside this process, so the figure is the committed measurement of exactly this workload
    (profiles/r1_traffic_apply_c4.json); any other workload reports null."""
    if (side, points, dtype) != (256, 1_000_000, "f32"):
        return None
    path = os.path.join(ROOT, "profiles", "r1_traffic_apply_c4.json")
    try:
        with open(path) as f:
            return json.load(f)["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def host_cores():
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota, at most 16 (a GPU box gives
    one GPU's job a 16-core share of a 256-thread host; more OpenMP threads than that only fight each other)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(side, tol):
    """Oracle ("port") on a bounded sample: config 4 scaled to side^3 with the same point density."""
    import numpy as np
    from field_interpolation_amd import synth
    from oracle import fi_oracle as fo
    npts = int(round(1_000_000 * (side / 256.0) ** 3))
    sizes, w, pos, val = synth.config4(side=side, num_points=npts, seed=3)
    t0 = time.perf_counter()
    f = fo.LatticeField(sizes)
    f.add_field_constraints(fo.Weights(model_2=w.model_2))
    f.add_value_constraints(pos, val, w.data_pos)
    t1 = time.perf_counter()
    res = f.solve_with_guess(np.zeros(f.num_unknowns, np.float32), 0, tol)
    t2 = time.perf_counter()
    iters = res[1] if res else -1
    port = {"value": f.num_unknowns / (t2 - t0), "unit": "lattice points/s", "cores": 1, "kind": "port",
            "sample": "config 4 at %d^3 (%d points, same density), assembly %.2f s + AtA/BiCGSTAB fp32 %.2f s, "
                      "%d iterations" % (side, npts, t1 - t0, t2 - t1, iters)}
    # SURVEY.md 8(d) "best-effort CPU": the same rows, Jacobi-PCG on A^T(A x) without forming AtA, OpenMP on every
    # host core -- not the reference's algorithm (that is the port above), reported beside it
    cores = host_cores()
    t3 = time.perf_counter()
    best = f.solve_pcg_rows_omp(np.zeros(f.num_unknowns, np.float32), 0, tol, cores)
    t4 = time.perf_counter()
    extra = None
    if best:
        extra = {"value": f.num_unknowns / ((t1 - t0) + (t4 - t3)), "unit": "lattice points/s", "cores": cores,
                 "kind": "matrix-free Jacobi-PCG on the oracle's rows (OpenMP), not the reference's algorithm",
                 "sample": "the same sample: assembly %.2f s (1 thread) + compressed rows/columns %.2f s + %d "
                           "iterations %.2f s" % (t1 - t0, best[3], best[1], best[4])}
    return port, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--side", type=int, default=256)
    ap.add_argument("--points", type=int, default=1_000_000, help="data points per 256^3-equivalent slab")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--cpu-side", type=int, default=112, help="lattice side of the CPU baseline sample (0: skip)")
    ap.add_argument("--levels", type=int, default=2, help="coarser levels for the coarse-to-fine start (0: plain Jacobi-PCG)")
    ap.add_argument("--coarse-tol", type=float, default=1e-5)
    ap.add_argument("--multigrid", action="store_true", help="V-cycle preconditioned CG instead of Jacobi-PCG")
    ap.add_argument("--poly", type=int, default=4, help="terms of the Chebyshev polynomial preconditioner (0: Jacobi-PCG)")
    ap.add_argument("--poly-ratio", type=float, default=10.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    import field_interpolation_amd as fi
    from field_interpolation_amd import dist as fdist
    from field_interpolation_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    dist = None
    # FI_BENCH_ONE_DEVICE=1 (functional test on a 1-GPU box only): every rank uses cuda:0, torch talks gloo
    one_device = os.environ.get("FI_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    side = args.side
    depth = side * world
    npts = args.points * world

    def build(slabs):
        """slabs: ONE lattice of 256 x 256 x (256 * world), a slab per rank, halo planes and dot products over RCCL.
        not slabs (only if the communicator cannot be set up): every rank solves its own 256^3 replica."""
        if slabs:
            sizes, w, pos, val = synth.config4(side=side, num_points=npts, seed=3, depth=depth)
            field = fi.LatticeField(sizes, dtype=args.dtype, rank=rank, nranks=world)
            if world > 1:
                fdist.init_comm(field, dev)      # RCCL unique id from rank 0, broadcast by torch.distributed
        else:
            sizes, w, pos, val = synth.config4(side=side, num_points=args.points, seed=3 + rank)
            field = fi.LatticeField(sizes, dtype=args.dtype)
        lo, hi = field.slab
        # each rank uploads the points whose cells touch its slab (the library drops the rest anyway)
        keep = fdist.points_of_slab(pos, 3, lo, hi)
        d_pos = torch.from_numpy(np.ascontiguousarray(pos[keep])).to(dev)
        d_val = torch.from_numpy(np.ascontiguousarray(val[keep])).to(dev)
        d_out = torch.empty(field.num_owned, dtype=torch.float32, device=dev)   # the solution stays in HBM
        torch.cuda.synchronize()
        field.add_field_constraints(w)
        if args.levels > 0:
            field.set_levels(args.levels, args.coarse_tol)
            if args.multigrid:
                field.set_multigrid(True)
        if args.poly > 1 and not args.multigrid:
            field.set_polynomial(args.poly, args.poly_ratio)

        def step():
            field.clear_points()
            field.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val)
            field.assemble()
            out = field.solve_cg(None, 0, args.tol, out=d_out)
            if out is None:
                raise RuntimeError("CG breakdown")
            return out
        return field, step

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    parallelism, note = "slab%d" % world, None
    try:
        field, step = build(True)
        for _ in range(args.warmup):
            step()
        ok = 1
    except Exception as e:          # noqa: BLE001 -- any failure of the exchange path is reported, not hidden
        if world == 1:
            raise
        ok, note = 0, "%s: %s" % (type(e).__name__, e)
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=("cpu" if one_device else dev))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:   # some rank could not run the slab exchange: independent replicas, said so in the line
            parallelism = "replicas%d (slab exchange failed: %s)" % (world, note or "on another rank")
            field, step = build(False)
            for _ in range(args.warmup):
                step()
    spmv_ms, spmv_n, asm_ms, solve_ms = 0.0, 0, 0.0, 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, iters, rel = step()
        st = field.stats()
        spmv_ms += st["spmv_ms_avg"] * st["spmv_samples"]
        spmv_n += st["spmv_samples"]
        asm_ms += st["assemble_ms"]
        solve_ms += st["solve_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if one_device else dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = field.stats()
    true_rel = field.true_residual()

    n_global = side * side * depth
    value = n_global * args.steps / elapsed
    spmv_avg_ms = spmv_ms / max(spmv_n, 1)
    achieved = st["spmv_bytes"] / (spmv_avg_ms * 1e-3) / 1e9 if spmv_avg_ms > 0 else 0.0
    line = {
        "metric": "solved lattice points/sec (assembly+CG to tol=%g)" % args.tol,
        "value": value, "unit": "lattice points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("config4: 3D %dx%dx%d lattice, %d scattered noisy value constraints, model_2=0.5, "
                                "Jacobi-PCG to rel. residual %g" % (side, side, depth, npts, args.tol)) if parallelism.startswith("slab")
                               else ("config4: %d independent 3D %dx%dx%d lattices, %d scattered noisy value constraints each, "
                                     "model_2=0.5, Jacobi-PCG to rel. residual %g" % (world, side, side, side, args.points, args.tol)),
                   "parallelism": parallelism, "iterations": iters, "rel_residual": rel,
                   "levels": st["num_levels"], "coarse_iterations": st["coarse_iterations"],
                   "solver": ("V-cycle PCG" if (args.multigrid and st["num_levels"] > 1) else
                              "Jacobi-PCG" + (" from a coarse-to-fine cascade" if st["num_levels"] > 1 else "")),
                   "true_rel_residual": true_rel, "assemble_ms": asm_ms / args.steps,
                   "solve_ms": solve_ms / args.steps, "occupied_cells": st["num_cells"],
                   "data_rows": st["num_data_rows"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(side, args.points, args.dtype) if world == 1 else None,
                     "kernel": "k_apply_march3d: AtA apply, matrix-free stencil + fused data cells (finest level)", "launch_ms": spmv_avg_ms,
                     "algorithmic_bytes": st["spmv_bytes"], "samples": spmv_n},
    }
    if rank == 0 and world == 1 and args.cpu_side > 0:
        line["cpu_baseline"], best_effort = cpu_baseline(args.cpu_side, args.tol)
        if best_effort:
            line["cpu_best_effort"] = best_effort
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
