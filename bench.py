#!/usr/bin/env python3
"""bench.py -- solved lattice points / second on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path on a synthetic lattice problem whose inputs already sit in HBM:
data-constraint assembly (fi_clear_points + fi_add_points + fi_assemble) + the iterative solve to
||Atb - AtA x|| <= tol * ||Atb|| (fi_solve_cg).

Workloads (--config, named in config.workload; SURVEY.md 8(d) recipes, field_interpolation_amd/synth.py):
  4 (default, the configuration BASELINE.json's metric is quoted on): 3-D 256^3 lattice, 1 M scattered noisy value
    constraints, model_2 = 0.5, tol 1e-5, fp32; coarse-to-fine start + CG preconditioned by a Chebyshev polynomial
  5: 3-D 512^3 SDF from 5 M oriented points, tol 1e-6: fp64 CG with the V-cycle preconditioner in fp32 (mixed)
  3: 2-D 4096^2 SDF from 200 k oriented points, tol 1e-5: the same solver
  2: 2-D 1024^2, 10 k noisy value constraints, model_2 = 10, tol 1e-5: the same solver
--gpus N (launched by torch.distributed.run, one rank per GPU): ONE lattice, one slab of its slowest axis per rank, halo
planes and dot products over RCCL.  --scaling strong (default): the lattice of the configuration itself, split N ways
(the form BASELINE configs 4 and 5 state); --scaling weak: the slab per GPU is fixed -- config 4 becomes
256 x 256 x (256 N) with N x 1 M points of the same density.  If the slab exchange cannot be set up the run FAILS (exit 3)
unless --allow-replicas is given; the line then says "replicasN" and its value is not the metric.

Prints ONE JSON line on rank 0 (see the task contract), including
  "roofline"       the dominant kernel of the timed region (largest share of GPU time): algorithmic bytes per launch
                    / mean launch duration measured with HIP events on the solver stream inside
                   the timed region; for the default solver that is the Chebyshev step of the preconditioner
                   (k_apply_march3d<..., EPI>: the model-operator apply with the polynomial recurrence in its epilogue)
  "roofline_apply" the same for the full operator apply with fused data cells (the CG SpMV the north-star names)
  "solution_rel_err" ||x - x64||_inf / ||x64||_inf against an fp64 solve of the same inputs to 1e-10 (outside the timed
                   region; N = 1 only)
  "accurate"       config 4, N = 1: the same step with the solver that meets the north-star's FIELD tolerance (fp64 CG +
                   fp32 V-cycle to --accurate-tol): value, ms_per_step, iterations, solution_rel_err <= 1e-5
  "cold_ms_per_step" one step on a fresh context (allocation, power method, first assemble, first solve);
  "cold_pooled_ms_per_step" the same with the device blocks of a destroyed context (fi_memory_pool)
  "cpu_baseline"   the C++ oracle (restatement of the reference's triplets -> AtA -> BiCGSTAB path, fp32, one thread)
                   timed on a bounded sample of the same workload (rank 0, N = 1, config 4 only)
  "cpu_best_effort" the same rows solved by a matrix-free Jacobi-PCG with OpenMP on all host cores
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
TRAFFIC_ROUND = "r3"       # profiles/<round>_traffic_<kernel>_c4.json: the PMC passes of the shipped kernels


def measured_traffic(kind, config, side, points, dtype):
    """HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, gfx950
    x2 read correction; tools/pmc_traffic.py).  Counters cannot be read from inside this process, so the figure is the
    committed measurement of exactly this workload (profiles/r2_traffic_*.json); any other workload reports null."""
    if (config, side, points, dtype) != (4, 256, 1_000_000, "f32"):
        return None
    path = os.path.join(ROOT, "profiles", "%s_traffic_%s_c4.json" % (TRAFFIC_ROUND, kind))
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


def workload(args, world):
    """-> dict(sizes, weights, positions, normals, values, tol, dtype, solver settings, text)"""
    from field_interpolation_amd import synth
    cfg = args.config
    weak = args.scaling == "weak" and world > 1
    if cfg == 4:
        side = args.side or 256
        depth = side * world if weak else side
        npts = args.points or int(round(1_000_000 * (side / 256.0) ** 3))
        npts = npts * world if weak else npts
        sizes, w, pos, val = synth.config4(side=side, num_points=npts, seed=3, depth=depth)
        return dict(sizes=sizes, w=w, pos=pos, nrm=None, val=val, tol=args.tol or 1e-5, dtype=args.dtype or "f32",
                    levels=(3 if args.multigrid else 1) if args.levels is None else args.levels, coarse_tol=args.coarse_tol or 1e-5,
                    multigrid=args.multigrid, mixed=False, poly=0 if args.multigrid else args.poly, points=npts,
                    text="config4: 3D %dx%dx%d lattice, %d scattered noisy value constraints, model_2=0.5" % (
                        sizes[0], sizes[1], sizes[2], npts))
    if cfg == 5:
        side = args.side or 512
        npts = args.points or int(round(5_000_000 * (side / 512.0) ** 2))
        if weak:
            raise SystemExit("config 5 is a fixed lattice: use --scaling strong")
        sizes, w, pos, nrm = synth.config5(side=side, num_points=npts, seed=4)
        return dict(sizes=sizes, w=w, pos=pos, nrm=nrm, val=None, tol=args.tol or 1e-6, dtype=args.dtype or "f64",
                    levels=6 if args.levels is None else args.levels, coarse_tol=args.coarse_tol or 1e-4,
                    multigrid=True, mixed=(args.dtype or "f64") == "f64", poly=0, points=npts,
                    text="config5: 3D %d^3 SDF from %d oriented points (sdf_from_points, default Weights)" % (side, npts))
    if cfg == 3:
        side = args.side or 4096
        pps = (args.points or 200_000) // 2
        if weak:
            raise SystemExit("config 3 is a fixed lattice: use --scaling strong")
        sizes, w, pos, nrm = synth.config3(side=side, points_per_shape=pps, seed=2)
        return dict(sizes=sizes, w=w, pos=pos, nrm=nrm, val=None, tol=args.tol or 1e-5, dtype=args.dtype or "f64",
                    levels=7 if args.levels is None else args.levels, coarse_tol=args.coarse_tol or 1e-4,
                    multigrid=True, mixed=(args.dtype or "f64") == "f64", poly=0, points=2 * pps,
                    text="config3: 2D %dx%d SDF from %d oriented points (triangle + inverted circle)" % (side, side, 2 * pps))
    if cfg == 2:
        side = args.side or 1024
        npts = args.points or 10_000
        if weak:
            raise SystemExit("config 2 is a fixed lattice: use --scaling strong")
        sizes, w, pos, val = synth.config2(side=side, num_points=npts, seed=1)
        return dict(sizes=sizes, w=w, pos=pos, nrm=None, val=val, tol=args.tol or 1e-5, dtype=args.dtype or "f64",
                    levels=7 if args.levels is None else args.levels, coarse_tol=args.coarse_tol or 1e-4,
                    multigrid=True, mixed=(args.dtype or "f64") == "f64", poly=0, points=npts,
                    text="config2: 2D %dx%d lattice, %d noisy value constraints, model_2=10" % (side, side, npts))
    raise SystemExit("--config must be 2, 3, 4 or 5 (config 1 is the CPU-runnable 1-D case: tests/)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json configuration: 2, 3, 4 (default) or 5")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="--gpus N: strong (default) = the configuration's own lattice split N ways, the form BASELINE "
                         "configs 4 and 5 state; weak = one 256^3 slab per GPU (256 x 256 x 256 N)")
    ap.add_argument("--allow-replicas", action="store_true",
                    help="if the slab exchange cannot be set up, run N independent replicas instead of failing")
    ap.add_argument("--side", type=int, default=0, help="lattice side (default: the configuration's)")
    ap.add_argument("--points", type=int, default=0, help="data points (default: the configuration's, scaled with the side)")
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--tol", type=float, default=0.0)
    ap.add_argument("--cpu-side", type=int, default=160, help="lattice side of the CPU baseline sample (0: skip)")
    ap.add_argument("--levels", type=int, default=None, help="coarser levels (config 4: coarse-to-fine start; 0: none)")
    ap.add_argument("--coarse-tol", type=float, default=0.0)
    ap.add_argument("--multigrid", action="store_true", help="config 4: V-cycle preconditioned CG")
    ap.add_argument("--poly", type=int, default=4, help="terms of the Chebyshev polynomial preconditioner (0: Jacobi-PCG)")
    ap.add_argument("--poly-ratio", type=float, default=30.0)
    ap.add_argument("--no-accuracy", action="store_true",
                    help="skip the fp64 comparison solve (solution_rel_err) and the accurate leg")
    ap.add_argument("--accurate-tol", type=float, default=0.0,
                    help="config 4, N = 1: residual tolerance of the leg that meets the north-star's 1e-5 FIELD tolerance "
                         "(default: 1e-7 at 256^3, tightened with the side -- the field error per unit of residual grows with "
                         "the lattice: 60 at 256^3, 190 at 512^3)")
    ap.add_argument("--accurate-levels", type=int, default=3)
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-step figure (fresh context)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import field_interpolation_amd as fi
    from field_interpolation_amd import dist as fdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    dist = None
    # FI_BENCH_ONE_DEVICE=1 (functional test on a 1-GPU box only): every rank uses cuda:0, torch talks gloo and the
    # slab exchange goes through the host-staged test transport (fi_comm_init_host) instead of RCCL
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

    def agree(ok, what):
        """All ranks learn whether every rank got through `what`; nobody enters the next collective otherwise."""
        if dist is None:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=("cpu" if one_device else dev))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    wl = workload(args, world)
    ndim = len(wl["sizes"])

    def configure(field, cfg):
        field.add_field_constraints(cfg["w"])
        if cfg["levels"] > 0:
            field.set_levels(cfg["levels"], cfg["coarse_tol"])
            if cfg["multigrid"]:
                field.set_multigrid(True)
                if cfg["mixed"]:
                    field.set_mixed_precision(True)
        if cfg["poly"] > 1:
            field.set_polynomial(cfg["poly"], args.poly_ratio)

    def build(slabs):
        """slabs: ONE lattice, a slab per rank, halo planes and dot products over RCCL.  not slabs (--allow-replicas, only
        if the communicator cannot be set up): every rank solves its own copy of the single-GPU workload."""
        note = None
        if slabs:
            field = fi.LatticeField(wl["sizes"], dtype=wl["dtype"], rank=rank, nranks=world)
            if world > 1:
                try:
                    fdist.init_comm(field, dev, host_staged=one_device)
                except Exception as e:          # noqa: BLE001 -- reported below, by every rank together
                    note = "%s: %s" % (type(e).__name__, e)
                if not agree(note is None, "communicator"):
                    raise RuntimeError(note or "communicator set-up failed on another rank")
        else:
            field = fi.LatticeField(wl["sizes"], dtype=wl["dtype"])
        configure(field, wl)
        # each rank uploads the points whose cells touch its slab on any level (the library drops the rest anyway)
        zlo, zhi = field.point_range()
        z = wl["pos"].reshape(-1, ndim)[:, ndim - 1]
        keep = (z >= zlo) & (z < zhi) if (slabs and world > 1) else np.ones(len(z), bool)
        d_pos = torch.from_numpy(np.ascontiguousarray(wl["pos"][keep])).to(dev)
        d_nrm = torch.from_numpy(np.ascontiguousarray(wl["nrm"][keep])).to(dev) if wl["nrm"] is not None else None
        d_val = torch.from_numpy(np.ascontiguousarray(wl["val"][keep])).to(dev) if wl["val"] is not None else None
        d_out = torch.empty(field.num_owned, dtype=torch.float32, device=dev)   # the solution stays in HBM
        torch.cuda.synchronize()
        w = wl["w"]

        def step(f=None, tol=None, out=None):
            f = field if f is None else f
            f.clear_points()
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient if d_nrm is not None else 0.0, w.gradient_kernel,
                         d_pos, d_nrm, None, values=d_val)
            f.assemble()
            res = f.solve_cg(None, 0, wl["tol"] if tol is None else tol, out=d_out if out is None else out)
            if res is None:
                raise RuntimeError("CG breakdown")
            return res
        return field, step, d_out

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    parallelism, note = "slab%d" % world, None
    field = step = d_out = None
    try:
        field, step, d_out = build(True)
        ok = True
    except Exception as e:          # noqa: BLE001 -- any failure of the exchange path is reported, not hidden
        if world == 1:
            raise
        ok, note = False, "%s: %s" % (type(e).__name__, e)
    if ok:
        for _ in range(args.warmup):
            try:
                step()
            except Exception as e:  # noqa: BLE001
                if world == 1:
                    raise
                ok, note = False, "%s: %s" % (type(e).__name__, e)
            if not agree(ok, "warm-up step"):   # after every step: a rank that failed must not leave the others in a collective
                ok = False
                break
    if not ok:
        if not args.allow_replicas:
            if rank == 0:
                print("bench.py: the slab exchange failed (%s); no metric measured (--allow-replicas runs independent "
                      "replicas instead)" % (note or "on another rank"), file=sys.stderr, flush=True)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
        parallelism = "replicas%d (slab exchange failed: %s)" % (world, note or "on another rank")
        field, step, d_out = build(False)
        for _ in range(args.warmup):
            step()

    acc = {"spmv_ms": 0.0, "spmv_n": 0, "prec_ms": 0.0, "prec_n": 0, "asm_ms": 0.0, "solve_ms": 0.0, "applies": 0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, iters, rel = step()
        st = field.stats()
        acc["spmv_ms"] += st["spmv_ms_avg"] * st["spmv_samples"]
        acc["spmv_n"] += st["spmv_samples"]
        acc["prec_ms"] += st["prec_ms_avg"] * st["prec_samples"]
        acc["prec_n"] += st["prec_samples"]
        acc["asm_ms"] += st["assemble_ms"]
        acc["solve_ms"] += st["solve_ms"]
        acc["applies"] += st["operator_applies"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if one_device else dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = field.stats()
    true_rel = field.true_residual()

    replicas = not parallelism.startswith("slab")
    n_global = int(np.prod(wl["sizes"])) * (world if replicas else 1)
    value = n_global * args.steps / elapsed
    spmv_avg_ms = acc["spmv_ms"] / max(acc["spmv_n"], 1)
    prec_avg_ms = acc["prec_ms"] / max(acc["prec_n"], 1)

    def roof(kernel, bytes_, ms, n, traffic):
        achieved = bytes_ / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic[0], "traffic_source": traffic[1], "kernel": kernel, "launch_ms": ms,
                "algorithmic_bytes": bytes_, "samples": n}

    def tr(kind):
        """(bytes per launch, where they come from): the committed PMC measurement of exactly this workload, or null."""
        t = measured_traffic(kind, args.config, wl["sizes"][0], wl["points"], wl["dtype"]) if world == 1 else None
        return (t, ("profiles/%s_traffic_%s_c4.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not "
                    "read in this run)" % (TRAFFIC_ROUND, kind)) if t is not None else None)

    apply_name = ("k_apply_march3d: AtA apply, matrix-free stencil + fused data cells (finest level)" if ndim == 3 else
                  "k_apply_tile2d: AtA apply, matrix-free stencil + fused data cells (finest level)")
    roof_apply = roof(apply_name, st["spmv_bytes"], spmv_avg_ms, acc["spmv_n"], tr("apply"))
    # the dominant kernel of the timed region: the Chebyshev step when the polynomial preconditioner runs (terms - 1
    # launches per outer iteration against one full apply), else the apply
    if acc["prec_n"] > 0 and wl["poly"] > 2:
        roof_main = roof("k_apply_march3d<EPI>: Chebyshev steps of the polynomial preconditioner = model-operator apply + "
                         "three-term recurrence in the epilogue; ALL steps of the sampled polynomials, timed between one pair "
                         "of HIP events per polynomial (first: r, bf16 scaling in, z out = 2.5 lattice passes; second 3.5; the "
                         "others 4.5): bytes of all sampled launches over their time, launch_ms = the mean per launch, finest level",
                         st["prec_bytes"], prec_avg_ms, acc["prec_n"], tr("cheb"))
    else:
        roof_main = roof_apply
    solver = ("V-cycle PCG" + (" (fp64 CG, fp32 V-cycle)" if wl["mixed"] else "") if (wl["multigrid"] and st["num_levels"] > 1) else
              ("CG preconditioned by a %d-term Chebyshev polynomial" % wl["poly"] if wl["poly"] > 1 else "Jacobi-PCG")
              + (" from a coarse-to-fine cascade" if st["num_levels"] > 1 else ""))
    line = {
        "metric": "solved lattice points/sec (assembly+CG to tol=%g)" % wl["tol"],
        "value": value, "unit": "lattice points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
        "config": {"workload": wl["text"] + ", %s to rel. residual %g" % (solver, wl["tol"]) + (
                       "; %d independent copies, one per GPU" % world if replicas else (
                           "; one lattice, %d slabs (%s scaling)" % (world, args.scaling) if world > 1 else "")),
                   "parallelism": parallelism, "iterations": iters, "rel_residual": rel,
                   "levels": st["num_levels"], "coarse_iterations": st["coarse_iterations"], "solver": solver,
                   "operator_applies": acc["applies"] // max(args.steps, 1),
                   "true_rel_residual": true_rel, "assemble_ms": acc["asm_ms"] / args.steps,
                   "solve_ms": acc["solve_ms"] / args.steps, "occupied_cells": st["num_cells"],
                   "data_rows": st["num_data_rows"]},
        "roofline": roof_main,
        "roofline_apply": roof_apply,
    }
    if world > 1 and st["iterations"] > 0:
        # what an outer iteration of the finest level costs in collectives (fi_stats counts them; the start and the
        # verification are in the totals)
        line["config"]["halo_exchanges_per_iteration"] = st["halo_exchanges"] / float(st["iterations"])
        line["config"]["allreduces_per_iteration"] = st["reductions"] / float(st["iterations"])
    if world == 1 and not args.no_cold:
        # Cold step: a FRESH context -- hipMalloc of every vector and list, the power method of the polynomial's bound
        # (fi_set_model: 16 marching launches + 2 host reads per level), the first assemble and a solve whose first look at
        # the stop flag is not scheduled by a previous solve of the same problem.  The timed steps above amortise all
        # of that (a per-frame caller does too: bipolar_2d.cpp:730); this is what a one-shot caller pays.
        def cold_step():
            torch.cuda.synchronize()
            t0c = time.perf_counter()
            cold = fi.LatticeField(wl["sizes"], dtype=wl["dtype"])
            configure(cold, wl)
            cold_out = torch.empty_like(d_out)
            step(cold, out=cold_out)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0c)
            del cold, cold_out          # (its device blocks go to the library's pool: fi_memory_pool)
            return ms

        fi.memory_pool(0)               # nothing left over from earlier contexts of this process
        line["cold_ms_per_step"] = cold_step()
        # the same once more: what a caller pays that creates and destroys a context per solve (the reference's stateless
        # solve_sparse_linear* used that way) once the pool holds the blocks of the context before
        line["cold_pooled_ms_per_step"] = cold_step()
        line["cold_note"] = ("fresh context: allocation, the polynomial's power method, first assemble and first solve "
                             "(cold_pooled: the device blocks come from the pool a destroyed context leaves); "
                             "ms_per_step is the steady state of a caller that re-solves on one context")
    if world == 1 and not args.no_accuracy:
        # the same inputs solved in fp64 to 1e-10 (V-cycle PCG where levels are available), outside the timed region
        x_run = d_out.cpu().numpy().astype(np.float64)
        ref = fi.LatticeField(wl["sizes"], dtype="f64")
        ref.add_field_constraints(wl["w"])
        if args.config == 4:
            ref.set_levels(max(wl["levels"], 1), 1e-6)
            ref.set_polynomial(4, args.poly_ratio)
        else:
            ref.set_levels(max(wl["levels"], 4), 1e-4)
            ref.set_multigrid(True)
            ref.set_mixed_precision(True)
        w = wl["w"]
        ref.add_points(w.data_pos, w.value_kernel, w.data_gradient if wl["nrm"] is not None else 0.0, w.gradient_kernel,
                       wl["pos"], wl["nrm"], None, values=wl["val"])
        ref.assemble()
        out = ref.solve_cg(None, 0, 1e-10)
        if out is not None:
            x64 = ref.solution_f64()
            line["solution_rel_err"] = float(np.abs(x_run - x64).max() / np.abs(x64).max())
            line["config"]["solution_check"] = ("||x - x64||_inf / ||x64||_inf against an fp64 solve of the same inputs to "
                                                "rel. residual %.1e (%d iterations)" % (ref.true_residual(), out[1]))
        del ref
        if args.config == 4 and out is not None:
            # The leg that meets the north-star's FIELD tolerance (values within 1e-5 of the converged solution; the
            # reference's ground truth is a double solve, sparse_linear.cpp:154-184): CG in fp64, preconditioned by one
            # fp32 V-cycle over cell-centred levels with the polynomial smoother.  Same step as the headline (clear,
            # add, assemble, solve), same inputs in HBM, timed the same way.
            if args.accurate_tol <= 0:
                side_max = max(wl["sizes"])
                args.accurate_tol = 1e-7 * min(1.0, (256.0 / side_max) ** 1.75)   # 512^3: 3e-8 (measured: field error 6.1e-6)
            acfg = dict(wl, levels=args.accurate_levels, coarse_tol=1e-5, multigrid=True, mixed=True, poly=0)
            af = fi.LatticeField(wl["sizes"], dtype="f64")
            configure(af, acfg)
            a_out = torch.empty_like(d_out)
            for _ in range(max(args.warmup, 1)):
                step(af, tol=args.accurate_tol, out=a_out)
            torch.cuda.synchronize()
            t0a = time.perf_counter()
            for _ in range(args.steps):
                _, a_it, a_rel = step(af, tol=args.accurate_tol, out=a_out)
            torch.cuda.synchronize()
            a_ms = 1e3 * (time.perf_counter() - t0a) / args.steps
            ast = af.stats()
            line["accurate"] = {
                "value": n_global / (a_ms * 1e-3), "unit": "lattice points/s", "ms_per_step": a_ms, "tol": args.accurate_tol,
                "dtype": "f64 CG (x, r, p, apply, dot products, stop test) + f32 V-cycle",
                "solver": "V-cycle PCG, %d levels (cell-centred), polynomial smoother; coarse-to-fine start" % ast["num_levels"],
                "iterations": a_it, "coarse_iterations": ast["coarse_iterations"], "true_rel_residual": af.true_residual(),
                "assemble_ms": ast["assemble_ms"], "solve_ms": ast["solve_ms"],
                "solution_rel_err": float(np.abs(af.solution_f64() - x64).max() / np.abs(x64).max())}
            del af, a_out
    if rank == 0 and world == 1 and args.cpu_side > 0 and args.config == 4:
        line["cpu_baseline"], best_effort = cpu_baseline(args.cpu_side, wl["tol"])
        if best_effort:
            line["cpu_best_effort"] = best_effort
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
