#!/usr/bin/env python3
"""bench.py -- solved lattice points / second on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path on a synthetic lattice problem whose inputs already sit in HBM:
data-constraint assembly (fi_clear_points + fi_add_points + fi_assemble) + the iterative solve (fi_solve_cg).

Workloads (--config, named in config.workload; SURVEY.md 8(d) recipes, field_interpolation_amd/synth.py):
  4 (default, the configuration BASELINE.json's metric is quoted on): 3-D 256^3 lattice, 1 M scattered noisy value
    constraints, model_2 = 0.5.  HEADLINE SOLVER (round 4): the one that meets the north-star's FIELD tolerance -- values
    within 1e-5 of the CPU reference's double solve (sparse_linear.cpp:154-184) -- fp64 CG preconditioned by an fp32
    V-cycle (a 1e-5 RESIDUAL leaves the field 2e-3 off: kappa ~ side^4).  STOP RULE (round 6, every configuration alike, no
    constant that depends on the workload): by the FIELD -- FI_OPT_FIELD_TOLERANCE = 1e-5, the solver's own estimate from
    consecutive iterates (include/fi_hip.h); `solution_rel_err` is measured against the ORACLE's fp64 solution of the same
    inputs (tests/golden/config4_256*_oracle_f64.npz).  Over slabs (--gpus N > 1) the same rule: every slab's two maxima
    travel with the r . r sum of the iteration's all-reduce (no collective of their own).
    --fast: the fp32 mode of rounds 1-3 as the line's value (coarse-to-fine start + CG preconditioned by a Chebyshev
    polynomial to residual 1e-5, field error 2e-3); the default run reports it in the `fast` sub-object.
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
                   the timed region: the Chebyshev steps of the polynomial (k_apply_march3d<..., EPI>: the model-operator
                   apply with the three-term recurrence in its epilogue) -- the V-cycle's smoother on the finest level
                   in the headline solver, the preconditioner in --fast
  "roofline_apply" the same for the full operator apply with fused data cells (the CG SpMV the north-star names; fp64 in
                   the headline solver)
  "roofline_512"   (N = 1, config 4) that fp64 apply ISOLATED at 512^3 -- the size the north-star's ">= 60 % of the HBM
                   roofline on the CG SpMV" is stated for -- on config 4's value data (8 M points) and on config 5's oriented
                   points (5 M), timed live by fi_time_apply (HIP events around 20 launches), with a plain device copy of a
                   512^3 fp64 vector timed the same way beside it: what one read stream + one write stream reach on this GPU
  "host_io"        (N = 1) the same step through the boundary's HOST buffers (field_interpolation.hpp:153-173: positions and
                   values as const float[], the solution as std::vector<float>): pinned staging, the upload of the next data
                   set and the download of the solution included in the timed region
  "roofline_assembly" SURVEY 8(d)'s assembly bytes (points in, occupied cells' blocks out, diag and Atb out) over the
                   time of fi_assemble (ALL levels of the hierarchy are built in that time)
  "solution_rel_err" ||x - x*||_inf / ||x*||_inf against the oracle's fp64 solution where the committed sample covers the
                   workload (config 4 at 256^3), else against an fp64 GPU solve to 1e-10 (said in config.solution_check)
  "fast"           config 4, N = 1: the fp32 / residual-1e-5 mode: value, ms_per_step, iterations, solution_rel_err
  "cold_ms_per_step" one step on a fresh context (allocation, power method, first assemble, first solve);
  "cold_pooled_ms_per_step" the same with the device blocks of a destroyed context (fi_memory_pool)
  "cpu_baseline"   the C++ oracle (restatement of the reference's triplets -> AtA -> BiCGSTAB path, fp32, one thread)
                   timed on the same workload at --cpu-side (default: the metric's own 256^3, about a minute; rank 0, N = 1)
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
# profiles/<round>_traffic_<kernel>_c4_<dtype>.json: the PMC passes of the shipped kernels.  The apply: round 6's passes.  The
# Chebyshev chain: round 5's launch-by-launch table -- those launches are unchanged since, and round 6's file of the same
# name is the mean over ALL launches of the two kernel names (residuals and power-method steps included), not the chain's.
TRAFFIC_ROUNDS = {"apply": ("r6", "r5"), "cheb": ("r5",)}   # (tried in turn: round 6 measured the fp64 workload only)


def measured_traffic(kind, config, side, points, dtype):
    """HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, gfx950
    x2 read correction; tools/pmc_traffic.py).  Counters cannot be read from inside this process, so the figure is the
    committed measurement of exactly this workload (TRAFFIC_ROUNDS) with the file it came from; any other workload reports null."""
    if (config, side, points) != (4, 256, 1_000_000):
        return None
    for rnd in TRAFFIC_ROUNDS.get(kind, ("r5",)):
        name = "profiles/%s_traffic_%s_c4_%s.json" % (rnd, kind, dtype)
        try:
            with open(os.path.join(ROOT, name)) as f:
                return json.load(f)["traffic_bytes"], name
        except (OSError, KeyError, ValueError):
            continue
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
                      "%d iterations to a RESIDUAL of %g -- fp32 BiCGSTAB on the explicit AtA goes no further (the GPU line above "
                      "runs its fp64 CG to the residual its metric names: not like for like, and a GPU/CPU ratio is no claim)"
                      % (side, npts, t1 - t0, t2 - t1, iters, tol)}
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


def roofline_512(fi, torch, dev):
    """The CG SpMV -- the full fp64 operator apply with fused data cells -- ISOLATED at 512^3 (the size the north-star's
    ">= 60 % of the HBM roofline on the CG SpMV" is stated for): config 4's value data at its density (8 M points) and config
    5's oriented points (5 M), fi_time_apply = HIP events around 20 back-to-back launches on the solver's stream; algorithmic
    bytes as the library counts them (2 x 8 N + the records of the occupied cells).  Beside it a plain
    device-to-device copy of one 512^3 fp64 vector timed the same way: one read stream + one write stream."""
    import numpy as np
    from field_interpolation_amd import synth
    out = {"side": 512, "dtype": "f64", "peak": HBM_PEAK_GBS, "unit": "GB/s", "target_frac": 0.60,
           "kernel": "k_apply_march3d<double>: AtA apply, matrix-free stencil + fused data cells, isolated (fi_time_apply: HIP events "
                     "around 20 launches; no solver around it)"}

    def leg(sizes, w, pos, nrm, val):
        f = fi.LatticeField(sizes, dtype="f64")
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, w.data_gradient if nrm is not None else 0.0, w.gradient_kernel, pos, nrm, None, values=val)
        f.assemble()
        f.time_apply(5)
        ms = min(f.time_apply(20) for _ in range(3))
        st = f.stats()
        ach = st["spmv_bytes"] / (ms * 1e-3) / 1e9
        return {"launch_ms": ms, "algorithmic_bytes": st["spmv_bytes"], "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                "occupied_cells": st["num_cells"], "data_rows": st["num_data_rows"]}

    sizes, w, pos, val = synth.config4(side=512, num_points=8_000_000, seed=3)
    out["value_data"] = dict(leg(sizes, w, pos, None, val), workload="config 4 at 512^3: 8 M scattered value constraints")
    del pos, val
    sizes, w5, pos5, nrm5 = synth.config5(side=512, num_points=5_000_000, seed=4)
    out["sdf_data"] = dict(leg(sizes, w5, pos5, nrm5, None), workload="config 5: 512^3 SDF from 5 M oriented points")
    del pos5, nrm5
    n = 512 ** 3
    a = torch.empty(n, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        b.copy_(a)
        e0.record()
        for _ in range(20):
            b.copy_(a)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20.0)
    out["stream_copy"] = {"what": "torch.Tensor.copy_ of one 512^3 fp64 vector (1 GiB read + 1 GiB written), torch events around 20 copies",
                          "launch_ms": best, "bytes": 2.0 * 8.0 * n, "achieved": 2.0 * 8.0 * n / (best * 1e-3) / 1e9,
                          "frac": 2.0 * 8.0 * n / (best * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del a, b
    torch.cuda.empty_cache()
    return out


def workload(args, world):
    """-> dict(sizes, weights, positions, normals, values, tol, dtype, solver settings, text)"""
    from field_interpolation_amd import bench_settings as bs
    from field_interpolation_amd import synth
    cfg = args.config
    weak = args.scaling == "weak" and world > 1
    st = bs.SETTINGS.get(cfg, {})
    if cfg == 4:
        side = args.side or 256
        depth = side * world if weak else side
        npts = args.points or int(round(1_000_000 * (side / 256.0) ** 3))
        npts = npts * world if weak else npts
        # the timed loop walks through args.datasets seeds of the workload (all resident in HBM): a re-solve on CHANGED data,
        # not the best case of a context that meets the same points again; every seed has an oracle golden at 256^3
        seeds = list(bs.CONFIG4_SEEDS[:max(1, args.datasets)])
        sets = [synth.config4(side=side, num_points=npts, seed=sd, depth=depth) for sd in seeds]
        sizes, w, pos, val = sets[0]
        more = [dict(pos=p_, nrm=None, val=v_, seed=sd) for sd, (_, _, p_, v_) in zip(seeds[1:], sets[1:])]
        text = "config4: 3D %dx%dx%d lattice, %d scattered noisy value constraints, model_2=0.5" % (sizes[0], sizes[1], sizes[2], npts)
        if args.fast:
            return dict(sizes=sizes, w=w, pos=pos, nrm=None, val=val, tol=args.tol or 1e-5, dtype=args.dtype or "f32", by_field=False,
                        levels=(3 if args.multigrid else 1) if args.levels is None else args.levels,
                        coarse_tol=args.coarse_tol or 1e-5, multigrid=args.multigrid, mixed=False,
                        poly=0 if args.multigrid else args.poly, points=npts, text=text, field_tol=None, more=more, seed=seeds[0])
        # the solver that meets the north-star's FIELD tolerance: fp64 CG + fp32 V-cycle.  The residual that buys a field
        # within 1e-5 tightens with the lattice (field error per unit of residual: 60 at 256^3, 190 at 512^3)
        dt = args.dtype or "f64"
        # The stop rule (field_interpolation_amd/bench_settings.py): by the FIELD -- the solver's estimate, the same rule for
        # every configuration, size and number of slabs (up to 16: the slabs' maxima travel with the r.r sum); at a residual
        # when --tol names one
        by_field = world <= bs.FIELD_RULE_MAX_SLABS and not args.tol
        tol = args.tol or (st["tol"] if by_field else bs.slab_residual(4, sizes))
        return dict(sizes=sizes, w=w, pos=pos, nrm=None, val=val, tol=tol, dtype=dt, by_field=by_field,
                    levels=st["levels"] if args.levels is None else args.levels, coarse_tol=args.coarse_tol or st["coarse_tol"],
                    multigrid=True, mixed=dt == "f64", poly=0, points=npts, text=text, field_tol=bs.FIELD_TOLERANCE, more=more,
                    seed=seeds[0], kcycle=(st.get("kcycle", 0) if args.kcycle is None else args.kcycle) if world == 1 else 0, cheb=st.get("cheb"))
    if cfg == 5:
        side = args.side or 512
        npts = args.points or int(round(5_000_000 * (side / 512.0) ** 2))
        if weak:
            raise SystemExit("config 5 is a fixed lattice: use --scaling strong")
        sizes, w, pos, nrm = synth.config5(side=side, num_points=npts, seed=4)
        return dict(sizes=sizes, w=w, pos=pos, nrm=nrm, val=None, tol=args.tol or st["tol"], dtype=args.dtype or "f64",
                    by_field=world <= bs.FIELD_RULE_MAX_SLABS and not args.tol,
                    levels=st["levels"] if args.levels is None else args.levels, coarse_tol=args.coarse_tol or st["coarse_tol"],
                    multigrid=True, mixed=(args.dtype or "f64") == "f64", poly=0, points=npts, kcycle=(st.get("kcycle", 0) if args.kcycle is None else args.kcycle) if world == 1 else 0, cheb=st.get("cheb"), field_tol=bs.FIELD_TOLERANCE, more=[], seed=4,
                    text="config5: 3D %d^3 SDF from %d oriented points (sdf_from_points, default Weights)" % (side, npts))
    if cfg == 3:
        side = args.side or 4096
        pps = (args.points or 200_000) // 2
        if weak:
            raise SystemExit("config 3 is a fixed lattice: use --scaling strong")
        sizes, w, pos, nrm = synth.config3(side=side, points_per_shape=pps, seed=2)
        return dict(sizes=sizes, w=w, pos=pos, nrm=nrm, val=None, tol=args.tol or st["tol"], dtype=args.dtype or "f64",
                    by_field=world <= bs.FIELD_RULE_MAX_SLABS and not args.tol,
                    levels=st["levels"] if args.levels is None else args.levels, coarse_tol=args.coarse_tol or st["coarse_tol"],
                    multigrid=True, mixed=(args.dtype or "f64") == "f64", poly=0, points=2 * pps, kcycle=(st.get("kcycle", 0) if args.kcycle is None else args.kcycle) if world == 1 else 0, cheb=st.get("cheb"), field_tol=bs.FIELD_TOLERANCE, more=[], seed=2,
                    text="config3: 2D %dx%d SDF from %d oriented points (triangle + inverted circle)" % (side, side, 2 * pps))
    if cfg == 2:
        side = args.side or 1024
        npts = args.points or 10_000
        if weak:
            raise SystemExit("config 2 is a fixed lattice: use --scaling strong")
        sizes, w, pos, val = synth.config2(side=side, num_points=npts, seed=1)
        return dict(sizes=sizes, w=w, pos=pos, nrm=None, val=val, tol=args.tol or st["tol"], dtype=args.dtype or "f64",
                    by_field=world <= bs.FIELD_RULE_MAX_SLABS and not args.tol,
                    levels=st["levels"] if args.levels is None else args.levels, coarse_tol=args.coarse_tol or st["coarse_tol"],
                    multigrid=True, mixed=(args.dtype or "f64") == "f64", poly=0, points=npts, kcycle=(st.get("kcycle", 0) if args.kcycle is None else args.kcycle) if world == 1 else 0, cheb=st.get("cheb"), field_tol=bs.FIELD_TOLERANCE, more=[], seed=1,
                    text="config2: 2D %dx%d lattice, %d noisy value constraints, model_2=10" % (side, side, npts))
    raise SystemExit("--config must be 2, 3, 4 or 5 (config 1 is the CPU-runnable 1-D case: tests/)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)  # (a context's first two assembles still allocate)
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
    ap.add_argument("--cpu-side", type=int, default=256,
                    help="lattice side of the CPU baseline (default: the metric's own 256^3, about a minute of host time; 0: skip)")
    ap.add_argument("--fast", action="store_true",
                    help="config 4: the fp32 / residual-1e-5 mode of rounds 1-3 as the line's value (field error 2e-3: outside "
                         "the north-star's tolerance); the default is the solver that meets it")
    ap.add_argument("--levels", type=int, default=None, help="coarser levels (config 4: coarse-to-fine start; 0: none)")
    ap.add_argument("--kcycle", type=int, default=None, help="FI_OPT_MG_KCYCLE: coarse levels corrected by two flexible-CG steps (default: the configuration's setting; "
                    "one GPU only -- over slabs every coarse dot product would be an all-reduce, those runs keep the V-cycle and its smoother)")
    ap.add_argument("--coarse-tol", type=float, default=0.0)
    ap.add_argument("--multigrid", action="store_true", help="config 4: V-cycle preconditioned CG")
    ap.add_argument("--poly", type=int, default=4, help="terms of the Chebyshev polynomial preconditioner (0: Jacobi-PCG)")
    ap.add_argument("--poly-ratio", type=float, default=30.0)
    ap.add_argument("--no-accuracy", action="store_true",
                    help="skip the comparison with the reference solution (solution_rel_err) and the `fast` sub-object")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-step figure (fresh context)")
    ap.add_argument("--no-roofline-512", action="store_true", help="skip the isolated 512^3 apply (roofline_512)")
    ap.add_argument("--no-host-io", action="store_true", help="skip the step through host buffers (host_io)")
    ap.add_argument("--datasets", type=int, default=3,
                    help="config 4: the timed steps walk through this many seeds of the workload (1-3; every one has an oracle "
                         "golden at 256^3): a re-solve on changed data each step.  1: the same points every step")
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
        if cfg.get("by_field"):
            from field_interpolation_amd import bench_settings as bs
            field.set_field_tolerance(bs.FIELD_TOLERANCE)
        if cfg.get("kcycle", 0) > 0 and cfg["multigrid"] and cfg["levels"] > 0:
            field.set_kcycle(cfg["kcycle"])
            if cfg.get("cheb"):
                field.set_cheb_smoother(*cfg["cheb"])

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
        data = []
        for ds in [dict(pos=wl["pos"], nrm=wl["nrm"], val=wl["val"], seed=wl["seed"])] + wl["more"]:
            z = ds["pos"].reshape(-1, ndim)[:, ndim - 1]
            keep = (z >= zlo) & (z < zhi) if (slabs and world > 1) else np.ones(len(z), bool)
            data.append(dict(
                seed=ds["seed"],
                pos=torch.from_numpy(np.ascontiguousarray(ds["pos"][keep])).to(dev),
                nrm=torch.from_numpy(np.ascontiguousarray(ds["nrm"][keep])).to(dev) if ds["nrm"] is not None else None,
                val=torch.from_numpy(np.ascontiguousarray(ds["val"][keep])).to(dev) if ds["val"] is not None else None))
        d_out = torch.empty(field.num_owned, dtype=torch.float32, device=dev)   # the solution stays in HBM
        torch.cuda.synchronize()
        w = wl["w"]
        counter = [0]

        def step(f=None, tol=None, out=None, which=None):
            """one pass of the hot path; which: the data set (default: the next one in turn)"""
            if which is None:
                which = counter[0] % len(data)
                counter[0] += 1
            d = data[which]
            f = field if f is None else f
            f.clear_points()
            f.add_points(w.data_pos, w.value_kernel, w.data_gradient if d["nrm"] is not None else 0.0, w.gradient_kernel,
                         d["pos"], d["nrm"], None, values=d["val"])
            f.assemble()
            res = f.solve_cg(None, 0, wl["tol"] if tol is None else tol, out=d_out if out is None else out)
            if res is None:
                raise RuntimeError("CG breakdown")
            return res
        step.data = data
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
        # (set-up, not a warm-up step: every data set once, so that the context's buffers have met their largest sizes)
        for _ in range((len(step.data) if len(step.data) > 1 else 0) + args.warmup):
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
        return (t[0], "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not read in this run)" % t[1]) if t else (None, None)

    mg_run = wl["multigrid"] and st["num_levels"] > 1
    apply_name = ("k_apply_march3d<%s>: AtA apply, matrix-free stencil + fused data cells (finest level%s)" % (
        "double" if wl["dtype"] == "f64" else "float", ": the SpMV of the fp64 CG" if wl["mixed"] else "") if ndim == 3 else
                  "k_apply_tile2d: AtA apply, matrix-free stencil + fused data cells (finest level)")
    roof_apply = roof(apply_name, st["spmv_bytes"], spmv_avg_ms, acc["spmv_n"], tr("apply"))
    # the dominant kernel of the timed region: the Chebyshev steps of the polynomial -- the V-cycle's smoother on the finest
    # level (two chains of terms - 1 launches per cycle against two residuals and one fp64 apply), or the preconditioner of
    # the polynomial PCG (terms - 1 launches per outer iteration against one full apply)
    if acc["prec_n"] > 0 and (mg_run or wl["poly"] > 2):
        roof_main = roof("k_apply_march3d<float, EPI>: Chebyshev steps of the polynomial (%s) = model-operator apply + three-term "
                         "recurrence in the epilogue; ALL steps of the sampled polynomials, timed between one pair of HIP events "
                         "per polynomial; bytes as the library counts what it launched -- with the iterates of the V-cycle's "
                         "polynomial stored as bfloat16 (undivided fp32 levels of >= 2^21 points) 8 / 10 / 12 / 14 bytes per point "
                         "for the four steps of 5 terms (r, the bf16 scaling, z, z_prev in, z_new out), 10 / 14 / 18 / 18 with fp32 "
                         "iterates (FI_NO_Z16, the polynomial PCG): bytes of all sampled launches over their time, launch_ms = "
                         "the mean per launch, finest level"
                         % ("the V-cycle's pre-smoother" if mg_run else "the preconditioner of the polynomial PCG"),
                         st["prec_bytes"], prec_avg_ms, acc["prec_n"], tr("cheb"))
    else:
        roof_main = roof_apply
    solver = ("V-cycle PCG" + (" (fp64 CG, fp32 V-cycle)" if wl["mixed"] else "") if mg_run else
              ("CG preconditioned by a %d-term Chebyshev polynomial" % wl["poly"] if wl["poly"] > 1 else "Jacobi-PCG")
              + (" from a coarse-to-fine cascade" if st["num_levels"] > 1 else ""))
    # SURVEY.md 8(d), assembly: read P (2D + 1) 4 bytes of points (positions, a value or a normal), write the occupied
    # cells' blocks (4 + s 2^D (2^D + 1) / 2 each), write diag and Atb (2 s N)
    s_el = 8.0 if wl["dtype"] == "f64" else 4.0
    asm_ms = acc["asm_ms"] / args.steps
    asm_bytes = (wl["points"] / (world if not replicas else 1) * (2 * ndim + 1) * 4.0 +
                 st["num_cells"] * (4.0 + s_el * (2 ** ndim) * (2 ** ndim + 1) / 2.0) + 2.0 * s_el * st["num_unknowns"])
    roof_asm = {"bound": "hbm", "achieved": asm_bytes / (asm_ms * 1e-3) / 1e9 if asm_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "traffic": None, "algorithmic_bytes": asm_bytes, "ms": asm_ms,
                "what": "fi_assemble of the finest level (SURVEY 8(d): points in, blocks of the occupied cells out, diag and "
                        "Atb out) over the time of the WHOLE fi_assemble -- sort, lists and every coarser level included; "
                        "a chain of sorts, scans and small launches, not one kernel"}
    roof_asm["frac"] = roof_asm["achieved"] / HBM_PEAK_GBS
    line = {
        "metric": "solved lattice points/sec (assembly+CG to %s)" % (
            "field within 1e-5 of the CPU reference by the solver's estimate: ended at rel. residual %.1e" % rel if wl.get("by_field") else (
                "field within 1e-5 of the CPU reference: rel. residual %.0e" % wl["tol"] if wl["field_tol"] and args.config == 4 else "tol=%g" % wl["tol"])),
        "value": value, "unit": "lattice points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None, "dtype": wl["dtype"], "data": "synthetic",
        "config": {"workload": wl["text"] + ", %s %s" % (solver, "stopped by the field (FI_OPT_FIELD_TOLERANCE = %g)" % wl["field_tol"]
                                                         if wl.get("by_field") else "to rel. residual %g" % wl["tol"]) + (
                       "; %d independent copies, one per GPU" % world if replicas else (
                           "; one lattice, %d slabs (%s scaling)" % (world, args.scaling) if world > 1 else "")),
                   "parallelism": parallelism, "iterations": iters, "rel_residual": rel,
                   "data_sets": ("the timed steps walk through %d data sets of the workload (synth seeds %s), all resident in HBM: "
                                 "every step re-solves on changed points and values" % (len(step.data), ", ".join(str(d["seed"]) for d in step.data))
                                 if len(step.data) > 1 else "one data set (seed %d), re-solved every step" % step.data[0]["seed"]),
                   "levels": st["num_levels"], "coarse_iterations": st["coarse_iterations"], "solver": solver,
                   "stop_rule": ("by the field: the last step ||x_k - x_(k-1)||_inf times sigma / (1 - sigma), sigma the slowest mean decay of the "
                                 "residual norm over the recent windows and the whole solve, "
                                 "x 2, <= %g x max |x| (FI_OPT_FIELD_TOLERANCE, include/fi_hip.h) -- every configuration alike"
                                 % wl["field_tol"]) if wl.get("by_field") else "relative residual <= %g" % wl["tol"],
                   "field_estimate": st["field_estimate"] if wl.get("by_field") else None,
                   "field_per_residual": st["field_per_residual"] if wl.get("by_field") else None,
                   "arithmetic": ("fp64: x, r, p, the operator apply, p . A p, r . r and the stop test; fp32: the V-cycle preconditioner, "
                                  "and r . z = t^2 (b . x of the fp32 cycle, summed four products at a time in float, then in double) "
                                  "from the cycle's last launch (FI_NO_TWIN_DOT: a pass in fp64 instead)" if wl["mixed"] else wl["dtype"]),
                   "operator_applies": acc["applies"] // max(args.steps, 1),
                   "true_rel_residual": true_rel, "assemble_ms": asm_ms,
                   "solve_ms": acc["solve_ms"] / args.steps, "occupied_cells": st["num_cells"],
                   "data_rows": st["num_data_rows"]},
        "roofline": roof_main,
        "roofline_apply": roof_apply,
        "roofline_assembly": roof_asm,
    }
    if world > 1:
        # what the transport looks like from INSIDE the library (fi_comm_info: RCCL's own count of the ranks, this rank's
        # index and device) and what an iteration of the finest level costs in collectives
        try:
            ci = field.comm_info()
            line["config"]["rccl_ranks"] = ci["ranks"]
            line["config"]["transport"] = ci["transport"]
            line["config"]["rank0_device"] = ci["device"]
            line["config"]["halo_planes_per_exchange"] = ci["halo_planes"]
            line["config"]["halo_bytes_per_exchange_and_neighbour"] = ci["halo_bytes_per_exchange"]
            devs = [None] * world
            dist.all_gather_object(devs, (rank, ci["rank"], ci["device"], ci["ranks"]))
            line["config"]["ranks_seen_by_rccl"] = [list(d) for d in devs]     # (launcher rank, RCCL rank, device, RCCL count)
        except Exception as e:      # noqa: BLE001 -- a diagnostic must not cost the measurement
            line["config"]["rccl_ranks"] = "unavailable: %s" % e
        if st["iterations"] > 0:
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
        from field_interpolation_amd import bench_settings as bs
        from field_interpolation_amd import synth
        gdir = os.path.join(ROOT, "tests", "golden")

        def against_sample(x, g):
            sizes_g = [int(v) for v in g["sizes"]]
            sd = int(g["stride"])
            got = np.asarray(x, np.float64).reshape(sizes_g[::-1])[tuple(slice(0, None, sd) for _ in sizes_g)]
            return float(np.abs(got - g["sample"]).max() / float(g["field_maxabs"]))

        def golden_note(g, name):
            how = ("Jacobi-PCG to a true residual of %.1e in %d iterations" % (float(g["true_rel_residual"]), int(g["iterations"]))
                   if int(g["iterations"]) > 0 else
                   "banded Cholesky in fp64 (the route of solve_sparse_linear_exact, sparse_linear.cpp:154-184), true residual %.1e"
                   % float(g["true_rel_residual"]))
            return ("||x - x*||_inf / ||x*||_inf against the ORACLE's fp64 solution of the same inputs (the reference's rows, explicit "
                    "AtA, %s; tests/golden/%s: every %dth point per axis, %d values)" % (how, name, int(g["stride"]), g["sample"].size))

        def run_solution(f):
            return f.solution_f64() if wl["dtype"] == "f64" else d_out.cpu().numpy().astype(np.float64)

        def gpu_reference(sizes_r, w_r, pos_r, nrm_r, val_r, levels_r):
            """the same inputs solved in fp64 to 1e-10 on the GPU (V-cycle PCG), outside the timed region"""
            ref = fi.LatticeField(sizes_r, dtype="f64")
            ref.add_field_constraints(w_r)
            bs.configure(ref, levels_r, 1e-5 if args.config == 4 else 1e-4)
            ref.add_points(w_r.data_pos, w_r.value_kernel, w_r.data_gradient if nrm_r is not None else 0.0, w_r.gradient_kernel,
                           pos_r, nrm_r, None, values=val_r)
            ref.assemble()
            out = ref.solve_cg(None, 0, 1e-10)
            return (ref.solution_f64(), ref.true_residual(), out[1]) if out is not None else None

        x64 = None
        use_golden = False
        default_size = (args.config == 4 and len(set(wl["sizes"])) == 1 and
                        wl["points"] == int(round(1_000_000 * (wl["sizes"][0] / 256.0) ** 3)))
        if default_size:
            # every data set of the timed loop, solved once more (untimed) and compared with the oracle's solution of ITS seed
            errs = {}
            for k, d in enumerate(step.data):
                name = ("config4_%d_oracle_f64.npz" % wl["sizes"][0]) if d["seed"] == 3 else (
                    "config4_%d_seed%d_oracle_f64.npz" % (wl["sizes"][0], d["seed"]))
                if not os.path.exists(os.path.join(gdir, name)):
                    continue
                g = np.load(os.path.join(gdir, name))
                step(which=k)
                errs[d["seed"]] = (against_sample(run_solution(field), g), field.true_residual(), name, g)
            if errs:
                use_golden = True
                worst = max(errs, key=lambda sd: errs[sd][0])
                line["solution_rel_err"] = errs[worst][0]
                line["config"]["solution_rel_err_by_seed"] = {str(sd): e[0] for sd, e in errs.items()}
                line["config"]["true_rel_residual_by_seed"] = {str(sd): e[1] for sd, e in errs.items()}
                line["config"]["solution_check"] = (golden_note(errs[worst][3], errs[worst][2]) + "; the worst of the %d data sets "
                                                    "the timed steps walk through (seeds %s), each against its own golden"
                                                    % (len(errs), ", ".join(str(sd) for sd in errs)))
        elif args.config == 2 and wl["sizes"] == [1024, 1024] and wl["points"] == 10_000 and os.path.exists(
                os.path.join(gdir, "config2_1024_oracle_f64.npz")):
            g = np.load(os.path.join(gdir, "config2_1024_oracle_f64.npz"))
            use_golden = True
            line["solution_rel_err"] = against_sample(run_solution(field), g)
            line["config"]["solution_check"] = golden_note(g, "config2_1024_oracle_f64.npz")
        if not use_golden:
            if len(step.data) > 1:
                step(which=0)      # (the reference below solves the first data set: the field must hold ITS solution)
            got = gpu_reference(wl["sizes"], wl["w"], wl["pos"], wl["nrm"], wl["val"],
                                max(wl["levels"], 3 if ndim == 3 and args.config == 4 else 4))
            if got is not None:
                x64, ref_rel, ref_it = got
                line["solution_rel_err"] = float(np.abs(run_solution(field) - x64).max() / np.abs(x64).max())
                line["config"]["solution_check"] = ("||x - x64||_inf / ||x64||_inf against an fp64 GPU solve of the same inputs to "
                                                    "rel. residual %.1e (%d iterations): the oracle cannot solve this size" % (ref_rel, ref_it))
        # What the configuration's residual buys in the FIELD, against the oracle: the same solver settings on the largest
        # size of the configuration's shape the oracle has solved (config 2: the full size; config 3: 1024^2; config 5: 128^3),
        # at the configuration's residual and at the residual that brings the field within 1e-5
        shape = {2: ("config2_1024_oracle_f64.npz", lambda g: synth.config2(side=1024, num_points=int(g["num_points"]), seed=1)),
                 3: ("config3_1024_oracle_f64.npz", lambda g: synth.config3(side=1024, points_per_shape=int(g["num_points"]) // 2, seed=2)),
                 5: ("config5_128_oracle_f64.npz", lambda g: synth.config5(side=128, num_points=int(g["num_points"]), seed=4))}
        if args.config in shape and os.path.exists(os.path.join(gdir, shape[args.config][0])):
            name, make = shape[args.config]
            g = np.load(os.path.join(gdir, name))
            sz, w_s, pos_s, second = make(g)
            nrm_s, val_s = (None, second) if args.config == 2 else (second, None)
            fs = fi.LatticeField(sz, dtype="f64")
            fs.add_field_constraints(w_s)
            side_ratio = max(wl["sizes"]) // max(sz)
            lv = max(1, wl["levels"] - int(round(np.log2(max(side_ratio, 1)))))     # the same coarsest lattice
            bs.configure(fs, lv, wl["coarse_tol"], by_field=wl.get("by_field", False), kcycle=wl.get("kcycle", 0), cheb=wl.get("cheb"))
            fs.add_points(w_s.data_pos, w_s.value_kernel, w_s.data_gradient if nrm_s is not None else 0.0, w_s.gradient_kernel,
                          pos_s, nrm_s, None, values=val_s)
            fs.assemble()
            res = fs.solve_cg(None, 0, wl["tol"])
            sts = fs.stats()
            err = against_sample(fs.solution_f64(), g) if res is not None else float("nan")
            line["config"]["field_accuracy_against_oracle"] = {
                "lattice": sz, "levels": lv, "golden": "tests/golden/" + name, "what": golden_note(g, name),
                "stop_rule": line["config"]["stop_rule"], "iterations": res[1] if res else -1, "rel_residual": sts["rel_residual"],
                "true_rel_residual": fs.true_residual(), "field_estimate": sts["field_estimate"], "solution_rel_err": err,
                "field_tolerance_met": bool(err <= bs.FIELD_TOLERANCE)}
            if side_ratio > 1:
                line["config"]["field_tolerance_met_on_the_oracles_size"] = bool(err <= bs.FIELD_TOLERANCE)
            del fs
        if wl["field_tol"]:
            line["config"]["field_tolerance"] = wl["field_tol"]
            line["config"]["field_tolerance_met"] = bool(line.get("solution_rel_err", 1.0) <= wl["field_tol"])
            line["config"]["field_tolerance_checked_against"] = "the oracle" if use_golden else "an fp64 GPU solve to 1e-10"
        if args.config == 4 and not args.fast:
            # The fp32 mode of rounds 1-3 beside it: coarse-to-fine start over one coarser level + CG preconditioned by the
            # 4-term Chebyshev polynomial, to a RESIDUAL of 1e-5 -- twice as fast, the field 2e-3 off.  Same step, same
            # inputs in HBM, timed the same way.
            fcfg = dict(wl, dtype="f32", levels=1, coarse_tol=1e-5, multigrid=False, mixed=False, poly=args.poly, tol=1e-5, by_field=False)
            ff = fi.LatticeField(wl["sizes"], dtype="f32")
            configure(ff, fcfg)
            f_out = torch.empty_like(d_out)
            for _ in range(max(args.warmup, 1)):
                step(ff, tol=1e-5, out=f_out)
            torch.cuda.synchronize()
            t0f = time.perf_counter()
            for _ in range(args.steps):
                _, f_it, f_rel = step(ff, tol=1e-5, out=f_out)
            torch.cuda.synchronize()
            f_ms = 1e3 * (time.perf_counter() - t0f) / args.steps
            fst = ff.stats()
            f_true = ff.true_residual()
            step(ff, tol=1e-5, out=f_out, which=0)    # (the first data set once more, untimed: the reference solutions are ITS)
            xf = f_out.cpu().numpy().astype(np.float64)
            if use_golden:
                f_err = against_sample(xf, np.load(os.path.join(gdir, "config4_%d_oracle_f64.npz" % wl["sizes"][0])))
            else:
                f_err = float(np.abs(xf - x64).max() / np.abs(x64).max()) if x64 is not None else None
            line["fast"] = {
                "value": n_global / (f_ms * 1e-3), "unit": "lattice points/s", "ms_per_step": f_ms, "tol": 1e-5, "dtype": "f32",
                "solver": "CG preconditioned by a %d-term Chebyshev polynomial from a coarse-to-fine cascade (1 coarser level)" % args.poly,
                "iterations": f_it, "coarse_iterations": fst["coarse_iterations"], "true_rel_residual": f_true,
                "assemble_ms": fst["assemble_ms"], "solve_ms": fst["solve_ms"], "solution_rel_err": f_err,
                "note": "a residual of 1e-5 leaves the field 2e-3 off (kappa ~ side^4): outside the north-star's field tolerance",
                "roofline": roof("k_apply_march3d<float, EPI>: Chebyshev steps of the polynomial preconditioner",
                                 fst["prec_bytes"], fst["prec_ms_avg"], fst["prec_samples"], (None, None)),
                "roofline_apply": roof("k_apply_march3d<float>: AtA apply with fused data cells", fst["spmv_bytes"],
                                       fst["spmv_ms_avg"], fst["spmv_samples"], (None, None))}
            del ff, f_out
    if world == 1 and args.config == 4 and wl.get("by_field") and not args.no_accuracy and not args.side:
        # Beside the rule: the residual rounds 4-5 had calibrated against the oracle for exactly this workload (3e-7: five
        # iterations, field 2.9e-6 off) -- what the headline cost before round 6 made the stop rule the same for every
        # workload.  Same steps, same data sets, the field rule switched off.
        field.set_field_tolerance(0.0)
        for _ in range(len(step.data)):
            step(tol=3e-7)
        torch.cuda.synchronize()
        t0r = time.perf_counter()
        for _ in range(args.steps):
            _, r_it, r_rel = step(tol=3e-7)
        torch.cuda.synchronize()
        r_ms = 1e3 * (time.perf_counter() - t0r) / args.steps
        line["calibrated_residual_r5"] = {
            "ms_per_step": r_ms, "value": n_global / (r_ms * 1e-3), "unit": "lattice points/s", "rel_residual_rule": 3e-7, "iterations": r_it,
            "rel_residual": r_rel,
            "note": "the stop rule of rounds 4-5 for THIS workload only (a residual calibrated against the oracle's solution of it); "
                    "not the line's value: the rule above is the same for every configuration and knows no workload"}
        field.set_field_tolerance(wl["field_tol"])
    if world == 1 and not args.no_host_io:
        # The boundary hands over HOST buffers (field_interpolation.hpp:153-173: `const float positions[]`, the solution a
        # std::vector<float>; the C ABI takes either kind).  The same step with its host traffic inside the timed region:
        # every data set lives in pinned host memory, is uploaded for its step, and the solution comes back into pinned host
        # memory.  `serial`: upload, step, download, one after the other.  `pipelined`: what a caller that re-solves does --
        # the upload of the NEXT data set and the download of the PREVIOUS solution run on a second stream beside the step
        # (two device buffers each); the last solution's download is inside the region.
        def host_io():
            w = wl["w"]
            nset = len(step.data)
            h_in = []
            for d in step.data:
                h_in.append({k: (d[k].cpu().pin_memory() if d[k] is not None else None) for k in ("pos", "nrm", "val")})
            n_own = field.num_owned
            h_out = [torch.empty(n_own, dtype=torch.float32).pin_memory() for _ in range(2)]
            d_in = [{k: (torch.empty_like(step.data[0][k]) if step.data[0][k] is not None else None) for k in ("pos", "nrm", "val")} for _ in range(2)]
            d_o = [torch.empty(n_own, dtype=torch.float32, device=dev) for _ in range(2)]
            side = torch.cuda.Stream(device=dev)

            def upload(slot, which, stream=None):
                for k in ("pos", "nrm", "val"):
                    if d_in[slot][k] is not None:
                        if h_in[which][k].numel() != d_in[slot][k].numel():
                            d_in[slot][k] = torch.empty(h_in[which][k].shape, dtype=h_in[which][k].dtype, device=dev)
                        d_in[slot][k].copy_(h_in[which][k], non_blocking=True)

            def run(slot):
                d = d_in[slot]
                field.clear_points()
                field.add_points(w.data_pos, w.value_kernel, w.data_gradient if d["nrm"] is not None else 0.0, w.gradient_kernel,
                                 d["pos"], d["nrm"], None, values=d["val"])
                field.assemble()
                if field.solve_cg(None, 0, wl["tol"], out=d_o[slot]) is None:
                    raise RuntimeError("CG breakdown")

            def timed(fn, reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t0) / reps

            for hb in h_out:            # (first touch of the pinned pages)
                hb.copy_(d_o[0], non_blocking=True)
            upload(0, 0)
            h2d_ms = timed(lambda: upload(0, 0), 5)
            d2h_ms = timed(lambda: h_out[0].copy_(d_o[0], non_blocking=True), 5)
            k = [0]

            def serial():
                upload(0, k[0] % nset)
                torch.cuda.synchronize()
                run(0)
                h_out[0].copy_(d_o[0], non_blocking=True)
                torch.cuda.synchronize()
                k[0] += 1
            serial()
            serial_ms = timed(serial, args.steps)
            # pipelined
            upload(0, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                cur, nxt = i % 2, (i + 1) % 2
                with torch.cuda.stream(side):
                    if i + 1 < args.steps:
                        upload(nxt, (i + 1) % nset)
                    if i >= 1:
                        h_out[nxt].copy_(d_o[nxt], non_blocking=True)     # (the previous step's solution sits in the other slot)
                run(cur)
                torch.cuda.synchronize()
            h_out[(args.steps - 1) % 2].copy_(d_o[(args.steps - 1) % 2], non_blocking=True)
            torch.cuda.synchronize()
            pipe_ms = 1e3 * (time.perf_counter() - t0) / args.steps
            return {"ms_per_step": pipe_ms, "ms_per_step_serial": serial_ms, "h2d_ms": h2d_ms, "d2h_ms": d2h_ms,
                    "device_resident_ms_per_step": 1e3 * elapsed / args.steps,
                    "h2d_bytes": sum(v.numel() * v.element_size() for v in h_in[0].values() if v is not None),
                    "d2h_bytes": n_own * 4,
                    "what": "the timed step with its host traffic inside: data sets in pinned host memory, the solution into pinned host "
                            "memory (float32, as the reference returns it).  serial: upload, step, download in turn; ms_per_step: the "
                            "next upload and the previous download on a second stream beside the step, the last download inside the region"}
        try:
            line["host_io"] = host_io()
        except Exception as e:      # noqa: BLE001 -- a diagnostic leg must not cost the measurement
            line["host_io"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and args.config == 4 and not args.no_roofline_512 and not args.side:
        line["roofline_512"] = roofline_512(fi, torch, dev)
    if rank == 0 and world == 1 and args.cpu_side > 0 and args.config == 4:
        line["cpu_baseline"], best_effort = cpu_baseline(args.cpu_side, 1e-5)
        if best_effort:
            line["cpu_best_effort"] = best_effort
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
