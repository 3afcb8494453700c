#!/usr/bin/env python3
"""The CPU baseline at the headline size: config 4 at 256^3 -- 1 M scattered value constraints,
model_2 = 0.5, tol 1e-5 -- through the oracle's restatement of the reference path (triplets -> CSC -> explicit AtA ->
Jacobi-preconditioned BiCGSTAB, fp32, ONE thread), and the same rows by the matrix-free OpenMP Jacobi-PCG.
Offline (about 10-20 minutes of host time); writes profiles/r2_cpu_baseline_256.json.  bench.py's own cpu_baseline leg
stays a 112^3 sample so that the default run finishes in minutes."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                            # noqa: E402
from bench import host_cores                                 # noqa: E402
from field_interpolation_amd import synth                     # noqa: E402
from oracle import fi_oracle as fo                            # noqa: E402

side = int(os.environ.get("SIDE", "256"))
tol = 1e-5
stop = False


def heartbeat():
    t0 = time.time()
    while not stop:
        time.sleep(45)
        print("... %d s" % (time.time() - t0), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
npts = int(round(1_000_000 * (side / 256.0) ** 3))
sizes, w, pos, val = synth.config4(side=side, num_points=npts, seed=3)
t0 = time.perf_counter()
f = fo.LatticeField(sizes)
f.add_field_constraints(fo.Weights(model_2=w.model_2))
f.add_value_constraints(pos, val, w.data_pos)
t1 = time.perf_counter()
print("assembly %.1f s" % (t1 - t0), flush=True)
res = f.solve_with_guess(np.zeros(f.num_unknowns, np.float32), 0, tol)
t2 = time.perf_counter()
out = {"workload": "config 4 at %d^3: %d scattered noisy value constraints, model_2=0.5, tol %g" % (side, npts, tol),
       "port": {"kind": "port", "cores": 1, "assembly_s": t1 - t0, "solve_s": t2 - t1, "iterations": res[1] if res else -1,
                "value": f.num_unknowns / (t2 - t0), "unit": "lattice points/s",
                "what": "oracle restatement of sparse_linear.cpp: triplets -> CSC -> explicit AtA -> BiCGSTAB + diagonal "
                        "preconditioner, fp32, one thread"}}
print(json.dumps(out["port"]), flush=True)
cores = host_cores()
t3 = time.perf_counter()
best = f.solve_pcg_rows_omp(np.zeros(f.num_unknowns, np.float32), 0, tol, cores)
t4 = time.perf_counter()
if best:
    out["best_effort"] = {"kind": "matrix-free Jacobi-PCG on the oracle's rows (OpenMP), not the reference's algorithm",
                          "cores": cores, "setup_s": best[3], "solve_s": best[4], "iterations": best[1],
                          "value": f.num_unknowns / ((t1 - t0) + (t4 - t3)), "unit": "lattice points/s"}
    print(json.dumps(out["best_effort"]), flush=True)
stop = True
dst = os.path.join(ROOT, "gpurun_out" if os.environ.get("GRAFT_REPO_ROOT") else "profiles", "r2_cpu_baseline_%d.json" % side)
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
