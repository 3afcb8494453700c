#!/usr/bin/env python3
"""FI_OPT_FIELD_TOLERANCE on every configuration that has an oracle golden (config 4: three seeds at 256^3; config 2 at full
size; config 3's shape at 1024^2; config 5's shape at 128^3): solves in turn on ONE context per configuration (the second and
third solve use the context's own calibration), each against the oracle's solution.  profiles/r6_field_rule.txt"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import bench_settings as bs
from field_interpolation_amd import synth

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def against(x, g):
    sizes = [int(s) for s in g["sizes"]]
    sd = int(g["stride"])
    got = np.asarray(x, np.float64).reshape(sizes[::-1])[tuple(slice(0, None, sd) for _ in sizes)]
    return float(np.abs(got - g["sample"]).max() / float(g["field_maxabs"]))


def run(name, f, adds, goldens, first_tol, repeats=3):
    for k in range(repeats):
        for add, g in zip(adds, goldens):
            f.clear_points()
            add(f)
            f.assemble()
            t0 = time.perf_counter()
            res = f.solve_cg(None, 0, first_tol)
            ms = 1e3 * (time.perf_counter() - t0)
            st = f.stats()
            err = against(f.solution_f64(), g)
            print("%-22s pass %d: %2d iterations in %d round(s), stop residual %.2e (reached %.2e), estimate %.2e, TRUE field error "
                  "%.2e %s, error per residual %.1f, solve %.2f ms (wall %.1f)" % (
                      name, k, st["iterations"], st["field_rounds"], st["stop_residual"], st["rel_residual"], st["field_estimate"], err,
                      "ok" if err <= 1e-5 else "MISSED", st["field_per_residual"], st["solve_ms"], ms), flush=True)


tol = float(os.environ.get("FIELD_TOL", "1e-5"))
# config 4, three seeds in turn on one context
sets = [synth.config4(seed=sd) for sd in bs.CONFIG4_SEEDS]
sizes, w = sets[0][0], sets[0][1]
f = bs.headline_field(fi, 4, sizes, w)
f.set_field_tolerance(tol)
adds = [(lambda ff, p=p, v=v: ff.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, p, None, None, values=v)) for _, _, p, v in sets]
gold = [np.load(os.path.join(G, "config4_256_oracle_f64.npz" if sd == 3 else "config4_256_seed%d_oracle_f64.npz" % sd)) for sd in bs.CONFIG4_SEEDS]
run("config 4 (256^3)", f, adds, gold, 1e-5)
del f
# config 2 at full size
g = np.load(os.path.join(G, "config2_1024_oracle_f64.npz"))
sizes, w2, pos, val = synth.config2()
f = bs.headline_field(fi, 2, sizes, w2)
f.set_field_tolerance(tol)
run("config 2 (1024^2)", f, [lambda ff: ff.add_points(w2.data_pos, w2.value_kernel, 0.0, w2.gradient_kernel, pos, None, None, values=val)], [g], 1e-5)
del f
# config 3's shape at 1024^2
g = np.load(os.path.join(G, "config3_1024_oracle_f64.npz"))
sizes, w3, pos3, nrm3 = synth.config3(side=1024, points_per_shape=int(g["num_points"]) // 2, seed=2)
f = fi.LatticeField(sizes, dtype="f64")
f.add_field_constraints(w3)
bs.configure(f, max(1, bs.SETTINGS[3]["levels"] - 2), bs.SETTINGS[3]["coarse_tol"], kcycle=bs.SETTINGS[3].get("kcycle", 0), cheb=bs.SETTINGS[3].get("cheb"))
f.set_field_tolerance(tol)
run("config 3 shape 1024^2", f, [lambda ff: ff.add_points(w3.data_pos, w3.value_kernel, w3.data_gradient, w3.gradient_kernel, pos3, nrm3, None)], [g], 1e-5)
del f
# config 5's shape at 128^3
g = np.load(os.path.join(G, "config5_128_oracle_f64.npz"))
sizes, w5, pos5, nrm5 = synth.config5(side=128, num_points=int(g["num_points"]), seed=4)
f = fi.LatticeField(sizes, dtype="f64")
f.add_field_constraints(w5)
bs.configure(f, max(1, bs.SETTINGS[5]["levels"] - 2), bs.SETTINGS[5]["coarse_tol"], kcycle=bs.SETTINGS[5].get("kcycle", 0), cheb=bs.SETTINGS[5].get("cheb"))
f.set_field_tolerance(tol)
run("config 5 shape 128^3", f, [lambda ff: ff.add_points(w5.data_pos, w5.value_kernel, w5.data_gradient, w5.gradient_kernel, pos5, nrm5, None)], [g], 1e-6)
