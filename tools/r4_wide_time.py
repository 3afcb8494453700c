#!/usr/bin/env python3
"""The untiled path once against the roofline: config 4's data (256^3, 1 M value rows) under the wide model stencils of
field_interpolation.cpp:282-315 (model_3, model_4, gradient_smoothness), fp32 and fp64: isolated apply (fi_time_apply),
algorithmic bytes 2 s N + C (4 + 36 s), bitwise reproducibility of two applies, and a solve."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import field_interpolation_amd as fi
from field_interpolation_amd import synth
side = int(os.environ.get("SIDE", "256"))
sizes, w0, pos, val = synth.config4(side=side, num_points=int(round(1e6 * (side / 256.0) ** 3)), seed=3)
x = np.random.default_rng(0).normal(size=int(np.prod(sizes)))
for name, kw in (("model_3 = 0.5", dict(model_2=0.0, model_3=0.5)), ("model_4 = 0.5", dict(model_2=0.0, model_4=0.5)),
                 ("model_2 = 0.5 + gradient_smoothness = 0.3", dict(model_2=0.5, gradient_smoothness=0.3)),
                 ("model_2 = 0.5 (the tiled kernel, for comparison)", dict(model_2=0.5))):
    for dt in ("f32", "f64"):
        w = fi.Weights(**kw)
        f = fi.LatticeField(sizes, dtype=dt)
        f.add_field_constraints(w)
        f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
        f.assemble()
        f.time_apply(3)
        ms = f.time_apply(20)
        st = f.stats()
        same = np.array_equal(f.apply_AtA(x), f.apply_AtA(x))
        out, it, rel = f.solve_cg(None, 200, 1e-5)
        print("%-50s %s: apply %7.1f us, %6.1f MB algorithmic = %.3f of 8 TB/s; two applies bitwise equal: %s; Jacobi-PCG %d iterations -> %.1e in %.1f ms"
              % (name, dt, ms * 1e3, st["spmv_bytes"] / 1e6, st["spmv_bytes"] / (ms * 1e-3) / 8e12, same, it, rel, f.stats()["solve_ms"]), flush=True)
        del f
