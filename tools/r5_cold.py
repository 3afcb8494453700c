"""Where a cold step of the HEADLINE solver goes (fp64 CG + fp32 V-cycle, bench settings): the calls of a bench step on a
fresh context, each timed with a device synchronisation; three contexts one after the other (the second and third find the
device blocks of the ones before in the pool), three steps each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import field_interpolation_amd as fi
from field_interpolation_amd import bench_settings as bs
from field_interpolation_amd import synth

sizes, w, pos, val = synth.config4(side=256, num_points=1000000, seed=3)
dev = torch.device("cuda", 0)
d_pos = torch.from_numpy(pos).to(dev)
d_val = torch.from_numpy(val).to(dev)
d_out = torch.empty(int(np.prod(sizes)), dtype=torch.float32, device=dev)
tol = bs.config4_tolerance(sizes, len(pos))


def timed(label, fn, acc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc.append((label, 1e3 * (time.perf_counter() - t0)))
    return r


for rep in range(3):
    acc = []
    f = timed("create+model+options", lambda: bs.headline_field(fi, 4, sizes, w), acc)
    for step in range(3):
        timed("clear+add %d" % step, lambda: (f.clear_points(), f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, d_pos, None, None, values=d_val)), acc)
        timed("assemble %d" % step, lambda: f.assemble(), acc)
        timed("solve %d" % step, lambda: f.solve_cg(None, 0, tol, out=d_out), acc)
    timed("destroy", lambda: f.__del__(), acc)
    f = None
    print("context %d: " % rep + "  ".join("%s %.2f" % a for a in acc), flush=True)
