"""Soak: contexts of varying shapes created, solved and destroyed in a loop; device memory in use and the host's resident set
must level off (the pool of fi_pool.hip holds at most 8 GiB; events, streams and host buffers go with their context)."""
import gc
import os
import resource
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import field_interpolation_amd as fi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(1)
shapes = [[48, 40, 36], [64, 64, 64], [33, 45, 29], [200, 180], [96, 80, 88], [500], [128, 128, 96]]
free0, total = torch.cuda.mem_get_info()
for i in range(n):
    sizes = shapes[i % len(shapes)]
    dtype = "f64" if i % 3 == 0 else "f32"
    f = fi.LatticeField(sizes, dtype=dtype)
    w = fi.Weights(data_gradient=0.0)
    f.add_field_constraints(w)
    if len(sizes) >= 2 and i % 2 == 0:
        f.set_levels(2, 1e-5 if dtype == "f32" else 1e-6)
        if i % 4 == 0:
            f.set_multigrid(True)
            if dtype == "f64":
                f.set_mixed_precision(True)
    if len(sizes) == 3 and i % 5 == 1:
        f.set_polynomial(4)
    m = 3000
    pos = np.stack([rng.uniform(0, s - 1, m) for s in sizes], axis=1).astype(np.float32)
    val = rng.normal(size=m).astype(np.float32)
    f.add_points(w.data_pos, w.value_kernel, 0.0, w.gradient_kernel, pos, None, None, values=val)
    f.assemble()
    x, it, rel = f.solve_cg(None, 0, 1e-5)
    assert rel <= 1e-5, (i, sizes, dtype, rel)
    del f
    gc.collect()
    if i % 50 == 49 or i == n - 1:
        free, _ = torch.cuda.mem_get_info()
        print("cycle %4d: device memory in use %.1f MiB (pool %.1f MiB), host RSS %.0f MiB" %
              (i + 1, (free0 - free) / 2 ** 20, fi.memory_pool() / 2 ** 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
fi.memory_pool(0)
free, _ = torch.cuda.mem_get_info()
print("after trimming the pool: device memory in use %.1f MiB" % ((free0 - free) / 2 ** 20))
