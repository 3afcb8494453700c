"""Shared builders for the parity tests: the same inputs go to the oracle and to the GPU library."""
import numpy as np


def random_points(rng, sizes, n, margin=1.5, with_edge_cases=True):
    D = len(sizes)
    pos = np.stack([rng.uniform(-margin, s - 1 + margin, n) for s in sizes], axis=1).astype(np.float32)
    if with_edge_cases and n >= 12:
        pos[: n // 6] = np.round(pos[: n // 6])                        # exact lattice hits (t == 0)
        pos[n // 6: n // 3] = np.floor(pos[n // 6: n // 3]) + 0.5      # cell centres / round-half cases
    nrm = rng.normal(size=(n, D)).astype(np.float32)
    pw = rng.uniform(0.2, 2.0, n).astype(np.float32)
    if with_edge_cases:
        pw[::7] = 0.0                                                   # dropped rows
    val = rng.normal(size=n).astype(np.float32)
    return pos, nrm, pw, val


def sphere_points(rng, sizes, n, noise=0.3):
    D = len(sizes)
    c = (np.array(sizes) - 1) / 2.0
    r = 0.3 * (min(sizes) - 1)
    d = rng.normal(size=(n, D))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = (c + r * d + rng.normal(scale=noise, size=(n, D))).astype(np.float32)
    return pos, d.astype(np.float32)


def oracle_weights(oracle, w):
    return oracle.Weights(data_pos=w.data_pos, data_gradient=w.data_gradient, model_0=w.model_0, model_1=w.model_1,
                          model_2=w.model_2, model_3=w.model_3, model_4=w.model_4,
                          gradient_smoothness=w.gradient_smoothness, value_kernel=int(w.value_kernel),
                          gradient_kernel=int(w.gradient_kernel))


def build_pair(oracle, fi, sizes, w, pos, nrm=None, pw=None, val=None, dtype="f64"):
    """The reference call sequence add_field_constraints + add_points on both sides.
    With `val` the value rows carry targets (add_value_constraint per point on the oracle side)."""
    ow = oracle_weights(oracle, w)
    fo = oracle.LatticeField(sizes)
    fo.add_field_constraints(ow)
    if val is None:
        fo.add_points(w.data_pos, int(w.value_kernel), w.data_gradient, int(w.gradient_kernel), pos, nrm, pw)
    else:
        for i in range(len(pos)):
            wi = np.float32(1.0 if pw is None else pw[i])
            if int(w.value_kernel) == 0:
                fo.add_value_constraint_nearest_neighbor(pos[i], nrm[i], float(val[i]), float(wi * np.float32(w.data_pos)))
            else:
                fo.add_value_constraint(pos[i], float(val[i]), float(wi * np.float32(w.data_pos)))
            if nrm is not None:
                fo.add_gradient_constraint(pos[i], nrm[i], float(wi * np.float32(w.data_gradient)), int(w.gradient_kernel))
    fg = fi.LatticeField(sizes, dtype=dtype)
    fg.add_field_constraints(w)
    fg.add_points(w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, pos, nrm, pw, values=val)
    return fo, fg


def rel_inf(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
