"""Portable synthetic inputs for the BASELINE.json configurations.

The reference's demos draw from libstdc++'s std::default_random_engine / std::normal_distribution
(sine_denoise_1d.cpp:35-36, sdf_field.cpp:308-310), whose streams are not portable, so the
workloads are regenerated from a counter-based generator defined here: splitmix64 -> 24-bit uniform,
Box-Muller -> normal.  Same seed => same bits everywhere (numpy integer arithmetic only).
All positions / normals are fp32, interleaved xyzxyz..., in LATTICE coordinates.
"""
import numpy as np

from .api import GradientKernel, ValueKernel, Weights

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(ctr):
    z = (ctr + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def uniform(seed, stream, n):
    """n uniform doubles in [0, 1) with 24 random bits each."""
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0x100000001B3)
               + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03) + np.uint64(stream) * np.uint64(0x2545F4914F6CDD1D))
        bits = _splitmix64(ctr)
    return (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def normal(seed, stream, n):
    u1 = uniform(seed, 2 * stream + 1000, n)
    u2 = uniform(seed, 2 * stream + 1001, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


# ---- C1: src/field_1d.cpp:20-29,98-107 at a given resolution ----------------------------------------

def config1(resolution=1024):
    """-> (sizes, weights, [(pos_lattice, value, gradient_lattice)])"""
    pts = []
    for pos, value, grad in [(0.2, 0.0, +1.0), (0.8, 0.0, -1.0)]:
        pos_l = np.float32(pos) * np.float32(resolution - 1)          # field_1d.cpp:101
        grad_l = np.float32(grad) / np.float32(resolution - 1)        # field_1d.cpp:102
        pts.append((pos_l, np.float32(value), grad_l))
    return [resolution], Weights(), pts


# ---- C2: 2-D noisy value constraints + smoothness prior (sine_denoise_1d.cpp:28-29,42-46 in 2-D) ---

def config2(side=1024, num_points=10000, seed=1):
    u = uniform(seed, 0, num_points)
    v = uniform(seed, 1, num_points)
    pos = np.stack([u * (side - 1), v * (side - 1)], axis=1).astype(np.float32)
    val = 0.5 * np.sin(10.0 * u * (1.0 + 2.0 * u)) * np.cos(7.0 * v) + 0.1 * normal(seed, 2, num_points)
    w = Weights(model_2=10.0)
    return [side, side], w, pos, val.astype(np.float32)


# ---- C3: 2-D SDF from oriented points: triangle r=0.35 plus inverted circle r=0.1 ------------------

def _polygon(t, sides, radius, center):
    """Point and outward normal on a regular polygon, t in [0,1) along the perimeter."""
    s = t * sides
    k = np.floor(s)
    f = s - k
    a0 = 2 * np.pi * k / sides
    a1 = 2 * np.pi * (k + 1) / sides
    p0 = np.stack([np.cos(a0), np.sin(a0)], 1)
    p1 = np.stack([np.cos(a1), np.sin(a1)], 1)
    p = p0 + (p1 - p0) * f[:, None]
    am = 0.5 * (a0 + a1)
    nrm = np.stack([np.cos(am), np.sin(am)], 1)
    return center + radius * p, nrm


def config3(side=4096, points_per_shape=100000, seed=2, pos_stddev=0.005, normal_stddev=0.05):
    n = points_per_shape
    t = (np.arange(n) + 0.5) / n
    tri_p, tri_n = _polygon(t, 3, 0.35, np.array([0.5, 0.5]))                  # sdf_field.cpp:27-38
    ang = 2 * np.pi * t
    cir_p = np.array([0.5, 0.5]) + 0.1 * np.stack([np.cos(ang), np.sin(ang)], 1)   # inverted: normals point in
    cir_n = -np.stack([np.cos(ang), np.sin(ang)], 1)
    pos = np.concatenate([tri_p, cir_p])
    nrm = np.concatenate([tri_n, cir_n])
    m = len(pos)
    pos = pos + pos_stddev * np.stack([normal(seed, 0, m), normal(seed, 1, m)], 1)  # sdf_field.cpp:312-315
    a = np.arctan2(nrm[:, 1], nrm[:, 0]) + normal_stddev * normal(seed, 2, m)       # sdf_field.cpp:316-321
    nrm = np.stack([np.cos(a), np.sin(a)], 1)
    pos = pos * (side - 1.0)                                                          # sdf_field.cpp:198-210
    return [side, side], Weights(), pos.astype(np.float32), nrm.astype(np.float32)


# ---- C4: 3-D scattered value constraints: signed distance to a sphere, noisy ----------------------

def config4(side=256, num_points=1000000, seed=3, depth=None):
    """`depth` (default side) is the extent of the slowest axis: weak scaling stacks slabs along it with
    the same point density."""
    depth = side if depth is None else depth
    n = int(num_points)
    pos = np.stack([uniform(seed, 0, n) * (side - 1), uniform(seed, 1, n) * (side - 1),
                    uniform(seed, 2, n) * (depth - 1)], axis=1)
    centre = np.array([(side - 1) / 2.0, (side - 1) / 2.0, (depth - 1) / 2.0])
    radius = 0.3 * (side - 1)
    val = np.linalg.norm(pos - centre, axis=1) - radius + 0.5 * normal(seed, 3, n)
    return [side, side, depth], Weights(model_2=0.5), pos.astype(np.float32), val.astype(np.float32)


# ---- C5: 3-D SDF from oriented points on a sphere ------------------------------------------------------

def config5(side=512, num_points=5000000, seed=4):
    n = int(num_points)
    d = np.stack([normal(seed, 0, n), normal(seed, 1, n), normal(seed, 2, n)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    centre = (side - 1) / 2.0
    radius = 0.3 * (side - 1)
    pos = centre + radius * d + 0.5 * np.stack([normal(seed, 3, n), normal(seed, 4, n), normal(seed, 5, n)], 1)
    nrm = d + 0.05 * np.stack([normal(seed, 6, n), normal(seed, 7, n), normal(seed, 8, n)], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return [side, side, side], Weights(), pos.astype(np.float32), nrm.astype(np.float32)


__all__ = ["uniform", "normal", "config1", "config2", "config3", "config4", "config5", "Weights", "ValueKernel",
           "GradientKernel"]
