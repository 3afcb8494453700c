#!/usr/bin/env python3
"""Golden solutions of the 2-D BASELINE configurations by the reference's EXACT route -- VERDICT r4 item 2(b).

  config 2 at its full size (1024 x 1024, 10 k noisy value constraints, model_2 = 10: synth.config2 seed 1)
  config 3's shape (SDF from oriented points: triangle + inverted circle, default Weights, synth.config3 seed 2) at 1024 x 1024
  (12 500 points: the full configuration's 200 k scaled by (1024 / 4096)^2)

The route is solve_sparse_linear_exact (sparse_linear.cpp:154-184): the reference's rows -> explicit AtA and Atb in fp64
(the oracle, oracle/fi_oracle.cpp: fio_normal_equations_f64, zeros dropped like :84) -> Cholesky in fp64.  The oracle's own
banded Cholesky (cholesky_solve, one thread, unblocked) would take hours on a band of 2 049 x 1 M; the factorisation here is
LAPACK's blocked banded Cholesky (dpbtrf / dpbtrs through scipy.linalg.cholesky_banded, same band layout, no reordering, all
cores) -- the same factorisation up to rounding, checked against the oracle's at a size both finish (--check-side) -- followed
by ONE step of iterative refinement with the residual formed from the ROWS (fio_apply_normal_f64: A^T(A x), no explicit AtA).
The residual that is stored is that rows-based one.  Build container only (17 GB of band storage); the GPU tests and bench.py
compare with the committed samples (every --stride-th point per axis, plus whole-field checksums).

Run:  python tests/golden/make_golden_2d.py [--which 2,3] [--side 1024]
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np                                            # noqa: E402
import scipy.linalg as sl                                     # noqa: E402
from field_interpolation_amd import synth                     # noqa: E402
from oracle import fi_oracle as fo                            # noqa: E402


def exact_banded(f, report):
    """x = (AtA)^-1 Atb by banded Cholesky of the oracle's explicit fp64 AtA, one refinement step through the rows."""
    t0 = time.perf_counter()
    AtA, atb, _ = f.normal_equations()
    n = AtA.shape[0]
    ptr, idx, val = AtA.indptr, AtA.indices, AtA.data
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
    low = idx >= cols
    kd = int((idx[low] - cols[low]).max())
    report("explicit AtA: %d unknowns, %d non-zeros, half bandwidth %d (%.0f s); band storage %.1f GB"
           % (n, AtA.nnz, kd, time.perf_counter() - t0, (kd + 1) * n * 8 / 1e9))
    ab = np.zeros((kd + 1, n), dtype=np.float64, order="F")          # LAPACK's lower band layout: ab[i - j, j] = A[i, j]
    ab[idx[low] - cols[low], cols[low]] = val[low]
    del AtA, cols, low
    t1 = time.perf_counter()
    c = sl.cholesky_banded(ab, overwrite_ab=True, lower=True, check_finite=False)
    report("dpbtrf: %.0f s" % (time.perf_counter() - t1))
    x = sl.cho_solve_banded((c, True), atb, check_finite=False)
    bnorm = np.linalg.norm(atb)
    r = atb - f.apply_normal(x)                                       # from the rows: A^T(A x)
    res0 = float(np.linalg.norm(r) / bnorm)
    x = x + sl.cho_solve_banded((c, True), r, check_finite=False)
    res1 = float(np.linalg.norm(atb - f.apply_normal(x)) / bnorm)
    report("true residual through the rows: %.2e after the solve, %.2e after one refinement step" % (res0, res1))
    return x, res1, kd


def store(name, sizes, x, stride, meta):
    g = x.reshape(sizes[::-1])
    s = np.ascontiguousarray(g[tuple(slice(0, None, stride) for _ in sizes)])
    path = os.path.join(HERE, name)
    np.savez_compressed(path, sizes=np.asarray(sizes, np.int32), stride=np.int32(stride), sample=s,
                        field_sum=np.float64(x.sum()), field_sumsq=np.float64((x * x).sum()),
                        field_maxabs=np.float64(np.abs(x).max()), **meta)
    print("wrote %s (%d sample values, %d bytes)" % (path, s.size, os.path.getsize(path)), flush=True)


def build(which, side):
    if which == 2:
        npts = int(round(10_000 * (side / 1024.0) ** 2))
        sizes, w, pos, val = synth.config2(side=side, num_points=npts, seed=1)
        f = fo.LatticeField(sizes)
        f.add_field_constraints(fo.Weights(model_2=w.model_2))
        f.add_value_constraints(pos, val, w.data_pos)
        return sizes, f, npts, 1, "config 2 (synth.config2 seed 1)"
    pps = int(round(100_000 * (side / 4096.0) ** 2))
    sizes, w, pos, nrm = synth.config3(side=side, points_per_shape=pps, seed=2)
    return sizes, fo.sdf_from_points(sizes, fo.Weights(), pos, nrm), 2 * pps, 2, "config 3's shape (synth.config3 seed 2, default Weights)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="2,3")
    ap.add_argument("--side", type=int, default=1024)
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--check-side", type=int, default=96, help="LAPACK's factorisation against the oracle's own at this side (0: skip)")
    args = ap.parse_args()

    def report(msg):
        print("  " + msg, flush=True)

    for which in [int(w) for w in args.which.split(",")]:
        if args.check_side:
            sizes, f, _, _, _ = build(which, args.check_side)
            xs, res, _ = exact_banded(f, lambda m: None)
            xo = f.solve_exact_f64()
            d = float(np.abs(xs - xo).max() / np.abs(xo).max())
            print("config %d at %d^2: LAPACK banded Cholesky + refinement against the oracle's cholesky_solve: %.1e" % (which, args.check_side, d), flush=True)
            assert d <= 1e-9
        sizes, f, npts, seed, what = build(which, args.side)
        print("config %d at %d^2: %d points, %d rows, %d triplets" % (which, args.side, npts, f.num_rows, f.num_triplets), flush=True)
        t0 = time.perf_counter()
        x, res, kd = exact_banded(f, report)
        print("  %.0f s in all" % (time.perf_counter() - t0), flush=True)
        assert res <= 1e-10
        store("config%d_%d_oracle_f64.npz" % (which, args.side), sizes, x, args.stride,
              dict(true_rel_residual=np.float64(res), num_points=np.int32(npts), seed=np.int32(seed), half_bandwidth=np.int32(kd),
                   iterations=np.int32(0),
                   what=what + ": the oracle's explicit fp64 AtA of the reference's rows, banded Cholesky (LAPACK dpbtrf, "
                        "sparse_linear.cpp:154-184's route) + one refinement step through the rows"))
        del f, x


if __name__ == "__main__":
    main()
