#!/usr/bin/env python3
"""CPU prototype (numpy / scipy, no GPU): iteration counts of preconditioner candidates for config 4 at a small side.
  poly     : CG preconditioned by the d-term Chebyshev polynomial in Dinv (A_model + diag A_data)   (what bench.py runs)
  add      : the same polynomial on every level, added up (BPX-like):  z = M0 r + P (M1 P^T r + P (...))
  vm       : V-cycle with the polynomial as pre- and post-smoother (two full applies per level and cycle)
  vc       : V-cycle with a degree-4 Chebyshev smoother in the full operator (what FI_OPT_MULTIGRID runs)
Coarse levels are re-discretisations from the same points (positions halved, model weight^2 * 2^D / 4^k), as in the
library.  Usage: proto_multilevel.py [side] [tol]"""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from field_interpolation_amd import synth

side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-9
CUBIC = "--cubic" in sys.argv


def second_diff(n):
    m = n - 2
    rows = np.repeat(np.arange(m), 3)
    cols = (np.arange(m)[:, None] + np.arange(3)[None, :]).ravel()
    vals = np.tile(np.array([1.0, -2.0, 1.0]), m)
    return sp.csr_matrix((vals, (rows, cols)), shape=(m, n))


def model_matrix(sizes, w2):
    nx, ny, nz = sizes
    Ix, Iy, Iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    Sx, Sy, Sz = second_diff(nx), second_diff(ny), second_diff(nz)
    # index = x + nx * (y + ny * z): kron(z, kron(y, x))
    Ax = sp.kron(Iz, sp.kron(Iy, Sx.T @ Sx))
    Ay = sp.kron(Iz, sp.kron(Sy.T @ Sy, Ix))
    Az = sp.kron(Sz.T @ Sz, sp.kron(Iy, Ix))
    return (w2 * (Ax + Ay + Az)).tocsr()


def data_rows(sizes, pos, val, wd):
    nx, ny, nz = sizes
    p = pos.reshape(-1, 3).astype(np.float64)
    f = np.floor(p)
    t = p - f
    f = f.astype(np.int64)
    rows, cols, vals = [], [], []
    for c in range(8):
        bx, by, bz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        ix, iy, iz = f[:, 0] + bx, f[:, 1] + by, f[:, 2] + bz
        wgt = (t[:, 0] if bx else 1 - t[:, 0]) * (t[:, 1] if by else 1 - t[:, 1]) * (t[:, 2] if bz else 1 - t[:, 2])
        ok = (ix >= 0) & (ix < nx) & (iy >= 0) & (iy < ny) & (iz >= 0) & (iz < nz)
        rows.append(np.nonzero(ok)[0])
        cols.append((ix + nx * (iy + ny * iz))[ok])
        vals.append(wgt[ok] * wd)
    B = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(len(p), nx * ny * nz))
    rhs = np.asarray(B @ np.ones(nx * ny * nz)).ravel() * val  # (sum of kept weights) * value, field_interpolation.cpp:57-80
    return B, rhs


def prolong_1d(nc, nf, cubic):
    rows, cols, vals = [], [], []
    for i in range(nf):
        if i % 2 == 0:
            rows.append(i); cols.append(min(i // 2, nc - 1)); vals.append(1.0)
        else:
            j = i // 2
            if cubic and j - 1 >= 0 and j + 2 < nc:
                for jj, v in ((j - 1, -1 / 16), (j, 9 / 16), (j + 1, 9 / 16), (j + 2, -1 / 16)):
                    rows.append(i); cols.append(jj); vals.append(v)
            elif j + 1 < nc:
                rows += [i, i]; cols += [j, j + 1]; vals += [0.5, 0.5]
            else:
                rows.append(i); cols.append(j); vals.append(1.0)
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf, nc))


class Level:
    def __init__(self, sizes, pos, val, w2, wd, terms, ratio):
        self.sizes = sizes
        Am = model_matrix(sizes, w2)
        B, rhs = data_rows(sizes, pos, val, wd)
        Ad = (B.T @ B).tocsr()
        self.A = (Am + Ad).tocsr()
        self.b = np.asarray(B.T @ rhs).ravel()
        self.d = self.A.diagonal()
        self.dinv = 1.0 / self.d
        self.At = (Am + sp.diags(Ad.diagonal())).tocsr()   # A~: model rows + the diagonal of the data rows
        m = Am.diagonal()
        v = np.random.default_rng(1).normal(size=Am.shape[0])
        for _ in range(30):
            v2 = (Am @ v) / m
            lam = np.linalg.norm(v2) / np.linalg.norm(v)
            v = v2 / np.linalg.norm(v2)
        self.lam_model = max(lam, 1.0)
        v = np.random.default_rng(2).normal(size=Am.shape[0])
        for _ in range(30):
            v2 = self.dinv * (self.A @ v)
            lam = np.linalg.norm(v2) / np.linalg.norm(v)
            v = v2 / np.linalg.norm(v2)
        self.lam_full = lam
        self.terms, self.ratio = terms, ratio

    def cheb(self, op, lam_hi, lam_lo, r, terms, x0=None):
        """terms-term Chebyshev approximation of op^-1 r in the Dinv-scaled recurrence (the library's cheb_smooth)."""
        theta, delta = 0.5 * (lam_hi + lam_lo), 0.5 * (lam_hi - lam_lo)
        sigma = theta / delta
        rho = 1.0 / sigma
        x = np.zeros_like(r) if x0 is None else x0.copy()
        res = r if x0 is None else r - op @ x
        dvec = self.dinv * res / theta
        x = x + dvec
        for _ in range(1, terms):
            rho_new = 1.0 / (2.0 * sigma - rho)
            res = r - op @ x
            dvec = rho_new * rho * dvec + 2.0 * rho_new / delta * (self.dinv * res)
            x = x + dvec
            rho = rho_new
        return x

    def M(self, r):
        hi = 1.1 * self.lam_model
        return self.cheb(self.At, hi, hi / self.ratio, r, self.terms)


def build(terms, ratio, nlev):
    sizes, w, pos, val = synth.config4(side=side, num_points=int(round(1e6 * (side / 256.0) ** 3)), seed=3)
    levels, transfers = [], []
    w2 = float(w.model_2) ** 2
    p = pos.reshape(-1, 3).astype(np.float64)
    sz = list(sizes)
    for l in range(nlev):
        levels.append(Level(sz, p / (2 ** l), val, w2 * (8.0 / 16.0) ** l, float(w.data_pos), terms, ratio))
        nsz = [(s + 1) // 2 for s in sz]
        if min(nsz) < 4:
            break
        if l + 1 < nlev:
            Px, Py, Pz = (prolong_1d(nsz[k], sz[k], CUBIC) for k in range(3))
            transfers.append(sp.kron(Pz, sp.kron(Py, Px)).tocsr())
        sz = nsz
    return levels, transfers


def pcg(L, prec, x0=None, maxit=2000):
    A, b = L.A, L.b
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    bb = np.linalg.norm(b)
    z = prec(r)
    p = z.copy()
    rz = r @ z
    hist = [np.linalg.norm(r) / bb]
    for it in range(1, maxit + 1):
        q = A @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        hist.append(np.linalg.norm(r) / bb)
        if hist[-1] <= tol:
            return x, it, hist
        z = prec(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, maxit, hist


def report(name, it, hist, cost_per_it):
    # iterations per decade over the last three decades
    h = np.array(hist)
    k3 = np.argmax(h <= tol * 1e3)
    print("%-34s %4d iterations   (last 3 decades: %5.1f per decade)   cost %6.0f B/pt" % (
        name, it, (it - k3) / 3.0, it * cost_per_it), flush=True)


def main():
    nlev = 1
    s = side
    while s >= 16:
        s = (s + 1) // 2
        nlev += 1
    print("side %d, tol %g, up to %d levels, %s transfers" % (side, tol, nlev, "cubic" if CUBIC else "trilinear"))
    OUTER = 92.0     # fp64 outer iteration: apply 16 + 12 records, resid 28, xp 36
    STEP = 14.0      # fp32 Chebyshev step, mean of 10 / 14 / 18
    FULL = 20.0      # fp32 full apply (8 + records)
    for terms, ratio in ((4, 30.0), (8, 100.0)):
        levels, P = build(terms, ratio, nlev)
        L0 = levels[0]
        sol = None
        # coarse-to-fine start like the library: solve the coarser level, interpolate
        x, it, hist = pcg(L0, L0.M)
        report("poly %d/%g" % (terms, ratio), it, hist, OUTER + (terms - 1) * STEP)

        def add_prec(r, l=0):
            z = levels[l].M(r)
            if l + 1 < len(levels):
                z = z + P[l] @ add_prec(P[l].T @ r, l + 1)
            return z
        x, it, hist = pcg(L0, add_prec)
        report("additive %d/%g, %d levels" % (terms, ratio, len(levels)), it, hist, OUTER + (terms - 1) * STEP * 8 / 7)

        def vm(r, l=0):
            Lv = levels[l]
            if l + 1 == len(levels):
                return spl.spsolve(Lv.A.tocsc(), r) if Lv.A.shape[0] < 6000 else Lv.M(r)
            z = Lv.M(r)
            r1 = r - Lv.A @ z
            z = z + P[l] @ vm(P[l].T @ r1, l + 1)
            return z + Lv.M(r - Lv.A @ z)
        x, it, hist = pcg(L0, vm)
        report("V(M,M) %d/%g" % (terms, ratio), it, hist, OUTER + (2 * FULL + 2 * (terms - 1) * STEP) * 8 / 7)

    def vc(r, l=0):
        Lv = levels[l]
        hi = 1.1 * Lv.lam_full
        if l + 1 == len(levels):
            return spl.spsolve(Lv.A.tocsc(), r) if Lv.A.shape[0] < 6000 else Lv.cheb(Lv.A, hi, hi / 10, r, 4)
        z = Lv.cheb(Lv.A, hi, hi / 10, r, 4)
        r1 = r - Lv.A @ z
        z = z + P[l] @ vc(P[l].T @ r1, l + 1)
        return Lv.cheb(Lv.A, hi, hi / 10, r, 4, x0=z)
    x, it, hist = pcg(L0, vc)
    report("V(cheb4 in A) [library]", it, hist, OUTER + (9 * (18 + 12)) * 8 / 7)


if __name__ == "__main__":
    main()
