#!/usr/bin/env python3
"""CPU prototype (numpy / scipy): does the ORDER of the V-cycle's transfers limit it on surface-type data?  The model is fourth
order (model_2: sum of squared second differences); trilinear interpolation and its transpose have orders 2 + 2 = 4, and the classical
rule for a 2m-th order operator wants m_P + m_R > 2m.  2-D, vertex-centred levels (sizes 2^k + 1), coarse levels re-discretised from
the same points like the library's, V(1,1) with a degree-4 Chebyshev smoother in Dinv A over [l / 10, 1.1 l], PCG to 1e-8.
  data: points on a circle (value 0) with bilinear value rows and cell-edge gradient rows -- an SDF like config 3's
usage: proto_transfer_order.py [k: side = 2^k + 1] [points]"""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 600
N = 2 ** k + 1


def second_diff(n):
    m = n - 2
    rows = np.repeat(np.arange(m), 3)
    cols = (np.arange(m)[:, None] + np.arange(3)[None, :]).ravel()
    return sp.csr_matrix((np.tile([1.0, -2.0, 1.0], m), (rows, cols)), shape=(m, n))


def operator(n, pos, nrm, w2, wv, wg):
    I = sp.identity(n)
    S = second_diff(n)
    A = w2 * (sp.kron(I, S.T @ S) + sp.kron(S.T @ S, I))
    f = np.floor(pos).astype(int)
    t = pos - f
    rows, cols, vals, rhs = [], [], [], []
    r = 0
    for i in range(len(pos)):
        x, y = f[i]
        if not (0 <= x < n - 1 and 0 <= y < n - 1):
            continue
        tx, ty = t[i]
        for (dx, dy, wgt) in ((0, 0, (1 - tx) * (1 - ty)), (1, 0, tx * (1 - ty)), (0, 1, (1 - tx) * ty), (1, 1, tx * ty)):
            rows.append(r); cols.append(x + dx + n * (y + dy)); vals.append(wv * wgt)
        rhs.append(0.0); r += 1
        # gradient rows on the cell's edges (field_interpolation.cpp:150-187): d/dx averaged over the two x edges, d/dy likewise
        for d in range(2):
            for e in range(2):
                a = (x + (e if d == 1 else 0)) + n * (y + (e if d == 0 else 0))
                b = a + (1 if d == 0 else n)
                we = wg * ((ty if e else 1 - ty) if d == 0 else (tx if e else 1 - tx))
                rows += [r, r]; cols += [a, b]; vals += [-we, we]
            rhs.append(wg * nrm[i, d]); r += 1
    B = sp.csr_matrix((vals, (rows, cols)), shape=(r, n * n))
    return (A + B.T @ B).tocsr(), B.T @ np.asarray(rhs)


def prolong_1d(nc, nf, cubic):
    rows, cols, vals = [], [], []
    for j in range(nc):
        rows.append(2 * j); cols.append(j); vals.append(1.0)
    for j in range(nc - 1):
        i = 2 * j + 1
        if cubic and 1 <= j < nc - 2:
            for dj, w in ((-1, -1 / 16), (0, 9 / 16), (1, 9 / 16), (2, -1 / 16)):
                rows.append(i); cols.append(j + dj); vals.append(w)
        else:
            rows += [i, i]; cols += [j, j + 1]; vals += [0.5, 0.5]
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf, nc))


def cheb(A, dinv, lam, b, x, deg, ratio):
    hi, lo = 1.1 * lam, 1.1 * lam / ratio
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta
    r = b - A @ x if x is not None else b.copy()
    if x is None:
        x = np.zeros_like(b)
    d = dinv * r / theta
    x = x + d
    rho = 1 / sigma
    for _ in range(1, deg):
        r = r - A @ d
        rho_new = 1 / (2 * sigma - rho)
        d = rho_new * rho * d + 2 * rho_new / delta * (dinv * r)
        x = x + d
        rho = rho_new
    return x


MODE = "rediscretised"   # or "W" (a second visit of every coarser level), "galerkin", "stiffer" (coarse model weight^2 x 2), "damped" (coarse correction x 0.5)


def build(cubic):
    rng = np.random.default_rng(2)
    a = rng.uniform(0, 2 * np.pi, npts)
    c, R = 0.5 * (N - 1), 0.3 * (N - 1)
    pos = np.stack([c + R * np.cos(a), c + R * np.sin(a)], 1) + rng.normal(scale=0.3, size=(npts, 2))
    nrm = np.stack([np.cos(a), np.sin(a)], 1)
    levels = []
    n, p, g, w2 = N, pos.copy(), nrm.copy(), 0.5 ** 2
    wg = 1.0
    while n >= 9:
        A, b = operator(n, p, g, w2, 1.0, wg)
        d = A.diagonal()
        dinv = 1.0 / d
        lam = spl.eigsh(sp.diags(dinv) @ A, k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]
        levels.append(dict(A=A, b=b, dinv=dinv, lam=lam, n=n))
        nc = (n + 1) // 2
        P1 = prolong_1d(nc, n, cubic)
        levels[-1]["P"] = sp.kron(P1, P1).tocsr()
        n, p, g, w2, wg = nc, p / 2, g * 2, w2 * (4 / 16) * (2.0 if MODE == "stiffer" else 1.0), wg / 2   # (library: positions halved, normals doubled, model w^2 * 2^D / 16, gradient w / 2)
    if MODE == "galerkin":
        for l in range(1, len(levels)):
            P = levels[l - 1]["P"]
            A = (P.T @ levels[l - 1]["A"] @ P).tocsr()
            levels[l]["A"] = A
            levels[l]["dinv"] = 1.0 / A.diagonal()
            levels[l]["lam"] = spl.eigsh(sp.diags(levels[l]["dinv"]) @ A, k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]
    return levels


def vcycle(L, l, b):
    lev = L[l]
    if l == len(L) - 1:
        return spl.spsolve(lev["A"].tocsc(), b)
    x = cheb(lev["A"], lev["dinv"], lev["lam"], b, None, 4, 10.0)
    r = b - lev["A"] @ x
    bc = lev["P"].T @ r
    if MODE == "K" and l + 1 < len(L) - 1:      # two FCG steps on the coarser level, each preconditioned by ITS cycle
        Ac = L[l + 1]["A"]
        c1 = vcycle(L, l + 1, bc); v1 = Ac @ c1; rho1 = c1 @ v1; a1 = c1 @ bc
        r1 = bc - (a1 / rho1) * v1
        c2 = vcycle(L, l + 1, r1); v2 = Ac @ c2; gam = c2 @ v1; beta = c2 @ v2; a2 = c2 @ r1
        rho2 = beta - gam * gam / rho1
        ec = (a1 / rho1 - gam * a2 / (rho1 * rho2)) * c1 + (a2 / rho2) * c2
        x = x + lev["P"] @ ec
        return cheb(lev["A"], lev["dinv"], lev["lam"], b, x, 4, 10.0)
    ec = vcycle(L, l + 1, bc)
    if MODE == "W" and l + 1 < len(L) - 1:      # the coarser level once more on what the first visit left
        ec = ec + vcycle(L, l + 1, bc - L[l + 1]["A"] @ ec)
    x = x + (0.5 if MODE == "damped" else 1.0) * (lev["P"] @ ec)
    return cheb(lev["A"], lev["dinv"], lev["lam"], b, x, 4, 10.0)


def pcg(L, tol=1e-8, maxit=400):
    A, b = L[0]["A"], L[0]["b"]
    x = np.zeros_like(b); r = b.copy(); z = vcycle(L, 0, r); p = z.copy(); rz = r @ z; bb = np.sqrt(b @ b)
    for it in range(1, maxit + 1):
        q = A @ p; al = rz / (p @ q); x += al * p; r -= al * q
        if np.sqrt(r @ r) <= tol * bb:
            return it
        z = vcycle(L, 0, r); rz2 = r @ z
        beta = -al * (z @ q) / rz if MODE == "K" else rz2 / rz     # flexible (Polak-Ribiere): z . (r_new - r_old) / (z_old . r_old) = -alpha z . q / rz_old
        p = z + beta * p; rz = rz2
    return maxit


for MODE in ("rediscretised", "W", "K"):
    for cubic in (False, True):
        L = build(cubic)
        print("side %d, %d levels, %d oriented points, coarse levels %s, %s interpolation (R = P^T): %d iterations" % (
            N, len(L), npts, MODE, "CUBIC" if cubic else "linear", pcg(L)), flush=True)
