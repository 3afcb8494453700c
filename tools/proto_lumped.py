#!/usr/bin/env python3
"""CPU prototype, part 3 (round 4): the V-cycle built on A~ = A_model + diag(lumped A_data) on EVERY level.
A value row a (trilinear weights, a >= 0) has a a^T <= (sum a) diag(a) (Cauchy-Schwarz), so A <= A~, and on smooth fields the
two agree to second order (mass lumping).  Coarse levels: model rows re-discretised (weights rescaled), data diagonal
d_c = P^T d_f (the row sums of the Galerkin product P^T diag(d) P) -- no data rows, no cells, no sort on any level but the
finest one's fp64 operator.  PCG on the exact A, preconditioned by
   lib   : the library's V(M, M): polynomial smoother in A_model + 4 diag(A_data), exact A in the residuals, coarse levels
           re-discretised from the points
   lump  : the same cycle run entirely on A~ (smoother: the polynomial in D~^-1 A~ itself)
usage: proto_lumped.py [side] [tol]     env DENS (point density factor), TERMS (4), RATIO (10)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1] + ["64", "1e-7"] if len(sys.argv) < 3 else sys.argv
import proto_multilevel as pm
import proto_cc as pc_  # noqa: F401  (runs its own report once; reuse its builders)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
from field_interpolation_amd import synth

side = int(sys.argv[1])
pm.tol = float(sys.argv[2])
dens = float(os.environ.get("DENS", "1"))
terms = int(os.environ.get("TERMS", "4"))
ratio = float(os.environ.get("RATIO", "10"))


def cheb(op, dinv, hi, lo, r, terms, x0=None):
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta
    rho = 1.0 / sigma
    x = np.zeros_like(r) if x0 is None else x0.copy()
    res = r if x0 is None else r - op @ x
    d = dinv * res / theta
    x = x + d
    for _ in range(1, terms):
        rho_new = 1.0 / (2.0 * sigma - rho)
        res = r - op @ x
        d = rho_new * rho * d + 2.0 * rho_new / delta * (dinv * res)
        x = x + d
        rho = rho_new
    return x


def power(op, dinv, n, it=30):
    v = np.random.default_rng(1).normal(size=n)
    lam = 1.0
    for _ in range(it):
        v2 = dinv * (op @ v)
        lam = np.linalg.norm(v2) / np.linalg.norm(v)
        v = v2 / np.linalg.norm(v2)
    return lam


sizes, w, pos, val = synth.config4(side=side, num_points=int(round(dens * 1e6 * (side / 256.0) ** 3)), seed=3)
p = pos.reshape(-1, 3).astype(np.float64)
w2 = float(w.model_2) ** 2
# finest level, exact
Am0 = pm.model_matrix([side] * 3, w2)
B, rhs = pm.data_rows([side] * 3, p, val, float(w.data_pos))
Ad0 = (B.T @ B).tocsr()
A = (Am0 + Ad0).tocsr()
b = np.asarray(B.T @ rhs).ravel()


class L0:
    pass


L0.A, L0.b = A, b
# lumped hierarchy
lev = []
n = side
d = np.asarray(Ad0 @ np.ones(A.shape[0])).ravel()   # row sums of the data term (>= its diagonal)
P = []
for l in range(8):
    Am = pm.model_matrix([n] * 3, w2 * (0.5 ** l))
    At = (Am + sp.diags(d)).tocsr()
    dinv = 1.0 / At.diagonal()
    lam = power(At, dinv, At.shape[0])
    lev.append(dict(A=At, dinv=dinv, lam=lam, n=n))
    if n // 2 < 8:
        break
    P1 = pc_.prolong_cc(n // 2, n)
    Pl = sp.kron(P1, sp.kron(P1, P1)).tocsr()
    P.append(Pl)
    d = np.maximum(np.asarray(Pl.T @ d).ravel(), 0.0)
    n //= 2
print("lumped hierarchy: sides", [q["n"] for q in lev], "lambda", ["%.2f" % q["lam"] for q in lev])


def vl(r, l=0):
    Lv = lev[l]
    hi = 1.1 * Lv["lam"]
    if l + 1 == len(lev):
        return spl.spsolve(Lv["A"].tocsc(), r)
    z = cheb(Lv["A"], Lv["dinv"], hi, hi / ratio, r, terms)
    r1 = r - Lv["A"] @ z
    z = z + P[l] @ vl(P[l].T @ r1, l + 1)
    return cheb(Lv["A"], Lv["dinv"], hi, hi / ratio, r, terms, x0=z)


x, it, hist = pm.pcg(L0, vl, maxit=100)
pm.report("lumped V-cycle, %d terms / %g" % (terms, ratio), it, hist, 0)
print("  residual history:", " ".join("%.1e" % h for h in hist))
# how far A~ is from A on the finest level: extreme eigenvalues of A~^-1 A (the best any cycle on A~ can do)
try:
    At0 = lev[0]["A"].tocsc()
    lu = spl.splu(At0) if At0.shape[0] <= 40000 else None
    if lu is not None:
        op = spl.LinearOperator(A.shape, matvec=lambda v: lu.solve(A @ v))
        hi_ = spl.eigs(op, k=1, which="LM", return_eigenvectors=False)[0].real
        lo_ = spl.eigs(op, k=1, which="SM", return_eigenvectors=False, maxiter=5000, tol=1e-3)[0].real
        print("  spectrum of A~^-1 A: [%.3f, %.3f]" % (lo_, hi_))
except Exception as e:  # noqa: BLE001
    print("  (spectrum skipped: %s)" % e)

# ---- hybrid: the finest level keeps the exact operator in its residuals (smoother: the polynomial in A~), coarse levels lumped
def vh(r, l=0):
    if l > 0:
        return vl(r, l)
    Lv = lev[0]
    hi = 1.1 * Lv["lam"]
    z = cheb(Lv["A"], Lv["dinv"], hi, hi / ratio, r, terms)
    r1 = r - A @ z
    z = z + P[0] @ vl(P[0].T @ r1, 1)
    r2 = r - A @ z
    return z + cheb(Lv["A"], Lv["dinv"], hi, hi / ratio, r2, terms)


x, it, hist = pm.pcg(L0, vh, maxit=100)
pm.report("hybrid (exact fine residuals)", it, hist, 0)

# ---- coarse-to-fine start on the lumped hierarchy: b_c = P^T b, each level solved (directly here), interpolated up
bs = [b]
for l in range(len(P)):
    bs.append(P[l].T @ bs[-1])
xs = spl.spsolve(lev[-1]["A"].tocsc(), bs[-1])
for l in range(len(P) - 1, -1, -1):
    x0 = P[l] @ xs
    if l > 0:
        xs, _ = spl.cg(lev[l]["A"], bs[l], x0=x0, rtol=1e-6, maxiter=500)
    else:
        xs = x0
r0 = np.linalg.norm(b - A @ xs) / np.linalg.norm(b)
x, it, hist = pm.pcg(L0, vl, x0=xs, maxit=100)
pm.report("lumped V-cycle from the lumped cascade (start residual %.1e)" % r0, it, hist, 0)
# ... and with the lumped FINE problem solved first (fp32 V-cycle PCG in the library; here: to 1e-4)
xf, _ = spl.cg(lev[0]["A"], b, x0=xs, rtol=1e-4, maxiter=500)
r0 = np.linalg.norm(b - A @ xf) / np.linalg.norm(b)
x, it, hist = pm.pcg(L0, vl, x0=xf, maxit=100)
pm.report("... from the lumped fine solution (start residual %.1e)" % r0, it, hist, 0)

# ---- mixed hierarchy: lumped on the sparse (fine) levels, re-discretised cell operators on the dense (coarse) ones
rl, rP = pc_.build(terms, ratio, 8, 4)   # the library's levels (exact operators, smoother in A_model + 4 diag A_data)
for K in (1, 2, 3):
    def vmix(r, l=0, K=K):
        last = l + 1 == len(lev)
        if l < K:
            Lv = lev[l]
            hi = 1.1 * Lv["lam"]
            Aop, sm = Lv["A"], (lambda rr, x0=None, Lv=Lv, hi=hi: cheb(Lv["A"], Lv["dinv"], hi, hi / ratio, rr, terms, x0=x0))
        else:
            Lr = rl[l]
            Aop = Lr.A
            def sm(rr, x0=None, Lr=Lr):
                if x0 is None:
                    return Lr.M(rr)
                return x0 + Lr.M(rr - Lr.A @ x0)
        if last:
            return spl.spsolve(Aop.tocsc(), r)
        z = sm(r)
        r1 = r - Aop @ z
        z = z + P[l] @ vmix(P[l].T @ r1, l + 1)
        return sm(r, x0=z)
    x, it, hist = pm.pcg(L0, vmix, maxit=100)
    pm.report("lumped on levels < %d, cells below" % K, it, hist, 0)

# ---- cycle variants on the K = 1 hierarchy (finest level lumped, cells below): more work on the cheap levels
def make_cycle(gamma=1, coarse_terms=None, coarse_sweeps=1):
    def cyc(r, l=0):
        last = l + 1 == len(lev)
        if l < 1:
            Lv = lev[l]
            hi = 1.1 * Lv["lam"]
            Aop = Lv["A"]
            def sm(rr, x0=None):
                return cheb(Lv["A"], Lv["dinv"], hi, hi / ratio, rr, terms, x0=x0)
        else:
            Lr = rl[l]
            Aop = Lr.A
            tt = coarse_terms or terms
            def sm(rr, x0=None):
                x = x0
                for _ in range(coarse_sweeps):
                    save = Lr.terms
                    Lr.terms = tt
                    try:
                        x = Lr.M(rr) if x is None else x + Lr.M(rr - Lr.A @ x)
                    finally:
                        Lr.terms = save
                return x
        if last:
            return spl.spsolve(Aop.tocsc(), r)
        z = sm(r)
        for g in range(gamma if l >= 1 else 1):
            r1 = r - Aop @ z
            z = z + P[l] @ cyc(P[l].T @ r1, l + 1)
        return sm(r, x0=z)
    return cyc


for name, kw in (("V", {}), ("W below the finest", dict(gamma=2)), ("6 terms below", dict(coarse_terms=6)),
                 ("2 sweeps below", dict(coarse_sweeps=2)), ("W + 2 sweeps below", dict(gamma=2, coarse_sweeps=2))):
    x, it, hist = pm.pcg(L0, make_cycle(**kw), maxit=100)
    pm.report("K=1, " + name, it, hist, 0)
    print("   ", " ".join("%.1e" % h for h in hist))
