#!/usr/bin/env python3
"""CPU prototype, part 2: cell-centred coarsening + V(M, M) with the polynomial smoother built on
A^ = A_model + f * diag(A_data)  (f = 2^D bounds the data blocks: a a^T <= 2^D diag(a_i^2), so M A <= M A^ < 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1] + ["64", "3e-9"] if len(sys.argv) < 3 else sys.argv
import proto_multilevel as pm
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from field_interpolation_amd import synth
side = int(sys.argv[1]); pm.tol = float(sys.argv[2])
dens = float(os.environ.get("DENS", "1"))

def prolong_cc(nc, nf):
    rows, cols, vals = [], [], []
    for i in range(nf):
        j = i // 2
        jn = j - 1 if i % 2 == 0 else j + 1
        if 0 <= jn < nc: rows += [i, i]; cols += [j, jn]; vals += [.75, .25]
        else:
            jo = j + 1 if jn < 0 else j - 1
            rows += [i, i]; cols += [j, jo]; vals += [1.25, -.25]
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf, nc))

class LevelF(pm.Level):
    def set_factor(self, f):
        m = self.A.diagonal() - (self.At.diagonal() - self.At.diagonal())  # placeholder
        Am = self.At - sp.diags(self.At.diagonal()) + sp.diags(self.m_diag)
        dd = self.d - self.m_diag
        self.Ahat = (Am + sp.diags(f * dd)).tocsr()
        self.dhat_inv = 1.0 / (self.m_diag + f * dd)
    def M(self, r):
        hi = 1.1 * self.lam_model
        save = self.dinv; self.dinv = self.dhat_inv
        try: return self.cheb(self.Ahat, hi, hi / self.ratio, r, self.terms)
        finally: self.dinv = save

def build(terms, ratio, nlev, f):
    sizes, w, pos, val = synth.config4(side=side, num_points=int(round(dens * 1e6 * (side / 256.0) ** 3)), seed=3)
    p = pos.reshape(-1, 3).astype(np.float64)
    w2 = float(w.model_2) ** 2
    levels, P = [], []
    n = side
    for l in range(nlev):
        L = LevelF([n] * 3, p, val, w2 * (0.5 ** l), float(w.data_pos), terms, ratio)
        L.m_diag = pm.model_matrix([n] * 3, w2 * (0.5 ** l)).diagonal()
        L.set_factor(f)
        levels.append(L)
        if n // 2 < 8 or l + 1 == nlev: break
        P1 = prolong_cc(n // 2, n)
        P.append(sp.kron(P1, sp.kron(P1, P1)).tocsr())
        n //= 2
        p = np.clip((p - 0.5) / 2, 0, n - 1 - 1e-9)
    return levels, P

for terms, ratio in ((4, 10.),):
    for f in (4,):
        levels, P = build(terms, ratio, 4, f)
        L0 = levels[0]
        def vm(r, l=0):
            Lv = levels[l]
            if l + 1 == len(levels):
                mode = os.environ.get("COARSE", "exact")
                if mode == "exact":
                    return spl.spsolve(Lv.A.tocsc(), r)
                if mode == "cheb20":
                    return Lv.cheb(Lv.A, 1.1 * Lv.lam_full, 1.1 * Lv.lam_full / 100, r, 20)
                z = Lv.M(r)
                for _ in range(int(mode[1:]) - 1):   # "m2": two sweeps of the polynomial smoother
                    z = z + Lv.M(r - Lv.A @ z)
                return z
            z = Lv.M(r); r1 = r - Lv.A @ z
            z = z + P[l] @ vm(P[l].T @ r1, l + 1)
            return z + Lv.M(r - Lv.A @ z)
        x, it, hist = pm.pcg(L0, vm, maxit=150)
        pm.report("dens %g cc V(M,M) %d/%g f=%d" % (dens, terms, ratio, f), it, hist, 92 + (2 * 20 + 2 * (terms - 1) * 14) * 8 / 7)
