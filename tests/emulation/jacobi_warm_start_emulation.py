"""CPU emulation: how many Jacobi sweeps does the step's SVD need if the Gram matrix is first rotated
by the eigenvectors found at the same (site, direction) one pass-pair earlier?  The oracle's SVD is
replaced by the emulated Jacobi so that the bond gauges are the ones the device would produce."""
import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))

import sys
import numpy as np
from oracle import mps_oracle as mo
from jacobi_emulation import pi_perm

TOL2, BIG2, FRAC = 1e-14, 1e-4, 0.2


def jacobi(G, m, V0=None, maxs=30, BIG2=None):
    BIG2 = BIG2 if BIG2 is not None else globals()['BIG2']
    n = G.shape[0]
    tr = np.trace(G)
    V = np.eye(n) if V0 is None else V0.copy()
    G = V.T @ G @ V
    pi = pi_perm(n); inv = np.argsort(pi)
    sweeps = 0
    if n == 2 and False:
        pass
    for s in range(maxs):
        lam_m = np.sort(np.diag(G))[::-1][m - 1]
        fl = (FRAC * max(lam_m, 0)) ** 2
        anyr = big = False
        for r in range(max(n - 1, 1)):
            J = np.eye(n)
            for k in range(n // 2):
                a, b_, g = G[2 * k, 2 * k], G[2 * k + 1, 2 * k + 1], G[2 * k, 2 * k + 1]
                sc = max(abs(a * b_), fl); g2 = g * g
                if not g2 > max(TOL2 * sc, (1e-15 * tr) ** 2):
                    continue
                anyr = True
                if g2 > BIG2 * sc:
                    big = True
                d = b_ - a; h = np.sqrt(d * d + 4 * g * g); t = 2 * g / (d + np.copysign(h, d))
                c = 1 / np.sqrt(1 + t * t); sn = c * t
                J[2 * k, 2 * k] = c; J[2 * k, 2 * k + 1] = sn; J[2 * k + 1, 2 * k] = -sn; J[2 * k + 1, 2 * k + 1] = c
            G = J.T @ G @ J; V = V @ J
            if n > 2:
                G = G[np.ix_(inv, inv)]; V = V[:, inv]
        sweeps += 1
        if not anyr or not big:
            break
    return sweeps, np.diag(G).copy(), V


class Harness:
    def __init__(self, warm, big2=None):
        self.warm = warm
        self.big2 = big2
        self.err = []
        self.device_order = True
        self.store = {}
        self.key = None
        self.log = []

    def svd(self, Bmat, m):
        Bm = Bmat.astype(np.float32).astype(np.float64)
        short_rows = Bm.shape[0] <= Bm.shape[1]
        W = Bm if short_rows else Bm.T
        n = W.shape[0]
        # the device orders the short index (bond, d); the oracle's matricisation is (d, bond)
        nb = n // 2
        perm = np.array([d_ * nb + a for a in range(nb) for d_ in range(2)]) if (n % 2 == 0 and self.device_order) else np.arange(n)
        W = W[perm]
        G = W @ W.T
        V0 = self.store.get(self.key) if self.warm else None
        if V0 is not None and V0.shape[0] != n:
            V0 = None
        sweeps, lam, V = jacobi(G, m, V0, BIG2=self.big2)
        o = np.argsort(-lam)
        # keep the SORTED eigenvector basis: next time the rotated Gram matrix is diagonal in descending order
        self.store[self.key] = V[:, o]
        self.log.append((self.key, n, sweeps))
        S = np.sqrt(np.maximum(lam[o], 0))
        Q = V[:, o[:m]]
        sq = np.sqrt(S[:m])
        long_f = (W.T @ Q) / sq[None, :]
        U_, S_, Vt_ = np.linalg.svd(W, full_matrices=False)
        best = (U_[:, :m] * S_[:m]) @ Vt_[:m]
        self.err.append((np.abs(Q @ (Q.T @ W) - best).max() / np.abs(W).max(), np.abs(S - S_).max() / S_[0], n))
        r32 = lambda a: a.astype(np.float32).astype(np.float64)        # the device keeps its cores in float32
        Qo = np.empty_like(Q); Qo[perm] = Q; Q = Qo                    # back to the oracle's row order
        if short_rows:
            return r32(Q * sq[None, :]), r32(long_f.T), S
        return r32(long_f), r32((Q * sq[None, :]).T), S


def run(warm, N=24, M=20, b=1500, L=2, passes=10, lr=1e-3, seed=0, big2=None):
    rng = np.random.default_rng(seed); D = 2
    p = rng.random((b, N)) * (rng.random((b, N)) > 0.81)
    X = np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1)
    y = rng.integers(0, L, b)
    st = mo.MPSState(N, D, L, M, mo.random_cores(N, M, D, L, rng=rng, scale=M * 0.64))
    mo.calibrate(st, X)
    y1h = mo.one_hot(y, L)
    H = Harness(warm, big2)
    orig = mo.tensor_svd
    mo.tensor_svd = H.svd
    out = []
    try:
        for sw in range(passes):
            f = mo.forward(st, X); left = st.l_pos == N - 1
            if left: st.Renv = {}
            else: st.Lenv = {}
            n0 = len(H.log)
            for j in range(N - 1):
                H.key = (st.l_pos, left)
                f = mo.sweep_step(st, f, y1h, lr, 1e-3, True, left, 'softmax', 'full_cross_ent', 0.1, 'fixed')
            sw_counts = [s for (_, n, s) in H.log[n0:] if n == 2 * M]
            out.append(np.mean(sw_counts))
    finally:
        mo.tensor_svd = orig
    e = np.array(H.err)
    return out, f, e


if __name__ == '__main__':
    for lr in (1e-3,):
        for big2 in (1e-4, 1e-8):
            for warm in (False, True):
                sw, f, e = run(warm, lr=lr, big2=big2, passes=12)
                late = e[len(e) * 6 // 10:]
                print('lr %g big2 %g warm %d | sweeps/pass: %s | late passes: worst product err %.1e, worst sigma err %.1e'
                      % (lr, big2, warm, ' '.join('%.2f' % v for v in sw), late[:, 0].max(), late[:, 1].max()), flush=True)
