"""CPU emulation (development aid): how many rounds does the two-sided Jacobi iteration of kernels_narrow.hip phase 7 need
  (A) with the device's stopping rule -- ne - 1 consecutive rounds without a "big" rotation (g^2 > stop2 * scale2), and
  (B) if it stopped as soon as EVERY pair of the current matrix is below a final threshold (g^2 <= fin2 * scale2), which a
      worker thread can evaluate on the block it rewrites anyway,
on merged tensors of a fresh network (the first two sweeps) and of a settled one (sweeps 7-8 over four rotating batches), and
what each leaves of the accuracy of the truncated product against LAPACK's best rank-m approximation.

    python3 tests/emulation/jacobi_direct_stop_emulation.py
"""
import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))
import numpy as np
from oracle import mps_oracle as mo
from jacobi_emulation import pi_perm

KEPT_FRAC, TOL2, ABS = 0.2, 1e-14, 1e-15


def training_mats(N=24, M=20, b=1500, L=2, seed=0, sweeps=8, nb=4):
    rng = np.random.default_rng(seed); D = 2
    Xs, ys = [], []
    for _ in range(nb):
        p = rng.random((b, N)) * (rng.random((b, N)) > 0.81)
        Xs.append(np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1)); ys.append(rng.integers(0, L, b))
    st = mo.MPSState(N, D, L, M, mo.random_cores(N, M, D, L, rng=rng, scale=M * 0.64))
    mo.calibrate(st, Xs[0])
    out = {}
    for sw in range(sweeps):
        X, y1h = Xs[sw % nb], mo.one_hot(ys[sw % nb], L)
        f = mo.forward(st, X); left = st.l_pos == N - 1
        if left: st.Renv = {}
        else: st.Lenv = {}
        for j in range(N - 1):
            rec = {}
            f = mo.sweep_step(st, f, y1h, 1e-3, 1e-3, True, left, 'softmax', 'full_cross_ent', 0.1, 'fixed', record=rec)
            Bm = rec['Bmat'].astype(np.float32).astype(np.float64)
            if min(Bm.shape) == 2 * M: out.setdefault(sw, []).append(Bm)
    return out


def run(W, m, stop2=1e-6, fin2=None, maxr=1200):
    """W: short side first (n x len).  Returns rounds, product error, kept-sigma error."""
    n = W.shape[0]
    G = W @ W.T
    G = G / np.trace(G)
    V = np.eye(n)
    pi = pi_perm(n); inv = np.argsort(pi)
    U, S, Vt = np.linalg.svd(W, full_matrices=False); best = (U[:, :m] * S[:m]) @ Vt[:m]
    abs2 = ABS * ABS
    def kept2_of(G):
        lam_m = np.sort(np.diag(G))[::-1][m - 1]
        return (KEPT_FRAC * max(lam_m, 0.0)) ** 2
    kept2 = kept2_of(G)
    last_big = -1
    rounds = 0
    iu = np.triu_indices(n, 1)
    while rounds < maxr:
        J = np.eye(n)
        for k in range(n // 2):
            a, b_, g = G[2 * k, 2 * k], G[2 * k + 1, 2 * k + 1], G[2 * k, 2 * k + 1]
            sc = max(abs(a * b_), kept2); g2 = g * g
            if not g2 > max(TOL2 * sc, abs2): continue
            if g2 > stop2 * sc: last_big = rounds
            d = b_ - a; h = np.sqrt(d * d + 4 * g * g); t = 2 * g / (d + np.copysign(h, d))
            t = float(np.float32(t))                      # the look-ahead chain is float32
            c = 1 / np.sqrt(1 + t * t); sn = c * t
            J[2 * k, 2 * k] = c; J[2 * k, 2 * k + 1] = sn; J[2 * k + 1, 2 * k] = -sn; J[2 * k + 1, 2 * k + 1] = c
        G = J.T @ G @ J; V = V @ J
        G = G[np.ix_(inv, inv)]; V = V[:, inv]
        rounds += 1
        if rounds % (n - 1) == 0: kept2 = kept2_of(G)
        if fin2 is None:
            if rounds - 1 - last_big >= n - 1: break
        else:
            dg = np.abs(np.diag(G))
            sc = np.maximum(np.outer(dg, dg), kept2)[iu]
            if (G[iu] ** 2 <= np.maximum(fin2 * sc, abs2)).all(): break
    lam = np.diag(G); o = np.argsort(-lam)[:m]; Q = V[:, o]
    prod = Q @ (Q.T @ W)
    perr = np.abs(prod - best).max() / np.abs(W).max()
    tr = (W * W).sum()
    serr = np.abs(np.sqrt(np.maximum(lam[o] * tr, 0)) - S[:m]).max() / S[0]
    return rounds, perr, serr


if __name__ == '__main__':
    mats = training_mats()
    for regime, sws in (('fresh network (sweeps 1-2)', (0, 1)), ('settled (sweeps 7-8)', (6, 7))):
        Ws = [Bm if Bm.shape[0] <= Bm.shape[1] else Bm.T for sw in sws for Bm in mats[sw][::2]]
        print(regime, len(Ws), 'matrices', Ws[0].shape)
        for name, kw in (('device rule, stop2 1e-6', dict()), ('direct stop fin2 1e-8', dict(fin2=1e-8)), ('direct stop fin2 1e-10', dict(fin2=1e-10)),
                         ('direct stop fin2 1e-11', dict(fin2=1e-11)), ('direct stop fin2 1e-12', dict(fin2=1e-12)), ('direct stop fin2 1e-13', dict(fin2=1e-13))):
            r = np.array([run(W, 20, **kw) for W in Ws])
            print('  %-26s rounds mean %6.1f max %4d | product error mean %.1e max %.1e | sigma error max %.1e' % (
                name, r[:, 0].mean(), r[:, 0].max(), r[:, 1].mean(), r[:, 1].max(), r[:, 2].max()))
