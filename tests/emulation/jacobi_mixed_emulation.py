"""CPU emulation of the mixed-precision decomposition of kernels_narrow.hip (phase 7a).

Merged tensors of an oracle training run (rotating batches, fresh network first) are decomposed by
  (1) two-sided Jacobi in float32 on float(G), position-space tournament and sliding-window stop as on the device;
  (2) float64: K = V32^T G V32, E = V32^T V32 - I, then repeated simultaneous steps Z = I + Y + Y^2/2, Y = X - E/2, where X holds the
      exact 2x2 Jacobi tangent of every pair (at least one index kept) that still violates g^2 <= final2 max(|a b|, kept2).
Printed per pass: float32 rounds, float64 steps, how often a step would have needed a tangent above the device's limit
(-> fallback to the float64 iteration), error of the truncated product against LAPACK's best rank-m approximation, error of the kept
and of all singular values.  The float64-only iteration's round count is printed beside it.

    python tests/emulation/jacobi_mixed_emulation.py            (about a minute)
"""
import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))

import numpy as np
from oracle import mps_oracle as mo
from jacobi_emulation import pi_perm

f32 = np.float32
TOL2, FRAC, MAXT = 1e-14, 0.2, 0.02


def collect(N=24, M=20, b=1500, L=2, passes=10, nb=4, seed=0):
    rng = np.random.default_rng(seed); D = 2
    Xs, ys = [], []
    for _ in range(nb):
        p = rng.random((b, N)) * (rng.random((b, N)) > 0.81)
        Xs.append(np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1)); ys.append(rng.integers(0, L, b))
    st = mo.MPSState(N, D, L, M, mo.random_cores(N, M, D, L, rng=rng, scale=M * 0.64))
    mo.calibrate(st, Xs[0])
    mats = []
    for sw in range(passes):
        X, y1h = Xs[sw % nb], mo.one_hot(ys[sw % nb], L)
        f = mo.forward(st, X); left = st.l_pos == N - 1
        if left: st.Renv = {}
        else: st.Lenv = {}
        for _ in range(N - 1):
            rec = {}
            f = mo.sweep_step(st, f, y1h, 1e-3, 1e-3, True, left, 'softmax', 'full_cross_ent', 0.1, 'fixed', record=rec)
            mats.append((sw, rec['Bmat'].astype(np.float32)))
    return mats


def rot_params(a, b, g, kept2, big2, dt):
    g2 = g * g
    sc = np.maximum(np.abs(a * b), dt(kept2))
    act = g2 > np.maximum(dt(TOL2) * sc, dt(1e-30))
    d = b - a
    den = d + np.copysign(np.sqrt(d * d + dt(4) * g2), d)
    den = np.where(den == 0, dt(1), den)
    return np.where(act, dt(2) * g / den, dt(0)), act & (g2 > dt(big2) * sc)


def jacobi_rounds(G, m, dt, big2, maxrounds=2000):
    n = G.shape[0]
    G = G.astype(dt).copy(); V = np.eye(n, dtype=dt)
    pi = pi_perm(n); inv = np.argsort(pi)
    ev = np.arange(0, n, 2); od = ev + 1
    rounds = last_big = 0; kept2 = 0.0
    while rounds < maxrounds:
        if rounds % (n - 1) == 0:
            kept2 = (FRAC * max(np.sort(np.diag(G).astype(np.float64))[::-1][m - 1], 0)) ** 2
        a = G[ev, ev]; b = G[od, od]; g = G[ev, od]
        t, big = rot_params(a, b, g, max(kept2, 1e-36), big2, dt)
        c = dt(1) / np.sqrt(dt(1) + t * t); s = c * t
        J = np.eye(n, dtype=dt)
        J[ev, ev] = c; J[od, od] = c; J[ev, od] = s; J[od, ev] = -s
        Gn = (J.T @ G @ J).astype(dt)
        rot = t != 0
        Gn[ev[rot], od[rot]] = 0; Gn[od[rot], ev[rot]] = 0
        Gn[ev[rot], ev[rot]] = (a - t * g)[rot]; Gn[od[rot], od[rot]] = (b + t * g)[rot]
        Gn = (Gn + Gn.T) * dt(0.5)
        G = Gn[np.ix_(inv, inv)]
        V = (V @ J).astype(dt)[:, inv]
        rounds += 1
        if big.any(): last_big = rounds
        if rounds - last_big >= n - 1: break
    return rounds, G, V


def tangents(K, E, m, final2):
    n = K.shape[0]
    d = np.diag(K) * (1 - np.diag(E))
    o = np.argsort(-d); kept = np.zeros(n, bool); kept[o[:m]] = True
    kept2 = (FRAC * max(d[o[m - 1]], 0)) ** 2
    g = K - E * (d[:, None] + d[None, :]) / 2; np.fill_diagonal(g, 0)
    sc = np.maximum(np.abs(np.outer(d, d)), kept2)
    act = (g * g > np.maximum(final2 * sc, 1e-30)) & (kept[:, None] | kept[None, :])
    dd = (d[None, :] - d[:, None]).astype(f32).astype(np.float64); gf = g.astype(f32).astype(np.float64)
    den = dd + np.copysign(np.sqrt(dd * dd + 4 * gf * gf), dd); den[den == 0] = 1
    T = np.where(act, 2 * gf / den, 0.0).astype(f32).astype(np.float64)
    X = np.triu(T, 1); X = X - X.T
    return X, np.where(act, g * g / sc, 0).max()


def svd_mixed(W, m, big32=1e-10, final2=1e-12):
    W = W.astype(np.float64); n = W.shape[0]
    G = W @ W.T; sc = 2.0 ** np.frexp(np.trace(G))[1]; G = G / sc
    r32, _, V = jacobi_rounds(G, m, f32, big32)
    V = V.astype(np.float64)
    K = V.T @ (G @ V); K = (K + K.T) / 2
    E = (V.T @ V - np.eye(n)).astype(f32).astype(np.float64)
    steps = 0; fallback = False
    for it in range(4):
        X, rel = tangents(K, E, m, final2)
        if it > 0 and rel == 0: break
        if np.abs(X).max() > MAXT or it >= 3: fallback = True; break
        Y = (X - E / 2).astype(f32).astype(np.float64)
        Z = np.eye(n) + Y + 0.5 * (Y @ Y)
        K = Z.T @ (K @ Z); K = (K + K.T) / 2
        V = V @ Z; E = np.zeros((n, n)); steps += 1
    return np.diag(K) * sc, V, r32, steps, fallback


def main():
    mats = collect()
    M = 20
    rows = []
    for sw, B in mats:
        n = min(B.shape)
        if n < 16: continue
        W = B if B.shape[0] <= B.shape[1] else B.T
        nb = n // 2
        W = W[np.array([d_ * nb + a for a in range(nb) for d_ in range(2)])].astype(np.float64)   # the device's (bond, d) pairing
        m = min(M, n)
        lam, V, r32, steps, fb = svd_mixed(W, m)
        G = W @ W.T; G /= 2.0 ** np.frexp(np.trace(G))[1]
        r64, _, _ = jacobi_rounds(G, m, np.float64, 1e-6)
        o = np.argsort(-lam); Q = V[:, o[:m]]
        U_, S_, Vt_ = np.linalg.svd(W, full_matrices=False)
        best = (U_[:, :m] * S_[:m]) @ Vt_[:m]
        S = np.sqrt(np.maximum(lam[o], 0))
        rows.append((sw, n, r32, steps, fb, r64, np.abs(Q @ (Q.T @ W) - best).max() / np.abs(W).max(),
                     (np.abs(S[:m] - S_[:m]) / S_[:m]).max(), np.abs(S - S_).max() / S_[0], np.abs(Q.T @ Q - np.eye(m)).max()))
    rows = np.array(rows, dtype=float)
    for p in sorted(set(rows[:, 0])):
        r = rows[(rows[:, 0] == p) & (rows[:, 1] == 2 * M)]
        print('pass %2d (n = %d, %2d matrices): float32 rounds %4.0f | float64 steps %.2f | fallbacks %d | float64-only rounds %4.0f | '
              'product err %.1e | kept sigma rel err %.1e | all sigma err / sigma_max %.1e | orthonormality %.1e'
              % (p, 2 * M, len(r), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].sum(), r[:, 5].mean(), r[:, 6].max(), r[:, 7].max(), r[:, 8].max(), r[:, 9].max()))


if __name__ == '__main__':
    main()
