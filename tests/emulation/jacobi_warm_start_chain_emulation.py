"""CPU emulation (development aid): does the two-sided Jacobi iteration of kernels_narrow.hip phase 7 need fewer rounds when it
starts from the eigenvector basis the SAME step of the previous sweep in the SAME direction ended with (G' = V_prev^T G V_prev,
V = V_prev . rotations) instead of the identity?  The training chain below takes its SVD split from the emulated iteration itself,
so the gauge each step hands to the next is the iteration's own (cold or warm), as it would be on the device.

    python3 tests/emulation/jacobi_warm_start_chain_emulation.py [sweeps] [M] [L]
"""
import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))
import numpy as np
from oracle import mps_oracle as mo
from jacobi_emulation import pi_perm

KEPT_FRAC, TOL2, ABS = 0.2, 1e-14, 1e-15


def jacobi(G, m, V0=None, stop2=1e-6, maxr=2000):
    """Device rule: stop after ne - 1 consecutive rounds without a rotation with g^2 > stop2 * scale2.  Returns lam, V, rounds."""
    n = G.shape[0]
    tr = np.trace(G)
    G = G / tr
    V = np.eye(n) if V0 is None else V0.copy()
    if V0 is not None:
        G = V0.T @ G @ V0
        G = 0.5 * (G + G.T)
    pi = pi_perm(n); inv = np.argsort(pi)
    abs2 = ABS * ABS

    def kept2_of(G):
        lam_m = np.sort(np.diag(G))[::-1][m - 1]
        return (KEPT_FRAC * max(lam_m, 0.0)) ** 2
    kept2 = kept2_of(G)
    last_big, rounds = -1, 0
    while rounds < maxr:
        J = np.eye(n)
        for k in range(n // 2):
            a, b_, g = G[2 * k, 2 * k], G[2 * k + 1, 2 * k + 1], G[2 * k, 2 * k + 1]
            sc = max(abs(a * b_), kept2); g2 = g * g
            if not g2 > max(TOL2 * sc, abs2): continue
            if g2 > stop2 * sc: last_big = rounds
            d = b_ - a; h = np.sqrt(d * d + 4 * g * g); t = 2 * g / (d + np.copysign(h, d))
            t = float(np.float32(t))
            c = 1 / np.sqrt(1 + t * t); sn = c * t
            J[2 * k, 2 * k] = c; J[2 * k, 2 * k + 1] = sn; J[2 * k + 1, 2 * k] = -sn; J[2 * k + 1, 2 * k + 1] = c
        G = J.T @ G @ J; V = V @ J
        G = G[np.ix_(inv, inv)]; V = V[:, inv]
        rounds += 1
        if rounds % (n - 1) == 0: kept2 = kept2_of(G)
        if rounds - 1 - last_big >= n - 1: break
    return np.diag(G) * tr, V, rounds


class Splitter:
    """stands in for mps_oracle.tensor_svd; remembers the basis per (direction, step) when warm"""
    def __init__(self, warm):
        self.warm, self.store, self.key, self.log = warm, {}, None, []

    def __call__(self, Bmat, m):
        short_rows = Bmat.shape[0] <= Bmat.shape[1]
        W = Bmat if short_rows else Bmat.T
        G = W @ W.T
        V0 = self.store.get(self.key) if self.warm else None
        if V0 is not None and V0.shape[0] != G.shape[0]: V0 = None
        lam, V, rounds = jacobi(G, m, V0)
        if self.warm: self.store[self.key] = V
        o = np.argsort(-lam)
        sig = np.sqrt(np.maximum(lam[o], 0.0))
        Q = V[:, o[:m]]
        sq = np.sqrt(sig[:m])
        short = Q * sq[None, :]
        long_ = (W.T @ Q) / sq[None, :]
        U, S, Vh = np.linalg.svd(Bmat, full_matrices=False)
        best = (U[:, :m] * S[:m]) @ Vh[:m]
        US, SVh = (short, long_.T) if short_rows else (long_, short.T)
        perr = np.abs(US @ SVh - best).max() / np.abs(Bmat).max()
        self.log.append((self.key, G.shape[0], rounds, V0 is not None, perr, np.abs(sig - S).max() / S[0]))
        return US, SVh, sig


def train(warm, sweeps, N=24, M=20, b=1500, L=2, seed=0, nb=4):
    rng = np.random.default_rng(seed); D = 2
    Xs, ys = [], []
    for _ in range(nb):
        p = rng.random((b, N)) * (rng.random((b, N)) > 0.81)
        Xs.append(np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1)); ys.append(rng.integers(0, L, b))
    st = mo.MPSState(N, D, L, M, mo.random_cores(N, M, D, L, rng=rng, scale=M * 0.64))
    mo.calibrate(st, Xs[0])
    sp = Splitter(warm)
    mo.tensor_svd = sp
    per_sweep, accs = [], []
    for sw in range(sweeps):
        X, y1h = Xs[sw % nb], mo.one_hot(ys[sw % nb], L)
        f = mo.forward(st, X); left = st.l_pos == N - 1
        if left: st.Renv = {}
        else: st.Lenv = {}
        n0 = len(sp.log)
        for j in range(N - 1):
            sp.key = (left, j)
            rec = {}
            f = mo.sweep_step(st, f, y1h, 1e-3, 1e-3, True, left, 'softmax', 'full_cross_ent', 0.1, 'fixed', record=rec)
        rows = [r for r in sp.log[n0:] if r[1] == 2 * M or r[1] == min(2 * M, 2 * M * L)]
        per_sweep.append((np.mean([r[2] for r in rows]), max(r[2] for r in rows), max(r[4] for r in rows), max(r[5] for r in rows)))
        accs.append(rec['accuracy'])
    return per_sweep, accs


if __name__ == '__main__':
    sweeps = int(_sys.argv[1]) if len(_sys.argv) > 1 else 12
    M = int(_sys.argv[2]) if len(_sys.argv) > 2 else 20
    L = int(_sys.argv[3]) if len(_sys.argv) > 3 else 2
    keep = mo.tensor_svd
    try:
        cold, acc_c = train(False, sweeps, M=M, L=L)
        warm, acc_w = train(True, sweeps, M=M, L=L)
    finally:
        mo.tensor_svd = keep
    print('full-size steps (n = %d): rounds mean / max | product error max | sigma error max' % (2 * M))
    for sw in range(sweeps):
        c, w = cold[sw], warm[sw]
        print('sweep %2d  cold %6.1f / %4d  %.1e %.1e acc %.3f | warm %6.1f / %4d  %.1e %.1e acc %.3f' % (sw + 1, c[0], c[1], c[2], c[3], acc_c[sw], w[0], w[1], w[2], w[3], acc_w[sw]))
