"""CPU emulation: first-order eigenvector correction after the early-stopped Jacobi iteration.
After the last sweep (all rotations small) the rotated Gram matrix G' = V^T G V has off-diagonals of
relative size ~1e-4.  V <- V (I + X), X_ij = g_ij / (g_jj - g_ii) removes them to second order for one GEMM
instead of one more sweep.  Evaluated on the matrices a device run produced (gpurun_out/svd_mats_*.npz)."""
import os as _os, sys as _sys
_HERE = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _HERE); _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_HERE)))

import sys
import numpy as np
import jacobi_warm_start_emulation as J


def correct(G0, V, m, guard=0.1):
    Gp = V.T @ G0 @ V
    lam = np.diag(Gp).copy()
    n = len(lam)
    order = np.argsort(-lam)
    rank = np.empty(n, int); rank[order] = np.arange(n)
    X = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i == j or (rank[i] >= m and rank[j] >= m):
                continue
            gap = lam[j] - lam[i]
            if abs(Gp[i, j]) <= guard * abs(gap):
                X[i, j] = Gp[i, j] / gap
    lam2 = lam + np.array([sum(Gp[i, j] * X[j, i] for j in range(n)) for i in range(n)])   # second-order eigenvalues
    return V @ (np.eye(n) + X), lam2


def evaluate(W, m, big2, do_corr):
    G = W @ W.T
    sweeps, lam, V = J.jacobi(G, m, BIG2=big2)
    if do_corr:
        V, lam = correct(G, V, m)
    o = np.argsort(-lam)[:m]
    Q = V[:, o]
    U_, S_, Vt_ = np.linalg.svd(W, full_matrices=False)
    best = (U_[:, :m] * S_[:m]) @ Vt_[:m]
    # what the kernel forms: short factor Q sqrt(s), long factor W^T Q / sqrt(s)  -> product Q Q^T W
    perr = np.abs(Q @ (Q.T @ W) - best).max() / np.abs(W).max()
    serr = (np.abs(np.sqrt(np.maximum(lam[o], 0)) - S_[:m]) / S_[:m]).max()
    orth = np.abs(Q.T @ Q - np.eye(m)).max()
    return sweeps, perr, serr, orth


if __name__ == '__main__':
    rows = []
    for side in ('right', 'left'):
        d = np.load(_os.path.join(_os.path.dirname(_os.path.dirname(_HERE)), 'gpurun_out', 'svd_mats_%s.npz' % side))
        for i, m in enumerate(d['m']):
            Bm = d['B%d' % i].astype(np.float64)
            W = Bm if Bm.shape[0] <= Bm.shape[1] else Bm.T
            if W.shape[0] < 8:
                continue
            for big2, corr in ((1e-4, False), (1e-4, True), (1e-3, True), (1e-8, False)):
                rows.append((big2, corr) + evaluate(W, int(m), big2, corr))
    rows = np.array(rows, dtype=object)
    for big2, corr in ((1e-4, False), (1e-4, True), (1e-3, True), (1e-8, False)):
        sel = [r for r in rows if r[0] == big2 and r[1] == corr]
        print('big2 %g correction %d: mean sweeps %.2f | worst product err %.1e | worst rel err of a kept sigma %.1e | worst |Q^T Q - I| %.1e'
              % (big2, corr, np.mean([r[2] for r in sel]), max(r[3] for r in sel), max(r[4] for r in sel), max(r[5] for r in sel)))
