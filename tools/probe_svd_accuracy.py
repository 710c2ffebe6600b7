#!/usr/bin/env python3
"""GPU probe: accuracy of the in-kernel Jacobi SVD late in training (the steady state, where it stops after
few sweeps).  Every step of one pass is run alone with capture on; the device's singular values and the
product of its two new cores are compared with LAPACK's SVD of the device's own updated merged tensor.
Usage: python tools/probe_svd_accuracy.py [N] [M] [b] [passes_before] [svd_stop2|-] [checked_passes] [rotating_batches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensornetworkforml_amd import _hip                      # noqa: E402
from tensornetworkforml_amd.Network_class import random_canonical_cores  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
P = int(sys.argv[4]) if len(sys.argv) > 4 else 10
STOP = float(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != '-' else None
CHK = int(sys.argv[6]) if len(sys.argv) > 6 else 2
NB = int(sys.argv[7]) if len(sys.argv) > 7 else 1
L, D = 2, 2
rng = np.random.default_rng(0)
batches = []
for _ in range(NB):
    p = rng.random((b, N), dtype=np.float32) * (rng.random((b, N), dtype=np.float32) > 0.81)
    batches.append((np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1).astype(np.float32), rng.integers(0, L, b).astype(np.int32)))
X, y = batches[0]
ctx = _hip.Context(N, D, L, M, b)
if STOP is not None:
    ctx.set_svd_stop(STOP)
ctx.set_cores(random_canonical_cores(N, M, D, L, scale=M * 0.5 * 0.64 * D, rng=rng), 0)
ctx.set_input(X, y)
ctx.scale_cores(1.0 / float(np.exp(ctx.forward_logabsmax() / N)))
hp = (1e-3, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, 'fixed')


def matricize(B, left):
    ml, _, _, mr, _ = B.shape
    if not left:
        return B.reshape(ml * D, D * mr * L)                                  # rows (a, d)
    return np.transpose(B, (0, 1, 4, 2, 3)).reshape(ml * D * L, D * mr)      # rows (a, d, l)


for ps in range(P + CHK):
    if NB > 1:
        ctx.set_input(*batches[ps % NB])
    ctx.forward(want_f=False)
    left = ctx.l_pos == N - 1
    if ps < P:
        ctx.svd_stats(reset=True)
        ctx.sweep(left, N - 1, True, *hp, want_metrics=False, want_f=False)
        s, n, _ = ctx.svd_stats()
        print('pass %2d: %.2f sweeps per SVD' % (ps, s / n))
        continue
    ctx.debug_enable(True)
    worst = dict(sig_abs=0.0, sig_rel_kept=0.0, prod=0.0)
    dump = []
    sweeps = []
    for k in range(N - 1):
        cores0, bond0, lp0 = ctx.get_cores()
        ctx.sweep(left, 1, k == 0, *hp)
        pp = lp0 - 1 if left else lp0
        ml = 1 if pp == 0 else int(bond0[pp - 1])
        mr = 1 if pp == N - 2 else int(bond0[pp + 1])
        Bn = ctx.step_debug('B_new').reshape(ml, D, D, mr, L)
        sig = ctx.step_debug('sigma')
        sc = ctx.step_debug('scalars')
        cores1, bond1, lp1 = ctx.get_cores()
        m = int(bond1[pp])
        A, C = cores1[pp].astype(np.float64), cores1[pp + 1].astype(np.float64)
        prod = np.einsum('adkl,kec->adecl', A, C) if A.ndim == 4 else np.einsum('adk,kecl->adecl', A, C)
        Bm = matricize(Bn, left)
        U, S, Vh = np.linalg.svd(Bm, full_matrices=False)
        best = matricize(prod, left) * 0 + (U[:, :m] * S[:m]) @ Vh[:m]
        worst['sig_abs'] = max(worst['sig_abs'], np.abs(sig - S).max() / S[0])
        worst['sig_rel_kept'] = max(worst['sig_rel_kept'], (np.abs(sig[:m] - S[:m]) / S[:m]).max())
        worst['prod'] = max(worst['prod'], np.abs(matricize(prod, left) - best).max() / np.abs(best).max())
        sweeps.append((int(sc[3]), int(sc[4])))
        dump.append((Bm.astype(np.float32), m, int(sc[3]), int(sc[55]) if len(sc) > 55 else -1))
    ctx.debug_enable(False)
    os.makedirs('gpurun_out', exist_ok=True)
    np.savez('gpurun_out/svd_mats_%s.npz' % (('left' if left else 'right') if CHK == 2 else 'pass%d' % ps), **{'B%d' % i: d[0] for i, d in enumerate(dump)},
             m=np.array([d[1] for d in dump]), sweeps=np.array([d[2] for d in dump]), rounds=np.array([d[3] for d in dump]))
    print('checked pass (%s): sweeps (count, n) %s' % ('left' if left else 'right', sweeps[:8] + ['...'] + sweeps[len(sweeps) // 2:len(sweeps) // 2 + 3]))
    print('   worst |sigma - lapack| / sigma_max = %.2e ; worst relative error of a kept sigma = %.2e ; '
          'worst |A.C - best rank-m| / max = %.2e' % (worst['sig_abs'], worst['sig_rel_kept'], worst['prod']))
ctx.close()
