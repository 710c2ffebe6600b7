#!/usr/bin/env python3
"""GPU: the mixed-precision decomposition (kernels_narrow.hip phase 7a) on merged tensors of an oracle training run, one
tnml_svd_split call per matrix, with the kernel's own diagnostics (float32 rounds, largest tangent / violation the first
float64 check saw, simultaneous steps, fallback, cycles of the three stages) next to the CPU emulation's figures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'emulation'))
import numpy as np
from tensornetworkforml_amd import _hip
import jacobi_mixed_emulation as em

M = 20
mats = em.collect(passes=int(os.environ.get('PASSES', '6')))
ctx = _hip.Context(24, 2, 2, M, 64)
ctx.debug_enable(2)
worst = dict(prod=0.0, kept=0.0, alls=0.0)
nfall = nmixed = 0
for idx, (sw, B) in enumerate(mats):
    n = min(B.shape)
    if n < 16:
        continue
    m = min(M, n)
    for mode in (True, False):
        ctx.set_svd_mode(mode)
        US, SVh, sig = ctx.svd_split(B, m)
        sc = ctx.step_debug('scalars')
        st = sc[5:]                      # p.stamps[i] = st[i]
        W = B.astype(np.float64)
        U_, S_, Vt_ = np.linalg.svd(W, full_matrices=False)
        best = (U_[:, :m] * S_[:m]) @ Vt_[:m]
        perr = np.abs(US.astype(np.float64) @ SVh.astype(np.float64) - best).max() / np.abs(W).max()
        kerr = (np.abs(sig[:m] - S_[:m]) / S_[:m]).max()
        aerr = np.abs(sig - S_).max() / S_[0]
        if mode:
            nmixed += 1; nfall += int(st[38])
            worst['prod'] = max(worst['prod'], perr); worst['kept'] = max(worst['kept'], kerr); worst['alls'] = max(worst['alls'], aerr)
            Wd = B if B.shape[0] <= B.shape[1] else B.T
            nb = n // 2
            Wd = Wd[np.array([d_ * nb + a for a in range(nb) for d_ in range(2)])].astype(np.float64)
            lam, V, r32, steps, fb = em.svd_mixed(Wd, m)
            print('pass %d #%3d %s n=%d | device: f32 rounds %3.0f t0 %.1e rel0 %.1e steps %.0f failed %.0f t_last %.1e rel_last %.1e cycles f32 %6.0f KE %6.0f steps %6.0f | total before/svd/after %.0f/%.0f/%.0f | '
                  'emulation: rounds %3d steps %d fb %d | perr %.1e kept %.1e all %.1e'
                  % (sw, idx, B.shape, n, st[34], st[35], st[36], st[37], st[38], st[42], st[43], st[39], st[40], st[41], st[0], st[1], st[2], r32, steps, fb, perr, kerr, aerr))
        else:
            print('             float64 iteration: rounds %3.0f cycles svd %.0f | perr %.1e kept %.1e all %.1e' % (st[50], st[1], perr, kerr, aerr))
print('mixed: %d matrices, %d fell back; worst product err %.2e, kept sigma rel err %.2e, all sigma err / sigma_max %.2e' % (nmixed, nfall, worst['prod'], worst['kept'], worst['alls']))
