#!/usr/bin/env python3
"""GPU: persistent sweep against the per-step launches, step by step (metrics), to find the first step where they part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tensornetworkforml_amd import _hip
from oracle import mps_oracle as mo

policy, M, N, b, L = os.environ.get('POLICY', 'fixed'), int(os.environ.get('M', 20)), int(os.environ.get('N', 48)), int(os.environ.get('B', 300)), int(os.environ.get('L', 2))
rng = np.random.default_rng(11); D = 2
p = rng.random((b, N)) * (rng.random((b, N)) > 0.6)
X = np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1).astype(np.float32)
y = rng.integers(0, L, b)
st = mo.MPSState(N, D, L, M, mo.random_cores(N, M, D, L, rng=rng, scale=M * 0.5 * 0.64 * D))
mo.calibrate(st, X.astype(np.float64))
cores32 = [c.astype(np.float32) for c in st.cores]
st = mo.MPSState(N, D, L, M, [c.astype(np.float64) for c in cores32])
ctxs = []
for persistent in (True, False):
    ctx = _hip.Context(N, D, L, M, b); ctx.set_cores(cores32, 0); ctx.set_input(X, y); ctx.set_persistent(persistent); ctxs.append(ctx)
X64 = X.astype(np.float64)
for sw in range(3):
    f_o = mo.forward(st, X64); left = st.l_pos == N - 1
    vh = [[], []]
    bonds_before = list(st.bond)
    f_o = mo.sweep(st, X64, y, f_o, 1e-2, 1e-3, L2_flag=True, left_dir=left, var_hist=vh, act_fn='softmax', loss_fn='full_cross_ent', T=0.1, trunc=policy)
    res = []
    for ctx in ctxs:
        fd0 = ctx.forward()
        met, f_d = ctx.sweep(left, N - 1, True, 1e-2, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, policy)
        res.append((met, f_d, fd0))
    print('sweep %d left=%d bonds before: %s' % (sw, left, bonds_before))
    print('   forward err persist %.1e perstep %.1e' % (np.abs(res[0][2] - mo.forward.__globals__['np'].asarray(res[1][2])).max() / np.abs(res[1][2]).max(), 0))
    mae_o = np.array(vh[1])
    for k in range(N - 1):
        d0, d1 = abs(res[0][0][k, 1] - mae_o[k]), abs(res[1][0][k, 1] - mae_o[k])
        flag = ' <<<' if d0 > 5 * max(d1, 1e-6) else ''
        if flag or k < 3 or k > N - 5:
            print('   step %2d MAE oracle %.6f persist %.6f (%.1e) perstep %.6f (%.1e)%s' % (k, mae_o[k], res[0][0][k, 1], d0, res[1][0][k, 1], d1, flag))
    for nm, r in zip(('persist', 'perstep'), res):
        print('   %s: f err vs oracle %.2e' % (nm, np.abs(r[1] - f_o).max() / np.abs(f_o).max()))
