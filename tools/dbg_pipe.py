#!/usr/bin/env python3
"""GPU debugging aid: pipelined vs classic step against the oracle, quantity by quantity, on a golden trajectory."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import golden_util as gu
from oracle import mps_oracle as mo
from tensornetworkforml_amd import _hip

def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

name = sys.argv[1] if len(sys.argv) > 1 else 'traj_fixed_softmax_full_cross_ent_L21'
d = gu.load(name)
N, M, L, D = int(d['N']), int(d['M']), int(d['L']), int(d['D'])
kw = dict(lr=float(d['lr']), weight_dec=float(d['wd']), L2_flag=bool(d['L2_flag']), act_fn=str(d['act_fn']), loss_fn=str(d['loss_fn']), T=float(d['T']), trunc=str(d['policy']))
X, y = d['X'], d['y']
for pipe in (False, True):
    print('==== pipeline', pipe, name, 'N', N, 'M', M, 'L', L, 'b', len(y))
    cores0 = gu.indexed(d, 'init_core', N)
    st = mo.MPSState(N, D, L, M, cores0, 0)
    ctx = _hip.Context(N, D, L, M, X.shape[0])
    ctx.set_cores(cores0, 0); ctx.set_input(X, y)
    ctx.set_step_pipeline(pipe)
    ctx.debug_enable(True)
    y1h = mo.one_hot(y, L)
    try:
        for sw in range(2):
            f_o = mo.forward(st, X); f_d = ctx.forward()
            left = st.l_pos == N - 1
            if left: st.Renv = {}
            else: st.Lenv = {}
            for j in range(N - 1):
                rec = {}
                f_o = mo.sweep_step(st, f_o, y1h, left_dir=left, record=rec, **kw)
                try:
                    met, f_d = ctx.sweep(left, 1, j == 0, kw['lr'], kw['weight_dec'], kw['L2_flag'], kw['act_fn'], kw['loss_fn'], kw['T'], kw['trunc'])
                    err = None
                except Exception as e:
                    err = str(e)[:80]
                    f_d = ctx.get_f(); met = np.zeros((1, 2))
                shp = rec['B'].shape
                q = {}
                for key, ok in (('B', 'B'), ('dB_raw', 'dB_raw'), ('B_new', 'B_new'), ('L2_grad', 'L2_grad')):
                    v = ctx.step_debug(key).reshape(shp)
                    sa, tc = gu.gauge_signs(ctx.step_debug('B').reshape(shp), rec['B'])
                    q[key] = relerr(v, gu.regauge(rec[ok], sa, tc))
                print(sw, j, 'f %.1e' % relerr(f_d, f_o), ' '.join('%s %.1e' % kv for kv in q.items()), 'acc', float(met[0, 0]), rec['accuracy'], 'mae', float(met[0, 1]), rec['MAE'], 'finite', bool(np.isfinite(f_d).all()), err or '')
                if err: raise SystemExit
    except SystemExit:
        pass
    ctx.close()
