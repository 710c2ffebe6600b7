"""GPU probe (development aid, drives the oracle as checker): precision of the two forward-chain kernels against the float64
oracle, and how the N = 196 / bond 20 re-based accuracy comparison of tests/test_true_shapes_gpu.py reacts to the choice.

    python3 tests/emulation/chain_precision_probe.py
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
import test_true_shapes_gpu as tt
from oracle import mps_oracle as mo
from tensornetworkforml_amd import _hip


def env_precision():
    N, M, b, L, D = 196, 20, 1000, 2, 2
    X, y = tt._diag14(5)
    st, cores32 = tt.calibrated_pair(N, M, D, L, X[:b], 3)
    f_o = mo.forward(st, X[:b].astype(np.float64))
    for plain in (1, 0):
        ctx = _hip.Context(N, D, L, M, b)
        ctx.set_chain_path(plain)
        ctx.set_cores(cores32, 0)
        ctx.set_input(X[:b], y[:b])
        f_d = ctx.forward()
        worst = 0.0
        for site in sorted(st.Renv):
            e_o = st.Renv[site]
            e_d = ctx.get_env(_hip.SIDE_RIGHT, site).astype(np.float64)
            worst = max(worst, np.abs(e_d - e_o).max() / np.abs(e_o).max())
        print('chain %s: f rel err %.2e, worst environment rel err over %d sites %.2e' % ('plain FMA' if plain else 'MFMA     ', tt.relerr(f_d, f_o), len(st.Renv), worst))
        ctx.close()


def n196(plain, chunk=15):
    N, M, b, L, D = 196, 20, 1000, 2, 2
    X, y = tt._diag14(5)
    st, cores32 = tt.calibrated_pair(N, M, D, L, X[:b], 3)
    ctx = _hip.Context(N, D, L, M, b)
    ctx.set_chain_path(plain)
    ctx.set_cores(cores32, 0)
    hp = (1e-2, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, 'fixed')
    okw = dict(L2_flag=True, act_fn='softmax', loss_fn='full_cross_ent', T=0.1, trunc='fixed')
    gaps = []
    for i in range(4):
        Xb, yb = X[i * b:(i + 1) * b], y[i * b:(i + 1) * b]
        X64, y1h = Xb.astype(np.float64), mo.one_hot(yb, L)
        ctx.set_input(Xb, yb)
        f_o = mo.forward(st, X64)
        ctx.forward(want_f=False)
        left = st.l_pos == N - 1
        if left:
            st.Renv = {}
        else:
            st.Lenv = {}
        for c0 in range(0, N - 1, chunk):
            accs = []
            for j in range(chunk):
                rec = {}
                f_o = mo.sweep_step(st, f_o, y1h, hp[0], hp[1], left_dir=left, record=rec, **okw)
                accs.append(rec['accuracy'])
            met, f_d = ctx.sweep(left, chunk, c0 == 0, *hp)
            gaps.append(np.abs(met[:, 0] - np.array(accs)))
            tt.resync(st, ctx, left)
            f_o = f_d.astype(np.float64)
    ctx.close()
    g = np.concatenate(gaps)
    print('n196 %s chunk %d: worst step gap %.4f, steps with gap > 0.005: %d of %d, > 0.002: %d; per-sweep worst %s' % (
        'plain FMA' if plain else 'MFMA     ', chunk, g.max(), int((g > 0.005 + 1e-6).sum()), g.size, int((g > 0.002 + 1e-6).sum()),
        [float(g[k * (N - 1):(k + 1) * (N - 1)].max()) for k in range(4)]))


if __name__ == '__main__':
    env_precision()
    for plain in (1, 0):
        n196(plain)
    n196(0, chunk=5)
