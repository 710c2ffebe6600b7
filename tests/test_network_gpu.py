"""GPU tests through the reference-style Python API (`Network_class.Network`), as a driver written
against the reference would use it."""
import contextlib
import io
import os
import pickle

import numpy as np
import pytest

import golden_util as gu
import tensornetworkforml_amd as pkg
from tensornetworkforml_amd import _hip
from Network_class import Network, _core_to_tensor
from Tensor_class import Tensor
import data_generator as gen
from oracle import mps_oracle as mo

pytestmark = pytest.mark.gpu


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def net_from_golden(d, **kw):
    N, M, L, D = int(d['N']), int(d['M']), int(d['L']), int(d['D'])
    with quiet():
        net = Network(N=N, M=M, D=D, L=L, T=float(d['T']), act_fn=str(d['act_fn']), loss_fn=str(d['loss_fn']),
                      trunc=str(d['policy']), **kw)
    net.As = [_core_to_tensor(c, i, N, i == 0) for i, c in enumerate(gu.indexed(d, 'init_core', N))]
    return net


@pytest.mark.parametrize('name', ['traj_reference_N16_script', 'traj_fixed_N16_script', 'traj_fixed_L3',
                                  'traj_reference_softmax_full_cross_ent_L21', 'traj_fixed_sigmoid_MSE_L20'])
def test_train_loop_matches_reference(name):
    """forward / accuracy / sweep exactly as Network.train drives them, against the reference's
    per-step accuracy, MAE and f."""
    d = gu.load(name)
    N = int(d['N'])
    net = net_from_golden(d)
    X, y = d['X'], d['y']
    k = 0
    for sw in range(int(d['n_sweeps'])):
        f = net.forward(X)
        assert relerr(f.elem, d['sw%d_f_forward' % sw]) < 2e-3
        assert list(f.axes_names) == ['l', 'b']
        left_dir = (net.l_pos == N - 1)
        assert left_dir == bool(d['sw%d_left_dir' % sw])
        vh = [[], []]
        f = net.sweep(X, y, f, float(d['lr']), float(d['wd']), L2_flag=bool(d['L2_flag']), left_dir=left_dir, var_hist=vh)
        ref_acc = [float(d['st%d_accuracy' % (k + j)]) for j in range(N - 1)]
        ref_mae = [float(d['st%d_MAE' % (k + j)]) for j in range(N - 1)]
        assert np.abs(np.array(vh[0]) - ref_acc).max() < 1e-6
        assert np.abs(np.array(vh[1]) - ref_mae).max() < 2e-3
        k += N - 1
        assert relerr(f.elem, d['st%d_f_new' % (k - 1)]) < 5e-3
        assert net.l_pos == (0 if left_dir else N - 1)
    assert relerr(net.forward(X).elem, d['final_f']) < 5e-3
    assert abs(net.accuracy(X, y) - mo.accuracy(d['final_f'], y)) < 1e-9


def test_sweep_step_api_and_env_lists():
    d = gu.load('traj_fixed_softmax_full_cross_ent_L21')
    N, L = int(d['N']), int(d['L'])
    net = net_from_golden(d)
    X, y = d['X'], d['y']
    f = net.forward(X)
    # r_cum_contraction as the reference leaves it after forward at l_pos = 0 (Network_class.py:240-242)
    r = net.r_cum_contraction
    assert len(r) == N and net.l_cum_contraction is None
    assert list(r[0].axes_names) == ['l', 'b'] and relerr(r[0].elem, d['sw0_f_forward']) < 2e-5
    for i in (1, N // 2, N - 1):
        assert list(r[i].axes_names) == ['left', 'b']
        assert relerr(r[i].elem.T, d['sw0_fw_Renv%d' % i]) < 2e-5
    assert len(net.TX) == N and list(net.TX[3].axes_names) == ['b', 'd3']
    one_hot = np.zeros((y.size, L)); one_hot[np.arange(y.size), y] = 1
    for k in range(N - 1):
        vh = [[], []]
        f = net.sweep_step(f, one_hot.T, float(d['lr']), len(y), float(d['wd']), L2_flag=True, left_dir=False, var_hist=vh)
        assert relerr(f.elem, d['st%d_f_new' % k]) < 5e-3
        assert abs(vh[0][0] - float(d['st%d_accuracy' % k])) < 1e-6
        assert net.l_pos == k + 1
        assert len(net.l_cum_contraction) == max(0, k)        # grown one entry per step from the second on
    with pytest.raises(Exception):
        net.sweep_step(f, one_hot.T, 0.1, len(y), 0.1)         # l_pos == N-1: not allowed for a right step
    lc = net.l_cum_contraction
    assert list(lc[0].axes_names) == ['right', 'b'] and lc[0].elem.shape[1] == len(y)
    # forward at an intermediate position raises, as the reference does (Network_class.py:258)
    f = net.forward(X)
    net.sweep_step(f, one_hot.T, 0.1, len(y), 0.1, left_dir=True)
    with pytest.raises(Exception):
        net.forward(X)
    with pytest.raises(AssertionError):
        net.forward(X[:, :-1])


def test_As_names_mutation_and_pickle():
    d = gu.load('traj_fixed_N16_script')
    N, M, L = int(d['N']), int(d['M']), int(d['L'])
    net = net_from_golden(d)
    X, y = d['X'], d['y']
    f0 = net.forward(X).elem
    As = net.As
    assert len(As) == N
    assert sorted(As[0].axes_names) == sorted(['d0', 'right', 'l']) and sorted(As[3].axes_names) == sorted(['left', 'd3', 'right'])
    assert sorted(As[N - 1].axes_names) == sorted(['left', 'd%d' % (N - 1)])
    # in-place edit of a handed-out Tensor is picked up by the next device call (Network.__init__'s
    # calibration loop edits As[i].elem this way, Network_class.py:175-176)
    net.As[5].elem = net.As[5].elem * 2.0
    assert relerr(net.forward(X).elem, 2.0 * f0) < 1e-5
    net.As[5].elem = net.As[5].elem / 2.0
    # a sweep moves the label: the handed-out list is rebuilt with the new shapes
    f = net.forward(X)
    net.sweep(X, y, f, 1e-3, 1e-3)
    As = net.As
    assert 'l' in As[N - 1].axes_names and 'l' not in As[0].axes_names
    # pickle round trip (training_diagonals.py:69-70, test_diagonals.py:41-42)
    blob = pickle.dumps(net)
    net2 = pickle.loads(blob)
    assert net2.l_pos == net.l_pos and net2.trunc == net.trunc
    assert relerr(net2.forward(X).elem, net.forward(X).elem) < 1e-6


def test_shipped_model_state_loads():
    """The reference's trained_diag_model.dat pickles a plain __dict__ whose As are Tensors in the
    reference's own axis orders; __setstate__ takes exactly that."""
    d = gu.load('shipped_diag_model')
    N, L = int(d['N']), int(d['L'])
    lp = int(d['l_pos'])
    As = []
    for i, c in enumerate(gu.indexed(d, 'core', N)):
        T = _core_to_tensor(c, i, N, i == lp)
        T.transpose(list(T.axes_names)[::-1])                    # some other axis order, as in the pickle
        As.append(T)
    state = dict(N=N, D=int(d['D']), L=L, M=int(d['M']), T=float(d['T']), As=As, l_pos=lp, act_fn=str(d['act_fn']),
                 loss_fn=str(d['loss_fn']), TX=None, r_cum_contraction=None, l_cum_contraction=None)
    net = Network.__new__(Network)
    net.__setstate__(state)
    f = net.forward(d['X'])
    assert relerr(f.elem, d['f']) < 2e-4
    assert net.accuracy(d['X'], d['y'], f) == 1.0
    act = net.apply_act_func(f)
    assert np.abs(act.elem - d['act']).max() < 1e-4


@pytest.mark.parametrize('act_fn,loss_fn', [('softmax', 'full_cross_ent'), ('sigmoid', 'MSE'), ('linear', 'cross_entropy'),
                                            ('softmax', 'cross_entropy')])
def test_activation_and_loss_derivative(act_fn, loss_fn):
    rng = np.random.default_rng(1)
    L, b = 3, 37
    with quiet():
        net = Network(N=4, M=2, L=L, act_fn=act_fn, loss_fn=loss_fn)
    f = Tensor(elem=rng.normal(size=(L, b)) * 0.3 + 0.5, axes_names=['l', 'b'])
    y = rng.integers(0, L, b)
    fa = net.apply_act_func(f)
    ref = mo.apply_act_func(f.elem, act_fn, 0.1)
    assert np.abs(fa.elem - ref).max() < 2e-6
    with quiet():
        g = net.compute_loss_derivate(fa, mo.one_hot(y, L))
    gref = mo.compute_loss_derivate(ref, mo.one_hot(y, L), act_fn, loss_fn, 0.1)
    assert relerr(g.elem, gref) < 2e-3


def test_calibration_and_diagonals_training_learns():
    """training_diagonals.py in miniature: the one task the reference learns (val acc -> 1.0)."""
    np.random.seed(0)
    data, label = gen.create_dataset(800, 8, 0.7)
    tr, va, te = gen.prepare_dataset(data, label, 1, 0.2, int(800 * 0.8), 64, 64)
    xcal = next(iter(tr)).X
    with quiet():
        net = Network(N=64, M=4, L=2, calibration_X=xcal, normalize=True, act_fn='softmax', loss_fn='full_cross_ent')
    fcal = net.forward(xcal)
    assert abs(np.abs(fcal.elem).max() - 1.0) < 1e-3               # calibrated: max |f| == 1
    with quiet():
        val_acc, var_hist = net.train(tr, va, lr=0.01, n_epochs=3, weight_dec=1)
    assert var_hist.shape == (3, 2, 63)
    assert val_acc[-1] >= 0.95, val_acc
    # all interior bonds collapsed to 2 under the reference truncation policy
    shapes = [A.elem.shape for A in net.As[2:-2]]
    assert all(sorted(s) == [2, 2, 2] for s in shapes if len(s) == 3)


def test_debug_var_hist_has_seven_series():
    d = gu.load('traj_fixed_softmax_full_cross_ent_L21')
    net = net_from_golden(d)
    X, y = d['X'], d['y']
    f = net.forward(X)
    vh = [[] for _ in range(7)]
    net.sweep(X, y, f, float(d['lr']), float(d['wd']), var_hist=vh, debug=True)
    assert all(len(v) == int(d['N']) - 1 for v in vh)
    ref_acc = [float(d['st%d_accuracy' % j]) for j in range(int(d['N']) - 1)]
    assert np.abs(np.array(vh[2]) - ref_acc).max() < 1e-6
    assert abs(vh[5][0] - float(d['st0_L2_loss'])) <= 1e-3 * abs(float(d['st0_L2_loss'])) + 1e-9


def _named_to_canon(Tn, p):
    """Merged Tensor on sites (p, p+1) -> (ml, D, D, mr, L) array, by axis name."""
    names = [str(a) for a in Tn.axes_names]
    want = ['left', 'd%d' % p, 'd%d' % (p + 1), 'right', 'l']
    present = [n for n in want if n in names]
    arr = np.transpose(Tn.elem, [names.index(n) for n in present])
    return arr.reshape([arr.shape[present.index(n)] if n in present else 1 for n in want])


@pytest.mark.parametrize('name', ['traj_fixed_softmax_full_cross_ent_L21', 'traj_reference_sigmoid_MSE_L21',
                                  'traj_fixed_linear_MSE_L20', 'traj_fixed_L3'])
def test_step_composed_of_update_B_L2_and_tensor_svd(name):
    """The reference's sweep_step written out with the three methods it is made of
    (contract -> update_B [-> compute_L2_reg] -> aggregate -> tensor_svd -> As, Network_class.py:484-566),
    each checked against the oracle's record of the same step, for a right and then a left sweep."""
    from custom_linalg_tools import contract
    d = gu.load(name)
    N, D, L, M = int(d['N']), int(d['D']), int(d['L']), int(d['M'])
    policy, L2_flag = str(d['policy']), bool(d['L2_flag'])
    lr, wd = float(d['lr']), float(d['wd'])
    X, y = d['X'], d['y']
    net = net_from_golden(d)
    st = mo.MPSState(N, D, L, M, gu.indexed(d, 'init_core', N), 0)
    y1h = mo.one_hot(y, L)
    kw = dict(act_fn=str(d['act_fn']), loss_fn=str(d['loss_fn']), T=float(d['T']), trunc=policy)
    for left_dir in (False, True):
        ldf = int(left_dir)
        f = net.forward(X)
        f_o = mo.forward(st, X)
        for k in range(N - 1):
            l = net.l_pos
            p = l - ldf
            assert l == st.l_pos
            rec = {}
            st_before = st.copy()
            st_before.Ln, st_before.Rn = {}, {}
            f_o = mo.sweep_step(st, f_o, y1h, lr, wd, L2_flag=L2_flag, left_dir=left_dir, record=rec, **kw)
            As = net.As
            B = contract(As[l - ldf], As[l + 1 - ldf], 'right', 'left')
            # singular vectors carry a sign gauge: the outer bonds of B differ by +-1 factors (s, t)
            sg, tg = gu.gauge_signs(_named_to_canon(B, p), rec['B'])
            assert relerr(_named_to_canon(B, p), gu.regauge(rec['B'], sg, tg)) < 2e-3
            if L2_flag:
                loss, grad = net.compute_L2_reg(B, wd, left_dir)
                loss_o, grad_o = mo.compute_L2_reg(st_before, rec['B'], p, wd)
                assert abs(loss - loss_o) < 2e-3 * abs(loss_o), (k, loss, loss_o)
                assert list(grad.axes_names) == list(B.axes_names)
                assert relerr(_named_to_canon(grad, p), gu.regauge(grad_o, sg, tg)) < 2e-3
            vh = [[], []]
            fT = Tensor(elem=np.asarray(f.elem), axes_names=['l', 'b'])
            Bn = net.update_B(B, fT, y1h, lr, wd, L2_flag=L2_flag, ldf=ldf, var_hist=vh)
            assert list(Bn.axes_names) == list(B.axes_names)
            assert relerr(_named_to_canon(Bn, p), gu.regauge(rec['B_new'], sg, tg)) < 2e-3, (left_dir, k)
            assert abs(vh[0][0] - rec['accuracy']) < 1e-6 and abs(vh[1][0] - rec['MAE']) < 2e-3
            assert net.l_pos == l                                   # update_B moves nothing
            # the SVD split, fed the way sweep_step feeds it (:528-560)
            Bm = Tensor(elem=Bn.elem.copy(), axes_names=list(Bn.axes_names))
            names = [str(a) for a in Bm.axes_names]
            if not left_dir:
                Bm.aggregate(axes_names=[n for n in ['d%d' % l, 'left'] if n in names], new_ax_name='i')
                Bm.aggregate(axes_names=[n for n in ['d%d' % (l + 1), 'right', 'l'] if n in names], new_ax_name='j')
            else:
                Bm.aggregate(axes_names=[n for n in ['d%d' % (l - 1), 'left', 'l'] if n in names], new_ax_name='i')
                Bm.aggregate(axes_names=[n for n in ['d%d' % l, 'right'] if n in names], new_ax_name='j')
            Bm.transpose(['i', 'j'])
            mat = Bm.elem.copy()
            TU, TSVh = net.tensor_svd(Bm, left_dir)
            m = rec['m']
            # disaggregate puts the components of 'i' / 'j' first (Tensor_class.py:162-199)
            assert [str(a) for a in TU.axes_names][-1] == 'right' and [str(a) for a in TSVh.axes_names][-1] == 'left'
            assert TU.elem.shape[-1] == m and TSVh.elem.shape[-1] == m
            US = TU.elem.reshape(-1, m)
            SVh = TSVh.elem.reshape(-1, m).T
            U, S, Vh = np.linalg.svd(mat, full_matrices=False)
            best = (U[:, :m] * S[:m]) @ Vh[:m]
            assert np.abs(US @ SVh - best).max() < 2e-3 * max(S[0], 1e-30), (left_dir, k)
            # sqrt(S) on both factors: column norms of U sqrt(S) and row norms of sqrt(S) Vh are sqrt(sigma)
            assert np.abs(np.linalg.norm(US, axis=0) ** 2 - S[:m]).max() < 2e-3 * S[0]
            assert np.abs(np.linalg.norm(SVh, axis=1) ** 2 - S[:m]).max() < 2e-3 * S[0]
            # carry on with the fused step so that both sides stay in lock-step
            f = net.sweep_step(f, y1h, lr, len(y), wd, L2_flag=L2_flag, left_dir=left_dir)
            assert relerr(f.elem, f_o) < 5e-3


def test_tensor_svd_argument_errors_match_reference():
    with quiet():
        net = Network(N=6, M=4, L=2)
    with pytest.raises(TypeError):
        net.tensor_svd(np.zeros((4, 4)))
    with pytest.raises(ValueError):
        net.tensor_svd(Tensor(elem=np.zeros((2, 2, 2)), axes_names=['i', 'j', 'k']))


def test_predict_equals_forward_and_leaves_the_resident_batch_alone():
    d = gu.load('traj_fixed_N16_script')
    N, L = int(d['N']), int(d['L'])
    X, y = d['X'], d['y']
    rng = np.random.default_rng(11)
    Xv = np.stack([np.sin(rng.random((37, N))), np.cos(rng.random((37, N)))], -1)
    lr, wd = float(d['lr']), float(d['wd'])

    def run(with_predict):
        net = net_from_golden(d)
        out = []
        for sw in range(2):
            f = net.forward(X)
            if with_predict:
                fp = net.predict(Xv)                               # a validation batch in between
                assert list(fp.axes_names) == ['l', 'b'] and fp.elem.shape == (L, 37)
                out.append(fp.elem.copy())
                assert len(net.TX) == N and net.TX[0].elem.shape[0] == len(y)   # TX still the training batch
            f = net.sweep(X, y, f, lr, wd, left_dir=(net.l_pos == N - 1))
            out.append(f.elem.copy())
        return net, out

    net_a, with_p = run(True)
    net_b, without = run(False)
    # the sweeps are bit-identical with and without the interleaved prediction
    assert np.array_equal(with_p[1], without[0]) and np.array_equal(with_p[3], without[1])
    # predict(X) == forward(X), at both label ends (same kernel, same arithmetic)
    for net in (net_a,):
        assert np.array_equal(net.predict(Xv).elem, net.forward(Xv).elem)
    with quiet():
        val_acc, _ = net_b.train([list(zip(X, y))], [list(zip(Xv, rng.integers(0, L, 37)))], lr, n_epochs=1, weight_dec=wd)
    assert 0.0 <= val_acc[0] <= 1.0
    with pytest.raises(AssertionError):
        net_b.predict(Xv[:, :-1])


# ---------------------------------------------------------------------------------------------------------------
# the shipped driver scripts, end to end (counterparts of training_diagonals.py:54-70, training_binary_MNIST.py:57-90)
# ---------------------------------------------------------------------------------------------------------------
def _write_idx(path, arr):
    import struct
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    with open(path, 'wb') as fh:
        fh.write(struct.pack('>HBB', 0, 0x08, arr.ndim))
        fh.write(struct.pack('>' + 'I' * arr.ndim, *arr.shape))
        fh.write(arr.tobytes())


def _synthetic_mnist(root, n_train, n_test, seed):
    """28 x 28 uint8 'digits' in MNIST's IDX format: class 0 a ring, class 1 a vertical bar (what makes 0 / 1 separable after
    2 x 2 max pooling), eight other labels as noise images that the binary script has to filter out."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:28, 0:28]
    ring = ((np.hypot(yy - 13.5, xx - 13.5) > 6) & (np.hypot(yy - 13.5, xx - 13.5) < 10)).astype(np.float64)
    bar = ((np.abs(xx - 13.5) < 2.5) & (np.abs(yy - 13.5) < 10)).astype(np.float64)

    def draw(n):
        lab = rng.integers(0, 10, n)
        lab[: n // 2] = rng.integers(0, 2, n // 2)                  # plenty of zeros and ones
        img = np.where((lab == 0)[:, None, None], ring[None], np.where((lab == 1)[:, None, None], bar[None], rng.random((n, 28, 28)) > 0.8))
        img = img * (160 + 95 * rng.random((n, 1, 1))) + 30 * rng.random((n, 28, 28))
        return np.clip(img, 0, 255).astype(np.uint8), lab.astype(np.uint8)

    tr, trl = draw(n_train)
    te, tel = draw(n_test)
    os.makedirs(root, exist_ok=True)
    _write_idx(os.path.join(root, 'train-images-idx3-ubyte'), tr)
    _write_idx(os.path.join(root, 'train-labels-idx1-ubyte'), trl)
    _write_idx(os.path.join(root, 't10k-images-idx3-ubyte'), te)
    _write_idx(os.path.join(root, 't10k-labels-idx1-ubyte'), tel)
    return int(((trl < 2).sum() + (tel < 2).sum()))


def test_training_diagonals_script(tmp_path, monkeypatch):
    """training_diagonals.main with the reference's defaults except the sample count: the pickle is written and loads, var_hist
    has the reference's shape (n_epochs, 2, n_batches * (N - 1)) (Network_class.py:314-318), the task is learnt."""
    import pickle
    from tensornetworkforml_amd import training_diagonals as script
    monkeypatch.chdir(tmp_path)
    np.random.seed(3)
    out = str(tmp_path / 'diag.dat')
    with quiet():
        val_acc, var_hist = script.main(['--n_samples', '2000', '--n_train_batch', '2', '--n_epochs', '3', '--out', out])
    n_batches, N = 2, 64
    assert var_hist.shape == (3, 2, n_batches * (N - 1))
    assert len(val_acc) == 3 and val_acc[-1] >= 0.95
    assert np.isfinite(var_hist).all() and var_hist[-1, 0].mean() > 0.9
    with open(out, 'rb') as fh:
        net = pickle.load(fh)
    assert net.N == N and net.M == 10 and len(net.As) == N
    X = gen.psi(gen.create_dataset(64, 8, 0.7)[0].reshape(64, -1))
    f = net.forward(X)                              # the unpickled network runs on the device again
    assert f.elem.shape == (2, 64) and np.isfinite(f.elem).all()


def test_training_binary_mnist_script(tmp_path, monkeypatch):
    """training_binary_MNIST.main on synthetic IDX files: the IDX reader, the 0/1 filter, 2 x 2 max pooling to N = 196, ten
    training batches per epoch as in the reference's defaults, pickle out.  Only the script path is asserted: whether the
    two-site optimiser LEARNS at N = 196 is the subject of the accuracy-parity tests against the oracle
    (tests/test_true_shapes_gpu.py::test_accuracy_parity_n196_*), and on these synthetic shapes three epochs at bond 3 or 10 leave
    the validation accuracy at chance for every hyper-parameter set tried (the learnable task of the reference is the diagonals)."""
    import pickle
    from tensornetworkforml_amd import training_binary_MNIST as script
    root = str(tmp_path / 'datasets')
    n01 = _synthetic_mnist(root, 3000, 600, 5)
    tr, trl, te, tel = gen.get_MNIST_dataset(root)
    assert tr.shape == (3000, 28, 28) and te.shape == (600, 28, 28) and trl.dtype == np.int64 and set(np.unique(trl)) <= set(range(10))
    assert script.pooling(tr[:3]).shape == (3, 14, 14) and script.pooling(tr[:3]).max() == tr[:3].max()
    monkeypatch.chdir(tmp_path)
    np.random.seed(4)
    out = str(tmp_path / 'mnist.dat')
    with quiet():
        val_acc, var_hist = script.main(['--data_dir', root, '--n_epochs', '2', '--normalise', '--lr', '0.01', '--L2_decay', '1e-3', '--out', out])
    n_batches, N = 10, 196
    assert int(n01 * 0.8 / 10) > 50                 # every batch has samples
    assert var_hist.shape == (2, 2, n_batches * (N - 1))
    assert len(val_acc) == 2 and np.isfinite(var_hist).all() and all(0.0 <= v <= 1.0 for v in val_acc)
    assert 0.0 <= var_hist[:, 0].min() and var_hist[:, 0].max() <= 1.0
    with open(out, 'rb') as fh:
        net = pickle.load(fh)
    assert net.N == N and net.M == 3 and net.L == 2
    # raw 0..255 pixels (the reference's own behaviour): runs, finite, nothing to learn from
    with quiet():
        val_raw, hist_raw = script.main(['--data_dir', root, '--n_epochs', '1', '--n_train_batch', '4', '--out', str(tmp_path / 'raw.dat')])
    assert hist_raw.shape == (1, 2, 4 * (N - 1)) and np.isfinite(hist_raw).all() and 0.0 <= val_raw[0] <= 1.0
