#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports /root/reference/TensorNetwork/{Tensor_class,custom_linalg_tools,Network_class}.py,
drives `Network.forward`, `Network.sweep_step` and whole sweeps on small seeded problems and
dumps inputs and outputs as float64 .npz files.  Nothing of the reference's text is stored:
the fixtures are arrays only.  The only code of ours mixed in is

  * `FixedBondNetwork.tensor_svd` -- an override that truncates to m = min(M, len(S)) on both
    factors.  Fixtures produced with it are labelled policy="fixed" and are NOT reference
    behaviour (the reference collapses every bond to the left bond of the merged tensor and
    crashes for L > 2); they pin the build's own fixed-bond policy
    against the reference's contraction / update / clipping arithmetic.
  * recorders wrapped around `update_B`, `compute_L2_reg` and `tensor_svd` that copy their
    arguments and results.

Every array is converted to the canonical layouts of oracle/mps_oracle.py before saving.
"""
import contextlib
import io
import os
import pickle
import re
import sys

import numpy as np

REF = '/root/reference/TensorNetwork'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, '..', '..'))

import Network_class as ref_tn            # noqa: E402  (the reference)
from Tensor_class import Tensor           # noqa: E402  (the reference)
from oracle.mps_oracle import core_from_named  # noqa: E402


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


class FixedBondNetwork(ref_tn.Network):
    """NOT reference behaviour: only the truncation rule of tensor_svd is replaced."""

    def tensor_svd(self, T, left_dir=False, threshold=0.999):
        U, S, Vh = np.linalg.svd(np.array(T.elem, copy=True), full_matrices=False)
        m = min(self.M, len(S))
        sq = np.sqrt(S[:m])
        TU = Tensor(elem=U[:, :m] * sq[None, :], axes_names=['i', 'right'])
        TV = Tensor(elem=sq[:, None] * Vh[:m, :], axes_names=['left', 'j'])
        TU.aggregations['i'] = T.aggregations['i']
        TV.aggregations['j'] = T.aggregations['j']
        TU.disaggregate('i')
        TV.disaggregate('j')
        return TU, TV


def psi(p):
    return np.transpose(np.array((np.sin(np.pi * p / 2), np.cos(np.pi * p / 2))), [1, 2, 0])


def canon_cores(net):
    return [core_from_named(A.elem, A.axes_names, i, net.N) for i, A in enumerate(net.As)]


def canon_B(T, p):
    """Reference merged tensor (named axes, some absent at the chain ends) -> (a, d, d', c, l)."""
    names = [str(a) for a in T.axes_names]
    elem = np.asarray(T.elem)
    want = ['left', 'd' + str(p), 'd' + str(p + 1), 'right', 'l']
    order = [names.index(w) for w in want if w in names]
    out = np.transpose(elem, order)
    pos = 0
    for k, w in enumerate(want):
        if w not in names:
            out = np.expand_dims(out, k)
    return np.ascontiguousarray(out)


def env_bm(T):
    """Reference environment Tensor ('left'|'right'|'l', 'b') -> (b, m)."""
    names = [str(a) for a in T.axes_names]
    e = np.asarray(T.elem)
    return np.ascontiguousarray(e.T if names[-1] == 'b' else e)


def f_lb(T):
    names = [str(a) for a in T.axes_names]
    e = np.asarray(T.elem)
    return np.ascontiguousarray(e if names[0] == 'l' else e.T)


class Recorder:
    """Wraps update_B / compute_L2_reg / tensor_svd of one reference Network instance."""

    def __init__(self, net):
        self.net = net
        self.rec = {}
        self._ub, self._l2, self._svd = net.update_B, net.compute_L2_reg, net.tensor_svd
        net.update_B = self.update_B
        net.compute_L2_reg = self.compute_L2_reg
        net.tensor_svd = self.tensor_svd

    def update_B(self, B, f, y, lr, weight_dec, **kw):
        ldf = kw.get('ldf', 0)
        self.p = self.net.l_pos - ldf
        self.rec['B'] = canon_B(B, self.p)
        out = self._ub(B, f, y, lr, weight_dec, **kw)
        self.rec['B_new'] = canon_B(out, self.p)
        return out

    def compute_L2_reg(self, B, weight_dec=0.001, left_dir=False):
        loss, der = self._l2(B, weight_dec, left_dir)
        self.rec['L2_loss'] = np.float64(loss)
        self.rec['L2_grad'] = canon_B(der, self.p)
        return loss, der

    def tensor_svd(self, T, left_dir=False, threshold=0.999):
        self.rec['Bmat'] = np.array(T.elem, copy=True)
        self.rec['S'] = np.linalg.svd(self.rec['Bmat'], compute_uv=False)
        return self._svd(T, left_dir, threshold)


def make_net(cls, N, M, L, X, act_fn, loss_fn, seed, T=0.1):
    np.random.seed(seed)
    with quiet():
        net = cls(N=N, M=M, L=L, T=T, normalize=True, calibration_X=X, act_fn=act_fn, loss_fn=loss_fn)
    return net


def dump_envs(net, out, prefix, only=None):
    """`only`: set of (side, site) to keep -- a step snapshot stores just the environments the
    step reads, to keep the fixtures small."""
    N = net.N
    full = {}
    _dump_envs(net, full, prefix)
    for k, v in full.items():
        m = re.match(r'.*(Lenv|Renv)(\d+)$', k)
        if only is None or (m.group(1), int(m.group(2))) in only:
            out[k] = v


def _dump_envs(net, out, prefix):
    N = net.N
    if net.r_cum_contraction is not None:
        r = net.r_cum_contraction
        if len(r) == N:        # built by forward: r[i] spans sites i..N-1 (r[0] is f)
            for i in range(1, N):
                out[prefix + 'Renv%d' % i] = env_bm(r[i])
        else:                  # grown by a left sweep: j-th appended spans sites N-1-j..N-1
            for j, t in enumerate(r):
                out[prefix + 'Renv%d' % (N - 1 - j)] = env_bm(t)
    if net.l_cum_contraction is not None:
        lc = net.l_cum_contraction
        n = N - 1 if len(lc) == N else len(lc)   # forward's last entry is f
        for i in range(n):
            out[prefix + 'Lenv%d' % i] = env_bm(lc[i])


def trajectory(name, cls, N, M, L, b, act_fn, loss_fn, lr, wd, L2_flag, n_sweeps, seed,
               policy, T=0.1, x_zero_frac=0.0):
    """Free-running sweeps of the reference with a full snapshot before every step."""
    rng = np.random.default_rng(seed)
    p = rng.random((b, N))
    if x_zero_frac:
        p = p * (rng.random((b, N)) > x_zero_frac)
    X = psi(p)
    y = rng.integers(0, L, b)
    net = make_net(cls, N, M, L, X, act_fn, loss_fn, seed, T)
    out = dict(N=N, M=M, L=L, D=2, b=b, T=T, lr=lr, wd=wd, L2_flag=L2_flag, n_sweeps=n_sweeps,
               act_fn=act_fn, loss_fn=loss_fn, policy=policy, X=X, y=y,
               numpy_version=np.__version__)
    for i, c in enumerate(canon_cores(net)):
        out['init_core%d' % i] = c
    rec = Recorder(net)
    step = 0
    for sw in range(n_sweeps):
        with quiet():
            f = net.forward(X)
        left_dir = (net.l_pos == N - 1)
        out['sw%d_left_dir' % sw] = left_dir
        out['sw%d_f_forward' % sw] = f_lb(f)
        dump_envs(net, out, 'sw%d_fw_' % sw)
        one_hot = np.zeros((y.size, L))
        one_hot[np.arange(y.size), y] = 1
        yT = one_hot.T
        if left_dir:
            net.r_cum_contraction = []
        else:
            net.l_cum_contraction = []
        for _ in range(N - 1):
            pre = 'st%d_' % step
            out[pre + 'l_pos'] = net.l_pos
            out[pre + 'left_dir'] = left_dir
            cc = canon_cores(net)
            out[pre + 'bond_before'] = np.array([c.shape[2] for c in cc][:-1])
            out[pre + 'cores_flat'] = np.concatenate([c.ravel() for c in cc])   # shapes follow from bond_before + l_pos
            pp = net.l_pos - int(left_dir)
            dump_envs(net, out, pre, only={('Lenv', pp - 2), ('Lenv', pp - 1), ('Renv', pp + 2), ('Renv', pp + 3)})
            out[pre + 'f_prev'] = f_lb(f)
            vh = [[], []]
            rec.rec = {}
            with quiet():
                f = net.sweep_step(f, yT, lr, b, wd, L2_flag=L2_flag, left_dir=left_dir, var_hist=vh)
            out[pre + 'f_new'] = f_lb(f)
            out[pre + 'accuracy'] = vh[0][0]
            out[pre + 'MAE'] = vh[1][0]
            for k, v in rec.rec.items():
                out[pre + k] = v
            out[pre + 'bond'] = np.array([core_from_named(A.elem, A.axes_names, i, N).shape[2]
                                          for i, A in enumerate(net.As)][:-1])
            step += 1
    out['n_steps'] = step
    for i, c in enumerate(canon_cores(net)):
        out['final_core%d' % i] = c
    out['final_l_pos'] = net.l_pos
    with quiet():
        out['final_f'] = f_lb(net.forward(X))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, 'steps', step)


def light_trajectory(name, cls, N, M, L, b, act_fn, loss_fn, lr, wd, L2_flag, n_sweeps, seed, policy, T=0.1,
                     x_zero_frac=0.0):
    """Free-running sweeps of the reference at a large bond dimension: inputs, initial cores and, per step, only the
    small gauge-invariant outputs (f, metrics, singular values, L2 loss, bonds).  The merged tensors of these shapes
    (100 x 1000 at M = 50, L = 10) would make the fixture tens of MB; the oracle is held to them element-wise on the
    small trajectories, here the point is the reference's arithmetic at the bench's bond dimensions."""
    rng = np.random.default_rng(seed)
    p = rng.random((b, N))
    if x_zero_frac:
        p = p * (rng.random((b, N)) > x_zero_frac)
    X = psi(p)
    y = rng.integers(0, L, b)
    net = make_net(cls, N, M, L, X, act_fn, loss_fn, seed, T)
    out = dict(N=N, M=M, L=L, D=2, b=b, T=T, lr=lr, wd=wd, L2_flag=L2_flag, n_sweeps=n_sweeps, act_fn=act_fn,
               loss_fn=loss_fn, policy=policy, X=X, y=y, numpy_version=np.__version__)
    for i, c in enumerate(canon_cores(net)):
        out['init_core%d' % i] = c
    rec = Recorder(net)
    step = 0
    for sw in range(n_sweeps):
        with quiet():
            f = net.forward(X)
        left_dir = (net.l_pos == N - 1)
        out['sw%d_left_dir' % sw] = left_dir
        out['sw%d_f_forward' % sw] = f_lb(f)
        one_hot = np.zeros((y.size, L))
        one_hot[np.arange(y.size), y] = 1
        yT = one_hot.T
        if left_dir:
            net.r_cum_contraction = []
        else:
            net.l_cum_contraction = []
        for _ in range(N - 1):
            pre = 'st%d_' % step
            vh = [[], []]
            rec.rec = {}
            with quiet():
                f = net.sweep_step(f, yT, lr, b, wd, L2_flag=L2_flag, left_dir=left_dir, var_hist=vh)
            out[pre + 'f_new'] = f_lb(f)
            out[pre + 'accuracy'] = vh[0][0]
            out[pre + 'MAE'] = vh[1][0]
            out[pre + 'S'] = rec.rec['S']
            if L2_flag:
                out[pre + 'L2_loss'] = rec.rec['L2_loss']
            out[pre + 'absB_new'] = np.abs(rec.rec['B_new']).sum()
            out[pre + 'bond'] = np.array([core_from_named(A.elem, A.axes_names, i, N).shape[2]
                                          for i, A in enumerate(net.As)][:-1])
            step += 1
    out['n_steps'] = step
    out['final_l_pos'] = net.l_pos
    with quiet():
        out['final_f'] = f_lb(net.forward(X))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, 'steps', step)


def forward_fixture(name, N, M, L, b, seed):
    """forward at l_pos = 0 and (after a fixed-bond right sweep) at l_pos = N-1."""
    rng = np.random.default_rng(seed)
    X = psi(rng.random((b, N)) * (rng.random((b, N)) > 0.5))
    X2 = psi(rng.random((b, N)))
    y = rng.integers(0, L, b)
    net = make_net(FixedBondNetwork, N, M, L, X, 'softmax', 'full_cross_ent', seed)
    out = dict(N=N, M=M, L=L, D=2, b=b, X=X, X2=X2, numpy_version=np.__version__)
    for i, c in enumerate(canon_cores(net)):
        out['a_core%d' % i] = c
    with quiet():
        f = net.forward(X)
    out['a_f'] = f_lb(f)
    dump_envs(net, out, 'a_')
    with quiet():
        net.sweep(X, y, f, 1e-3, 1e-3, L2_flag=False)
    assert net.l_pos == N - 1
    for i, c in enumerate(canon_cores(net)):
        out['b_core%d' % i] = c
    with quiet():
        f = net.forward(X2)
    out['b_f'] = f_lb(f)
    dump_envs(net, out, 'b_')
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name)


def contract_fixture():
    """Known answers of custom_linalg_tools.contract on the operand patterns of the hot path
    plus the notebook smoke case (1,2,3,4)x(3,4,5,6), contracted='k', common='l'."""
    from custom_linalg_tools import contract, partial_trace
    rng = np.random.default_rng(7)
    out = {}

    def case(tag, e1, n1, e2, n2, **kw):
        T1, T2 = Tensor(elem=e1.copy(), axes_names=n1), Tensor(elem=e2.copy(), axes_names=n2)
        T3 = contract(T1, T2, **kw)
        out[tag + '_e1'], out[tag + '_e2'] = e1, e2
        out[tag + '_n1'], out[tag + '_n2'] = np.array(n1), np.array(n2)
        out[tag + '_out'] = T3.elem
        out[tag + '_names'] = np.array([str(a) for a in T3.axes_names])
        out[tag + '_n1_after'] = np.array([str(a) for a in T1.axes_names])
        out[tag + '_n2_after'] = np.array([str(a) for a in T2.axes_names])
        out[tag + '_kw'] = np.array(repr(kw))

    case('nb', rng.random((1, 2, 3, 4)), ['i', 'j', 'k', 'l'], rng.random((3, 4, 5, 6)), ['k', 'l', 'n', 'm'],
         contracted='k', common='l')
    case('atx', rng.random((3, 4, 2)), ['left', 'right', 'd5'], rng.random((7, 2)), ['b', 'd5'], contracted='d5')
    case('env', rng.random((3, 4, 7)), ['left', 'right', 'b'], rng.random((4, 7)), ['left', 'b'],
         contracted_axis1='right', contracted_axis2='left', common='b')
    case('merge', rng.random((2, 4, 2, 3)), ['d1', 'right', 'l', 'left'], rng.random((4, 5, 2)), ['left', 'right', 'd2'],
         contracted_axis1='right', contracted_axis2='left')
    case('outer', rng.random((7, 2)), ['b', 'd1'], rng.random((7, 2)), ['b', 'd2'], common='b')
    case('grad', rng.random((2, 7)), ['l', 'b'], rng.random((2, 2, 3, 4, 7)), ['d1', 'd2', 'left', 'right', 'b'],
         contracted='b')
    case('multi', rng.random((3, 3, 4, 4)), ['left', 'L_2', 'right', 'R_2'], rng.random((4, 4, 5, 5)),
         ['left', 'L_2', 'right', 'R_2'], contracted_axis1=[2, 3], contracted_axis2=[0, 1])
    e = rng.random((3, 4, 3, 2))
    T = Tensor(elem=e.copy(), axes_names=['a', 'b', 'c', 'd'])
    out['pt_e'] = e
    out['pt_out'] = partial_trace(T, 'a', 'c').elem
    # aggregate / disaggregate known answers (Tensor_class.py:97-199)
    e = rng.random((2, 3, 4, 5))
    T = Tensor(elem=e.copy(), axes_names=['p', 'q', 'r', 's'])
    T.aggregate(axes_names=['r', 'p'], new_ax_name='i')
    out['agg_e'] = e
    out['agg_out'] = T.elem.copy()
    out['agg_names'] = np.array([str(a) for a in T.axes_names])
    T.disaggregate('i')
    out['dis_out'] = T.elem.copy()
    out['dis_names'] = np.array([str(a) for a in T.axes_names])
    np.savez_compressed(os.path.join(HERE, 'contract_known_answers.npz'), **out)
    print('wrote contract_known_answers')


class _Whitelist(pickle.Unpickler):
    OK = {('Network_class', 'Network'), ('Tensor_class', 'Tensor'),
          ('numpy.core.multiarray', '_reconstruct'), ('numpy', 'ndarray'), ('numpy', 'dtype'),
          ('numpy._core.multiarray', '_reconstruct')}

    def find_class(self, module, name):
        if (module, name) not in self.OK:
            raise pickle.UnpicklingError('forbidden global %s.%s' % (module, name))
        return super().find_class(module, name)


def shipped_model_fixture():
    """trained_diag_model.dat -> cores + forward output on a seeded diagonal dataset."""
    with open(os.path.join(REF, 'trained_diag_model.dat'), 'rb') as fh:
        net = _Whitelist(fh).load()
    rng = np.random.default_rng(0)
    n, ld, sigma = 256, int(np.sqrt(net.N)), 0.6
    one = np.eye(ld)
    zero = one[::-1, :]
    labels = rng.integers(0, 2, n)
    data = np.where((labels == 0)[:, None, None], zero, one) * (1 - sigma) + rng.random((n, ld, ld)) * sigma
    X = psi(data.reshape(n, -1))
    out = dict(N=net.N, M=net.M, L=net.L, D=net.D, T=net.T, l_pos=net.l_pos, act_fn=net.act_fn,
               loss_fn=net.loss_fn, X=X, y=labels)
    for i, c in enumerate(canon_cores(net)):
        out['core%d' % i] = c
    with quiet():
        f = net.forward(X)
    out['f'] = f_lb(f)
    out['accuracy'] = net.accuracy(X, labels, f)
    out['act'] = f_lb(net.apply_act_func(f))
    np.savez_compressed(os.path.join(HERE, 'shipped_diag_model.npz'), **out)
    print('wrote shipped_diag_model, accuracy', out['accuracy'])


def main_light():
    light_trajectory('ltraj_fixed_M20', FixedBondNetwork, N=12, M=20, L=2, b=64, act_fn='softmax',
                     loss_fn='full_cross_ent', lr=1e-3, wd=1e-3, L2_flag=True, n_sweeps=2, seed=106, policy='fixed',
                     x_zero_frac=0.5)
    light_trajectory('ltraj_fixed_M10', FixedBondNetwork, N=14, M=10, L=2, b=40, act_fn='softmax',
                     loss_fn='full_cross_ent', lr=1e-3, wd=1e-3, L2_flag=True, n_sweeps=2, seed=108, policy='fixed',
                     x_zero_frac=0.5)
    light_trajectory('ltraj_fixed_M50_L10', FixedBondNetwork, N=14, M=50, L=10, b=40, act_fn='softmax',
                     loss_fn='full_cross_ent', lr=1e-3, wd=1e-3, L2_flag=True, n_sweeps=2, seed=107, policy='fixed')


COMBOS = [('softmax', 'full_cross_ent'), ('linear', 'MSE'), ('sigmoid', 'MSE'), ('softmax', 'MSE'),
          ('softmax', 'cross_entropy'), ('sigmoid', 'cross_entropy'), ('linear', 'cross_entropy'),
          ('sigmoid', 'full_cross_ent'), ('linear', 'full_cross_ent')]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'light':
        return main_light()
    contract_fixture()
    forward_fixture('forward_N6_M2', 6, 2, 2, 7, 1)
    forward_fixture('forward_N16_M4', 16, 4, 2, 32, 2)
    forward_fixture('forward_N64_M10', 64, 10, 2, 7, 3)
    seed = 10
    for policy, cls in (('reference', ref_tn.Network), ('fixed', FixedBondNetwork)):
        for L2_flag in (True, False):
            for act_fn, loss_fn in (COMBOS if L2_flag else COMBOS[:3]):
                seed += 1
                trajectory('traj_%s_%s_%s_L2%d' % (policy, act_fn, loss_fn, int(L2_flag)), cls,
                           N=8, M=4, L=2, b=12, act_fn=act_fn, loss_fn=loss_fn, lr=0.05, wd=0.01,
                           L2_flag=L2_flag, n_sweeps=2, seed=seed, policy=policy)
    # script-like hyper-parameters, more sites, sparse (MNIST-like) inputs, several sweeps
    trajectory('traj_reference_N16_script', ref_tn.Network, N=16, M=6, L=2, b=32, act_fn='softmax',
               loss_fn='full_cross_ent', lr=1e-3, wd=1e-3, L2_flag=True, n_sweeps=4, seed=101,
               policy='reference', x_zero_frac=0.8)
    trajectory('traj_fixed_N16_script', FixedBondNetwork, N=16, M=6, L=2, b=32, act_fn='softmax',
               loss_fn='full_cross_ent', lr=1e-3, wd=1e-3, L2_flag=True, n_sweeps=4, seed=102,
               policy='fixed', x_zero_frac=0.8)
    trajectory('traj_fixed_N12_biglr', FixedBondNetwork, N=12, M=5, L=2, b=20, act_fn='sigmoid',
               loss_fn='MSE', lr=0.5, wd=0.1, L2_flag=True, n_sweeps=2, seed=103,
               policy='fixed')
    # label counts the unmodified reference cannot run (it raises at Network_class.py:914)
    trajectory('traj_fixed_L3', FixedBondNetwork, N=8, M=4, L=3, b=12, act_fn='softmax',
               loss_fn='full_cross_ent', lr=0.05, wd=0.01, L2_flag=True, n_sweeps=2, seed=104,
               policy='fixed')
    trajectory('traj_fixed_L10', FixedBondNetwork, N=8, M=6, L=10, b=24, act_fn='softmax',
               loss_fn='full_cross_ent', lr=0.05, wd=0.01, L2_flag=True, n_sweeps=2, seed=105,
               policy='fixed')
    main_light()     # the bench's bond dimensions on short chains (small outputs only)
    shipped_model_fixture()


if __name__ == '__main__':
    main()
