"""`Network` -- the MPS classifier and its DMRG-style two-site sweep optimiser with the API of the
reference's `Network_class.Network` (/root/reference/TensorNetwork/Network_class.py), executed by
hand-written HIP kernels on an MI355X through the C ABI of include/tnml.h (ctypes, `_hip.py`).

Design: the cores, the batch, both environment stacks and f live in HBM for the whole
life of the object; `forward`, `sweep`, `sweep_step`, `accuracy`, `apply_act_func` and
`compute_loss_derivate` are device calls; `As`, `TX`, `r_cum_contraction` and `l_cum_contraction`
are materialised as `Tensor` objects only when somebody reads them.  There is no NumPy fallback:
without the library or without a gfx950 device every compute method raises.

Differences from the reference, all opt-in or unavoidable:
  * arithmetic is float32 on the device (float64 inside the merged-tensor update and the SVD);
  * `trunc='adaptive'` (not reference behaviour) keeps min(M, index + 1) singular values, `index` being
    the cumulative-share index the reference computes and never uses (Network_class.py:889-891,
    `threshold=0.999`);
  * `trunc='reference'` (default) reproduces the reference's truncation rule, including its
    ValueError for L > 2 (Network_class.py:914); `trunc='fixed'` keeps m = min(M, len(S)) on both
    factors -- the only policy under which the bond dimension M survives the first sweep;
  * the softmax subtracts the per-sample maximum (same function, no overflow at f/T > 709);
  * `update_B`, `tensor_svd` and `compute_L2_reg` are fused into one kernel per sweep step; the
    methods of that name run the same device code on the operands they are given.
"""
import numpy as np

from Tensor_class import Tensor
from custom_linalg_tools import contract, partial_trace  # noqa: F401  (re-exported like the reference)

try:
    from tensornetworkforml_amd import _hip
except ImportError:                                       # running from inside the package directory
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tensornetworkforml_amd import _hip

new = np.newaxis

_ACTS = ['linear', 'sigmoid', 'softmax']
_LOSSES = ['MSE', 'cross_entropy', 'full_cross_ent']


def random_canonical_cores(N, M, D, L, scale=1.0, rng=None):
    """U[0,1)/scale cores in the canonical device layout (ml, D, mr[, L]), label on site 0.

    The draws follow Network.__init__ (Network_class.py:145-148, 186-189): (L,M,D) for site 0,
    (M,M,D) for the interior, (M,D) for the last site, in site order, from the legacy global
    generator when `rng` is None -- so `np.random.seed(s); Network(...)` starts from the same
    numbers as the reference."""
    draw = np.random.random if rng is None else rng.random
    cores = [np.transpose(draw((L, M, D)) / scale, (2, 1, 0))[None]]          # (1, D, M, L)
    for _ in range(1, N - 1):
        cores.append(np.transpose(draw((M, M, D)) / scale, (0, 2, 1)))        # (M, D, M)
    cores.append((draw((M, D)) / scale)[:, :, None])                          # (M, D, 1)
    return [np.ascontiguousarray(c) for c in cores]


def _core_to_tensor(core, site, N, has_label):
    """canonical (ml, D, mr[, L]) -> Tensor with the reference's axis names for that site."""
    names, sl = [], []
    if site > 0:
        names.append('left'); sl.append(slice(None))
    else:
        sl.append(0)
    names.append('d' + str(site)); sl.append(slice(None))
    if site < N - 1:
        names.append('right'); sl.append(slice(None))
    else:
        sl.append(0)
    if has_label:
        names.append('l'); sl.append(slice(None))
    return Tensor(elem=np.array(core[tuple(sl)], dtype=np.float64), axes_names=names)


def _tensor_to_core(T, site, N):
    """Tensor of one site (any axis order, reference names) -> canonical (ml, D, mr[, L])."""
    names = [str(a) for a in T.axes_names]
    order = [names.index(n) for n in ('left', 'd' + str(site), 'right', 'l') if n in names]
    c = np.transpose(np.asarray(T.elem), order)
    if 'left' not in names:
        c = c[None]
    if 'right' not in names:
        c = np.expand_dims(c, 2)
    return np.ascontiguousarray(c), ('l' in names)


class _EnvList:
    """Lazy stand-in for `r_cum_contraction` / `l_cum_contraction`: a list of Tensors that is
    downloaded from the device only when it is indexed.  `entries` is a list of ('env', side, site)
    or ('f',) descriptors in the reference's list order."""

    def __init__(self, net, entries):
        self._net, self._entries, self._cache = net, list(entries), {}
        self._epoch = net._env_epoch

    def __len__(self):
        return len(self._entries)

    def _get(self, i):
        if self._epoch != self._net._env_epoch:
            raise RuntimeError("this environment list belongs to an earlier forward/sweep; read "
                               "net.r_cum_contraction / net.l_cum_contraction again")
        if i not in self._cache:
            e = self._entries[i]
            if e[0] == 'f':
                self._cache[i] = Tensor(elem=self._net._ctx.get_f().astype(np.float64), axes_names=['l', 'b'])
            else:
                _, side, site = e
                arr = self._net._ctx.get_env(side, site).astype(np.float64).T      # (m, b)
                self._cache[i] = Tensor(elem=arr, axes_names=['left' if side == _hip.SIDE_RIGHT else 'right', 'b'])
        return self._cache[i]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._get(i)

    def __iter__(self):
        return (self._get(i) for i in range(len(self)))


class Network():
    """Matrix Product State classifier trained by alternating two-site sweeps.

    Attributes (as in the reference, Network_class.py:14-47): N, D, L, M, T, act_fn, loss_fn,
    As, l_pos, TX, r_cum_contraction, l_cum_contraction.
    """

    def __init__(self, N, M, D=2, L=10, T=0.1, normalize=False, calibration_X=None, act_fn='linear',
                 loss_fn='cross_entropy', check=False, trunc='reference', device=0, svd_stop=None, threshold=0.999):
        self.N, self.D, self.L, self.M, self.T = N, D, L, M, T
        assert act_fn in _ACTS, "Please select an activation function between 'linear', 'sigmoid', 'softmax'"
        assert loss_fn in _LOSSES, "Please select a loss function between 'MSE', 'cross_entropy', 'full_cross_ent'"
        assert trunc in _hip.TRUNC, "trunc must be 'reference', 'fixed' or 'adaptive'"
        self.act_fn, self.loss_fn, self.trunc = act_fn, loss_fn, trunc
        self._device = device
        self._threshold = threshold        # cumulative-share threshold of trunc='adaptive'
        self._svd_stop = svd_stop          # None: the library default (include/tnml.h, tnml_set_svd_stop)
        self._init_runtime()
        self._l_pos = 0

        if normalize:
            print('Normalizing weights...')
            # output ~ [M E(A) E(x) D]^N with E(A) = 0.5, E(x) = 0.64 (Network_class.py:139-142)
            scale = float(self.M) * 0.5 * 0.64 * self.D
            print('Scaling factor: %.2f' % scale)
            self._host_cores = random_canonical_cores(N, M, D, L, scale)
            self._host_newer = True
            if calibration_X is None:
                B = 16
                u = np.random.random((B, self.N))
                X = np.transpose(np.array((np.sin(np.pi * u / 2), np.cos(np.pi * u / 2))), [1, 2, 0])
            else:
                X = calibration_X
                B = X.shape[0]
            print('\nCalibrating weights on dataset...')
            # order of magnitude of the output (Network_class.py:168-170).  max|f| of the raw chain is
            # ~1e-66 at N = 784, below float32: the device evaluates log max|f| with per-site
            # renormalisation instead of max|forward(X)|.
            ctx = self._sync_to_device(B)
            self._X_host, self._b = X, B
            ctx.set_input(X, None)
            self._y_dev = None
            log_fmax = ctx.forward_logabsmax()
            F2 = float(np.exp(log_fmax / self.N))
            if check:
                print('f_max for random input of %d samples : ' % (B), float(np.exp(log_fmax)))
            print("Rescaling factor for calibration: ", F2)
            ctx.scale_cores(1.0 / F2)
            self._device_newer = True
            f = self.forward(X)
            if check:
                print('f_max for random input of %d samples (after): ' % (B), float(np.abs(f.elem).max()))
        else:
            self._host_cores = random_canonical_cores(N, M, D, L)
            self._host_newer = True

    # ------------------------------------------------------------------------------------------
    # runtime state (not pickled)
    # ------------------------------------------------------------------------------------------
    def _init_runtime(self):
        self._ctx = None
        self._host_cores = None       # canonical cores as last seen on the host
        self._host_newer = False      # host copy must be uploaded before the next device call
        self._device_newer = False    # device copy changed since the host copy was taken
        self._As = None               # Tensor list handed to the user (may have been mutated)
        self._As_snapshot = None      # what that list looked like when it was built
        self._X_host = None
        self._b = 0
        self._y_dev = None
        self._env_epoch = 0
        self._r_entries = None
        self._l_entries = None
        self._r_user = None
        self._l_user = None

    def _context(self, b):
        if self._ctx is None:
            self._ctx = _hip.Context(self.N, self.D, self.L, self.M, max(int(b), 1), self._device)
            if getattr(self, '_svd_stop', None) is not None:
                self._ctx.set_svd_stop(self._svd_stop)
            if getattr(self, '_threshold', None) is not None:
                self._ctx.set_trunc_threshold(self._threshold)
        return self._ctx

    def _collect_user_edits(self):
        """If `net.As` was handed out, fold any edit of those Tensors back into the host cores."""
        if self._As is None:
            return
        cores, lab_site = [], None
        for i, Tn in enumerate(self._As):
            c, has_l = _tensor_to_core(Tn, i, self.N)
            if has_l:
                lab_site = i
            cores.append(c)
        changed = (self._As_snapshot is None or len(cores) != len(self._As_snapshot) or
                   any(a.shape != b.shape or not np.array_equal(a, b) for a, b in zip(cores, self._As_snapshot)))
        if changed:
            if lab_site is None:
                raise Exception("no core carries the label axis 'l'")
            self._host_cores = cores
            self._l_pos = lab_site
            self._host_newer = True
            self._As_snapshot = [c.copy() for c in cores]

    def _sync_to_device(self, b=None):
        self._collect_user_edits()
        ctx = self._context(b if b is not None else max(self._b, 1))
        if self._host_newer:
            ctx.set_cores(self._host_cores, self._l_pos)
            self._host_newer = False
            self._device_newer = False
            self._invalidate_envs()
        return ctx

    def _sync_to_host(self):
        self._collect_user_edits()
        if self._device_newer and not self._host_newer:
            cores, _, lp = self._ctx.get_cores()
            self._host_cores = [c.astype(np.float64) for c in cores]
            self._l_pos = lp
            self._device_newer = False
            self._As = None

    def _invalidate_envs(self):
        self._env_epoch += 1
        self._r_entries = self._l_entries = None
        self._r_user = self._l_user = None

    # ------------------------------------------------------------------------------------------
    # reference attributes, materialised on demand
    # ------------------------------------------------------------------------------------------
    @property
    def l_pos(self):
        if self._ctx is not None and not self._host_newer:
            return self._ctx.l_pos if self._device_newer else self._l_pos
        return self._l_pos

    @l_pos.setter
    def l_pos(self, v):
        self._sync_to_host()
        self._l_pos = int(v)
        self._host_newer = True

    @property
    def As(self):
        self._sync_to_host()
        if self._As is None:
            self._As = [_core_to_tensor(c, i, self.N, i == self._l_pos) for i, c in enumerate(self._host_cores)]
            self._As_snapshot = [np.array(c, dtype=np.float64) for c in self._host_cores]
        return self._As

    @As.setter
    def As(self, tensors):
        self._sync_to_host()
        self._As = list(tensors)
        self._As_snapshot = None

    @property
    def TX(self):
        if self._X_host is None:
            return None
        return [Tensor(elem=self._X_host[:, i, :], axes_names=['b', 'd' + str(i)]) for i in range(self.N)]

    def _env_list(self, which):
        user = self._r_user if which == 'r' else self._l_user
        if user is not None:
            return user
        entries = self._r_entries if which == 'r' else self._l_entries
        return None if entries is None else _EnvList(self, entries)

    @property
    def r_cum_contraction(self):
        return self._env_list('r')

    @r_cum_contraction.setter
    def r_cum_contraction(self, v):
        self._r_user = v

    @property
    def l_cum_contraction(self):
        return self._env_list('l')

    @l_cum_contraction.setter
    def l_cum_contraction(self, v):
        self._l_user = v

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward(self, X):
        """Predictions (pre-activation) for X (b, N, D); builds the environment stack for the next
        sweep: right environments at l_pos == 0, left ones at l_pos == N-1
        (Network_class.py:195-258).  Returns a Tensor ('l', 'b')."""
        assert self.N == X.shape[1], "The 1 dimension of the input data must be the flattened number of pixels"
        lp = self.l_pos
        if lp != 0 and lp != self.N - 1:
            raise Exception('### Error ###\n l =', lp, ' -> forward should not be called if l has an intermediate position')
        ctx = self._sync_to_device(X.shape[0])
        self._X_host = X
        self._b = X.shape[0]
        ctx.set_input(X, None)
        self._y_dev = None
        f = ctx.forward()
        self._invalidate_envs()
        S_R, S_L = _hip.SIDE_RIGHT, _hip.SIDE_LEFT
        if lp == 0:
            self._r_entries = [('f',)] + [('env', S_R, i) for i in range(1, self.N)]
            self._l_entries = None
        else:
            self._l_entries = [('env', S_L, i) for i in range(self.N - 1)] + [('f',)]
            self._r_entries = None
        return Tensor(elem=f.astype(np.float64), axes_names=['l', 'b'])

    def predict(self, X):
        """forward(X)'s return value without making X the resident batch: no environment list is
        rebuilt, `TX` and the training batch stay as they are (the validation loop of `train`,
        Network_class.py:339-346, only needs f).  Not in the reference."""
        assert self.N == X.shape[1], "The 1 dimension of the input data must be the flattened number of pixels"
        lp = self.l_pos
        if lp != 0 and lp != self.N - 1:
            raise Exception('### Error ###\n l =', lp, ' -> forward should not be called if l has an intermediate position')
        ctx = self._sync_to_device(max(self._b, 1))
        return Tensor(elem=ctx.predict(X).astype(np.float64), axes_names=['l', 'b'])

    # ------------------------------------------------------------------------------------------
    # training
    # ------------------------------------------------------------------------------------------
    def train(self, train_loader, val_loader, lr, n_epochs=10, weight_dec=0.001, L2_flag=True, debug=False):
        """One sweep per training batch, direction alternating with the label position, then a
        validation pass per epoch (Network_class.py:261-350).  Returns (val_acc, var_hist) with
        var_hist of shape (n_epochs, 2 | 7, n_batches * (N-1))."""
        val_acc, var_hist = [], []
        print("\n --- TRAINING PROCEDURE ---")
        for epoch in range(n_epochs):
            epoch_train_acc = np.zeros(len(train_loader))
            var_hist.append([[] for _ in range(7 if debug else 2)])
            for i, data in enumerate(train_loader, 0):
                x, y = _unpack_batch(data)
                f = self.forward(x)
                epoch_train_acc[i] = self.accuracy(x, y, f)
                left_dir = (self.l_pos == self.N - 1)
                f = self.sweep(x, y, f, lr, weight_dec, L2_flag=L2_flag, left_dir=left_dir,
                               var_hist=var_hist[epoch], debug=debug)
                print('\r' + "Epoch %d/%d - train accuracy : %.4f - completed : %.2f "
                      % (epoch, n_epochs, epoch_train_acc[i], (i + 1) * 100 / len(train_loader)) + '%', end=' ')
            epoch_val_acc = np.zeros(len(val_loader))
            for i, data in enumerate(val_loader, 0):
                x, y = _unpack_batch(data)
                epoch_val_acc[i] = self.accuracy(x, y, self.predict(x))      # same f as forward(x), nothing rebuilt
            val_acc.append(epoch_val_acc.mean())
            print('\r' + "Epoch %d/%d - train accuracy : %.4f - val accuracy: %.4f"
                  % (epoch, n_epochs, epoch_train_acc.mean(), val_acc[-1]))
        return val_acc, np.array(var_hist)

    def accuracy(self, X, y, f=None):
        """Fraction of samples whose argmax over labels equals y (Network_class.py:354-380)."""
        if f is None:
            f = self.forward(X)
        y_pred = np.argmax(f.elem, axis=0)
        errors = (np.asarray(y) != y_pred).sum()
        return (len(y_pred) - errors) / len(y_pred)

    def _upload_labels(self, y_int):
        y_int = np.ascontiguousarray(y_int, dtype=np.int32)
        if self._y_dev is None or not np.array_equal(self._y_dev, y_int):
            if self._X_host is None:
                raise Exception("forward(X) must run before a sweep: no batch is resident")
            assert y_int.shape[0] == self._b, "labels and resident batch differ in length"
            self._ctx.set_labels(y_int)
            self._y_dev = y_int.copy()

    def _first_of_sweep(self, left_dir):
        return self.l_pos == (self.N - 1 if left_dir else 0)

    def _run_steps(self, f, y_int, n_steps, lr, weight_dec, L2_flag, left_dir, var_hist, debug):
        ctx = self._sync_to_device()
        self._upload_labels(y_int)
        ctx.set_f(np.asarray(f.elem, dtype=np.float32))
        first = self._first_of_sweep(left_dir)
        if first:
            # the list this direction grows starts empty (Network_class.py:426-429)
            if left_dir:
                self._r_entries, self._r_user = [], None
            else:
                self._l_entries, self._l_user = [], None
        out = None
        if debug and var_hist is not None:
            ctx.debug_enable(True)
            for k in range(n_steps):
                f_before = ctx.get_f()
                met, out = ctx.sweep(left_dir, 1, first and k == 0, lr, weight_dec, L2_flag, self.act_fn,
                                     self.loss_fn, self.T, self.trunc)
                B, dB, L2g = ctx.step_debug('B'), ctx.step_debug('dB_raw'), ctx.step_debug('L2_grad')
                sc = ctx.step_debug('scalars')
                var_hist[0].append(np.abs(B).mean())
                var_hist[1].append(np.abs(dB - L2g).mean())
                var_hist[2].append(float(met[0, 0]))
                var_hist[3].append(np.abs(f_before).mean())
                var_hist[4].append(float(met[0, 1]))
                var_hist[5].append(sc[0] if L2_flag else None)
                var_hist[6].append(np.abs(L2g).mean())
            ctx.debug_enable(False)
        else:
            met, out = ctx.sweep(left_dir, n_steps, first, lr, weight_dec, L2_flag, self.act_fn, self.loss_fn,
                                 self.T, self.trunc, want_metrics=var_hist is not None)
            if var_hist is not None:
                var_hist[0].extend(float(v) for v in met[:, 0])
                var_hist[1].extend(float(v) for v in met[:, 1])
        self._device_newer = True
        self._As = None
        self._env_epoch += 1
        # environment lists as the reference leaves them: the grown one gains one entry per step
        lp = ctx.l_pos
        S_R, S_L = _hip.SIDE_RIGHT, _hip.SIDE_LEFT
        if left_dir:
            # steps at l = N-2 .. 1 append Renv[l+1]; after reaching l_pos the deepest is Renv[lp+2]
            self._r_entries = [('env', S_R, i) for i in range(self.N - 1, lp + 1, -1)]
        else:
            self._l_entries = [('env', S_L, i) for i in range(0, lp - 1)]
        return Tensor(elem=out.astype(np.float64), axes_names=['l', 'b'])

    def sweep(self, X, y, f, lr, weight_dec, L2_flag=True, left_dir=False, var_hist=None, debug=False):
        """N-1 two-site steps in one direction (Network_class.py:384-436).  `forward(X)` must have
        run on the same batch (the reference's `train` does that); X itself is not read here,
        exactly as in the reference."""
        return self._run_steps(f, np.asarray(y), self.N - 1, lr, weight_dec, L2_flag, left_dir, var_hist, debug)

    def sweep_step(self, f, y, lr, batch_size, weight_dec, L2_flag=True, left_dir=False, var_hist=None,
                   debug=False):
        """One two-site step (Network_class.py:440-573).  `y` is the one-hot target (L, b) the
        reference passes here; the returned Tensor is f recomputed from the updated, un-truncated
        merged tensor."""
        y = np.asarray(y)
        y_int = np.argmax(y, axis=0) if y.ndim == 2 else y
        l = self.l_pos
        if left_dir and not (1 <= l <= self.N - 1):
            raise Exception('### Error ###\n l =', l, ' -> position not allowed for left sweep step')
        if not left_dir and not (0 <= l <= self.N - 2):
            raise Exception('### Error ###\n l =', l, ' -> position not allowed for right sweep step')
        return self._run_steps(f, y_int, 1, lr, weight_dec, L2_flag, left_dir, var_hist, debug)

    # ------------------------------------------------------------------------------------------
    # pieces of a step the reference exposes as methods
    # ------------------------------------------------------------------------------------------
    def _on_device_f(self, f):
        ctx = self._sync_to_device(f.elem.shape[1])
        names = [str(a) for a in f.axes_names]
        arr = np.asarray(f.elem if names[0] == 'l' else f.elem.T, dtype=np.float32)
        if self._X_host is None or arr.shape[1] != self._b:
            # no resident batch of that size: park a dummy one so that f has a home on the device
            self._b = arr.shape[1]
            ctx.set_input(np.zeros((self._b, self.N, self.D), dtype=np.float32), None)
            self._X_host = None
            self._y_dev = None
            self._invalidate_envs()
        ctx.set_f(arr)
        return ctx

    def apply_act_func(self, f):
        """Activation of the network output (Network_class.py:767-796)."""
        ctx = self._on_device_f(f)
        a, _ = ctx.activation(self.act_fn, self.loss_fn, self.T, want_act=True, want_der=False)
        out = Tensor(elem=a.astype(np.float64), axes_names=['l', 'b'])
        if [str(n) for n in f.axes_names][0] != 'l':
            out.transpose(f.axes_names)
        return out

    def compute_loss_derivate(self, f, y):
        """Derivative of the loss w.r.t. the (activated) output f; y one-hot (L, b)
        (Network_class.py:800-835).  `f` is the ACTIVATED output, as in the reference."""
        fa = np.asarray(f.elem if [str(n) for n in f.axes_names][0] == 'l' else f.elem.T, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        ctx = self._on_device_f(Tensor(elem=fa, axes_names=['l', 'b']))
        ctx.set_labels(np.argmax(y, axis=0).astype(np.int32))
        self._y_dev = None
        d = ctx.loss_derivative_of_activated(self.act_fn, self.loss_fn, self.T)
        return Tensor(elem=d.astype(np.float64), axes_names=['l', 'b'])

    def tensor_svd(self, T, left_dir=False, threshold=0.999):
        """SVD split of a 2-D Tensor ('i', 'j') with sqrt(S) on both factors (Network_class.py:839-962),
        run by the device's Jacobi kernel.  The rank kept follows `self.trunc` at the current l_pos:
        'reference' keeps all singular values next to the chain ends and `aggregations['i']['left']`
        of them elsewhere (:894-945, `threshold` is computed but unused there); 'fixed' keeps
        min(M, len(S))."""
        if type(T) != Tensor:
            raise TypeError("This function only support object from the class Tensor")
        if len(T.shape) != 2:
            raise ValueError("This function only support a 2D tensors")
        rows, cols = T.elem.shape
        nS = min(rows, cols)
        lp = self.l_pos
        adaptive = self.trunc == 'adaptive'
        if self.trunc in ('fixed', 'adaptive'):
            m = min(self.M, nS)
        else:
            interior = (1 < lp < self.N - 1) if left_dir else (0 < lp < self.N - 2)
            m = int(T.aggregations['i']['left']) if interior else nS
            if m > nS:
                # np.eye(m, m) * S[:m] in the reference (:914 / :949)
                raise ValueError("operands could not be broadcast together with shapes (%d,%d) (%d,) " % (m, m, nS))
        ctx = self._sync_to_device()
        US, SVh, sig = ctx.svd_split(np.asarray(T.elem, dtype=np.float32), m)
        if adaptive:
            # the index the reference computes and never uses (:889-891), here it decides the rank
            thr = threshold if getattr(self, '_threshold', None) is None else self._threshold
            m = min(m, int(np.argmax(np.cumsum(sig) / sig.sum() > thr)) + 1)
            US, SVh = US[:, :m], SVh[:m]
        TU = Tensor(elem=US.astype(np.float64), axes_names=['i', 'right'])
        TSVh = Tensor(elem=SVh.astype(np.float64), axes_names=['left', 'j'])
        TU.aggregations['i'] = T.aggregations['i']
        TSVh.aggregations['j'] = T.aggregations['j']
        TU.disaggregate('i')
        TSVh.disaggregate('j')
        return TU, TSVh

    def _merged_to_canonical(self, B, p):
        """Named merged Tensor on sites (p, p+1) -> (array (ml, D, D, mr, L), function mapping a canonical
        array back to a Tensor with B's own axis order)."""
        names = [str(a) for a in B.axes_names]
        want = ['left', 'd%d' % p, 'd%d' % (p + 1), 'right', 'l']
        for n in names:
            if n not in want:
                raise ValueError("axis '%s' does not belong to the merged tensor of sites (%d, %d)" % (n, p, p + 1))
        present = [n for n in want if n in names]
        arr = np.transpose(np.asarray(B.elem), [names.index(n) for n in present])
        full_shape = [arr.shape[present.index(n)] if n in present else 1 for n in want]
        canon = np.ascontiguousarray(arr.reshape(full_shape), dtype=np.float32)

        def back(c):
            t = np.asarray(c, dtype=np.float64).reshape([full_shape[want.index(n)] for n in present])
            return Tensor(elem=np.transpose(t, [present.index(n) for n in names]).copy(), axes_names=list(B.axes_names))
        return canon, back

    def update_B(self, B, f_orig, y, lr, weight_dec, L2_flag=True, ldf=0, var_hist=None, debug=False):
        """Gradient step on the merged tensor B of sites (l-ldf, l+1-ldf) (Network_class.py:577-763):
        extends the environment list behind the sweep, builds dB = sum_b loss'(f_b) phi_b, subtracts
        the L2 / weight-decay term, clips by the L1 ratio and returns B + lr dB.  `forward(X)` must
        have built the environments for this batch; cores and l_pos are left as they are."""
        left_dir = bool(ldf)
        l = self.l_pos
        if left_dir and not (1 <= l <= self.N - 1):
            raise Exception('### Error ###\n l =', l, ' -> position not allowed for left sweep step')
        if not left_dir and not (0 <= l <= self.N - 2):
            raise Exception('### Error ###\n l =', l, ' -> position not allowed for right sweep step')
        p = l - int(left_dir)
        canon, back = self._merged_to_canonical(B, p)
        y = np.asarray(y)
        y_int = np.argmax(y, axis=0) if y.ndim == 2 else y
        ctx = self._sync_to_device()
        self._upload_labels(y_int)
        fnames = [str(a) for a in f_orig.axes_names]
        f32 = np.asarray(f_orig.elem if fnames[0] == 'l' else f_orig.elem.T, dtype=np.float32)
        ctx.set_f(f32)
        if self._first_of_sweep(left_dir):
            if left_dir:
                self._r_entries, self._r_user = [], None
            else:
                self._l_entries, self._l_user = [], None
        Bnew, met = ctx.update_B(canon, left_dir, lr, weight_dec, L2_flag, self.act_fn, self.loss_fn, self.T)
        if var_hist is not None:
            if debug:
                ctx.debug_enable(True)
                Bd, dB, L2g = ctx.step_debug('B'), ctx.step_debug('dB_raw'), ctx.step_debug('L2_grad')
                sc = ctx.step_debug('scalars')
                ctx.debug_enable(False)
                var_hist[0].append(np.abs(Bd).mean())
                var_hist[1].append(np.abs(dB - L2g).mean())
                var_hist[2].append(float(met[0]))
                var_hist[3].append(np.abs(f32).mean())
                var_hist[4].append(float(met[1]))
                var_hist[5].append(sc[0] if L2_flag else None)
                var_hist[6].append(np.abs(L2g).mean())
            else:
                var_hist[0].append(float(met[0]))
                var_hist[1].append(float(met[1]))
        self._env_epoch += 1
        S_R, S_L = _hip.SIDE_RIGHT, _hip.SIDE_LEFT
        if left_dir:
            self._r_entries = [('env', S_R, i) for i in range(self.N - 1, l, -1)]
        else:
            self._l_entries = [('env', S_L, i) for i in range(0, l)]
        return back(Bnew)

    def compute_L2_reg(self, B, weight_dec=0.001, left_dir=False):
        """L2 term of the loss, weight_dec * <psi|psi> with B in place of the two cores at
        (l-ldf, l+1-ldf), and its gradient w.r.t. B (Network_class.py:966-1179).  Returns
        (float, Tensor shaped like B)."""
        left_dir = bool(left_dir)
        l = self.l_pos
        p = l - int(left_dir)
        if p < 0 or p > self.N - 2:
            raise Exception('### Error ###\n l =', l, ' -> position not allowed for %s sweep step'
                            % ('left' if left_dir else 'right'))
        canon, back = self._merged_to_canonical(B, p)
        ctx = self._sync_to_device()
        loss, grad = ctx.l2_term(canon, left_dir, weight_dec)
        return loss, back(grad)

    # ------------------------------------------------------------------------------------------
    # persistence: the whole object is pickled (training_diagonals.py:69-70)
    # ------------------------------------------------------------------------------------------
    def __getstate__(self):
        As = self.As
        return dict(N=self.N, D=self.D, L=self.L, M=self.M, T=self.T, As=As, l_pos=self._l_pos,
                    act_fn=self.act_fn, loss_fn=self.loss_fn, TX=None, r_cum_contraction=None,
                    l_cum_contraction=None, trunc=self.trunc)

    def __setstate__(self, state):
        # accepts both our own pickles and the reference's (plain __dict__ with As as Tensors)
        self.N, self.D, self.L, self.M, self.T = state['N'], state['D'], state['L'], state['M'], state['T']
        self.act_fn, self.loss_fn = state['act_fn'], state['loss_fn']
        self.trunc = state.get('trunc', 'reference')
        self._device = 0
        self._init_runtime()
        self._l_pos = state['l_pos']
        cores = []
        for i, Tn in enumerate(state['As']):
            c, _ = _tensor_to_core(Tn, i, self.N)
            cores.append(np.array(c, dtype=np.float64))
        self._host_cores = cores
        self._host_newer = True


def _unpack_batch(data):
    """A loader batch is a list of (x_i (N, D), y_i) tuples (data_generator.py:190-192); loaders of
    this package attach the stacked arrays to the list to skip the Python loop."""
    X = getattr(data, 'X', None)
    if X is not None:
        return X, data.y
    x = np.array([data[i][0] for i in range(len(data))])
    y = np.array([data[i][1] for i in range(len(data))])
    return x, y
