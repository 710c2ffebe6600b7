"""CPU oracle for the MPS two-site sweep hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the algorithm of the reference
(francescovidaich964/TensorNetworkForML, `TensorNetwork/Network_class.py`), written on
plain ndarrays in fixed canonical layouts instead of the reference's named-axis
`Tensor` objects.  It is the checker the HIP path is compared against; it is never the
product.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product path (`tensornetworkforml_amd`) never does.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the unmodified reference
in the build container and dumps inputs/outputs of `forward`, `sweep_step` and whole
sweeps; `tests/test_oracle_golden.py` checks every function below against those vectors
(fp64, rtol 1e-9 on gauge-invariant quantities).

Canonical layouts (shared with the device library):
  bond[i]          dimension of the bond between site i and i+1      (i = 0..N-2)
  cores[i]         (ml, D, mr)      ml = bond[i-1] (1 at i=0), mr = bond[i] (1 at i=N-1)
  cores[l_pos]     (ml, D, mr, L)   the label axis is last on the site that carries it
  X                (b, N, D)
  Lenv[i]          (b, mr_i)   contraction of sites 0..i   with the input
  Renv[i]          (b, ml_i)   contraction of sites i..N-1 with the input
  B                (ml, D, D, mr, L)   merged two-site tensor, axes (a, d, d', c, l)
  f, y_onehot, g   (L, b)

Reference call sites restated (file:line in /root/reference/TensorNetwork):
  forward            Network_class.py:195-258
  sweep              Network_class.py:384-436
  sweep_step         Network_class.py:440-573
  update_B           Network_class.py:577-763
  apply_act_func     Network_class.py:767-796
  compute_loss_derivate  Network_class.py:800-835
  tensor_svd         Network_class.py:839-962
  compute_L2_reg     Network_class.py:966-1179
"""
import numpy as np

ACTS = ('linear', 'sigmoid', 'softmax')
LOSSES = ('MSE', 'cross_entropy', 'full_cross_ent')
TRUNCS = ('reference', 'fixed', 'adaptive')


class MPSState:
    """Cores + bond bookkeeping of a linear MPS with one label-carrying site."""

    def __init__(self, N, D, L, M, cores, l_pos=0, dtype=np.float64):
        self.N, self.D, self.L, self.M = int(N), int(D), int(L), int(M)
        self.dtype = dtype
        self.l_pos = int(l_pos)
        self.cores = [np.ascontiguousarray(c, dtype=dtype) for c in cores]
        assert len(self.cores) == N
        self.bond = [self.cores[i].shape[2] for i in range(N - 1)]
        for i in range(N):
            c = self.cores[i]
            assert c.ndim == (4 if i == self.l_pos else 3), (i, c.shape)
            assert c.shape[0] == self.ml(i) and c.shape[1] == D and c.shape[2] == self.mr(i)
        # per-batch caches (filled by forward / grown by sweep_step)
        self.X = None
        self.Lenv = {}
        self.Renv = {}
        # batch-independent norm environments for the L2 term
        self.Ln = {}
        self.Rn = {}

    def ml(self, i):
        return 1 if i == 0 else self.bond[i - 1]

    def mr(self, i):
        return 1 if i == self.N - 1 else self.bond[i]

    def copy(self):
        s = MPSState(self.N, self.D, self.L, self.M, [c.copy() for c in self.cores],
                     self.l_pos, self.dtype)
        s.X = self.X
        s.Lenv = dict(self.Lenv)
        s.Renv = dict(self.Renv)
        return s


def random_cores(N, M, D, L, rng=None, scale=1.0, dtype=np.float64):
    """U[0,1)/scale cores in the canonical layout, label on site 0.

    Draw order and shapes follow Network.__init__ (Network_class.py:145-148,186-189):
    site 0 is drawn as (l,right,d), interior sites as (left,right,d), the last as (left,d),
    so that with the legacy global RNG (`rng=None` -> np.random.random) the same seed gives
    the same numbers as the reference.
    """
    draw = np.random.random if rng is None else rng.random
    cores = []
    a0 = draw((L, M, D)) / scale                      # (l, right, d)
    cores.append(np.transpose(a0, (2, 1, 0))[None].astype(dtype))  # (1, D, M, L)
    for _ in range(1, N - 1):
        a = draw((M, M, D)) / scale                   # (left, right, d)
        cores.append(np.transpose(a, (0, 2, 1)).astype(dtype))      # (ml, D, mr)
    aN = draw((M, D)) / scale                         # (left, d)
    cores.append(aN[:, :, None].astype(dtype))        # (ml, D, 1)
    return cores


def site_matrix(core, x):
    """T[b, a, c, (l)] = sum_d core[a, d, c, (l)] * x[b, d]   (A_TX of Network_class.py:227)."""
    return np.tensordot(x, core, axes=([1], [1]))


def forward(state, X):
    """Build the environment stack for the current label position and return f (L, b).

    Network_class.py:195-258.  l_pos == 0  -> all right environments Renv[1..N-1];
    l_pos == N-1 -> all left environments Lenv[0..N-2].  Anything else raises, as the
    reference does (:258).
    """
    s = state
    X = np.asarray(X, dtype=s.dtype)
    assert X.shape[1] == s.N, "The 1 dimension of the input data must be the flattened number of pixels"
    b = X.shape[0]
    s.X = X
    s.Lenv, s.Renv = {}, {}
    if s.l_pos == 0:
        env = site_matrix(s.cores[s.N - 1], X[:, s.N - 1])[:, :, 0]          # (b, ml)
        s.Renv[s.N - 1] = env
        for i in range(s.N - 2, 0, -1):
            T = site_matrix(s.cores[i], X[:, i])                              # (b, ml, mr)
            env = np.einsum('bac,bc->ba', T, env)
            s.Renv[i] = env
        T = site_matrix(s.cores[0], X[:, 0])[:, 0]                            # (b, mr, L)
        f = np.einsum('bcl,bc->lb', T, env)
        return f
    elif s.l_pos == s.N - 1:
        env = site_matrix(s.cores[0], X[:, 0])[:, 0]                          # (b, mr)
        s.Lenv[0] = env
        for i in range(1, s.N - 1):
            T = site_matrix(s.cores[i], X[:, i])
            env = np.einsum('ba,bac->bc', env, T)
            s.Lenv[i] = env
        T = site_matrix(s.cores[s.N - 1], X[:, s.N - 1])[:, :, 0]             # (b, ml, L)
        f = np.einsum('ba,bal->lb', env, T)
        return f
    raise Exception('forward should not be called if l has an intermediate position')


def apply_act_func(f, act_fn, T):
    """Network_class.py:767-796.  The softmax is the reference's formula
    exp(f/T)/sum exp(f/T); the per-sample max is subtracted first, which is the same
    function but does not overflow (the reference overflows for f/T > 709)."""
    if act_fn == 'linear':
        return f.copy()
    if act_fn == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-f / T))
    if act_fn == 'softmax':
        z = f / T
        z = z - z.max(axis=0, keepdims=True)
        e = np.exp(z)
        return e / e.sum(axis=0, keepdims=True)
    raise AssertionError(act_fn)


def compute_loss_derivate(fa, y1h, act_fn, loss_fn, T):
    """Network_class.py:800-835 (fa = activated output, y1h = one-hot (L, b))."""
    if loss_fn == 'MSE':
        return y1h - fa
    if loss_fn == 'cross_entropy':
        if act_fn == 'softmax':
            return (y1h - y1h * fa) / T
        return y1h / fa
    if loss_fn == 'full_cross_ent':
        z = fa - (y1h == 0)
        return 1.0 / (z + 1e-4)
    raise AssertionError(loss_fn)


def one_hot(y, L, dtype=np.float64):
    """Network_class.py:421-423."""
    y = np.asarray(y)
    oh = np.zeros((L, y.size), dtype=dtype)
    oh[y, np.arange(y.size)] = 1
    return oh


def norm_env_left(state, i):
    """Ln_i[a, a'] over sites 0..i  (compute_L2_reg, Network_class.py:1004-1029); Ln_{-1} = [[1]].
    Cached; sweep_step drops the entries a new core invalidates."""
    s = state
    if i < 0:
        return np.ones((1, 1), dtype=s.dtype)
    j = i
    while j >= 0 and j not in s.Ln:
        j -= 1
    env = np.ones((1, 1), dtype=s.dtype) if j < 0 else s.Ln[j]
    for k in range(j + 1, i + 1):
        c = s.cores[k]
        assert c.ndim == 3, "norm environment crosses the label site"
        env = np.einsum('adc,ae,edf->cf', c, env, c)
        s.Ln[k] = env
    return env


def norm_env_right(state, i):
    """Rn_i[c, c'] over sites i..N-1 (Network_class.py:1035-1061); Rn_N = [[1]]."""
    s = state
    if i > s.N - 1:
        return np.ones((1, 1), dtype=s.dtype)
    j = i
    while j <= s.N - 1 and j not in s.Rn:
        j += 1
    env = np.ones((1, 1), dtype=s.dtype) if j > s.N - 1 else s.Rn[j]
    for k in range(j - 1, i - 1, -1):
        c = s.cores[k]
        assert c.ndim == 3, "norm environment crosses the label site"
        env = np.einsum('adc,cf,edf->ae', c, env, c)
        s.Rn[k] = env
    return env


def compute_L2_reg(state, B, p, weight_dec):
    """(wd * <B, G>, 2 wd G) with G = Ln . B . Rn   (Network_class.py:966-1179).
    B acts on sites (p, p+1); Ln spans sites 0..p-1, Rn spans p+2..N-1, both built from
    the cores as they are NOW (left ones already updated in this sweep)."""
    Ln = norm_env_left(state, p - 1)
    Rn = norm_env_right(state, p + 2)
    G = np.einsum('ae,axycl,cf->exyfl', Ln, B, Rn)
    loss = weight_dec * float(np.sum(B * G))
    return loss, 2.0 * weight_dec * G


def trunc_rank(policy, left_dir, p, N, ml, D, mr, L, M):
    """Bond dimension kept by tensor_svd (Network_class.py:894-910, 931-945) and whether the
    reference's un-truncated factor has a compatible shape.  Returns (m, ok)."""
    if not left_dir:
        rows, cols = D * ml, D * mr * L
    else:
        rows, cols = D * ml * L, D * mr
    nS = min(rows, cols)
    if policy in ('fixed', 'adaptive'):        # adaptive: this is the cap, adaptive_rank() decides below it
        return min(M, nS), True
    first = (p == 0)
    last = (p == N - 2)
    if not left_dir:
        if first:                       # l_pos == 0: only Vh cut, U is rows x rows
            return nS, rows == nS
        if not last:                    # interior: m = left bond of the merged tensor
            return ml, ml <= nS
        return nS, cols == nS           # l_pos == N-2: only U cut, Vh is cols x cols
    else:
        if last:                        # l_pos == N-1: only U cut
            return nS, cols == nS
        if not first:                   # interior
            return ml, ml <= nS
        return nS, rows == nS           # l_pos == 1: only Vh cut


def matricize(B, left_dir):
    """(a,d,d',c,l) -> 2-D for the SVD (aggregate calls at Network_class.py:528-556).
    right sweep: rows (d, a), cols (d', c, l);  left sweep: rows (d, a, l), cols (d', c)."""
    ml, D, _, mr, L = B.shape
    if not left_dir:
        return np.transpose(B, (1, 0, 2, 3, 4)).reshape(D * ml, D * mr * L)
    return np.transpose(B, (1, 0, 4, 2, 3)).reshape(D * ml * L, D * mr)


def adaptive_rank(S, cap, threshold=0.999):
    """NOT reference behaviour: the reference computes `index = argmax(cumsum(S)/S.sum() > threshold)`
    (Network_class.py:889-891) and never uses it.  Policy 'adaptive' keeps min(cap, index + 1)."""
    cum = np.cumsum(S) / S.sum()
    return int(min(cap, int(np.argmax(cum > threshold)) + 1))


def tensor_svd(Bmat, m):
    """Full SVD, keep m, split sqrt(S) on both factors (Network_class.py:887, 912-915)."""
    U, S, Vh = np.linalg.svd(Bmat, full_matrices=False)
    sq = np.sqrt(S[:m])
    return U[:, :m] * sq[None, :], sq[:, None] * Vh[:m, :], S


def sweep_step(state, f_prev, y1h, lr, weight_dec, L2_flag=True, left_dir=False,
               act_fn='linear', loss_fn='cross_entropy', T=0.1, trunc='reference',
               record=None, threshold=0.999):
    """One two-site optimisation step (Network_class.py:440-573 incl. update_B :577-763).

    Returns f_new (L, b): the output recomputed from the updated, UN-truncated B (:494-523).
    `record`, if a dict, receives the intermediate quantities the parity tests compare.
    """
    s = state
    assert trunc in TRUNCS
    N, D, L = s.N, s.D, s.L
    l = s.l_pos
    p = l - 1 if left_dir else l          # B acts on sites (p, p+1)
    if left_dir:
        if not (1 <= l <= N - 1):
            raise Exception('position not allowed for left sweep step')
    else:
        if not (0 <= l <= N - 2):
            raise Exception('position not allowed for right sweep step')
    X = s.X
    b = X.shape[0]
    ml, mr = s.ml(p), s.mr(p + 1)

    # merged tensor (:484)
    if not left_dir:
        B = np.einsum('adkl,kec->adecl', s.cores[p], s.cores[p + 1])
    else:
        B = np.einsum('adk,kecl->adecl', s.cores[p], s.cores[p + 1])

    # environments either side of B; the one on the trailing side is grown here (:637-652, :669-684)
    if not left_dir:
        if p >= 1:
            T1 = site_matrix(s.cores[p - 1], X[:, p - 1])                 # (b, ml', ml)
            s.Lenv[p - 1] = T1[:, 0] if p == 1 else np.einsum('ba,bac->bc', s.Lenv[p - 2], T1)
        E = s.Lenv[p - 1] if p >= 1 else np.ones((b, 1), dtype=s.dtype)
        R = s.Renv[p + 2] if p + 2 <= N - 1 else np.ones((b, 1), dtype=s.dtype)
    else:
        if p + 2 <= N - 1:
            T1 = site_matrix(s.cores[p + 2], X[:, p + 2])                 # (b, mr, mr')
            s.Renv[p + 2] = T1[:, :, 0] if p + 2 == N - 1 else np.einsum('bac,bc->ba', T1, s.Renv[p + 3])
        E = s.Lenv[p - 1] if p >= 1 else np.ones((b, 1), dtype=s.dtype)
        R = s.Renv[p + 2] if p + 2 <= N - 1 else np.ones((b, 1), dtype=s.dtype)

    # activation, metrics, loss derivative (:694-707)
    fa = apply_act_func(f_prev, act_fn, T)
    accuracy = float(np.mean(np.argmax(fa, axis=0) == np.argmax(y1h, axis=0)))
    MAE = float(np.abs(y1h - fa).mean())
    g = compute_loss_derivate(fa, y1h, act_fn, loss_fn, T)

    # bond gradient (:625-655, :710-724)
    x0, x1 = X[:, p], X[:, p + 1]
    GE = np.einsum('lb,ba,bd->ladb', g, E, x0).reshape(L * ml * D, b)
    XR = np.einsum('be,bc->bec', x1, R).reshape(b, D * mr)
    dB_raw = (GE @ XR).reshape(L, ml, D, D, mr).transpose(1, 2, 3, 4, 0)

    # weight decay (:728-734)
    if L2_flag:
        L2_loss, L2_grad = compute_L2_reg(s, B, p, weight_dec)
    else:
        L2_loss, L2_grad = None, weight_dec * B
    dB = dB_raw - L2_grad

    # clip + update (:755-761)
    B_measure = np.abs(B).sum()
    dB_measure = np.abs(dB).sum()
    dB_clipped = dB / (dB_measure / B_measure) if dB_measure > B_measure else dB
    B_new = B + lr * dB_clipped

    # output from the updated, un-truncated B (:494-523)
    W = np.einsum('adecl,bd,be->bacl', B_new, x0, x1)
    f_new = np.einsum('ba,bacl,bc->lb', E, W, R)

    # SVD split (:528-563, :839-962)
    m, ok = trunc_rank(trunc, left_dir, p, N, ml, D, mr, L, s.M)
    if not ok:
        raise ValueError("shapes not aligned: the reference's un-truncated SVD factor does not fit "
                         "(Network_class.py:914 / :949)")
    Bmat = matricize(B_new, left_dir)
    if trunc == 'adaptive':
        m = adaptive_rank(np.linalg.svd(Bmat, compute_uv=False), m, threshold)
    US, SVh, S = tensor_svd(Bmat, m)
    if not left_dir:
        s.cores[p] = np.ascontiguousarray(US.reshape(D, ml, m).transpose(1, 0, 2))
        s.cores[p + 1] = np.ascontiguousarray(SVh.reshape(m, D, mr, L))
        s.l_pos = l + 1
    else:
        s.cores[p] = np.ascontiguousarray(US.reshape(D, ml, L, m).transpose(1, 0, 3, 2))
        s.cores[p + 1] = np.ascontiguousarray(SVh.reshape(m, D, mr))
        s.l_pos = l - 1
    s.bond[p] = m
    # norm environments that contained the two rewritten cores are stale now
    for k in [k for k in s.Ln if k >= p]:
        del s.Ln[k]
    for k in [k for k in s.Rn if k <= p + 1]:
        del s.Rn[k]

    if record is not None:
        record.update(dict(p=p, B=B, E=E, R=R, g=g, fa=fa, accuracy=accuracy, MAE=MAE,
                           dB_raw=dB_raw, L2_loss=L2_loss, L2_grad=L2_grad, dB=dB,
                           B_measure=B_measure, dB_measure=dB_measure, B_new=B_new,
                           f_new=f_new, S=S, m=m,
                           trunc_product=(US @ SVh), Bmat=Bmat))
    return f_new


def sweep(state, X, y, f, lr, weight_dec, L2_flag=True, left_dir=False, var_hist=None,
          **kw):
    """N-1 sweep steps in one direction (Network_class.py:384-436).  `forward(state, X)`
    must have been called on the same X (the reference's train loop does, :327)."""
    s = state
    y1h = one_hot(y, s.L, s.dtype)
    if left_dir:
        s.Renv = {}
    else:
        s.Lenv = {}
    for _ in range(s.N - 1):
        rec = {} if var_hist is not None else None
        f = sweep_step(s, f, y1h, lr, weight_dec, L2_flag=L2_flag, left_dir=left_dir,
                       record=rec, **kw)
        if var_hist is not None:
            var_hist[0].append(rec['accuracy'])
            var_hist[1].append(rec['MAE'])
    return f


def accuracy(f, y):
    """Network_class.py:354-380."""
    return float(np.mean(np.argmax(f, axis=0) == np.asarray(y)))


def calibrate(state, X):
    """Network.__init__ calibration (Network_class.py:168-176): divide every core by
    max|f|^(1/N).  Returns the factor."""
    f = forward(state, X)
    F2 = float(np.abs(f).max()) ** (1.0 / state.N)
    state.cores = [c / F2 for c in state.cores]
    state.Ln, state.Rn = {}, {}
    return F2


# ---------------------------------------------------------------------------------------
# conversion between the canonical core layout and the reference's named-axis cores
# (used by the golden generator and by the host API layer's tests)
# ---------------------------------------------------------------------------------------
def core_from_named(elem, axes_names, site, N):
    """Reference Tensor (elem, axes_names) of site `site` -> canonical (ml, D, mr[, L])."""
    names = [str(a) for a in axes_names]
    elem = np.asarray(elem)
    order = []
    if 'left' in names:
        order.append(names.index('left'))
    order.append(names.index('d' + str(site)))
    if 'right' in names:
        order.append(names.index('right'))
    if 'l' in names:
        order.append(names.index('l'))
    c = np.transpose(elem, order)
    if 'left' not in names:
        c = c[None]
    if 'right' not in names:
        c = np.expand_dims(c, 2)
    return np.ascontiguousarray(c)
