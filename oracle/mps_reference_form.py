"""CPU baseline in the REFERENCE'S OWN COMPUTATIONAL FORM -- TEST / BENCHMARK INFRASTRUCTURE ONLY.

`oracle/mps_oracle.py` restates the reference's algorithm with einsum/BLAS and cached norm environments (the "honest
optimised CPU form", B4 of BASELINE.md section 3).  This file restates the same step the way the reference itself computes
it, so that its cost -- not only its result -- stands in for "the reference CPU path" on the GPU box, where the
reference's Python files do not travel (forms B1, B2, B3):

  * every contraction is the reference's primitive: operands permuted to (unique, common, contracted), reshaped for
    broadcasting, multiplied elementwise into the full (unique1 x unique2 x common x contracted) array and summed over the
    last axis (custom_linalg_tools.py:57-84) -- no BLAS;
  * the contractions of a step happen in the reference's order: phi = x_l (x) x_{l+1} (x) R (x) E over the batch, then
    loss_der . phi (Network_class.py:625-710); f from the updated B by four successive contractions (:494-523);
  * the L2 term recomputes both norm environments from the chain ends at EVERY step, one (M, M, M, M) site tensor and one
    two-axis contraction per site, on deep copies (:1000-1135);
  * float64 throughout; truncation rule selectable: "reference" (B1/B2) or "fixed" (B3).

Only `tests/` and `bench.py`'s cpu_baseline leg import it.  Parity status: PINNED through tests/test_oracle_golden.py::
test_reference_form_matches_goldens (same golden vectors as mps_oracle, rtol 1e-9).
"""
import copy

import numpy as np

from . import mps_oracle as mo


def bms(A, B, a_unique, b_unique, a_common=(), b_common=(), a_contr=(), b_contr=()):
    """The reference's `_contract_` (custom_linalg_tools.py:10-87) on plain arrays: result axes = A-unique, B-unique,
    common.  Axis arguments are index tuples."""
    A = np.transpose(A, tuple(a_unique) + tuple(a_common) + tuple(a_contr))
    B = np.transpose(B, tuple(b_unique) + tuple(b_common) + tuple(b_contr))
    ua, ub = len(a_unique), len(b_unique)
    sa = A.shape[:ua] + (1,) * ub + A.shape[ua:]
    sb = (1,) * ua + B.shape
    T = A.reshape(sa) * B.reshape(sb)                  # the full broadcast product is materialised (:81)
    for _ in range(len(a_contr)):
        T = T.sum(axis=-1)                             # (:82-84)
    return T


def _norm_site(c):
    """(left, right, L_2, R_2) tensor of one site: A (x) A contracted over d (Network_class.py:1015-1019)."""
    a, a2 = copy.deepcopy(c), copy.deepcopy(c)         # the reference deep-copies both operands (:1010-1011)
    return bms(a, a2, (0, 2), (0, 2), (), (), (1,), (1,))          # (left, right, L_2, R_2)


def l2_term_from_scratch(state, B, p, weight_dec):
    """compute_L2_reg (Network_class.py:966-1179): both norm environments rebuilt from the chain ends."""
    s = state
    left = None
    if p >= 1:
        c0 = s.cores[0][0]                                         # (d, right)
        left = bms(copy.deepcopy(c0), copy.deepcopy(c0), (1,), (1,), (), (), (0,), (0,))       # (right, R_2)
        for i in range(1, p):
            site = _norm_site(s.cores[i])                          # (left, right, L_2, R_2)
            left = bms(left, site, (), (1, 3), (), (), (0, 1), (0, 2))
    right = None
    if p + 2 <= s.N - 1:
        cN = s.cores[s.N - 1][:, :, 0]                             # (left, d)
        right = bms(copy.deepcopy(cN), copy.deepcopy(cN), (0,), (0,), (), (), (1,), (1,))       # (left, L_2)
        for i in range(s.N - 2, p + 1, -1):
            site = _norm_site(s.cores[i])
            right = bms(site, right, (0, 2), (), (), (), (1, 3), (0, 1))
    G = B                                                          # (a, d, e, c, l)
    if right is not None:
        G = bms(G, right, (0, 1, 2, 4), (1,), (), (), (3,), (0,))  # (a, d, e, l, c')
        G = np.transpose(G, (0, 1, 2, 4, 3))
    if left is not None:
        G = bms(left, G, (1,), (1, 2, 3, 4), (), (), (0,), (0,))   # (a', d, e, c, l)
    loss = weight_dec * float(bms(B, G, (), (), (), (), (0, 1, 2, 3, 4), (0, 1, 2, 3, 4)))
    return loss, 2.0 * weight_dec * G


def sweep_step(state, f_prev, y1h, lr, weight_dec, L2_flag=True, left_dir=False, act_fn='linear',
               loss_fn='cross_entropy', T=0.1, trunc='reference'):
    """One two-site step in the reference's computational form; state and return value as mps_oracle.sweep_step."""
    s = state
    N, D, L = s.N, s.D, s.L
    l = s.l_pos
    p = l - 1 if left_dir else l
    X = s.X
    ml, mr = s.ml(p), s.mr(p + 1)
    # merged tensor (:484)
    if not left_dir:
        B = bms(s.cores[p], s.cores[p + 1], (0, 1, 3), (1, 2), (), (), (2,), (0,))            # (a, d, l, e, c)
        B = np.transpose(B, (0, 1, 3, 4, 2))
    else:
        B = bms(s.cores[p], s.cores[p + 1], (0, 1), (1, 2, 3), (), (), (2,), (0,))            # (a, d, e, c, l)
    # environment on the trailing side grows by one site (:637-652, :669-684)
    if not left_dir and p >= 1:
        T1 = bms(s.cores[p - 1], X[:, p - 1], (0, 2), (0,), (), (), (1,), (1,))                # (a', a, b)
        s.Lenv[p - 1] = T1[0].T if p == 1 else bms(s.Lenv[p - 2], T1, (), (1,), (0,), (2,), (1,), (0,)).T
    if left_dir and p + 2 <= N - 1:
        T1 = bms(s.cores[p + 2], X[:, p + 2], (0, 2), (0,), (), (), (1,), (1,))                # (c, c', b)
        s.Renv[p + 2] = T1[:, 0].T if p + 2 == N - 1 else bms(T1, s.Renv[p + 3], (0,), (), (2,), (0,), (1,), (1,)).T
    b = X.shape[0]
    E = s.Lenv[p - 1] if p >= 1 else None
    R = s.Renv[p + 2] if p + 2 <= N - 1 else None
    # phi over the batch (:625-655)
    phi = bms(X[:, p], X[:, p + 1], (1,), (1,), (0,), (0,))                                    # (d, e, b)
    if R is not None:
        phi = bms(phi, R, (0, 1), (1,), (2,), (0,))                                            # (d, e, c, b)
    else:
        phi = phi[:, :, None, :]
    if E is not None:
        phi = bms(phi, E, (0, 1, 2), (1,), (3,), (0,))                                         # (d, e, c, a, b)
    else:
        phi = phi[:, :, :, None, :]
    fa = mo.apply_act_func(f_prev, act_fn, T)
    g = mo.compute_loss_derivate(fa, y1h, act_fn, loss_fn, T)
    dB = bms(g, phi, (0,), (0, 1, 2, 3), (), (), (1,), (4,))                                   # (l, d, e, c, a)
    dB = np.transpose(dB, (4, 1, 2, 3, 0))                                                     # (a, d, e, c, l)
    if L2_flag:
        _, L2_grad = l2_term_from_scratch(s, B, p, weight_dec)
    else:
        L2_grad = weight_dec * copy.deepcopy(B)
    dB = dB - L2_grad
    B_measure = np.abs(B).sum()
    if np.abs(dB).sum() > B_measure:
        dB = dB / (np.abs(dB).sum() / B_measure)
    B_new = B + lr * dB
    # f from the updated, un-truncated B (:494-523)
    out = bms(B_new, X[:, p], (0, 2, 3, 4), (0,), (), (), (1,), (1,))                          # (a, e, c, l, b)
    out = bms(out, X[:, p + 1], (0, 2, 3), (), (4,), (0,), (1,), (1,))                         # (a, c, l, b)
    if E is not None:
        out = bms(E, out, (), (1, 2), (0,), (3,), (1,), (0,))                                  # (c, l, b)
    else:
        out = out[0]
    if R is not None:
        out = bms(out, R, (1,), (), (2,), (0,), (0,), (1,))                                    # (l, b)
    else:
        out = out[0]
    f_new = out
    # SVD split (:839-962)
    m, ok = mo.trunc_rank(trunc, left_dir, p, N, ml, D, mr, L, s.M)
    if not ok:
        raise ValueError("shapes not aligned: the reference's un-truncated SVD factor does not fit")
    US, SVh, _ = mo.tensor_svd(mo.matricize(B_new, left_dir), m)
    if not left_dir:
        s.cores[p] = np.ascontiguousarray(US.reshape(D, ml, m).transpose(1, 0, 2))
        s.cores[p + 1] = np.ascontiguousarray(SVh.reshape(m, D, mr, L))
        s.l_pos = l + 1
    else:
        s.cores[p] = np.ascontiguousarray(US.reshape(D, ml, L, m).transpose(1, 0, 3, 2))
        s.cores[p + 1] = np.ascontiguousarray(SVh.reshape(m, D, mr))
        s.l_pos = l - 1
    s.bond[p] = m
    return f_new


def forward(state, X):
    """Network.forward in the reference's form (Network_class.py:227-255): the site matrices A_TX of ALL sites are
    materialised first, then chained by broadcast-multiply-sum."""
    s = state
    X = np.asarray(X, dtype=s.dtype)
    s.X = X
    s.Lenv, s.Renv = {}, {}
    N = s.N
    ATX = [bms(s.cores[i], X[:, i], (0, 2) + ((3,) if s.cores[i].ndim == 4 else ()), (0,), (), (), (1,), (1,)) for i in range(N)]
    if s.l_pos == 0:
        env = ATX[N - 1][:, 0].T                                     # (b, ml)
        s.Renv[N - 1] = env
        for i in range(N - 2, 0, -1):
            env = bms(ATX[i], env, (0,), (), (2,), (0,), (1,), (1,)).T
            s.Renv[i] = env
        return bms(ATX[0][0], env, (1,), (), (2,), (0,), (0,), (1,))          # (l, b)
    if s.l_pos == N - 1:
        env = ATX[0][0].T                                            # (b, mr)
        s.Lenv[0] = env
        for i in range(1, N - 1):
            env = bms(env, ATX[i], (), (1,), (0,), (2,), (1,), (0,)).T
            s.Lenv[i] = env
        return bms(env, ATX[N - 1][:, 0], (), (1,), (0,), (2,), (1,), (0,))   # (l, b)
    raise Exception('forward should not be called if l has an intermediate position')
