"""Loading helpers for the fixtures written by tests/golden/make_golden.py."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: d[k] for k in d.files}


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def unflatten_cores(flat, bond, l_pos, D, L):
    """Inverse of the generator's `cores_flat` packing: canonical (ml, D, mr[, L]) cores."""
    N = len(bond) + 1
    cores, off = [], 0
    for i in range(N):
        ml = 1 if i == 0 else int(bond[i - 1])
        mr = 1 if i == N - 1 else int(bond[i])
        shape = (ml, D, mr, L) if i == l_pos else (ml, D, mr)
        n = int(np.prod(shape))
        cores.append(flat[off:off + n].reshape(shape).copy())
        off += n
    assert off == flat.size
    return cores


def indexed(d, prefix, n):
    return [d['%s%d' % (prefix, i)] for i in range(n)]


def step_envs(d, pre):
    Lenv, Renv = {}, {}
    for k in d:
        if k.startswith(pre + 'Lenv'):
            Lenv[int(k[len(pre) + 4:])] = d[k]
        elif k.startswith(pre + 'Renv'):
            Renv[int(k[len(pre) + 4:])] = d[k]
    return Lenv, Renv


def gauge_signs(B_dev, B_ref):
    """Singular vectors are defined up to a sign, so the two outer bond indices (a, c) of a merged
    tensor (a, d, d', c, l) carry independent +-1 factors between two correct implementations.
    Returns (s_a, t_c) with B_dev ~ s_a t_c B_ref, read off the data."""
    Mx = np.einsum('adecl,adecl->ac', B_dev, B_ref)
    a0 = int(np.argmax(np.abs(Mx).sum(axis=1)))
    t = np.where(Mx[a0] < 0, -1.0, 1.0)
    s = np.where((Mx * t[None, :]).sum(axis=1) < 0, -1.0, 1.0)
    return s, t


def regauge(T, s, t):
    return T * s[:, None, None, None, None] * t[None, None, None, :, None]
