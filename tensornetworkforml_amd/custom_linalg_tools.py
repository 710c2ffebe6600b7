"""Name-based pairwise tensor contraction with the API of the reference's
`custom_linalg_tools` (/root/reference/TensorNetwork/custom_linalg_tools.py), written from scratch.

Host-side utility only.  The reference evaluates a contraction by broadcasting the two operands
against each other and summing the last axes (custom_linalg_tools.py:81-84), which materialises
|unique1| x |unique2| x |common| x |contracted| elements; here the same result comes from one batched
matrix product, `out[u1, u2, c] = sum_k A[u1, c, k] B[u2, c, k]`.

Kept semantics (drivers and notebooks written against the reference may rely on them):
  * result axes = unique axes of T1, unique axes of T2, then the common axes (custom_linalg_tools.py:77);
  * both operands are permuted IN PLACE to (unique, common, contracted) order (:74-75);
  * several axes can be contracted at once by passing lists of indices; none at all gives the
    outer product over the non-common axes.
"""
import numpy as np

from Tensor_class import Tensor


def _as_list(v):
    return list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v]


def _contract_(T1, T2, contracted_axis1, contracted_axis2, common_axis1=[], common_axis2=[]):
    """Index-based contraction (custom_linalg_tools.py:10-87): all axes are positions."""
    assert len(common_axis1) == len(common_axis2), "number of common axes is different"
    if type(contracted_axis1) != list:
        assert T1.shape[contracted_axis1] == T2.shape[contracted_axis2], "dimensions of contracted axes do not match"
    k1, k2 = [int(a) for a in _as_list(contracted_axis1)], [int(a) for a in _as_list(contracted_axis2)]
    c1, c2 = [int(a) for a in common_axis1], [int(a) for a in common_axis2]
    for a, b in zip(c1, c2):
        assert T1.shape[a] == T2.shape[b], "dimensions of common axes do not match"

    def split(T, common, contracted):
        tail = common + contracted
        unique = [i for i in range(T.rank) if i not in tail]
        return unique, unique + tail

    u1, perm1 = split(T1, c1, k1)
    u2, perm2 = split(T2, c2, k2)
    # the in-place permutation of the operands is part of the reference's observable behaviour
    T1.transpose(T1.axes_names[perm1])
    T2.transpose(T2.axes_names[perm2])
    nu1, nu2, nc, nk = len(u1), len(u2), len(c1), len(k1)
    s1, s2 = T1.elem.shape, T2.elem.shape
    U1 = int(np.prod(s1[:nu1], dtype=np.int64))
    U2 = int(np.prod(s2[:nu2], dtype=np.int64))
    Cc = int(np.prod(s1[nu1:nu1 + nc], dtype=np.int64))
    K = int(np.prod(s1[nu1 + nc:], dtype=np.int64))
    A = np.ascontiguousarray(T1.elem).reshape(U1, Cc, K)
    B = np.ascontiguousarray(T2.elem).reshape(U2, Cc, K)
    # out[c, u1, u2] = A[:, c, :] @ B[:, c, :].T  -> (u1, u2, c)
    out = np.matmul(np.transpose(A, (1, 0, 2)), np.transpose(B, (1, 2, 0)))
    out = np.transpose(out, (1, 2, 0)).reshape(tuple(s1[:nu1]) + tuple(s2[:nu2]) + tuple(s1[nu1:nu1 + nc]))
    names = np.concatenate([T1.axes_names[:nu1], T2.axes_names[:nu2 + nc]])
    return Tensor(elem=out, axes_names=names)


def contract(T1, T2, contracted_axis1=[], contracted_axis2=[], common_axis1=[], common_axis2=[],
             contracted=None, common=None):
    """Name-based front end (custom_linalg_tools.py:90-161).  `contracted` / `common` are shortcuts
    for axes that carry the same name in both operands."""
    if contracted is not None:
        contracted_axis1 = contracted_axis2 = contracted
    if common is not None:
        common_axis1 = common_axis2 = common
    if type(common_axis1) != list:
        common_axis1 = [common_axis1]
    if type(common_axis2) != list:
        common_axis2 = [common_axis2]
    if type(contracted_axis1) == str:
        contracted_axis1 = T1.ax_to_index(contracted_axis1)
    if type(contracted_axis2) == str:
        contracted_axis2 = T2.ax_to_index(contracted_axis2)
    common_axis1 = [T1.ax_to_index(a) if type(a) == str else a for a in common_axis1]
    common_axis2 = [T2.ax_to_index(a) if type(a) == str else a for a in common_axis2]
    return _contract_(T1, T2, contracted_axis1, contracted_axis2, common_axis1, common_axis2)


def partial_trace(T, ax1, ax2):
    """Trace over two named axes of one tensor (custom_linalg_tools.py:164-189).  T is permuted in
    place so that the traced axes lead."""
    traced = [ax1, ax2]
    keep = [n for n in T.axes_names if n not in traced]
    T.transpose(np.array(traced + keep))
    return Tensor(elem=np.trace(T.elem, axis1=0, axis2=1), axes_names=np.array(keep))
