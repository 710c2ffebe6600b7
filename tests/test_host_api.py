"""CPU tests of the host-side layer: named-axis Tensor, contract (against known answers produced by
the reference, tests/golden/contract_known_answers.npz), the data module, the bond bookkeeping
helper of the C ABI and the C-ABI library itself (loads, exports every declared symbol, fails loudly
without a GPU)."""
import os
import pickle
import re

import numpy as np
import pytest

import golden_util as gu
import tensornetworkforml_amd as pkg
from tensornetworkforml_amd import _hip
from Tensor_class import Tensor
from custom_linalg_tools import contract, partial_trace
import data_generator as gen
from oracle import mps_oracle as mo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_module_aliases():
    import Network_class
    import Tensor_class
    assert pkg.Network is Network_class.Network
    assert pkg.Tensor_class is Tensor_class


@pytest.mark.parametrize('tag', ['nb', 'atx', 'env', 'merge', 'outer', 'grad', 'multi'])
def test_contract_known_answers(tag):
    d = gu.load('contract_known_answers')
    kw = eval(str(d[tag + '_kw']))          # the keyword arguments the generator used (repr of a dict)
    T1 = Tensor(elem=d[tag + '_e1'].copy(), axes_names=[str(a) for a in d[tag + '_n1']])
    T2 = Tensor(elem=d[tag + '_e2'].copy(), axes_names=[str(a) for a in d[tag + '_n2']])
    T3 = contract(T1, T2, **kw)
    assert [str(a) for a in T3.axes_names] == [str(a) for a in d[tag + '_names']]
    np.testing.assert_allclose(T3.elem, d[tag + '_out'], rtol=1e-12, atol=1e-14)
    # the operands are permuted in place exactly as the reference leaves them
    assert [str(a) for a in T1.axes_names] == [str(a) for a in d[tag + '_n1_after']]
    assert [str(a) for a in T2.axes_names] == [str(a) for a in d[tag + '_n2_after']]


def test_notebook_contract_case():
    # old_files/tn_develpment.ipynb:413-423: (1,2,3,4)x(3,4,5,6), contracted='k', common='l'
    T1 = Tensor(elem=np.random.rand(1, 2, 3, 4), axes_names=['i', 'j', 'k', 'l'])
    T2 = Tensor(elem=np.random.rand(3, 4, 5, 6), axes_names=['k', 'l', 'n', 'm'])
    T3 = contract(T1, T2, contracted='k', common='l')
    assert list(T3.axes_names) == ['i', 'j', 'n', 'm', 'l'] and T3.shape == (1, 2, 5, 6, 4)


def test_partial_trace_and_aggregate():
    d = gu.load('contract_known_answers')
    T = Tensor(elem=d['pt_e'].copy(), axes_names=['a', 'b', 'c', 'd'])
    np.testing.assert_allclose(partial_trace(T, 'a', 'c').elem, d['pt_out'], rtol=1e-13)
    T = Tensor(elem=d['agg_e'].copy(), axes_names=['p', 'q', 'r', 's'])
    T.aggregate(axes_names=['r', 'p'], new_ax_name='i')
    assert [str(a) for a in T.axes_names] == [str(a) for a in d['agg_names']]
    np.testing.assert_array_equal(T.elem, d['agg_out'])
    assert dict((k, int(v)) for k, v in T.aggregations['i'].items()) == {'r': 4, 'p': 2}
    T.disaggregate('i')
    assert [str(a) for a in T.axes_names] == [str(a) for a in d['dis_names']]
    np.testing.assert_array_equal(T.elem, d['dis_out'])
    assert T.aggregations == {}


def test_tensor_errors_and_arithmetic():
    with pytest.raises(Exception):
        Tensor()
    T = Tensor(shape=(2, 3), axes_names=['a', 'b'], scale=2.)
    assert T.shape == (2, 3) and T.rank == 2 and T.elem.max() <= 0.5
    with pytest.raises(ValueError):
        T.aggregate(axes_names=['a'])                      # new_ax_name missing
    with pytest.raises(AssertionError):
        T.aggregate(axes_names=['zz'], new_ax_name='i')
    U = Tensor(elem=T.elem.T.copy(), axes_names=['b', 'a'])
    S = T + U
    np.testing.assert_allclose(S.elem, 2 * T.elem)
    assert list(U.axes_names) == ['a', 'b']                # `o` is permuted in place
    np.testing.assert_allclose((T - U).elem, 0 * T.elem)
    bad = Tensor(elem=np.zeros((2, 2)), axes_names=['a'])  # wrong number of names -> warning, None
    assert bad.axes_names is None


def test_tensor_pickles_like_the_reference():
    T = Tensor(elem=np.arange(6.).reshape(2, 3), axes_names=['left', 'd3'])
    T2 = pickle.loads(pickle.dumps(T))
    assert set(vars(T2)) >= {'elem', 'shape', 'rank', 'aggregations', 'history_axes_names', 'axes_names'}


def test_create_dataset_matches_reference_stream():
    # same draws as data_generator.py:41-50: choice(labels) then rand(noise)
    np.random.seed(3)
    data, labels = gen.create_dataset(50, 8, 0.7)
    np.random.seed(3)
    lab = np.random.choice([0, 1], size=50, p=[0.5, 0.5])
    noise = np.random.rand(50, 8, 8) * 0.7
    one = np.eye(8)
    ref = np.where((lab == 0)[:, None, None], one[::-1], one) * 0.3 + noise
    np.testing.assert_array_equal(labels, lab)
    np.testing.assert_allclose(data, ref)


def test_loader_protocol():
    data, labels = gen.create_dataset(103, 4, 0.5)
    tr, va, te = gen.prepare_dataset(data, labels, 1, 0.2, 20, 8, 128)
    assert len(tr) == int(103 * 0.8) // 20 and len(va) == (103 - int(103 * 0.8)) // 8
    seen = []
    for batch in tr:
        assert isinstance(batch, list) and len(batch) == 20
        x0, y0 = batch[0]
        assert x0.shape == (16, 2) and batch.X.shape == (20, 16, 2)
        np.testing.assert_array_equal(batch.X[0], x0)
        seen.extend(id(b) for b in batch)
    # psi = [sin, cos] on the last axis
    np.testing.assert_allclose(gen.psi(np.array([[0., 1.]])), [[[0., 1.], [1., 6.123e-17]]], atol=1e-12)
    assert len(te) == 0 or True


def test_mnist_reader_needs_local_files(tmp_path):
    with pytest.raises(FileNotFoundError):
        gen.get_MNIST_dataset(str(tmp_path))
    # a two-image IDX pair round-trips
    import struct
    for nm, arr in (('train-images-idx3-ubyte', np.arange(2 * 28 * 28, dtype=np.uint8).reshape(2, 28, 28)),
                    ('t10k-images-idx3-ubyte', np.zeros((1, 28, 28), np.uint8))):
        with open(tmp_path / nm, 'wb') as fh:
            fh.write(struct.pack('>HBB', 0, 8, 3) + struct.pack('>III', *arr.shape) + arr.tobytes())
    for nm, arr in (('train-labels-idx1-ubyte', np.array([3, 1], np.uint8)), ('t10k-labels-idx1-ubyte', np.array([7], np.uint8))):
        with open(tmp_path / nm, 'wb') as fh:
            fh.write(struct.pack('>HBB', 0, 8, 1) + struct.pack('>I', arr.size) + arr.tobytes())
    a, b, c, d = gen.get_MNIST_dataset(str(tmp_path))
    assert a.shape == (2, 28, 28) and list(b) == [3, 1] and c.shape == (1, 28, 28) and list(d) == [7]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'tnml.h')).read()
    declared = set(re.findall(r'\b(tnml_[A-Za-z0-9_]+)\s*\(', header))
    declared -= {'tnml_status'}
    lib = _hip.lib()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_hip.SYMBOLS) == declared, set(_hip.SYMBOLS) ^ declared
    assert lib.tnml_version().startswith(b'tnml-hip')


def test_no_cpu_fallback():
    if _hip.device_count() > 0:
        pytest.skip('GPU present')
    with pytest.raises(_hip.TnmlError) as ei:
        _hip.Context(8, 2, 2, 4, 16)
    assert ei.value.code == -4
    np.random.seed(0)
    net = pkg.Network(N=6, M=3, L=2)           # construction is host-only ...
    with pytest.raises(_hip.TnmlError):
        net.forward(np.random.rand(4, 6, 2))   # ... the first compute call needs the device


@pytest.mark.parametrize('policy', ['reference', 'fixed'])
def test_trunc_rank_matches_oracle(policy):
    rng = np.random.default_rng(0)
    for _ in range(300):
        N = int(rng.integers(3, 12)); D = 2; L = int(rng.integers(1, 5)); M = int(rng.integers(1, 9))
        left = bool(rng.integers(0, 2)); p = int(rng.integers(0, N - 1))
        ml = 1 if p == 0 else int(rng.integers(1, 9)); mr = 1 if p == N - 2 else int(rng.integers(1, 9))
        m_o, ok = mo.trunc_rank(policy, left, p, N, ml, D, mr, L, M)
        m_c = _hip.trunc_rank(policy, left, p, N, ml, D, mr, L, M)
        assert (m_c == m_o) if ok else (m_c == -6), (policy, left, p, N, ml, mr, L, M, m_o, ok, m_c)


def test_random_init_matches_reference_draw_order():
    from Network_class import random_canonical_cores
    np.random.seed(11)
    mine = random_canonical_cores(5, 3, 2, 2, scale=1.7)
    np.random.seed(11)
    ref = mo.random_cores(5, 3, 2, 2, scale=1.7)
    for a, b in zip(mine, ref):
        np.testing.assert_array_equal(a, b)


def test_host_side_under_sanitizers():
    """csrc/Makefile target `san`, both halves under -fsanitize=address,undefined:
      * the pure host arithmetic (truncation ranks, layout permutations: host_plan.inc) over a grid of shapes;
      * the WHOLE host side of the library -- tnml_api.hip and the launch wrappers of kernels_*.hip, built --cuda-host-only --
        against the stand-in HIP / RCCL runtime of csrc/san/hip_stub.cpp, which checks every copy and the argument block of
        every launch (pointers AND the extents the kernels touch) against its allocation registry: whole sweeps of C2, C3
        and C5 at their true sizes in both directions through the C ABI, persistent / per-step / classic / large-tensor
        paths, ragged and tiny chains, bond 64 (csrc/san/plan_san_main.cpp)."""
    import shutil
    import subprocess
    if shutil.which('g++') is None or not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no g++ / hipcc')
    csrc = os.path.join(ROOT, 'tensornetworkforml_amd', 'csrc')
    out = subprocess.run(['make', '-C', csrc, '-j4', 'san'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'sanitizer test ok' in out.stdout
    assert 'host planning under ASan + UBSan: ok' in out.stdout
    import re
    m = re.search(r'san-stub: (\d+) launches checked \((\d+) kernels\), (\d+) pointer extents checked, 0 live allocations', out.stdout)
    assert m and int(m.group(1)) > 50000 and int(m.group(2)) >= 25 and int(m.group(3)) > 10 ** 6, out.stdout[-2000:]
