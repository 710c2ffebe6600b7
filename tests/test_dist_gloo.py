"""World-size-2 tests of the batch-sharded path on CPU (gloo): the shard helper, the file-store rendezvous (unique-id
hand-off, barrier, max over ranks), and -- with the oracle standing in for the per-rank device work -- that the one
message a step exchanges (sum over the shards of the PRE-gradient Z, the batch sum taken before the behind environment is
extended with the previous step's new core, + metric slots) reproduces the single-rank step exactly once every rank has
contracted it with that core (dB = A^T . Z), so that replicated update + SVD keeps the ranks identical."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from tensornetworkforml_amd import dist as tdist
    from oracle import mps_oracle as mo
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)      # the test's own all-reduce below
    # 1. the library's rendezvous: a file store, no torch (a fake 128-byte id: RCCL itself needs GPUs)
    grp = tdist.init_process_group(rank, world)
    uid = tdist.broadcast_unique_id(lambda: bytes(range(128)), rank)
    assert uid == bytes(range(128))
    grp.barrier()
    assert grp.max_float(1.5 + rank) == 1.5 + world - 1
    assert grp.broadcast_bytes(b'from-one' if rank == 1 else b'', src=1) == b'from-one'
    # 2. one sweep step, sharded
    rng = np.random.default_rng(0)                     # same stream on every rank
    N, M, D, L, b = 10, 4, 2, 3, 37                    # odd batch: ragged shards
    p = rng.random((b, N))
    X = np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1)
    y = rng.integers(0, L, b)
    cores = mo.random_cores(N, M, D, L, rng=rng, scale=M * 0.64)
    hp = dict(lr=0.05, weight_dec=0.01, L2_flag=True, act_fn='softmax', loss_fn='full_cross_ent', T=0.1, trunc='fixed')
    full = mo.MPSState(N, D, L, M, [c.copy() for c in cores])
    f_full = mo.forward(full, X)
    Xs, ys = tdist.shard_batch(X, y, rank, world)
    mine = mo.MPSState(N, D, L, M, [c.copy() for c in cores])
    f_mine = mo.forward(mine, Xs)
    lo, hi = tdist.shard_bounds(b, rank, world)
    np.testing.assert_allclose(f_mine, f_full[:, lo:hi], rtol=1e-12)
    for step in range(3):
        rec_full, rec = {}, {}
        f_full = mo.sweep_step(full, f_full, mo.one_hot(y, L), record=rec_full, **hp)
        probe = mo.MPSState(N, D, L, M, [c.copy() for c in mine.cores], mine.l_pos)
        probe.X, probe.Lenv, probe.Renv = mine.X, dict(mine.Lenv), dict(mine.Renv)
        mo.sweep_step(probe, f_mine, mo.one_hot(ys, L), record=rec, **hp)     # local quantities only
        fa = rec['fa']
        correct = float((np.argmax(fa, 0) == ys).sum())
        sum_abs = float(np.abs(mo.one_hot(ys, L) - fa).sum())
        # pre-gradient of this shard: the batch sum BEFORE the extension with the previous step's new core A_{p-1}
        #   Z[(a', x), d, e, c, l] = sum_s g[l,s] E_{p-2}[s,a'] x_{p-1}[s,x] x_p[s,d] x_{p+1}[s,e] R[s,c]      (p >= 1)
        p_ = rec['p']
        if p_ >= 1:
            Eprev = mine.Lenv[p_ - 2] if p_ >= 2 else np.ones((len(ys), 1))
            Z = np.einsum('lb,ba,bx,bd,be,bc->axdecl', rec['g'], Eprev, mine.X[:, p_ - 1], mine.X[:, p_], mine.X[:, p_ + 1], rec['R'])
        else:
            Z = rec['dB_raw']
        msg = torch.from_numpy(tdist.pack_payload(Z, correct, sum_abs, 0, len(ys)).astype(np.float64))
        dist.all_reduce(msg)                                                   # THE exchange of the step
        Zsum, acc, mae, bad = tdist.unpack_payload(msg.numpy(), L)
        if p_ >= 1:                                                            # every rank alike: dB = A_{p-1}^T . Z
            dB = np.einsum('axh,axdecl->hdecl', mine.cores[p_ - 1], Zsum.reshape(Z.shape))
        else:
            dB = Zsum.reshape(Z.shape)
        np.testing.assert_allclose(dB.reshape(rec_full['dB_raw'].shape), rec_full['dB_raw'], rtol=1e-5, atol=1e-6 * np.abs(rec_full['dB_raw']).max())
        assert abs(acc - rec_full['accuracy']) < 1e-9 and abs(mae - rec_full['MAE']) < 1e-6 and not bad
        # replicated update from the summed gradient: every rank must land on the full-batch cores
        f_mine = _replicated_step(mo, mine, f_mine, ys, dB.reshape(rec_full['dB_raw'].shape).astype(np.float64), hp)
        for a, c in zip(mine.cores, full.cores):
            np.testing.assert_allclose(np.abs(a), np.abs(c), rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(f_mine, f_full[:, lo:hi], rtol=1e-4, atol=1e-6)
    dist.barrier()
    open(os.path.join(out_dir, 'ok%d' % rank), 'w').write('ok')
    grp.destroy_process_group()
    # a second group of the same processes (same key): a new generation, nothing of the first one is read again
    grp2 = tdist.init_process_group(rank, world)
    assert grp2 is not grp and grp2.gen != grp.gen
    assert grp2.broadcast_bytes(b'again' if rank == 0 else b'') == b'again'
    grp2.destroy_process_group()
    dist.destroy_process_group()


def _replicated_step(mo, st, f_prev, ys, dB_raw_global, hp):
    """What every rank does after the all-reduce: the oracle's step with the local gradient replaced
    by the global one (the device's narrow kernel reads the reduced buffer in the same way)."""
    # the oracle's step with the gradient overridden: its tail is re-implemented here
    N, D, L = st.N, st.D, st.L
    p = st.l_pos
    rec = {}
    snapshot = mo.MPSState(N, D, L, st.M, [c.copy() for c in st.cores], st.l_pos)
    snapshot.X, snapshot.Lenv, snapshot.Renv = st.X, dict(st.Lenv), dict(st.Renv)
    mo.sweep_step(snapshot, f_prev, mo.one_hot(ys, L), record=rec, **hp)
    st.Lenv, st.Renv = snapshot.Lenv, snapshot.Renv
    B = rec['B']
    _, L2g = mo.compute_L2_reg(st, B, p, hp['weight_dec'])
    dB = dB_raw_global - L2g
    Bm, Dm = np.abs(B).sum(), np.abs(dB).sum()
    if Dm > Bm:
        dB = dB / (Dm / Bm)
    B_new = B + hp['lr'] * dB
    E, R = rec['E'], rec['R']
    x0, x1 = st.X[:, p], st.X[:, p + 1]
    W = np.einsum('adecl,bd,be->bacl', B_new, x0, x1)
    f_new = np.einsum('ba,bacl,bc->lb', E, W, R)
    ml, mr = B.shape[0], B.shape[3]
    m, ok = mo.trunc_rank(hp['trunc'], False, p, N, ml, D, mr, L, st.M)
    US, SVh, S = mo.tensor_svd(mo.matricize(B_new, False), m)
    st.cores[p] = np.ascontiguousarray(US.reshape(D, ml, m).transpose(1, 0, 2))
    st.cores[p + 1] = np.ascontiguousarray(SVh.reshape(m, D, mr, L))
    st.l_pos = p + 1
    st.bond[p] = m
    st.Ln = {k: v for k, v in st.Ln.items() if k < p}
    st.Rn = {k: v for k, v in st.Rn.items() if k > p + 1}
    return f_new


def test_shard_bounds_cover_the_batch():
    from tensornetworkforml_amd import dist as tdist
    for n in (1, 7, 64, 5000, 20000):
        for world in (1, 2, 3, 8):
            spans = [tdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_step(tmp_path):
    world, port = 2, _free_port()
    # leftovers of a crashed earlier job under the very key this job will use (same port, same parent pid): a stale
    # generation file, a stale hello and a stale unique id must not be read by anybody
    stale = '/tmp/tnml_rdzv_%d_none_%d_u%d' % (port, os.getpid(), os.getuid())
    os.makedirs(stale, mode=0o700, exist_ok=True)
    for name, payload in (('gen', b'{"gen": "dead", "nonces": {"0": "x", "1": "y"}, "error": ""}'), ('hello_1', b'{"nonce": "y", "host": "elsewhere"}'),
                          ('dead_s1_r0', bytes(128)), ('s1_r0', bytes(128))):
        with open(os.path.join(stale, name), 'wb') as fh:
            fh.write(payload)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ('ok%d' % r)) for r in range(world))
    assert not os.path.exists(stale)          # the last group removed its directory


def test_rendezvous_refuses_foreign_directory(tmp_path):
    from tensornetworkforml_amd import dist as tdist
    # a FILE where the directory should be: rank 0 must not follow or delete it
    key = 'refuse_%d' % os.getpid()
    path = '/tmp/tnml_rdzv_%s_u%d' % (key, os.getuid())
    with open(path, 'w') as fh:
        fh.write('not a directory')
    try:
        with pytest.raises(RuntimeError):
            tdist.FileGroup(0, 1, key)
    finally:
        os.remove(path)
    # world size 1 needs no peer: the handshake completes alone and the directory goes away again
    g = tdist.FileGroup(0, 1, 'solo_%d' % os.getpid())
    assert g.broadcast_bytes(b'x') == b'x'
    g.destroy_process_group()
    assert not os.path.exists(g.dir)
