"""Multi-GPU plumbing for the batch-sharded sweep.

One process per GPU.  Every rank holds the full MPS and a contiguous shard of the minibatch with
its own environment stacks; per sweep step the ranks exchange exactly one message: an RCCL
all-reduce (sum) of the batch-summed gradient tensor plus four metric slots, issued by libtnml_hip.so
itself on the context's stream.  Every rank then runs the identical update + SVD on identical
inputs, so the cores stay bit-identical without a broadcast.

No PyTorch anywhere: the rendezvous (handing the 128-byte RCCL unique id from rank 0 to the others, host-side barriers,
max-over-ranks of a timing) is a small file store on the node (FileGroup); the data path links RCCL directly.

Since round 2 the message of a step is the PRE-gradient Z (wide_pipe_device.h): the sum over the batch taken before the
extension of the behind environment with the core the previous SVD produced (each rank forms it beside that SVD, inside
the same launch); the all-reduce sits between two step launches and every rank then forms the gradient proper,
dB = A^T . Z, alike
    [ Z ((h' D) * D*D*g*L floats) | correct count | sum |y - act(f)| | non-finite count | sample count ]
"""
import os

import numpy as np

METRIC_SLOTS = 4


def env_rank():
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n samples for `rank`; the first n % world ranks get one extra."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(X, y, rank, world):
    lo, hi = shard_bounds(len(X), rank, world)
    return X[lo:hi], (None if y is None else y[lo:hi])


class FileGroup:
    """Host-side rendezvous of the ranks of ONE node without torch: a directory under /tmp keyed by the launcher's
    MASTER_PORT and the launcher's PID (torch.distributed.run is the parent of every rank, so the key is common to the
    job and unique to it).  Ranks exchange small values as files written atomically (write + rename) and poll for
    each other's; every collective carries a sequence number, so nothing is ever read twice.  Used for the 128-byte
    RCCL unique id, for barriers around timed regions and for max-over-ranks of a timing: never on the data path."""

    def __init__(self, rank, world, key=None, root='/tmp', timeout=300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            key = '%s_%d' % (os.environ.get('MASTER_PORT', '29500'), os.getppid())
        self.dir = os.path.join(root, 'tnml_rdzv_%s' % key)
        os.makedirs(self.dir, exist_ok=True)
        self.seq = 0

    def _put(self, name, payload):
        tmp = os.path.join(self.dir, '.%s.%d.tmp' % (name, self.rank))
        with open(tmp, 'wb') as fh:
            fh.write(payload)
        os.replace(tmp, os.path.join(self.dir, name))

    def _get(self, name):
        import time
        path = os.path.join(self.dir, name)
        t_end = time.monotonic() + self.timeout
        while not os.path.exists(path):
            if time.monotonic() > t_end:
                raise TimeoutError('rank %d: %s never appeared (a peer died?)' % (self.rank, path))
            time.sleep(0.0002)
        with open(path, 'rb') as fh:
            return fh.read()

    def all_gather_bytes(self, payload):
        self.seq += 1
        self._put('s%d_r%d' % (self.seq, self.rank), payload)
        return [self._get('s%d_r%d' % (self.seq, r)) for r in range(self.world)]

    def barrier(self):
        self.all_gather_bytes(b'1')

    def max_float(self, v):
        import struct
        return max(struct.unpack('d', x)[0] for x in self.all_gather_bytes(struct.pack('d', float(v))))

    def broadcast_bytes(self, payload, src=0):
        return self.all_gather_bytes(payload if self.rank == src else b'')[src]

    def destroy_process_group(self):
        import shutil
        self.barrier()
        # rank 0 removes the directory only after every rank has left the barrier (said so with a file it never reads back)
        self._put('bye_r%d' % self.rank, b'1')
        if self.rank == 0:
            for r in range(self.world):
                self._get('bye_r%d' % r)
            shutil.rmtree(self.dir, ignore_errors=True)


_group = None


def init_process_group(rank, world, key=None):
    """Rendezvous of the ranks of this node (no torch.distributed: a file store, see FileGroup)."""
    global _group
    if _group is None or _group.world != world or _group.rank != rank:
        _group = FileGroup(rank, world, key)
    return _group


def broadcast_unique_id(make_uid, rank):
    """rank 0 calls make_uid() (-> 128 bytes) and every rank receives it."""
    assert _group is not None, "init_process_group first"
    uid = _group.broadcast_bytes(make_uid() if rank == 0 else b'')
    assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
    return bytes(uid)


def attach_comm(ctx, rank, world):
    """Create the RCCL communicator of a `_hip.Context` (no-op for world == 1)."""
    from . import _hip
    if world == 1:
        if os.environ.get('TNML_FORCE_COMM') == '1':          # test hook: 1-rank RCCL communicator
            ctx.comm_init(0, 1, _hip.comm_unique_id())
        return
    uid = broadcast_unique_id(_hip.comm_unique_id, rank)
    ctx.comm_init(rank, world, uid)


def pack_payload(dB_raw, correct, sum_abs, nonfinite, count):
    """The per-rank message of one sweep step, as a flat float32 array (layout above)."""
    return np.concatenate([np.asarray(dB_raw, np.float32).ravel(),
                           np.array([correct, sum_abs, nonfinite, count], np.float32)])


def unpack_payload(buf, L):
    """-> (dB_raw_flat, accuracy, MAE, any_nonfinite) from the summed message."""
    buf = np.asarray(buf)
    tail = buf[-METRIC_SLOTS:]
    cnt = float(tail[3])
    return buf[:-METRIC_SLOTS], float(tail[0]) / cnt, float(tail[1]) / (cnt * L), bool(tail[2] != 0)
