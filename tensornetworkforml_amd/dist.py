"""Multi-GPU plumbing for the batch-sharded sweep.

One process per GPU.  Every rank holds the full MPS and a contiguous shard of the minibatch with
its own environment stacks; per sweep step the ranks exchange exactly one message: an RCCL
all-reduce (sum) of the raw bond gradient plus four metric slots
    [ dB_raw (h*D*D*g*L floats) | correct count | sum |y - act(f)| | non-finite count | sample count ]
issued by libtnml_hip.so itself on the context's stream between the slab reduction and the
single-workgroup update/SVD kernel.  Every rank then runs the identical update + SVD on identical
inputs, so the cores stay bit-identical without a broadcast.

torch.distributed is used for the rendezvous only (handing the 128-byte RCCL unique id from rank 0
to the others, host-side barriers): the gloo backend, no GPU tensors.
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


def init_process_group(rank, world, backend='gloo'):
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def broadcast_unique_id(make_uid, rank):
    """rank 0 calls make_uid() (-> 128 bytes) and every rank receives it."""
    import torch.distributed as dist
    box = [make_uid() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = box[0]
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
