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
    MASTER_PORT, its run id and its PID (torch.distributed.run is the parent of every rank).  Ranks exchange small values as
    files written atomically (write + rename) and poll for each other's.  Used for the 128-byte RCCL unique id, for barriers
    around timed regions and for max-over-ranks of a timing: never on the data path.

    The key alone is not trusted (a crashed earlier job, PID reuse, the same parent spawning ranks twice): the group opens
    with a handshake that gives it a GENERATION.  Every rank draws a nonce and keeps publishing `hello_<rank>` (nonce, host
    name); rank 0 first wipes and recreates the directory (mode 0700, owned by this user), collects one hello per rank and
    publishes `gen` = its own nonce plus the nonces it saw; a rank accepts a `gen` file only if it lists the rank's own
    nonce, so files of any other job -- older or concurrent -- are never read: every later file name carries the generation
    and a sequence number.  Ranks on different hosts cannot meet in /tmp; the handshake then times out with that hint, and
    ranks that do meet but report different host names (a shared /tmp) are refused."""

    def __init__(self, rank, world, key=None, root='/tmp', timeout=300.0):
        import json, secrets, socket, time
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            key = '%s_%s_%d' % (os.environ.get('MASTER_PORT', '29500'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.getppid())
        key = ''.join(ch if (ch.isalnum() or ch in '_-.') else '_' for ch in str(key))
        self.dir = os.path.join(root, 'tnml_rdzv_%s_u%d' % (key, os.getuid()))
        self.seq = 0
        self.gen = None
        nonce = secrets.token_hex(8)
        host = socket.gethostname()
        t_end = time.monotonic() + self.timeout
        if self.rank == 0:
            self._fresh_dir()
            seen = {0: (nonce, host)}
            while len(seen) < self.world:
                for r in range(1, self.world):
                    if r not in seen:
                        v = self._try_read('hello_%d' % r)
                        if v:
                            try:
                                d = json.loads(v.decode())
                                seen[r] = (d['nonce'], d['host'])
                            except (ValueError, KeyError):
                                pass
                if time.monotonic() > t_end:
                    raise TimeoutError('rendezvous: rank 0 saw %d of %d ranks in %s (one process per GPU on ONE node is '
                                       'required: ranks on other hosts cannot meet in /tmp)' % (len(seen), self.world, self.dir))
                time.sleep(0.001)
            hosts = sorted({h for _, h in seen.values()})
            err = '' if len(hosts) == 1 else 'ranks on different hosts share %s: %s' % (self.dir, hosts)
            self.gen = nonce
            self._put('gen', json.dumps({'gen': nonce, 'nonces': {str(r): n for r, (n, _) in seen.items()}, 'error': err}).encode())
            if err:
                raise RuntimeError('rendezvous: ' + err)
        else:
            last = 0.0
            while self.gen is None:
                now = time.monotonic()
                if now - last > 0.05:            # rank 0 may wipe the directory after this rank's first hello: keep saying it
                    try:
                        os.makedirs(self.dir, mode=0o700, exist_ok=True)
                        self._put('hello_%d' % self.rank, json.dumps({'nonce': nonce, 'host': host}).encode())
                    except OSError:
                        pass                      # the directory is being wiped under us: try again
                    last = now
                v = self._try_read('gen')
                if v:
                    try:
                        d = json.loads(v.decode())
                        if d['nonces'].get(str(self.rank)) == nonce:      # this job's generation, not a stale one
                            if d.get('error'):
                                raise RuntimeError('rendezvous: ' + d['error'])
                            self.gen = d['gen']
                    except (ValueError, KeyError):
                        pass
                if self.gen is None:
                    if now > t_end:
                        raise TimeoutError('rendezvous: rank %d never saw its generation in %s (rank 0 dead, or on another host?)'
                                           % (self.rank, self.dir))
                    time.sleep(0.001)

    def _fresh_dir(self):
        import shutil, stat
        if os.path.lexists(self.dir):
            st = os.lstat(self.dir)
            if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
                raise RuntimeError('rendezvous: %s exists and is not a directory of this user' % self.dir)
            shutil.rmtree(self.dir, ignore_errors=True)
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        st = os.lstat(self.dir)
        if st.st_uid != os.getuid():
            raise RuntimeError('rendezvous: %s is not owned by this user' % self.dir)
        os.chmod(self.dir, 0o700)

    def _try_read(self, name):
        try:
            with open(os.path.join(self.dir, name), 'rb') as fh:
                return fh.read()
        except OSError:
            return None

    def _put(self, name, payload):
        tmp = os.path.join(self.dir, '.%s.%d.tmp' % (name, self.rank))
        with open(tmp, 'wb') as fh:
            fh.write(payload)
        os.replace(tmp, os.path.join(self.dir, name))

    def _get(self, name):
        import time
        t_end = time.monotonic() + self.timeout
        while True:
            v = self._try_read(name)
            if v is not None:
                return v
            if time.monotonic() > t_end:
                raise TimeoutError('rank %d: %s never appeared in %s (a peer died?)' % (self.rank, name, self.dir))
            time.sleep(0.0002)

    def all_gather_bytes(self, payload):
        self.seq += 1
        self._put('%s_s%d_r%d' % (self.gen, self.seq, self.rank), payload)
        return [self._get('%s_s%d_r%d' % (self.gen, self.seq, r)) for r in range(self.world)]

    def barrier(self):
        self.all_gather_bytes(b'1')

    def max_float(self, v):
        import struct
        return max(struct.unpack('d', x)[0] for x in self.all_gather_bytes(struct.pack('d', float(v))))

    def broadcast_bytes(self, payload, src=0):
        return self.all_gather_bytes(payload if self.rank == src else b'')[src]

    def destroy_process_group(self):
        import shutil
        global _group
        self.barrier()
        # rank 0 removes the directory only after every rank has left the barrier (said so with a file it never reads back)
        self._put('%s_bye_r%d' % (self.gen, self.rank), b'1')
        if self.rank == 0:
            for r in range(self.world):
                self._get('%s_bye_r%d' % (self.gen, r))
            shutil.rmtree(self.dir, ignore_errors=True)
        if _group is self:
            _group = None                 # a later init_process_group opens a new generation


_group = None


def init_process_group(rank, world, *, key=None, timeout=300.0):
    """Rendezvous of the ranks of this node (no torch.distributed: a file store, see FileGroup).  `key` is keyword-only: the
    third positional argument of torch.distributed.init_process_group is a backend name, which must not become a directory."""
    global _group
    if _group is None or _group.world != world or _group.rank != rank:
        _group = FileGroup(rank, world, key, timeout=timeout)
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
