"""Data layer with the API of the reference's `data_generator`
(/root/reference/TensorNetwork/data_generator.py), without torch / torchvision.

`create_dataset` draws from the legacy global NumPy generator in the reference's order, so the same
seed yields the same images.  The loaders are plain Python objects that follow the protocol
`Network.train` relies on (data_generator.py:183-192 + Network_class.py:323-325): `len(loader)` is
the number of batches and iterating yields lists of `(x_i (N, D), y_i)` tuples; here each list also
carries the stacked arrays as `.X` / `.y` so that the hot path skips the per-sample Python loop.
MNIST is read from local IDX files (there is no download path in this package).
"""
import gzip
import os
import struct

import numpy as np


def create_dataset(n_samples, linear_dim=5, sigma=0.5, prob_zero=0.5):
    """Noisy images of the two diagonals of a square and their labels (data_generator.py:6-52)."""
    one = np.eye(linear_dim)
    zero = one[::-1, :]
    labels = np.random.choice([0, 1], size=n_samples, p=[prob_zero, 1 - prob_zero])
    data = np.where((labels == 0)[:, None, None], zero[None], one[None]).astype(float)
    noise = np.random.rand(n_samples, linear_dim, linear_dim) * sigma
    return data * (1 - sigma) + noise, labels


def _read_idx(path):
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as fh:
        zero, dtype, ndim = struct.unpack('>HBB', fh.read(4))
        if zero != 0 or dtype != 0x08:
            raise ValueError('%s is not an unsigned-byte IDX file' % path)
        shape = struct.unpack('>' + 'I' * ndim, fh.read(4 * ndim))
        return np.frombuffer(fh.read(), dtype=np.uint8).reshape(shape)


def get_MNIST_dataset(data_root_dir='./datasets', download=True):
    """MNIST as uint8 arrays (60000,28,28), (60000,), (10000,28,28), (10000,)
    (data_generator.py:55-87).  Reads the four IDX files from `data_root_dir` (also looked up under
    MNIST/raw/, the layout torchvision leaves behind); `download` is accepted for signature
    compatibility but nothing is ever fetched."""
    names = ['train-images-idx3-ubyte', 'train-labels-idx1-ubyte', 't10k-images-idx3-ubyte', 't10k-labels-idx1-ubyte']
    out = []
    for nm in names:
        for cand in (os.path.join(data_root_dir, nm), os.path.join(data_root_dir, nm + '.gz'),
                     os.path.join(data_root_dir, 'MNIST', 'raw', nm), os.path.join(data_root_dir, 'MNIST', 'raw', nm + '.gz')):
            if os.path.exists(cand):
                out.append(_read_idx(cand))
                break
        else:
            raise FileNotFoundError("MNIST file %s not found under %s (this package does not download)" % (nm, data_root_dir))
    return out[0], out[1].astype(np.int64), out[2], out[3].astype(np.int64)


class NumpyDataset:
    """(data[index], label[index]) pairs (data_generator.py:90-122)."""

    def __init__(self, data, label):
        self.data = data
        self.label = label

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        return (self.data[index], self.label[index])


class Batch(list):
    """A list of (x_i, y_i) tuples that also carries the stacked arrays."""
    X = None
    y = None


class SubsetRandomSampler:
    """Indices of a subset in a fresh random order on every pass."""

    def __init__(self, indices):
        self.indices = np.asarray(indices)

    def __iter__(self):
        return iter(self.indices[np.random.permutation(len(self.indices))].tolist())

    def __len__(self):
        return len(self.indices)


class DataLoader:
    """Minimal loader: batches of `batch_size` samples drawn through `sampler` (sequential when
    None), `drop_last` as in torch, one `Batch` per iteration (the reference uses
    `collate_fn=lambda x: x`, i.e. the raw list of samples)."""

    def __init__(self, dataset, batch_size=1, sampler=None, drop_last=False, collate_fn=None):
        self.dataset, self.batch_size, self.sampler, self.drop_last = dataset, int(batch_size), sampler, drop_last

    def _n(self):
        return len(self.sampler) if self.sampler is not None else len(self.dataset)

    def __len__(self):
        n = self._n()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = list(iter(self.sampler)) if self.sampler is not None else list(range(len(self.dataset)))
        for k in range(len(self)):
            idx = np.asarray(order[k * self.batch_size:(k + 1) * self.batch_size])
            out = Batch((self.dataset.data[i], self.dataset.label[i]) for i in idx)
            out.X = np.asarray(self.dataset.data)[idx]
            out.y = np.asarray(self.dataset.label)[idx]
            yield out


def psi(x):
    """Feature map [sin(pi x / 2), cos(pi x / 2)] on the last axis (data_generator.py:165-167)."""
    x = np.array((np.sin(np.pi * x / 2), np.cos(np.pi * x / 2)))
    return np.transpose(x, [1, 2, 0])


def prepare_dataset(data, label, train_perc, val_perc, train_batch_size, val_batch_size, test_batch_size):
    """Embed, split and wrap in loaders (data_generator.py:125-192).  As in the reference the
    pixels go through psi un-normalised: pass data in [0, 1]."""
    x = psi(data.reshape(len(data), -1))
    m = int(len(x) * train_perc)
    train_set = NumpyDataset(x[:m], label[:m])
    test_set = NumpyDataset(x[m:], label[m:])
    train_len = int(m * (1 - val_perc))
    train_loader = DataLoader(train_set, train_batch_size, sampler=SubsetRandomSampler(np.arange(train_len)), drop_last=True)
    val_loader = DataLoader(train_set, val_batch_size, sampler=SubsetRandomSampler(np.arange(train_len, m)), drop_last=True)
    test_loader = DataLoader(test_set, test_batch_size, drop_last=False)
    return train_loader, val_loader, test_loader
