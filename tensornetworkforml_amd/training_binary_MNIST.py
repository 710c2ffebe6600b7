#!/usr/bin/env python3
"""Binary (0 vs 1) MNIST classifier on the MI355X backend: the counterpart of the reference's
training_binary_MNIST.py (same flags and defaults, training_binary_MNIST.py:38-46): digits 0/1,
2x2 max-pooling to 14x14 (N = 196).  MNIST is read from local IDX files under --data_dir (nothing is
downloaded).  As in the reference the raw 0..255 pixels go through psi un-normalised unless
--normalise is given (with raw pixels the features are in {-1, 0, 1}
and the reference does not learn).
"""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetworkforml_amd  # noqa: E402,F401
import data_generator as gen   # noqa: E402
import Network_class as tn     # noqa: E402


def pooling(X):
    """2x2 max pooling of a stack of images (the reference uses skimage.measure.block_reduce)."""
    n, h, w = X.shape
    return X[:, :h - h % 2, :w - w % 2].reshape(n, h // 2, 2, w // 2, 2).max(axis=(2, 4))


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train the Tensor Network to classify a binary MNIST dataset')
    ap.add_argument('--data_dir', type=str, default='datasets')
    ap.add_argument('--n_train_batch', type=int, default=10)
    ap.add_argument('--M', type=int, default=3)
    ap.add_argument('--n_epochs', type=int, default=3)
    ap.add_argument('--lr', type=float, default=0.001)
    ap.add_argument('--L2_decay', type=float, default=1e-56)
    ap.add_argument('--act_fn', type=str, default='softmax')
    ap.add_argument('--loss_fn', type=str, default='full_cross_ent')
    ap.add_argument('--trunc', type=str, default='reference', choices=['reference', 'fixed'])
    ap.add_argument('--normalise', action='store_true', help='scale pixels to [0, 1] before the feature map')
    ap.add_argument('--out', type=str, default='trained_MNIST_model.dat')
    args = ap.parse_args(argv)

    train_data, train_labels, test_data, test_labels = gen.get_MNIST_dataset(args.data_dir)
    data = pooling(np.concatenate((train_data, test_data)))
    labels = np.concatenate((train_labels, test_labels))
    mask = (labels == 0) | (labels == 1)
    data01, labels01 = data[mask], labels[mask]
    if args.normalise:
        data01 = data01 / 255.0
    train_batch = int(len(data01) * 0.8 / args.n_train_batch)
    train_loader, val_loader, _ = gen.prepare_dataset(data01, labels01, 1, 0.2, train_batch, 128, 128)
    x_cal = next(iter(train_loader)).X
    net = tn.Network(N=data[0].size, M=args.M, L=2, calibration_X=x_cal, normalize=True, act_fn=args.act_fn,
                     loss_fn=args.loss_fn, trunc=args.trunc)
    val_acc, var_hist = net.train(train_loader, val_loader, lr=args.lr, n_epochs=args.n_epochs,
                                  weight_dec=args.L2_decay)
    with open(args.out, 'wb') as fh:
        pickle.dump(net, fh)
    print('validation accuracy per epoch:', ['%.4f' % v for v in val_acc])
    return val_acc, var_hist


if __name__ == '__main__':
    main()
