#!/usr/bin/env python3
"""Diagonals classifier on the MI355X backend: the counterpart of the reference's
training_diagonals.py (same flags and defaults, training_diagonals.py:33-44; same outputs: a pickled
network and, when matplotlib is available, the accuracy / MAE curves).

    python tensornetworkforml_amd/training_diagonals.py [--M 10 --n_epochs 5 ...]
"""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetworkforml_amd  # noqa: E402,F401  (registers the bare module names)
import data_generator as gen   # noqa: E402
import Network_class as tn     # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train the Tensor Network to classify the dataset of diagonals')
    ap.add_argument('--n_samples', type=int, default=5000, help='Number of samples to generate')
    ap.add_argument('--linear_dim', type=int, default=8, help='Size of both dimensions of the samples')
    ap.add_argument('--sigma', type=float, default=0.7, help='Sigma of the noise added to the dataset')
    ap.add_argument('--n_train_batch', type=int, default=1, help='Number of batches the training set is split in')
    ap.add_argument('--M', type=int, default=10, help='Size of the bond between tensors of the network')
    ap.add_argument('--n_epochs', type=int, default=5, help='Number of epochs')
    ap.add_argument('--lr', type=float, default=0.01, help='Learning Rate')
    ap.add_argument('--L2_decay', type=float, default=1, help='Weight decay value for L2 regularization')
    ap.add_argument('--act_fn', type=str, default='softmax')
    ap.add_argument('--loss_fn', type=str, default='full_cross_ent')
    ap.add_argument('--trunc', type=str, default='reference', choices=['reference', 'fixed'],
                    help="truncation policy of the SVD split ('reference' = the original's rule)")
    ap.add_argument('--out', type=str, default='trained_diag_model.dat')
    args = ap.parse_args(argv)

    train_batch = int(args.n_samples * 0.8 / args.n_train_batch)
    data, label = gen.create_dataset(args.n_samples, args.linear_dim, args.sigma)
    train_loader, val_loader, _ = gen.prepare_dataset(data, label, 1, 0.2, train_batch, 128, 128)
    x_cal = next(iter(train_loader)).X
    net = tn.Network(N=args.linear_dim ** 2, M=args.M, L=2, calibration_X=x_cal, normalize=True,
                     act_fn=args.act_fn, loss_fn=args.loss_fn, trunc=args.trunc)
    val_acc, var_hist = net.train(train_loader, val_loader, lr=args.lr, n_epochs=args.n_epochs,
                                  weight_dec=args.L2_decay)
    with open(args.out, 'wb') as fh:
        pickle.dump(net, fh)
    print('validation accuracy per epoch:', ['%.4f' % v for v in val_acc])
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
    except ImportError:
        print('(matplotlib not installed: curves not drawn)')
        return val_acc, var_hist
    os.makedirs('results', exist_ok=True)
    xs = np.arange(args.n_epochs * var_hist.shape[2]) / var_hist.shape[2]
    for row, name, ylabel in ((0, 'accuracy', 'Accuracy'), (1, 'MAE', '| f(x) - y |')):
        plt.figure()
        plt.plot(xs, var_hist[:, row].reshape(-1), label='Train ' + name)
        if row == 0:
            plt.plot(np.arange(1, args.n_epochs + 1), val_acc, 'ro', label='Validation acc')
        plt.xlabel('Epoch'); plt.ylabel(ylabel); plt.legend()
        plt.savefig('results/diag_%s.png' % name)
        plt.close()
    return val_acc, var_hist


if __name__ == '__main__':
    main()
