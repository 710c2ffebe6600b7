#!/usr/bin/env python3
"""GPU: the pipelined large-tensor step against the classic launch sequence (tnml_set_step_pipeline(0)) on whole sweeps: f, metrics, device time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tensornetworkforml_amd import _hip
from tensornetworkforml_amd.Network_class import random_canonical_cores

N, M, b, L, D = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (24, 50, 5000, 10, 2))]
rng = np.random.default_rng(0)
p = rng.random((b, N), dtype=np.float32) * (rng.random((b, N), dtype=np.float32) > 0.6)
X = np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1).astype(np.float32)
y = rng.integers(0, L, b).astype(np.int32)
cores = random_canonical_cores(N, M, D, L, scale=M * 0.5 * 0.64 * D, rng=rng)
hp = (1e-3, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, 'fixed')
res = {}
for pipe in (0, 1):
    ctx = _hip.Context(N, D, L, M, b)
    ctx.set_step_pipeline(pipe)
    ctx.set_cores(cores, 0)
    ctx.set_input(X, y)
    ctx.scale_cores(1.0 / float(np.exp(ctx.forward_logabsmax() / N)))
    outs = []
    for sw in range(4):
        ctx.forward(want_f=False)
        met, f = ctx.sweep(ctx.l_pos == N - 1, N - 1, True, *hp)
        outs.append((met, f))
    ctx.profile_reset(); ctx.profile_enable(2)
    for sw in range(4):
        ctx.forward(want_f=False)
        ctx.sweep(ctx.l_pos == N - 1, N - 1, True, *hp, want_metrics=False, want_f=False)
    ms, nl = ctx.profile_get(4)
    print('pipeline %d: %.1f us per step, %d launches in 4 sweeps' % (pipe, 1e3 * ms / (4 * (N - 1)), nl))
    res[pipe] = outs
    ctx.close()
for sw in range(4):
    (m0, f0), (m1, f1) = res[0][sw], res[1][sw]
    print('sweep %d: max|f| %.3e  |f_pipe - f_classic| / max %.2e ; metrics diff acc %.2e mae %.2e ; f_pipe zeros: %s' % (
        sw, np.abs(f0).max(), np.abs(f1 - f0).max() / max(np.abs(f0).max(), 1e-30), np.abs(m0[:, 0] - m1[:, 0]).max(), np.abs(m0[:, 1] - m1[:, 1]).max(), bool((f1 == 0).all())))
