#!/usr/bin/env python3
"""GPU: step-by-step comparison of the pipelined large-tensor step against the classic sequence (one tnml_sweep call per step)."""
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
ctxs = []
for pipe in (0, 1):
    ctx = _hip.Context(N, D, L, M, b)
    ctx.set_step_pipeline(pipe)
    ctx.set_cores(cores, 0)
    ctx.set_input(X, y)
    ctx.scale_cores(1.0 / float(np.exp(ctx.forward_logabsmax() / N)))
    ctx.forward(want_f=False)
    ctxs.append(ctx)
for k in range(N - 1):
    out = []
    for ctx in ctxs:
        met, f = ctx.sweep(False, 1, k == 0, *hp)
        cs, bond, lp = ctx.get_cores()
        out.append((met, f, cs, bond))
    (m0, f0, c0, b0), (m1, f1, c1, b1) = out
    if 5 <= k <= 7:
        for site in (k - 2, k - 1):
            e0 = ctxs[0].get_env(_hip.SIDE_LEFT, site); e1 = ctxs[1].get_env(_hip.SIDE_LEFT, site)
            print('    env L site %d: shape %s max %.3e diff %.3e nan %s' % (site, e0.shape, np.abs(e0).max(), np.abs(e0 - e1).max(), bool(np.isnan(e1).any())))
    cd = max(np.abs(a - b_).max() / max(np.abs(a).max(), 1e-30) for a, b_ in zip(c0, c1))
    print('step %2d bonds %s: f diff %.2e (max|f| %.2e, pipe zeros %s), metrics %s vs %s, cores diff %.2e' % (
        k, (int(b0[max(k - 1, 0)]), int(b0[k])), np.abs(f1 - f0).max() / np.abs(f0).max(), np.abs(f0).max(), bool((f1 == 0).all()), m0[0], m1[0], cd))
