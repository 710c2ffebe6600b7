#!/usr/bin/env python3
"""GPU: in-kernel cycle stamps of the update + SVD workgroup at the middle step of a C3 sweep, persistent launch vs one launch per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tensornetworkforml_amd import _hip

N, M, b, L, D = 784, int(os.environ.get('M', '20')), int(os.environ.get('B', '5000')), 2, 2
hp = (1e-3, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, 'fixed')
for persistent in (True, False):
    ctx = _hip.Context(N, D, L, M, b)
    ctx.set_persistent(persistent)
    batches = [bench.synth(N, b, L, 1234 + 97 * k) for k in range(4)]
    for k, (X, y) in enumerate(batches):
        ctx.stage_batch(k, X, y)
    ctx.select_batch(0)
    ctx.set_cores(bench.init_cores(N, M, D, L, 99), 0)
    ctx.scale_cores(1.0 / float(np.exp(ctx.forward_logabsmax() / N)))
    for it in range(6):
        ctx.select_batch(it % 4); ctx.forward(want_f=False)
        ctx.sweep(ctx.l_pos == N - 1, N - 1, True, *hp, want_metrics=False, want_f=False)
    ctx.synchronize()
    ctx.debug_enable(2)
    ctx.select_batch(2); ctx.forward(want_f=False)
    left = ctx.l_pos == N - 1
    if persistent:
        ctx.sweep(left, N - 1, True, *hp, want_metrics=False, want_f=False)
    else:
        ctx.sweep(left, (N - 1) // 2 + 1, True, *hp, want_metrics=False, want_f=False)
    ctx.synchronize()
    st = ctx.step_debug('scalars')[5:]
    print('%s: cycles pre %.0f / jacobi %.0f / post %.0f ; rounds %.0f ; kernel-or-step us %.1f | pre split: front %.0f (flags seen at %.0f) / B %.0f / L2 %.0f / sums+update %.0f / gram %.0f | post split: sort %.0f / cores %.0f / norm env %.0f'
          % ('persistent' if persistent else 'per-step  ', st[0], st[1], st[2], st[50], st[3] / 100.0, st[9], st[20], st[10], st[11], st[12], st[13], st[6], st[7], st[8]))
    if persistent:
        print('   helper 0, part 2 of that step, us relative to the update workgroup starting the step: enter %.2f | core flag seen %.2f | T seen %.2f | Z seen %.2f | operands in LDS %.2f | products issued %.2f | arrived %.2f | update workgroup saw all arrivals %.2f'
              % tuple((st[i] - st[28]) / 100.0 for i in (22, 23, 30, 24, 25, 26, 27, 29)))
        print('   batch-side workgroup 0 in the iteration of that step (f of step k, Z of step k+1), same clock: enter %.2f | core flag seen %.2f | B_new flag seen %.2f | tiles done %.2f | partial stored + ticket %.2f | reduced Z published (last arriver) %.2f'
              % tuple((st[i] - st[28]) / 100.0 for i in (32, 33, 34, 35, 36, 37)))
    ctx.close()
