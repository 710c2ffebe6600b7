#!/usr/bin/env python3
"""GPU: diagnostics of the mixed-precision decomposition inside a real C3-shaped sweep (step by step over a stretch of sites after
a few warm-up passes): float32 rounds, first check's largest tangent / violation, steps, fallback, and the spectrum."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tensornetworkforml_amd import _hip

N, M, b, L, D = 784, int(os.environ.get('M', '20')), int(os.environ.get('B', '5000')), 2, 2
warm = int(os.environ.get('WARM', '5'))
ctx = _hip.Context(N, D, L, M, b)
batches = [bench.synth(N, b, L, 1234 + 97 * k) for k in range(4)]
for k, (X, y) in enumerate(batches):
    ctx.stage_batch(k, X, y)
ctx.select_batch(0)
ctx.set_cores(bench.init_cores(N, M, D, L, 99), 0)
ctx.scale_cores(1.0 / float(np.exp(ctx.forward_logabsmax() / N)))
hp = (1e-3, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, 'fixed')
for it in range(warm):
    ctx.select_batch(it % 4)
    ctx.forward(want_f=False)
    ctx.sweep(ctx.l_pos == N - 1, N - 1, True, *hp, want_metrics=False, want_f=False)
ctx.synchronize()
print('after %d passes:' % warm, ctx.svd_stats(reset=True), 'mixed', ctx.mixed_svds, 'steps', ctx.mixed_steps, 'fallbacks', ctx.mixed_fallbacks)
ctx.select_batch(warm % 4)
ctx.forward(want_f=False)
left = ctx.l_pos == N - 1
ctx.debug_enable(3)
first = True
for step in range(int(os.environ.get('STEPS', '60'))):
    ctx.sweep(left, 1, first, *hp, want_metrics=False, want_f=False)
    first = False
    ctx.synchronize()
    st = ctx.step_debug('scalars')[5:]
    sig = ctx.step_debug('sigma')
    if step % 4 == 0 or st[38]:
        print('step %3d n=%2.0f rounds32 %3.0f t0 %.1e rel0 %.1e steps %.0f failed %.0f t_last %.1e rel_last %.1e kept2f %.1e | cycles f32 %.0f KE %.0f steps %.0f | sigma/s1: %s'
              % (step, st[5], st[34], st[35], st[36], st[37], st[38], st[42], st[43], st[44], st[39], st[40], st[41],
                 ' '.join('%.0e' % v for v in ((sig / sig[0])[[1, 5, 10, 15, 18, 19, 20, 21, 25, 30, 39]] if len(sig) >= 40 else sig[:4] / sig[0]))))
