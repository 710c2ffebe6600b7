"""GPU: where does a production sweep spend its time?  Two deterministic runs of the same passes on one context:
  --trace   : P passes back to back (run it under `rocprofv3 --kernel-trace --output-format csv`)
  --rounds  : the same passes, the last one step by step with the Jacobi statistics of every step  -> JSON
  --merge T R : per-dispatch durations of the last pass from the trace CSV T against the rounds of JSON R
"""
import json
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
N, M, D, L, b, P = 784, 20, 2, 2, 5000, 7


def make_ctx():
    from tensornetworkforml_amd import _hip
    import bench
    X, y = bench.synth(N, b, L, 1000)
    ctx = _hip.Context(N, D, L, M, b)
    ctx.set_cores(bench.init_cores(N, M, D, L, 99), 0)
    ctx.set_input(X, y)
    F2 = float(np.exp(ctx.forward_logabsmax() / N))
    ctx.scale_cores(1.0 / F2)
    if os.environ.get('TL_CLASSIC'):          # the classic launch sequence: the update + SVD kernel runs with nothing beside it
        ctx.set_step_pipeline(False)
    return ctx


hp = (1e-3, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, 'fixed')

if sys.argv[1] == '--trace':
    ctx = make_ctx()
    for _ in range(P):
        ctx.forward(want_f=False)
        ctx.sweep(ctx.l_pos == N - 1, N - 1, True, *hp, want_metrics=False, want_f=False)
    ctx.synchronize()
    ctx.close()
elif sys.argv[1] == '--rounds':
    ctx = make_ctx()
    for _ in range(P - 1):
        ctx.forward(want_f=False)
        ctx.sweep(ctx.l_pos == N - 1, N - 1, True, *hp, want_metrics=False, want_f=False)
    ctx.forward(want_f=False)
    left = ctx.l_pos == N - 1
    ctx.debug_enable(2)
    rows = []
    for k in range(N - 1):
        ctx.svd_stats(reset=True)
        ctx.sweep(left, 1, k == 0, *hp, want_metrics=False, want_f=False)
        sw, nsvd, rounds = ctx.svd_stats()
        sc = ctx.step_debug('scalars')
        rows.append({'k': k, 'sweeps': sw, 'rounds': rounds, 'chol': ctx.cholesky_steps, 'n': sc[10], 'wg0_us': sc[8] / 100.0})
    json.dump(rows, open(sys.argv[2], 'w'))
    ctx.close()
else:
    import csv
    kname = 'narrow_step_kernel' if os.environ.get('TL_CLASSIC') else 'step_pipe_kernel'
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if kname in r['Kernel_Name']]
    rd = json.load(open(sys.argv[3]))
    last = rows[-(N - 1):]                       # the N-1 step launches of the last pass (its prologue launch precedes them)
    dur = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last])
    gap = np.array([(int(last[i + 1]['Start_Timestamp']) - int(last[i]['End_Timestamp'])) / 1e3 for i in range(len(last) - 1)])
    rounds = np.array([r['rounds'] for r in rd])
    n = np.array([r['n'] for r in rd])
    wg0 = np.array([r['wg0_us'] for r in rd])
    mid = n == 2 * M
    print('last pass: %d launches, total %.2f ms kernel time, %.2f ms of gaps; mean duration %.1f us, mean gap %.2f us (max %.1f)'
          % (len(last), dur.sum() / 1e3, gap.sum() / 1e3, dur.mean(), gap.mean(), gap.max()))
    A = np.vstack([np.ones(mid.sum()), rounds[mid]]).T
    coef = np.linalg.lstsq(A, dur[mid], rcond=None)[0]
    print('mid-chain launches (n = %d): duration = %.1f us + %.3f us x rounds   (rounds %.1f mean, %d..%d)' % (2 * M, coef[0], coef[1], rounds[mid].mean(), rounds[mid].min(), rounds[mid].max()))
    coef2 = np.linalg.lstsq(A, wg0[mid], rcond=None)[0]
    print('the same steps run one per call, workgroup 0 alone (realtime counter): %.1f us + %.3f us x rounds' % (coef2[0], coef2[1]))
    for lo, hi in ((0, 60), (60, 80), (80, 120), (120, 400)):
        sel = mid & (rounds >= lo) & (rounds < hi)
        if sel.any():
            print('   rounds %3d..%3d: %4d launches, traced duration %.1f us, stepwise workgroup 0 %.1f us' % (lo, hi, sel.sum(), dur[sel].mean(), wg0[sel].mean()))
