#!/usr/bin/env python3
"""GPU probe: per-kernel time and Jacobi sweep counts of the sweep step on a headline-shaped chain
(short N so that it runs in seconds).  Usage: python tools/probe_step.py [N] [M] [b] [L]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensornetworkforml_amd import _hip                      # noqa: E402
from tensornetworkforml_amd.Network_class import random_canonical_cores  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2
D = 2
rng = np.random.default_rng(1)
p = rng.random((b, N), dtype=np.float32) * (rng.random((b, N), dtype=np.float32) > 0.81)
X = np.stack([np.sin(np.pi * p / 2), np.cos(np.pi * p / 2)], -1).astype(np.float32)
y = rng.integers(0, L, b).astype(np.int32)
ctx = _hip.Context(N, D, L, M, b)
ctx.set_cores(random_canonical_cores(N, M, D, L, scale=M * 0.5 * 0.64 * D, rng=rng), 0)
ctx.set_input(X, y)
ctx.scale_cores(1.0 / float(np.exp(ctx.forward_logabsmax() / N)))
hp = (1e-3, 1e-3, True, 'softmax', 'full_cross_ent', 0.1, 'fixed')


def stamp_line(tag, sc):
    cyc = sc[5] + sc[6] + sc[7]
    print('%s: cycles pre/jacobi/post = %d / %d / %d ; kernel %.1f us ; clock %.2f GHz ; sweeps %d n %d ; %.0f cycles per round'
          % (tag, sc[5], sc[6], sc[7], sc[8] / 100.0, cyc / max(sc[8] / 100.0, 1e-9) / 1e3, sc[9], sc[10],
             sc[6] / max(1, sc[55] if len(sc) > 55 and sc[55] > 0 else sc[9] * (sc[10] - 1))))
    if len(sc) > 18:
        print('   pre split: load %d / B %d / L2 %d / sums+update %d / gram %d cycles' % tuple(sc[14:19]))
    if len(sc) > 21:
        print('   one round, parameter wave: loads+new elements %d / rotation %d / stores+barrier %d cycles' % tuple(sc[19:22]))
    if len(sc) > 28:
        print('   wide kernel WG0: x-stage %d / operand stage %d / f+env MFMA %d / activation %d / gP %d / dB MFMA+store %d cycles (own activation %d)' % tuple(sc[22:29]))
    if len(sc) > 52 and sc[46] > 0:
        print('   round-timing experiment (cycles per round): all %d | params only %d | G only %d | params+G %d | V only %d | params+V %d | G+V %d' % tuple(sc[46:53]))
    if len(sc) > 38 and sc[33] > 0:
        print('   hand-off, 10 ns ticks after workgroup 0 started: wait over %d | (classic: reduce helpers; pipelined: poll starts, payload in LDS) %d..%d | slice helpers %d..%d | polls %d' % tuple(sc[33:39]))
    if len(sc) > 45 and sc[39] > 0:
        print('   fine stamps: phase-4 loop %d / block sums %d / update loop %d / barrier after update %d | zero fill+barrier %d / gram GEMM %d / symmetrise+tables %d' % tuple(sc[39:46]))
    if len(sc) > 54 and sc[53] > 0:
        print('   Cholesky step: factorisation %d / L^T L + symmetrise %d cycles' % (sc[53], sc[54]))
    if len(sc) > 84 and sc[61] > 0:
        names = ['loads issued', 'lds stored', 'barrier', 'contraction', 'prep loaded', '', 'sigma pow+barrier', 'gather+barrier', 'T2 gemm first pass', '', 'phase10 start', 'T2 gemm', 'Nh_new gemm', "B' gemm", 'T gemm', "G' gemm"]
        print('   y-stamps (cycles since kernel start):', ' | '.join('%s %d' % (names[i], sc[61 + i]) for i in range(16) if names[i] and sc[61 + i] > 0))
    if len(sc) > 93:
        print('   per wave (work cycles, barrier wait, role rank): ' + ' '.join('w%d:%d/%d/r%d' % (i, int(sc[77 + i] % 1e5), int(sc[77 + i] // 1e5), round((sc[77 + i] % 1) * 1e3) - 1) for i in range(16)))
    if len(sc) > 103 and sc[93 + 9] > 0:
        hs = sc[93:93 + 10]
        print('   slice helper 0, 10 ns ticks after workgroup 0 started: block start %d | body start %d | loads issued %d | operands in LDS %d | level 1 done %d | barrier %d | level 2 done %d | stores issued %d | drained %d'
              % tuple(int(hs[i] - hs[9]) for i in (8, 0, 1, 2, 3, 4, 5, 6, 7)))
        print('   slice helper 0: %d shader cycles from block start to drained = %.2f GHz' % (sc[93 + 10], sc[93 + 10] / max(hs[7] - hs[8], 1) / 10.0))
    if len(sc) > 115 and sc[93 + 12] > 0 and sc[93 + 12 + 6] > 0:
        h2 = sc[93 + 12:93 + 12 + 7]
        print('   slice helper 0, SECOND pass through the same code (ticks since its own start): loads issued %d | operands in LDS %d | level 1 done %d | barrier %d | level 2 done %d | stores issued %d'
              % tuple(int(h2[i] - h2[0]) for i in (1, 2, 3, 4, 5, 6)))
        h1 = sc[93:93 + 7]
        print('   first pass, same reference:                                                        loads issued %d | operands in LDS %d | level 1 done %d | barrier %d | level 2 done %d | stores issued %d'
              % tuple(int(h1[i] - h1[0]) for i in (1, 2, 3, 4, 5, 6)))
    if len(sc) >= 120 + 192 and sc[120] > 0:
        w = np.array(sc[120:120 + 192]).reshape(16, 12)
        base = w[:, 0][w[:, 0] > 0].min()
        print('   per-wave probe points (cycles since wave 0 started; rows = probe points, columns = waves 0..15):')
        for i in range(12):
            if w[:, i].max() > 0:
                print('     p%-2d ' % i + ' '.join('%6d' % (x - base if x > 0 else -1) for x in w[:, i]))
    if len(sc) > 13:
        print('   post split: sort %d / cores %d / norm env + metrics %d cycles' % (sc[11], sc[12], sc[13]))


def one_pass(**kw):
    ctx.forward(want_f=False)
    left = ctx.l_pos == N - 1
    return ctx.sweep(left, N - 1, True, *hp, **kw)


one_pass(want_metrics=False, want_f=False)
one_pass(want_metrics=False, want_f=False)
# sweep counts, step by step
ctx.debug_enable(True)
ctx.forward(want_f=False)
left = ctx.l_pos == N - 1
sw = []
for k in range(N - 1):
    ctx.sweep(left, 1, k == 0, *hp)
    sc = ctx.step_debug('scalars')
    sw.append((int(sc[3]), int(sc[4])))
    if k == (N - 1) // 2:
        stamp_line('isolated mid step', sc)
ctx.debug_enable(False)
print('jacobi (sweeps, n) per step:', sw[:6], '...', sw[len(sw) // 2], '...', sw[-3:])
print('mean sweeps (interior):', np.mean([s for s, n in sw if n == 2 * M]))
# timing: whole passes, then per kernel
ctx.synchronize()
import time
t0 = time.perf_counter()
ctx.timer_start()
for _ in range(4):
    one_pass(want_metrics=False, want_f=False)
t_enq = time.perf_counter() - t0
ms = ctx.timer_stop()
print('host enqueue: %.1f us per sweep step (GPU pass time below)' % (1e6 * t_enq / 4 / (N - 1)))
ctx.synchronize()
t0 = time.perf_counter()
ctx.forward(want_f=False)
t1 = time.perf_counter()
ctx.sweep(ctx.l_pos == N - 1, N - 1, True, *hp, want_metrics=False, want_f=False)
t2 = time.perf_counter()
ctx.synchronize()
t3 = time.perf_counter()
print('single pass from idle: forward enqueue %.1f us, sweep enqueue %.1f us per step, drain %.1f us per step'
      % (1e6 * (t1 - t0), 1e6 * (t2 - t1) / (N - 1), 1e6 * (t3 - t2) / (N - 1)))
print('pass: %.3f ms  -> %.1f us per sweep step (incl. forward)' % (ms / 4, 1e3 * ms / 4 / (N - 1)))
# stamps of the last step of a full-speed sweep (stops half way so that the last step is an interior one)
ctx.debug_enable(2)
ctx.forward(want_f=False)
left = ctx.l_pos == N - 1
ctx.sweep(left, N // 2, True, *hp, want_metrics=False, want_f=False)
ctx.synchronize()
stamp_line('back-to-back mid step', ctx.step_debug('scalars'))
ctx.sweep(left, N - 1 - N // 2, False, *hp, want_metrics=False, want_f=False)
ctx.debug_enable(0)
ctx.profile_reset()
ctx.profile_enable(True)
one_pass(want_metrics=False, want_f=False)
ctx.profile_enable(False)
for i, nm in enumerate(['env_chain', 'wide', 'reduce', 'narrow']):
    t, n = ctx.profile_get(i)
    print('%-10s avg %.1f us over %d launches' % (nm, 1e3 * t / max(n, 1), n))
ctx.close()
