#!/usr/bin/env python3
"""GPU, experiment build -DTNML_ROUND32_TIMING (TNML_LIB=build_exp/libtnml_r32.so): cycles per float32 Jacobi round with roles off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tensornetworkforml_amd import _hip
rng = np.random.default_rng(0)
ctx = _hip.Context(24, 2, 2, 32, 64)
ctx.debug_enable(2)
for (r, c, m) in ((40, 80, 20), (20, 40, 10), (64, 128, 32)):
    W = rng.standard_normal((r, c)).astype(np.float32)
    ctx.svd_split(W, m)
    st = ctx.step_debug('scalars')[5:]
    print('n=%d: all %.0f | params only %.0f | G only %.0f | V only %.0f | barrier only %.0f cycles per round; production loop %.0f / %d rounds = %.0f'
          % (min(r, c), st[45], st[46], st[47], st[48], st[49], st[39], st[34], st[39] / max(st[34], 1)))
