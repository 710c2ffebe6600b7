#!/usr/bin/env python3
"""GPU: the in-LDS update + SVD workgroup alone (tnml_svd_split of a rows x cols matrix: one workgroup, no helpers), to be run under
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... for the LDS picture of the critical workgroup.
   python tools/probe_svd_lds.py [rows] [cols] [m] [repeats]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensornetworkforml_amd import _hip
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 80
m = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 200
kind = sys.argv[5] if len(sys.argv) > 5 else 'lowrank'     # lowrank | random | orth (rows already orthogonal: one quiet sweep)
rng = np.random.default_rng(0)
ctx = _hip.Context(8, 2, 2, max(rows, cols) // 2, 64)
# a matrix like a settled merged tensor: low rank + small update
A = rng.standard_normal((rows, m)).astype(np.float32) @ rng.standard_normal((m, cols)).astype(np.float32)
A += 1e-3 * rng.standard_normal((rows, cols)).astype(np.float32) * np.abs(A).mean()
if kind == 'random':
    A = rng.standard_normal((rows, cols)).astype(np.float32)
elif kind == 'orth':
    Q, _ = np.linalg.qr(rng.standard_normal((cols, rows)))
    A = (Q.T * np.linspace(1.0, 2.0, rows)[:, None]).astype(np.float32)
ctx.svd_stats(reset=True)
for _ in range(rep):
    US, SVh, sig = ctx.svd_split(A, m)
st = ctx.svd_stats()
print('ok', kind, 'rounds per svd %.1f' % (st[2] / max(st[1], 1)), sig[:3])
ctx.close()
