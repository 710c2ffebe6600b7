#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of a C5 run: per pipelined large-tensor step, where the time between the end of one step's chain
(big_norm_out_kernel) and the start of the next (big_front_kernel) goes, and which of the two streams the step waited for.
   python3 tools/c5_gaps.py <kernel_trace.csv>"""
import csv, re, sys
import numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'^void\s+', '', n).replace('(anonymous namespace)::', '').replace('tnml::', '')
    return re.sub(r'[<(].*$', '', n)
ev = [(short(r['Kernel_Name']), int(r['Start_Timestamp']) / 1e3, int(r['End_Timestamp']) / 1e3) for r in rows]
gap, slack, period, jac, pre, hop2 = [], [], [], [], [], []
last_norm_end = last_reduce_end = last_front = last_upd_end = None
for name, s, e in ev:
    if name == 'big_norm_out_kernel': last_norm_end = e
    elif name == 'reduce_slabs_kernel': last_reduce_end = e
    elif name == 'big_update_kernel': last_upd_end = e
    elif name == 'wide_step_mfma_tiled_kernel' and last_upd_end is not None: hop2.append(s - last_upd_end)
    elif name == 'big_jacobi_kernel':
        jac.append(e - s)
        if last_front is not None: pre.append(s - last_front)
    elif name == 'big_front_kernel':
        if last_norm_end is not None and last_reduce_end is not None and s - last_norm_end < 200:
            gap.append(s - last_norm_end); slack.append(last_reduce_end - last_norm_end)
        if last_front is not None and s - last_front < 2000: period.append(s - last_front)
        last_front = s
gap, slack, period, jac, pre, hop2 = map(np.array, (gap, slack, period, jac, pre, hop2))
print('steps %d: period mean %.1f us (median %.1f)' % (len(period), period.mean(), np.median(period)))
print('front start - norm_out end: mean %.1f median %.1f p90 %.1f' % (gap.mean(), np.median(gap), np.percentile(gap, 90)))
print('side stream (reduce end) - norm_out end: mean %.1f median %.1f; side stream later than the chain on %.0f %% of steps' % (slack.mean(), np.median(slack), 100.0 * (slack > 0).mean()))
late = slack > 0
print('  chain later: gap mean %.1f (n %d) | side stream later: front start - reduce end mean %.1f (n %d)' % (gap[~late].mean(), (~late).sum(), (gap[late] - slack[late]).mean(), late.sum()))
print('front start -> jacobi start: mean %.1f | jacobi mean %.1f | tiled start - update end: mean %.1f median %.1f' % (pre.mean(), jac.mean(), hop2.mean(), np.median(hop2)))
