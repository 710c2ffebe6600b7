#!/usr/bin/env python3
"""Print the kernel timeline (start, duration, queue, name) of a few consecutive sweep steps from a rocprofv3 kernel trace:
   python3 tools/trace_window.py <kernel_trace.csv> [first big_jacobi index] [how many jacobi kernels]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 400
count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
def short(n):
    n = re.sub(r'^void\s+', '', n).replace('(anonymous namespace)::', '').replace('tnml::', '')
    return re.sub(r'[<(].*$', '', n)
jac = [i for i, r in enumerate(rows) if 'big_jacobi' in r['Kernel_Name']]
lo, hi = jac[first], jac[first + count]
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo - 8:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  +%7.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), short(r['Kernel_Name'])))
