#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/run_profiles.sh into the small files kept under profiles/:
  gpurun_out/prof_kt_<cfg>/**/*kernel_stats.csv   -> gpurun_out/summary_kernel_stats_<cfg>.csv        (whole process, copied as is)
  gpurun_out/prof_kt_<cfg>/**/*kernel_trace.csv   -> gpurun_out/summary_kernel_stats_<cfg>_timed.csv  (the TIMED passes only)
  gpurun_out/prof_{fetch,write,mfma}_<cfg>/**/*counter_collection.csv -> gpurun_out/summary_pmc_<cfg>.json
     per kernel and counter: launches and mean value per launch, whole process and timed passes (FETCH_SIZE / WRITE_SIZE in
     KB as rocprofv3 reports them).

"Timed passes" = the dispatches between bench.py's phase markers 2 and 3 (tnml_marker: an empty kernel
`tnml_phase_marker_kernel` whose grid is 64 x id threads).  Markers are ordinary dispatches, so the same cut works on a
kernel trace (by time stamps) and on a counter pass (by dispatch order), and no profiler marker API is involved."""
import csv
import glob
import json
import os
import re
import shutil
import sys

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'gpurun_out')
MARK = 'tnml_phase_marker_kernel'
PHASES = {(1, 2): 'warmup', (2, 3): 'timed', (4, 5): 'resident', (6, 7): 'cold'}


def short(name):
    name = re.sub(r'^void\s+', '', name).replace('(anonymous namespace)::', '').replace('tnml::', '')
    return re.sub(r'[<(].*$', '', name)


def grid_threads(row):
    if 'Grid_Size' in row and row['Grid_Size'] not in ('', None):
        return int(row['Grid_Size'])
    g = 1
    for ax in ('X', 'Y', 'Z'):
        g *= int(row.get('Grid_Size_' + ax, 1) or 1)
    return g


def phase_of(marks_seen):
    """name of the phase the dispatches after the last seen marker belong to (None outside the named phases)"""
    if not marks_seen:
        return None
    last = marks_seen[-1]
    for (a, b), nm in PHASES.items():
        if last == a:
            return nm
    return None


# ---- kernel trace -------------------------------------------------------------------------------------------------------------
stats = glob.glob(os.path.join(out, 'prof_kt_' + cfg, '**', '*kernel_stats.csv'), recursive=True)
if stats:
    dst = os.path.join(out, 'summary_kernel_stats_%s.csv' % cfg)
    shutil.copy(stats[0], dst)
    print('kernel stats (whole process) ->', dst)
    for row in list(csv.DictReader(open(stats[0])))[:8]:
        print('  %-40s calls %7s  avg %12.1f ns  total %6.1f %%' % (short(row['Name'])[:40], row['Calls'], float(row['AverageNs']), float(row['Percentage'])))
else:
    print('no kernel_stats.csv found')

traces = glob.glob(os.path.join(out, 'prof_kt_' + cfg, '**', '*kernel_trace.csv'), recursive=True)
if traces:
    rows = sorted(csv.DictReader(open(traces[0])), key=lambda r: int(r['Start_Timestamp']))
    per_phase = {}
    seen = []
    for r in rows:
        if MARK in r['Kernel_Name']:
            seen.append(grid_threads(r) // 64)
            continue
        ph = phase_of(seen)
        if ph is None:
            continue
        d = per_phase.setdefault(ph, {}).setdefault(short(r['Kernel_Name']), [])
        d.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for ph, kern in per_phase.items():
        total = sum(sum(v) for v in kern.values())
        dst = os.path.join(out, 'summary_kernel_stats_%s_%s.csv' % (cfg, ph))
        with open(dst, 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
            for k, v in sorted(kern.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k, len(v), sum(v), '%.1f' % (sum(v) / len(v)), '%.2f' % (100.0 * sum(v) / max(total, 1)), min(v), max(v)])
        print('kernel stats of the %s passes (between the phase markers) -> %s' % (ph, dst))
        for k, v in sorted(kern.items(), key=lambda kv: -sum(kv[1]))[:5]:
            print('  %-40s calls %7d  avg %12.1f ns  total %6.1f %%' % (k[:40], len(v), sum(v) / len(v), 100.0 * sum(v) / max(total, 1)))
    if not per_phase:
        print('no phase markers in the kernel trace (bench.py too old?)')
else:
    print('no kernel_trace.csv found')

# ---- counter passes -----------------------------------------------------------------------------------------------------------
pmc = {}
for tag in ('fetch', 'write', 'mfma'):
    for f in glob.glob(os.path.join(out, 'prof_%s_%s' % (tag, cfg), '**', '*counter_collection.csv'), recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r['Dispatch_Id']))
        seen, last_disp = [], None
        for row in rows:
            if MARK in row['Kernel_Name']:
                if row['Dispatch_Id'] != last_disp:
                    seen.append(grid_threads(row) // 64)
                    last_disp = row['Dispatch_Id']
                continue
            k = short(row['Kernel_Name'])
            c = row['Counter_Name']
            for scope in ('whole_process', phase_of(seen)):
                if scope is None:
                    continue
                d = pmc.setdefault(scope, {}).setdefault(k, {}).setdefault(c, {'launches': 0, 'sum': 0.0})
                d['launches'] += 1
                d['sum'] += float(row['Counter_Value'])
res = {}
for scope, kerns in pmc.items():
    for k, cs in kerns.items():
        for c, d in cs.items():
            key = 'mean_KB' if c in ('FETCH_SIZE', 'WRITE_SIZE') else 'mean'
            res.setdefault(scope, {}).setdefault(k, {})[c] = {'launches': d['launches'], key: d['sum'] / max(d['launches'], 1)}
dst = os.path.join(out, 'summary_pmc_%s.json' % cfg)
json.dump(res, open(dst, 'w'), indent=1, sort_keys=True)
print('pmc summary ->', dst)
for scope in ('timed', 'whole_process'):
    for k, cs in sorted(res.get(scope, {}).items()):
        print('  [%s] %-36s %s' % (scope, k[:36], ' | '.join('%s %.4g (x%d)' % (c, list(v.values())[1], v['launches']) for c, v in sorted(cs.items()))))
