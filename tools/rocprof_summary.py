#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/run_profiles.sh into the small files kept under profiles/:
  gpurun_out/prof_kt_<cfg>/**/*kernel_stats.csv        -> gpurun_out/summary_kernel_stats_<cfg>.csv (copied as is)
  gpurun_out/prof_{fetch,write,mfma}_<cfg>/**/*counter_collection.csv -> gpurun_out/summary_pmc_<cfg>.json
     per kernel and counter: launches and mean value per launch (FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports)."""
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


def short(name):
    name = re.sub(r'^void\s+', '', name).replace('(anonymous namespace)::', '').replace('tnml::', '')
    return re.sub(r'\(.*$', '', name)


stats = glob.glob(os.path.join(out, 'prof_kt_' + cfg, '**', '*kernel_stats.csv'), recursive=True)
if stats:
    dst = os.path.join(out, 'summary_kernel_stats_%s.csv' % cfg)
    shutil.copy(stats[0], dst)
    print('kernel stats ->', dst)
    for row in list(csv.DictReader(open(stats[0])))[:12]:
        print('  %-40s calls %7s  avg %10.1f ns  total %6.1f %%' % (short(row['Name'])[:40], row['Calls'], float(row['AverageNs']), float(row['Percentage'])))
else:
    print('no kernel_stats.csv found')

pmc = {}
for tag in ('fetch', 'write', 'mfma'):
    for f in glob.glob(os.path.join(out, 'prof_%s_%s' % (tag, cfg), '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row['Kernel_Name'])
            c = row['Counter_Name']
            d = pmc.setdefault(k, {}).setdefault(c, {'launches': 0, 'sum': 0.0})
            d['launches'] += 1
            d['sum'] += float(row['Counter_Value'])
res = {}
for k, cs in pmc.items():
    res[k] = {}
    for c, d in cs.items():
        key = 'mean_KB' if c in ('FETCH_SIZE', 'WRITE_SIZE') else 'mean'
        res[k][c] = {'launches': d['launches'], key: d['sum'] / max(d['launches'], 1)}
dst = os.path.join(out, 'summary_pmc_%s.json' % cfg)
json.dump(res, open(dst, 'w'), indent=1, sort_keys=True)
print('pmc summary ->', dst)
for k, cs in sorted(res.items()):
    print('  %-36s %s' % (k[:36], ' | '.join('%s %.4g (x%d)' % (c, list(v.values())[1], v['launches']) for c, v in sorted(cs.items()))))
