#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small per-kernel tables kept under profiles/.

    python profiles/summarise.py stats  <dir> <out.csv>       kernel-trace --stats run
    python profiles/summarise.py pmc    <dir> <out.csv>       one --pmc pass (per kernel, per counter:
                                                              launches, mean and total)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    return name.split('(')[0][:90]


def stats(d, out):
    rows = []
    for f in find(d, 'kernel_stats.csv'):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'percent'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], '%.3f' % (float(r['TotalDurationNs']) / 1e6),
                        '%.2f' % (float(r['AverageNs']) / 1e3), '%.2f' % (float(r['MinNs']) / 1e3),
                        '%.2f' % (float(r['MaxNs']) / 1e3), r['Percentage']])


def pmc(d, out):
    acc = defaultdict(lambda: [0, 0.0])
    meta = {}
    for f in find(d, 'counter_collection.csv'):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (short(r['Kernel_Name']), r['Counter_Name'])
                acc[k][0] += 1
                acc[k][1] += float(r['Counter_Value'])
                meta[short(r['Kernel_Name'])] = (r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''),
                                                 r.get('SGPR_Count', ''), r.get('LDS_Block_Size', ''),
                                                 r.get('Scratch_Size', ''))
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'counter', 'launches', 'mean_per_launch', 'total', 'vgpr', 'agpr', 'sgpr', 'lds', 'scratch'])
        for (k, c), (n, tot) in sorted(acc.items(), key=lambda kv: (-kv[1][1], kv[0])):
            w.writerow([k, c, n, '%.6g' % (tot / n), '%.6g' % tot] + list(meta[k]))


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
