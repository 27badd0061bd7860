#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small per-kernel tables kept under profiles/.

    python profiles/summarise.py stats  <dir> <out.csv>       kernel-trace --stats run
    python profiles/summarise.py stats_db <dir> <out.csv>     the same from the rocpd database
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


def stats_db(d, out):
    """Same table from the rocpd database (*_results.db) that rocprofv3 writes when no CSV output is asked for."""
    import sqlite3
    db = sqlite3.connect(find(d, '_results.db')[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(db.execute('select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), '
                           'max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by 1 order by 3 desc' % (kd, ks)))
    total = sum(r[2] for r in rows)
    import subprocess
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'percent'])
        for r in rows:
            name = r[0][:-3] if r[0].endswith('.kd') else r[0]
            try:
                name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            w.writerow([short(name), r[1], '%.3f' % (r[2] / 1e6), '%.2f' % (r[3] / 1e3), '%.2f' % (r[4] / 1e3),
                        '%.2f' % (r[5] / 1e3), '%.2f' % (100. * r[2] / total)])


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


def traffic(fetch_csv, write_csv, out, kernel='k_stage_'):
    """HBM bytes per launch of the dominant kernel from the two PMC passes.  FETCH_SIZE and
    WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so it is
    doubled (checked on k_invist, whose read volume is known: 2 x 74.9 MiB reported = 258^3 doubles
    rounded up to 128-byte lines per row)."""
    import json
    tot = {'fetch': [0, 0.0], 'write': [0, 0.0]}
    per = {}
    for key, f in (('fetch', fetch_csv), ('write', write_csv)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel not in r['kernel']:
                    continue
                n = int(r['launches']); b = float(r['total']) * 1024.0 * (2.0 if key == 'fetch' else 1.0)
                tot[key][0] += n; tot[key][1] += b
                per.setdefault(r['kernel'], {})[key + '_bytes_per_launch'] = b / n
    res = {'kernel': kernel, 'launches': tot['fetch'][0],
           'fetch_bytes_per_launch': tot['fetch'][1] / max(tot['fetch'][0], 1),
           'write_bytes_per_launch': tot['write'][1] / max(tot['write'][0], 1),
           'per_variant': per,
           'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py; '
                     'KiB -> bytes; FETCH_SIZE x2 (gfx950)'}
    res['traffic_bytes_per_launch'] = res['fetch_bytes_per_launch'] + res['write_bytes_per_launch']
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)


if __name__ == '__main__':
    {'stats': stats, 'stats_db': stats_db, 'pmc': pmc, 'traffic': traffic}[sys.argv[1]](*sys.argv[2:])
