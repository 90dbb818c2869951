#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) the launch count and the average duration (first launch of every
group dropped as warm-up).  python tools/kernel_trace_summary.py <dir or csv> [substring ...]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    pats = sys.argv[2:]
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    for f in files:
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        seen = {}
        for r in rows:
            name = r['Kernel_Name']
            if pats and not any(p in name for p in pats):
                continue
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            key = (name[:70], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''), r.get('Workgroup_Size_X', ''))
            seen.setdefault(key, []).append(d)
        for k, v in seen.items():
            w = v[1:] if len(v) > 1 else v
            print('%-72s grid %8s x %3s  n=%4d  avg %9.1f us  min %9.1f' % (k[0], k[1], k[2], len(v), sum(w) / len(w), min(w)))


if __name__ == '__main__':
    main()
