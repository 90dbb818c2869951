#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV into the per-kernel table kept under profiles/:
    python tools/kernel_stats.py <..._kernel_trace.csv> <iterations in the trace> "<header line>" > profiles/<name>.txt"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void\s+', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name if len(name) <= 96 else name[:93] + '...'


def main():
    path, iters, header = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row['Kernel_Name'])
            tot[k] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6
            cnt[k] += 1
    total = sum(tot.values())
    print('# ' + header)
    print('# %d iterations in the trace; total kernel time %.1f ms = %.1f ms / iteration' % (iters, total, total / iters))
    print('%-98s %7s %11s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', '%'))
    for k in sorted(tot, key=lambda k: -tot[k]):
        print('%-98s %7d %11.2f %10.1f %6.2f' % (k, cnt[k], tot[k], 1000.0 * tot[k] / cnt[k], 100.0 * tot[k] / total))


if __name__ == '__main__':
    main()
