#!/usr/bin/env python
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: python tools/pmc_summary.py <dir> [kernel substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            if pat and pat not in k:
                continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
    for k, v in agg.items():
        print(k)
        for c, val in sorted(v.items()):
            d = max(len(n[(k, c)]), 1)
            print('    %-28s %14.4g   per dispatch %14.4g  (%d dispatches)' % (c, val, val / d, d))


if __name__ == '__main__':
    main()
