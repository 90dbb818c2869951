#!/usr/bin/env python
"""GPU-busy fraction and launch gaps of the timed region from a rocprofv3 --kernel-trace CSV of bench.py (the question a hipGraph
capture would answer: how much of the wall time is the GPU waiting for the next launch?).

    python tools/gpu_busy.py <kernel_trace.csv> <iterations in the trace> [warm-up iterations to skip = 3]
"""
import csv
import sys


def main():
    path, iters = sys.argv[1], int(sys.argv[2])
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(path)))
    n = len(iv)
    iv = iv[int(n * skip / float(iters)):]                     # drop the warm-up iterations' launches
    t0, t1 = iv[0][0], max(e for _, e in iv)
    busy, gaps = 0, []
    cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    timed = iters - skip
    gaps.sort()
    print('kernel launches in the trace: %d (%.0f per iteration)' % (n, n / float(iters)))
    print('timed window (%d iterations): %.1f ms wall, %.1f ms with a kernel running = %.2f %% GPU-busy' %
          (timed, (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
    print('idle: %.2f ms per iteration in %d gaps (median %.1f us, 95th percentile %.1f us); gaps > 20 us: %d, %.2f ms per iteration' %
          (sum(gaps) / 1e6 / timed, len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.95)] / 1e3,
           sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6 / timed))
    print('=> a graph capture of the iteration could recover at most the idle share above')


if __name__ == '__main__':
    main()
