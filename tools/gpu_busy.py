#!/usr/bin/env python
"""GPU-busy fraction and launch gaps of the timed region from a rocprofv3 --kernel-trace CSV of bench.py (the question a hipGraph
capture would answer: how much of the wall time is the GPU waiting for the next launch?).

    python tools/gpu_busy.py <kernel_trace.csv> <iterations in the trace> [warm-up iterations to skip = 3] [--gaps]
        --gaps: also list which kernels the long gaps (> 20 us) follow and precede
"""
import csv
import sys


def main():
    argv = [a for a in sys.argv if a != '--gaps']
    path, iters = argv[1], int(argv[2])
    skip = int(argv[3]) if len(argv) > 3 else 3
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path)))
    iv = [(s, e) for s, e, _ in rows]
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
    if '--gaps' in sys.argv:
        rows = rows[int(n * skip / float(iters)):]
        after, ce, prev = {}, rows[0][1], rows[0][2]
        for s, e, name in rows[1:]:
            if s > ce and s - ce > 20000:
                key = (prev.split('(')[0][:60], name.split('(')[0][:60])
                d = after.setdefault(key, [0, 0])
                d[0] += 1
                d[1] += s - ce
            if e >= ce:
                ce, prev = e, name
        print('long gaps (> 20 us) by (kernel before -> kernel after): count per iteration, ms per iteration')
        for key, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
            print('  %6.1f  %7.3f   %s  ->  %s' % (c / float(timed), t / 1e6 / timed, key[0], key[1]))


if __name__ == '__main__':
    main()
