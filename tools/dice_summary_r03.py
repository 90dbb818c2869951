#!/usr/bin/env python
"""Round-3 summary of the free-running Dice experiment: every implementation / realisation of the SAME training
process (tools/dice_seeds.py: 500 iterations at 64 x 64, batch 4, lr 1e-3, SWA over iterations 350-500, 256 validation pairs) as one
GROUP of per-seed results, and the question "do the groups sample one distribution?".

    python tools/dice_summary_r03.py profiles/r02_dice_*seed*.log profiles/r03_dice_*.log

Groups (label of the RESULT line, prefixed by the round of the log): product (HIP kernels, GPU), alt / biasgrad product (round-2
re-samplings), oracle (torch-CPU), oracle2 (torch-CPU, one thread: another summation order), oracle_gpu (the oracle's code on torch-ROCm's
library kernels), standin (the product's host logic on torch-CPU arithmetic).
Prints per-group mean / sd / standard error, a one-way analysis of variance over the groups that cover seeds 0-24, and the pooled
"product code" vs "oracle code" difference with a seed-blocked standard error.
"""
import math
import re
import sys

import numpy as np


def load(paths):
    res = {}
    for f in paths:
        rnd = 'r02' if '/r02_' in f or f.startswith('r02_') else 'r03'
        kind = re.search(r'r02_dice_(alt_|biasgrad_|)seed', f)
        for l in open(f):
            if l.startswith('RESULT'):
                _, seed, side, d1, d2, dm = l.split()
                res.setdefault('%s:%s%s' % (rnd, kind.group(1) if kind else '', side), {})[int(seed)] = float(dm)
    return res


def main():
    res = load(sys.argv[1:])
    print('%-26s %4s %8s %8s %8s %s' % ('group', 'n', 'mean', 'sd', 'se', 'runs below 0.25 ("not taken off")'))
    for k in sorted(res):
        a = np.array([res[k][s] for s in sorted(res[k])])
        sd = a.std(ddof=1) if len(a) > 1 else float('nan')
        print('%-26s %4d %8.4f %8.4f %8.4f %d' % (k, len(a), a.mean(), sd, sd / math.sqrt(len(a)), int((a < 0.25).sum())))
    # ---- one-way ANOVA over the groups with all of seeds 0..24 -------------------------------------------------------------------
    full = [k for k in sorted(res) if all(s in res[k] for s in range(25))]
    X = np.array([[res[k][s] for s in range(25)] for k in full])          # groups x seeds
    g, n = X.shape
    grand = X.mean()
    ssb = n * ((X.mean(1) - grand) ** 2).sum()
    ssw = ((X - X.mean(1, keepdims=True)) ** 2).sum()
    F = (ssb / (g - 1)) / (ssw / (g * (n - 1)))
    print('\none-way ANOVA over %d groups x seeds 0-24 (%s):' % (g, ', '.join(full)))
    print('  group means %s; sd of the group means %.4f, expected from the within-group scatter alone %.4f' % (
        ' '.join('%.3f' % v for v in X.mean(1)), X.mean(1).std(ddof=1), math.sqrt(ssw / (g * (n - 1)) / n)))
    try:
        from scipy import stats
        print('  F(%d, %d) = %.3f, p = %.3f' % (g - 1, g * (n - 1), F, 1 - stats.f.cdf(F, g - 1, g * (n - 1))))
        # seeds as blocks (the seed fixes data order and initial weights and explains part of the variance)
        sm = X.mean(0, keepdims=True)
        resid = X - X.mean(1, keepdims=True) - sm + grand
        Fb = (ssb / (g - 1)) / ((resid ** 2).sum() / ((g - 1) * (n - 1)))
        print('  with the seeds as blocks: F(%d, %d) = %.3f, p = %.3f' % (g - 1, (g - 1) * (n - 1), Fb, 1 - stats.f.cdf(Fb, g - 1, (g - 1) * (n - 1))))
    except ImportError:
        print('  F = %.3f' % F)
    # ---- product code vs oracle code, per seed (mean over the realisations available for that seed) ------------------------------------
    prod = [k for k in res if k.endswith('product')]
    orac = [k for k in res if 'oracle' in k]
    seeds = sorted(s for s in set().union(*[set(res[k]) for k in prod]) if any(s in res[k] for k in orac))
    d = []
    for s in seeds:
        p = [res[k][s] for k in prod if s in res[k]]
        o = [res[k][s] for k in orac if s in res[k]]
        d.append(np.mean(p) - np.mean(o))
    d = np.array(d)
    se = d.std(ddof=1) / math.sqrt(len(d))
    print('\nproduct code (%s) minus oracle code (%s), per-seed means over the realisations of a seed, %d seeds:' % (', '.join(sorted(prod)), ', '.join(sorted(orac)), len(d)))
    print('  mean %+.4f, se %.4f, 95 %% CI [%+.4f, %+.4f]' % (d.mean(), se, d.mean() - 1.99 * se, d.mean() + 1.99 * se))
    allp = np.array([v for k in prod for v in res[k].values()])
    allo = np.array([v for k in orac for v in res[k].values()])
    print('  pooled runs: product %d runs mean %.4f; oracle %d runs mean %.4f' % (len(allp), allp.mean(), len(allo), allo.mean()))
    if any(k.endswith('standin') for k in res):
        k = [k for k in res if k.endswith('standin')][0]
        ss = sorted(res[k])
        o = np.array([np.mean([res[q][s] for q in orac if s in res[q]]) for s in ss])
        p = np.array([np.mean([res[q][s] for q in prod if s in res[q]]) for s in ss])
        a = np.array([res[k][s] for s in ss])
        print('\nstand-in (product host logic, torch-CPU arithmetic) on seeds %s: mean %.4f; oracle code on the same seeds %.4f; product (GPU) on the same seeds %.4f' % (
            ss, a.mean(), o.mean(), p.mean()))


if __name__ == '__main__':
    main()
