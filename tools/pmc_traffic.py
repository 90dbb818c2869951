#!/usr/bin/env python
"""HBM traffic per launch of every convolution kernel instance from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-timer [workload flags]`, as MI355X_MICROARCH.md
prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced reads -> doubled.

    python tools/pmc_traffic.py <FETCH_SIZE dir> <WRITE_SIZE dir> <out.json> <workload key>

The JSON holds one entry per workload key (bench.py's `<model>-<decoder>-<size>-bs<batch>-<dtype>-lmix<l>`): per kernel name as
rocprofv3 prints it, and per family (conv_fwd = forward + data-gradient launches, conv_wgrad = weight gradient + slab reduce).
"""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        agg[k][0] += float(r['Counter_Value'])
        agg[k][1] += 1
    return agg


def family(k):
    if k.startswith(('conv16h_kernel', 'conv16_kernel', 'conv_fast_kernel', 'conv_fast_batched_kernel', 'conv_fwd_kernel', 'conv_direct_kernel', 'conv_direct_mfma_kernel', 'conv_dgrad', 'pw_reduce_kernel',
                     'smallk_conv_kernel', 'tapsum_kernel')):
        return 'conv_fwd'
    if k.startswith(('wgrad32h_kernel', 'conv_wgrad', 'slab_reduce', 'pw_reduce_wgrad_kernel', 'smallk_wgrad_kernel')):
        return 'conv_wgrad'
    return None


if __name__ == '__main__':
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out_path, wkey = sys.argv[3], sys.argv[4]
    kernels, fam = {}, collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k in fetch:
        if family(k) is None:
            continue
        n = fetch[k][1]
        rd = 2.0 * fetch[k][0] * 1024 / n          # gfx950: FETCH_SIZE counts 64 B per 128-B request
        wr = write[k][0] * 1024 / max(write[k][1], 1)
        kernels[k] = {'launches_profiled': n, 'hbm_read_bytes_per_launch': rd, 'hbm_write_bytes_per_launch': wr,
                      'hbm_bytes_per_launch': rd + wr}
        f = fam[family(k)]
        f[0] += 2.0 * fetch[k][0] * 1024
        f[1] += write[k][0] * 1024
        if not k.startswith('slab_reduce'):
            f[2] += n
    fams = {name: {'launches_profiled': v[2], 'hbm_read_bytes_per_launch': v[0] / v[2], 'hbm_write_bytes_per_launch': v[1] / v[2],
                   'hbm_bytes_per_launch': (v[0] + v[1]) / v[2]} for name, v in fam.items() if v[2]}
    doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
    doc.setdefault('method', 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); KiB; FETCH x2 (gfx950); the slab '
                             'reduction launches are charged to the weight-gradient family')
    doc.setdefault('workloads', {})[wkey] = {'kernels': kernels, 'families': fams}
    json.dump(doc, open(out_path, 'w'), indent=1, sort_keys=True)
    print(json.dumps(fams, indent=1))
