#!/usr/bin/env python
"""HBM traffic per launch of the convolution kernels from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-timer`, as MI355X_MICROARCH.md prescribes:
counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced reads -> doubled.

    python tools/pmc_traffic.py gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE profiles/r01_conv_traffic.json
"""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        fam = 'conv_fwd' if k.startswith(('conv_fast_kernel', 'conv_fast_batched_kernel', 'conv_fwd_kernel', 'conv_direct_kernel')) else \
              ('conv_wgrad' if k.startswith('conv_wgrad') else k)
        agg[fam][0] += float(r['Counter_Value'])
        agg[fam][1] += 1
    return agg


if __name__ == '__main__':
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for fam in ('conv_fwd', 'conv_wgrad'):
        n = fetch[fam][1]
        rd = 2.0 * fetch[fam][0] * 1024 / n          # gfx950: FETCH_SIZE counts 64 B per 128-B request
        wr = write[fam][0] * 1024 / write[fam][1]
        out[fam] = {'launches_profiled': n, 'hbm_read_bytes_per_launch': rd, 'hbm_write_bytes_per_launch': wr,
                    'hbm_bytes_per_launch': rd + wr}
    out['method'] = 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace; KiB; FETCH x2 (gfx950)'
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out, indent=1))
