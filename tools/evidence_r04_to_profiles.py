#!/usr/bin/env python
"""Turn the output of tools/evidence_run_r04.sh (gpurun_out/ev6) into the summaries tracked under profiles/ (round 4):

    python tools/evidence_r04_to_profiles.py [gpurun_out/ev6]

  r04_bench_lines.jsonl                                   every bench.py JSON line of the session, labelled
  r04_final_kernel_stats_bench_dafnet_film_256_bs8.txt    per-kernel table of the rocprofv3 kernel trace of the headline workload
  r04_kernel_stats_bench_dafnet_spade_256_bs8_bf16_act16.txt   the same for BASELINE config #3's model
  r04_gpu_busy_bench_dafnet_film_256_bs8.txt              GPU-busy share / launch gaps of that trace
  r04_conv_traffic.json                                   HBM bytes per launch of every convolution kernel (PMC FETCH_SIZE / WRITE_SIZE passes)
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'ev6')
P = os.path.join(ROOT, 'profiles')

LINES = [('bench_f32', 'python bench.py --steps 20 --warmup 5 (headline: DAFNet-FiLM 256x256 bs8 fp32)'),
         ('bench_f32_multistream', '--multi-stream (fp32, discriminator phases on concurrent streams)'),
         ('bench_bf16_act16', '--dtype bf16 --act16 (FiLM; multi-stream is the default of the 16-bit modes)'),
         ('bench_spade_f32', '--decoder spade (fp32)'),
         ('bench_spade_bf16_act16', '--decoder spade --dtype bf16 --act16 (BASELINE config #3 model, per GPU)'),
         ('bench_spade_bf16_act16_onestream', '--decoder spade --dtype bf16 --act16 --no-multi-stream'),
         ('bench_mmsdnet3_320_f16_act16', '--model mmsdnet --modalities 3 --size 320 --batch 16 --dtype f16 --act16 (BASELINE config #5 model, per GPU)'),
         ('bench_spade_bf16_act16_conv16off', '--decoder spade --dtype bf16 --act16 --conv16 0 (the round-3 16-bit kernels only)'),
         ('bench_f32_conv16off', '--conv16 0 (fp32 headline workload on the round-3 kernels only)'),
         ('bench_lmix01', '--l_mix 0.1 (BASELINE config #4 schedule, per GPU)')]


def main():
    with open(os.path.join(P, 'r04_bench_lines.jsonl'), 'w') as out:
        for name, label in LINES:
            f = os.path.join(EV, name + '.json')
            if not os.path.exists(f):
                continue
            txt = open(f).read().strip().splitlines()
            if not txt:
                continue
            d = json.loads(txt[-1])
            d['_label'] = label
            # bench.py read gpu_busy_frac / traffic from the PREVIOUS session's files under profiles/ (this session's trace and PMC passes
            # ran after it): restate them from this session's summaries so that the line and the files committed beside it agree
            busy = os.path.join(EV, 'gpu_busy.txt')
            if 'gpu_busy_frac' in d.get('from_committed_profile', {}) and os.path.exists(busy):
                import re
                m = re.search(r'= ([0-9.]+) % GPU-busy', open(busy).read())
                if m:
                    d['from_committed_profile']['gpu_busy_frac']['value'] = float(m.group(1)) / 100.0
            # ... and the HBM bytes per launch from this session's PMC passes (bench.py read the previous session's file)
            tj = os.path.join(EV, 'conv_traffic.json')
            if name == 'bench_f32' and os.path.exists(tj):
                tw = json.load(open(tj)).get('workloads', {}).get('dafnet-film-256-bs8-f32-lmix1', {})
                kern, fams = tw.get('kernels', {}), tw.get('families', {})
                for e in [d.get('roofline', {})] + d.get('roofline_kernels', []):
                    if e.get('kernel') in kern:
                        e['traffic'] = kern[e['kernel']]['hbm_bytes_per_launch']
                for k, fk in (('conv_fwd_kernel', 'conv_fwd'), ('conv_wgrad_kernel', 'conv_wgrad')):
                    if k in d.get('roofline_family', {}) and fk in fams:
                        d['roofline_family'][k]['traffic'] = fams[fk]['hbm_bytes_per_launch']
            out.write(json.dumps(d) + '\n')
            print('%-36s %8.2f slices/s  %8.2f ms' % (name, d['value'], d['ms_per_step']))
    for src, dst in (('final_kernel_stats.txt', 'r04_final_kernel_stats_bench_dafnet_film_256_bs8.txt'),
                     ('kernel_stats_spade_bf16_act16.txt', 'r04_kernel_stats_bench_dafnet_spade_256_bs8_bf16_act16.txt'),
                     ('gpu_busy.txt', 'r04_gpu_busy_bench_dafnet_film_256_bs8.txt'), ('conv_traffic.json', 'r04_conv_traffic.json'),
                     ('conv16_ab.txt', 'r04_conv16_ab.txt'), ('conv16_ab_f32.txt', 'r04_conv16_ab_f32.txt')):
        if os.path.exists(os.path.join(EV, src)):
            shutil.copy(os.path.join(EV, src), os.path.join(P, dst))


if __name__ == '__main__':
    main()
