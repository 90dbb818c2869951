#!/usr/bin/env python
"""Turn the raw output of tools/evidence_run.sh (gpurun_out/ev4) into the summaries tracked under profiles/ (round 2):

    python tools/evidence_to_profiles.py [gpurun_out/ev4]

  r02_bench_lines.jsonl                         every bench.py JSON line of the session, labelled
  r02_final_kernel_stats_<workload>.{txt,csv}    per-kernel table of the rocprofv3 kernel trace of `python3 bench.py` (headline workload)
  r02_kernel_stats_<workload>.txt               the same for the 16-bit workloads
  r02_gpu_busy_<workload>.txt                   GPU-busy share / launches per iteration, eager and replayed from hipGraphs
  r02_conv_traffic.json                         HBM bytes per launch of every convolution kernel (PMC FETCH_SIZE / WRITE_SIZE passes)
  r02_pmc_mfma_lds_conv_kernels.txt             MFMA-busy / wait / LDS-conflict counters of the dominant kernels, fp32 and 16-bit
"""
import glob
import io
import json
import os
import subprocess
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
EV = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'ev4')
P = os.path.join(ROOT, 'profiles')
W = 'bench_dafnet_film_256_bs8'


def run(*cmd):
    return subprocess.run([sys.executable] + list(cmd), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout


def bench_lines():
    order = ['bench_f32', 'bench_f32_graphs', 'bench_f32_multistream', 'bench_bf16', 'bench_bf16_act16', 'bench_bf16_act16_multistream',
             'bench_spade_bf16_act16_multistream', 'bench_spade_f32', 'bench_spade_bf16',
             'bench_spade_bf16_act16', 'bench_mmsdnet3_320_f16', 'bench_mmsdnet3_320_f16_act16', 'bench_lmix01']
    out = []
    for name in order:
        for d in (EV, EV + 'b'):
            f = os.path.join(d, name + '.json')
            if os.path.exists(f):
                for line in open(f):
                    if line.startswith('{'):
                        rec = json.loads(line)
                        rec = dict([('_label', name)] + list(rec.items()))
                        out.append(json.dumps(rec))
    open(os.path.join(P, 'r02_bench_lines.jsonl'), 'w').write('\n'.join(out) + '\n')
    print('bench lines:', len(out))


def trace_csv(name):
    f = glob.glob(os.path.join(EV, name, '**', '*kernel_trace.csv'), recursive=True)
    return f[0] if f else None


def kernel_tables():
    for tr, iters, fname, header in (
            ('trace', 10, 'r02_final_kernel_stats_%s' % W,
             'rocprofv3 --kernel-trace --output-format csv : python3 bench.py --no-cpu-baseline --steps 7 --warmup 3 (round 2, DAFNet-FiLM 256x256 bs8 fp32, 1 x MI355X)'),
            ('trace_bf16_act16', 10, 'r02_kernel_stats_%s_bf16_act16' % W,
             'rocprofv3 --kernel-trace : python3 bench.py --no-cpu-baseline --dtype bf16 --act16 --steps 7 --warmup 3 (DAFNet-FiLM 256x256 bs8, bf16 MFMA operands, 16-bit trunk storage)'),
            ('trace_spade_bf16_act16', 7, 'r02_kernel_stats_bench_dafnet_spade_256_bs8_bf16_act16',
             'rocprofv3 --kernel-trace : python3 bench.py --no-cpu-baseline --decoder spade --dtype bf16 --act16 --steps 4 --warmup 3 (BASELINE config #3 model)')):
        c = trace_csv(tr)
        if not c:
            continue
        txt = run('tools/kernel_stats.py', c, str(iters), header)
        open(os.path.join(P, fname + '.txt'), 'w').write(txt)
        if tr == 'trace':
            rows = ['Name,Calls,TotalDurationMs,AverageUs,Percentage']
            for line in txt.splitlines()[3:]:
                parts = line.rsplit(None, 4)
                if len(parts) == 5:
                    rows.append('"%s",%s,%s,%s,%s' % (parts[0].strip(), parts[1], parts[2], parts[3], parts[4]))
            open(os.path.join(P, fname + '.csv'), 'w').write('\n'.join(rows) + '\n')
    busy = ['# tools/gpu_busy.py over the rocprofv3 kernel traces of `python3 bench.py --no-cpu-baseline --steps 7 --warmup 3 [--graphs]` (fp32 headline workload)']
    for tr, label in (('trace', 'eager (one host call per launch)'), ('trace_graphs', 'conf.hip_graphs: trainer steps replayed from hipGraphs'),
                      ('trace_bf16_act16', 'eager, --dtype bf16 --act16')):
        c = trace_csv(tr)
        if c:
            busy.append('\n== ' + label)
            busy.append(run('tools/gpu_busy.py', c, '10').rstrip())
    open(os.path.join(P, 'r02_gpu_busy_%s.txt' % W), 'w').write('\n'.join(busy) + '\n')


def traffic():
    out = os.path.join(P, 'r02_conv_traffic.json')
    if os.path.exists(out):
        os.remove(out)
    for fd, wd, key in (('pmc_fetch', 'pmc_write', 'dafnet-film-256-bs8-f32-lmix1'),
                        ('pmc_fetch_bf16a', 'pmc_write_bf16a', 'dafnet-film-256-bs8-bf16-act16-lmix1')):
        if os.path.isdir(os.path.join(EV, fd)):
            print(run('tools/pmc_traffic.py', os.path.join(EV, fd), os.path.join(EV, wd), out, key)[:300])


def pmc_text():
    hdr = '''# rocprofv3 --pmc passes over tools/conv_one.py / tools/wgrad_one.py (5 launches each, B = 8, 3x3 stride 1 'same'), 1 x MI355X, round 2.
# pass A: SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE (+ --kernel-trace)
# pass B: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
# MFMA pipe utilisation (cycle based) = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); effective clock = GRBM_GUI_ACTIVE / 8 / kernel duration
# 16-bit sections: DTYPE=bf16, IO=0 fp32 tensors in HBM / IO=5 16-bit input and output (forward) or x and dy (weight gradient)
'''
    out = [hdr]
    for sec in ('fwd_128_128_128', 'fwd_256_64_64', 'wgrad_128_128_128', 'wgrad_256_64_64'):
        out.append('==== fp32 ' + sec)
        for kind in ('mfma', 'lds'):
            d = os.path.join(EV, 'pmc_%s_%s' % (kind, sec))
            if os.path.isdir(d):
                if kind == 'mfma':
                    out.append(run('tools/kernel_trace_summary.py', d, 'conv_').rstrip())
                out.append(run('tools/pmc_summary.py', d, 'conv_').rstrip())
        out.append('')
    for sec in ('fwd_io0', 'fwd_io5', 'wgrad_io0', 'wgrad_io5'):
        out.append('==== bf16 ' + sec + ' (128 x 128, 128 -> 128)')
        for kind in ('mfma', 'lds'):
            d = os.path.join(EV, 'pmc16_%s_%s' % (kind, sec))
            if os.path.isdir(d):
                if kind == 'mfma':
                    out.append(run('tools/kernel_trace_summary.py', d, 'conv_').rstrip())
                out.append(run('tools/pmc_summary.py', d, 'conv_').rstrip())
        out.append('')
    before = os.path.join(ROOT, 'profiles', 'r02_pmc_lds_16bit_before.txt')
    if os.path.exists(before):
        out.append('==== LDS conflicts of the 16-bit kernels BEFORE the conflict-free images / swizzle (same commands; kept for reference)')
        out.append(open(before).read().rstrip())
    open(os.path.join(P, 'r02_pmc_mfma_lds_conv_kernels.txt'), 'w').write('\n'.join(out) + '\n')


if __name__ == '__main__':
    bench_lines()
    kernel_tables()
    traffic()
    pmc_text()
