#!/usr/bin/env python
"""Benchmark of the hot path: one DAFNet training iteration (BASELINE.json config[1]: dafnet_config_chaos, FiLM
decoder, 256x256 two-modality slices, batch 8 per GPU, fp32).

A "step" = DAFNetExecutor.train_batch = supervised_trainer.fit + 2 x D_Mask_trainer.fit + D_Image1/2_trainer.fit
incl. the fake-pool generation, Adam updates and BatchNorm moving-average updates -- nothing is skipped.  Inputs
(synthetic, seeded) are resident in HBM before the timed region.  Metric: paired 2-D slices per second, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W]         # N > 1: starts its N ranks itself (child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Extra fields of the JSON line:
  roofline         -- the DOMINANT kernel instance of the step (largest GPU time), under the name rocprofv3 reports for it, e.g.
                      conv_fast_kernel<128, 128, 2, 2, 0>: algorithmic FLOPs (2*M*K*N per launch, from the launch geometry) / HIP-event
                      time of the sampled launches inside the timed region, against the MFMA peak of the compute dtype;
                      `traffic` = HBM bytes per launch of that kernel from rocprofv3 PMC passes of this same workload
                      (profiles/r03_conv_traffic.json, keyed by workload and kernel name; null when the workload was not profiled)
  roofline_kernels -- the same entry for every convolution kernel instance with >= 0.5 ms of GPU time per step
  roofline_family  -- the two families of round 1 (all forward + data-gradient launches / all weight-gradient launches)
  cpu_baseline     -- the oracle (torch-CPU restatement, "port") timed on this box's host cores on a bounded sample
  multi_stream     -- informational: the same iterations re-timed with conf.multi_stream (discriminator phases on concurrent HIP streams,
                      bit-identical results); NOT `value` -- time-shared kernels would misstate the per-kernel roofline entries
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16); never the 2:1-sparsity figure
DTYPE_NAME = {'f32': 'fp32', 'bf16': 'bf16 MFMA operands (fp32 accumulate)', 'f16': 'fp16 MFMA operands (fp32 accumulate)'}
TFLOP_PER_PAIR = {('film', 256): 1.630, ('spade', 256): 2.871, ('mmsdnet', 256): 1.469, ('mmsdnet', 320): 2.296}   # BASELINE.md section 4 (conv MACs x 2)


_FAMILY = {1: 'conv_fast_kernel', 2: 'conv_fwd_kernel', 3: 'conv_direct_kernel', 4: 'conv_fast_batched_kernel',
           5: 'conv_dgrad_s2k4_smallc_kernel', 6: 'conv_wgrad_tr_kernel', 7: 'conv_wgrad_fast_kernel', 8: 'conv_wgrad_kernel',
           9: 'conv_wgrad_c8m_kernel', 10: 'pw_reduce_kernel', 11: 'smallk_conv_kernel', 12: 'pw_reduce_wgrad_kernel',
           13: 'smallk_wgrad_kernel', 22: 's2k3c9_wgrad_kernel', 24: 'locnet5_wgrad_kernel'}
HBM_BOUND_FAMILIES = ('pw_reduce_kernel', 'smallk_conv_kernel', 'pw_reduce_wgrad_kernel', 'smallk_wgrad_kernel', 'conv_direct_kernel', 'conv_direct_mfma_kernel',
                      'conv_wgrad_c8m_kernel', 'conv_dgrad_s2k4_smallc_kernel', 'conv8h_kernel', 's2k3c9_fwd_kernel', 's2k3c9_dgrad_kernel',
                      's2k3c9_wgrad_kernel', 'locnet5_fwd_kernel')      # (locnet5_f32_kernel is MFMA-bound: priced against the fp32 peak)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~ 8 TB/s


def kernel_name(kid, prec):
    """mmseg_conv2d_last_kernel() id -> the kernel name rocprofv3 reports (template arguments included)"""
    fam, rest = kid // 1000000, kid % 1000000
    if fam == 18:              # <input-channel planes, output-channel planes> of the patch-resident fp32 weight gradient
        return 'wgrad32h_kernel<%d, %d>' % (rest // 1000, rest % 1000)
    if fam == 19:
        return 'wgrad16h_kernel<%d>' % prec
    if fam == 23:              # the localisation network's first layer in the 16-bit modes (csrc/s2conv.hpp)
        cin, cout = rest // 1000, rest % 1000
        if prec and cin == 16 and cout == 20:
            return 'locnet5_fwd_kernel<%d>' % prec          # (a padded launch of this shape would be the fp32 kernel; the models have none)
        return 'locnet5_f32_kernel<%s, %d, %d>' % ('8, 8' if cin == 16 else '20, 0', cout, prec)
    if fam == 21:              # the modality encoder's first layer (csrc/s2conv.hpp): N field 16 = forward, 9 = data gradient
        return 's2k3c9_fwd_kernel' if rest % 1000 == 16 else 's2k3c9_dgrad_kernel'
    if fam == 20:              # <precision, column tiles per wave, 16-bit input, 16-bit output, ReLU>: the M tile field carries the three flags
        fl = rest // 1000
        return 'conv8h_kernel<%d, 4, %s, %s, %s>' % (prec, 'true' if fl & 1 else 'false', 'true' if fl & 2 else 'false', 'true' if fl & 4 else 'false')
    if fam in (16, 17):        # the large-tile 16-bit kernels: plain <pixels per block> * 1000 + <N tile> (no flag fields)
        bm, bn = rest // 1000, rest % 1000
        if fam == 16:          # <M tile, N tile, waves M, waves N, ring stages, precision>
            return 'conv16_kernel<256, %d, %d, %d, %d, %d>' % (bn, 2 if bn == 256 else 4, 4 if bn == 256 else (2 if bn == 128 else 1),
                                                              2 if bn == 256 else 3, prec)
        rows = bm // 32        # <N tile, waves M, waves N, ring stages, precision, K tiles per barrier, image rows, patch buffers>
        wm_, wn_ = (8, 1) if bn == 64 else ((2, 4) if bn == 256 else (4, 2))
        return 'conv16h_kernel<%d, %d, %d, %d, %d, 1, %d, %d>' % (bn, wm_, wn_, 2 if bn == 256 else 3, prec, rows, 1 if rows == 16 else 2)
    flag, rest = rest // 500000, rest % 500000
    k64, rest = rest // 250000, rest % 250000
    bm, bn = rest // 1000, rest % 1000
    wm, wn = (4, 1) if bn == 32 else (2, 2)
    tf = 'true' if flag else 'false'
    if fam == 1:               # <M tile, N tile, waves M, waves N, precision, 16-bit input, K tile>
        return '%s<%d, %d, %d, %d, %d, %s, %d>' % (_FAMILY[fam], bm, bn, wm, wn, prec, tf, 64 if k64 else 32)
    if fam == 4:
        return '%s<%d, %d, %d, %d, %d, %s>' % (_FAMILY[fam], bm, bn, wm, wn, prec, tf)
    if fam == 7:
        return '%s<%d, %d, %d, %d, %d>' % (_FAMILY[fam], bm, bn, wm, wn, prec)
    if fam == 2:               # generic kernel: <..., 4-channel gather, precision> (16-bit MFMA variant only for forward launches)
        return '%s<%d, %d, %d, %d, %s, %d>' % (_FAMILY[fam], bm, bn, wm, wn, tf, prec if k64 else 0)
    if fam == 8:
        return '%s<%d, %d, %d, %d, %s>' % (_FAMILY[fam], bm, bn, wm, wn, tf)
    if fam == 6:
        return '%s<%d, %d, %d, %d, %d, %s>' % (_FAMILY[fam], bm, bn, wm, wn, prec, tf)
    if fam == 14:
        return 'conv_wgrad_tr_anyw_kernel<%d, %d, %d, %d, %d>' % (bm, bn, wm, wn, prec)
    if fam == 3:
        return 'conv_direct_mfma_kernel<3>'
    if fam == 5:
        return 'conv_dgrad_s2k4_smallc_kernel<%d>' % bm
    if fam in (10, 12):             # <lanes per pixel, outputs, 16-byte vectors per lane>
        return '%s<%d, %d, %d>' % (_FAMILY[fam], bm, bn, 2 if (bm, bn) == (8, 8) else 1)
    if fam in (11, 13):             # <kernel size, input channels, lanes per pixel>
        return '%s<%d, %d, %d>' % (_FAMILY[fam], bm // 20, bn, bm % 20)
    return _FAMILY.get(fam, 'kernel_%d' % kid)


class ConvTimer(object):
    """HIP-event timing of every convolution launch on the compute stream (torch's current stream IS the stream
    the kernels are launched on).  Events are only read after the timed region."""

    def __init__(self, stride=8):
        self.records = []     # (kind, flops, bytes, start_event, end_event, shape)
        self.enabled = False
        # Bracketing EVERY launch with events costs ~8 % of the step (the event packets serialise back-to-back kernels),
        # which would distort the very throughput being reported; every `stride`-th convolution launch is timed instead
        # (a uniform sample over the timed region; stride 7/8/9 does not alias with the launch pattern of a step).
        self.stride = max(1, int(stride))
        self.seen = 0
        self.counts = {}      # kind -> total launches in the timed region (sampled or not)
        self.kcounts = {}     # rocprof kernel name -> total launches in the timed region
        self.kwork = {}       # rocprof kernel name -> [flops, bytes] over ALL its launches while enabled (sampled or not)
        self.prec = 0

    def install(self):
        from multimodal_segmentation_amd import _native
        self._orig = _native.call
        timer = self

        def call(name, *args):
            full_name = name
            io = 0
            if name in ('mmseg_conv2d_fwd_t', 'mmseg_conv2d_fwd_scaled_t', 'mmseg_conv2d_wgrad_t'):
                name, io = name[:-2], args[-1]       # same leading arguments + the 16-bit storage bits (x1, x2, y / dy)
            e1, e2, e3 = (2.0 if io & 1 else 4.0), (2.0 if io & 2 else 4.0), (2.0 if io & 4 else 4.0)
            if timer.enabled and name in ('mmseg_conv2d_fwd', 'mmseg_conv2d_fwd_scaled', 'mmseg_conv2d_wgrad',
                                          'mmseg_conv2d_dgrad_parity', 'mmseg_conv2d_dgrad_parity_all', 'mmseg_conv8h_fwd_t'):
                if name == 'mmseg_conv8h_fwd_t':        # 3x3 'same' convolution of an 8-channel tensor (16-bit modes): x, w, bias, y, B, H, W, Cout, ..., hx, hy
                    (B, H, W, Cout) = args[4:8]
                    hx, hy = args[10], args[11]
                    flops = 2.0 * B * H * W * 72 * Cout
                    nbytes = (2.0 if hx else 4.0) * B * H * W * 8 + 4.0 * 72 * Cout + (2.0 if hy else 4.0) * B * H * W * Cout
                    kind = 'conv_fwd_kernel'
                    shape = ('fwd', B, H, W, 8, 0, Cout, 3, 3, '8h')
                elif name == 'mmseg_conv2d_dgrad_parity_all':
                    (B, Ho, Wo, Cout, H, W, Cin, KH, KW, stride) = args[3:13]
                    # the parity classes together touch every (output pixel, tap) pair of the strided convolution once
                    flops = 2.0 * B * Ho * Wo * KH * KW * Cin * Cout
                    nbytes = 4.0 * (B * Ho * Wo * Cout + KH * KW * Cin * Cout + B * H * W * Cin)
                    kind = 'conv_fwd_kernel'
                    shape = ('dgrad_parity_all', B, Ho, Wo, Cout, Cin, KH, KW, stride)
                elif name == 'mmseg_conv2d_dgrad_parity':
                    (B, Ho, Wo, Cout, H, W, Cin, TH, TW, stride, ph, pw) = args[3:15]
                    hs, ws = (H - ph + stride - 1) // stride, (W - pw + stride - 1) // stride
                    flops = 2.0 * B * hs * ws * Cin * TH * TW * Cout     # exact taps of this parity class
                    nbytes = 4.0 * (B * Ho * Wo * Cout + TH * TW * Cin * Cout + B * hs * ws * Cin)
                    kind = 'conv_fwd_kernel'
                    shape = ('dgrad_parity', B, Ho, Wo, Cout, Cin, TH, TW, stride)
                elif name == 'mmseg_conv2d_fwd_scaled':          # convolution + folded inference BatchNorm (+ReLU)
                    (B, H, W, C1, C2, Ho, Wo, Cout, KH, KW) = args[7:17]
                    ups = args[20]
                    flops = 2.0 * B * Ho * Wo * Cout * KH * KW * (C1 + C2)
                    nbytes = e1 * B * (H >> ups) * (W >> ups) * C1 + e2 * B * H * W * C2 + 4.0 * KH * KW * (C1 + C2) * Cout + e3 * B * Ho * Wo * Cout
                    kind = 'conv_fwd_kernel'
                    shape = ('fwd+bn', B, H, W, C1, C2, Cout, KH, KW, 'ups' if ups else '')
                elif name == 'mmseg_conv2d_fwd':
                    (B, H, W, C1, C2, Ho, Wo, Cout, KH, KW) = args[7:17]
                    transposed = args[21]
                    # algorithmic FLOPs: a fractionally-strided (data-gradient) launch only has the taps of the strided
                    # forward convolution it differentiates, i.e. B*H*W (its INPUT pixels) x KH*KW x Cin x Cout
                    pix = B * H * W if transposed else B * Ho * Wo
                    flops = 2.0 * pix * Cout * KH * KW * (C1 + C2)
                    ups = args[20]
                    # algorithmic bytes: every input element, weight and output element once
                    nbytes = e1 * B * (H >> ups) * (W >> ups) * C1 + e2 * B * H * W * C2 + 4.0 * KH * KW * (C1 + C2) * Cout + e3 * B * Ho * Wo * Cout
                    kind = 'conv_fwd_kernel'
                    shape = ('dgrad_T' if transposed else 'fwd', B, H, W, C1, C2, Cout, KH, KW, 'ups' if ups else '')
                else:
                    (B, H, W, C1, C2, Ho, Wo, Cout, KH, KW) = args[6:16]
                    flops = 2.0 * B * Ho * Wo * Cout * KH * KW * (C1 + C2)
                    ups = args[19]
                    nbytes = e1 * B * (H >> ups) * (W >> ups) * C1 + e2 * B * H * W * C2 + 4.0 * KH * KW * (C1 + C2) * Cout + e3 * B * Ho * Wo * Cout
                    kind = 'conv_wgrad_kernel'
                    shape = ('wgrad', B, H, W, C1, C2, Cout, KH, KW, 'ups' if ups else '')
                timer.counts[kind] = timer.counts.get(kind, 0) + 1
                timer.seen += 1
                sampled = timer.seen % timer.stride == 0
                if sampled:
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                rc = timer._orig(full_name, *args)
                kname = kernel_name(timer._orig('mmseg_conv2d_last_kernel'), timer.prec)     # which template instance ran
                timer.kcounts[kname] = timer.kcounts.get(kname, 0) + 1
                kw = timer.kwork.setdefault(kname, [0.0, 0.0])
                kw[0] += flops
                kw[1] += nbytes
                if sampled:
                    e.record()
                    timer.records.append((kind, flops, nbytes, s, e, shape, kname))
                return rc
            return timer._orig(full_name, *args)
        _native.call = call

    def breakdown(self, steps):
        """per-shape table (stderr): launches/step, ms/step, TFLOP/s -- where the convolution time of a step goes"""
        agg = {}
        for kind, flops, nbytes, s, e, shape, _k in self.records:
            d = agg.setdefault(shape, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e)
            d[2] += flops
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        sys.stderr.write('%-60s %8s %9s %9s\n' % ('shape', 'n/step', 'ms/step', 'TFLOP/s'))
        for shape, (n, ms, fl) in rows:
            sys.stderr.write('%-60s %8.1f %9.3f %9.1f\n' % (' '.join(str(v) for v in shape), n / steps, ms / steps,
                                                           fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0))

    def summary(self):
        out = {}
        for kind, flops, nbytes, s, e, _, kname in self.records:
            ms = s.elapsed_time(e)
            for key in (kind, ('kernel', kname)):
                d = out.setdefault(key, {'flops': 0.0, 'ms': 0.0, 'launches': 0, 'bytes': 0.0})
                d['flops'] += flops
                d['bytes'] += nbytes
                d['ms'] += ms
                d['launches'] += 1
        return out


def host_cores():
    """cores this process may actually use: the scheduler affinity (the GPU box exposes far more CPUs than its share), at
    most 16 (the box's share for one GPU)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline_measure(H, decoder, B):
    """The oracle ("port") on the host cores: one full DAFNet iteration (generator fit + 2 mask-D fits + 2 image-D
    fits incl. pools) at the benchmark's image size and batch (about 10-30 s of CPU work).  Runs in a child process (see
    cpu_baseline)."""
    from oracle import dafnet as OD, models as OM
    from tests import helpers as Hh
    cores = host_cores()
    torch.set_num_threads(cores)
    P = OM.build_dafnet_params(10, H, H, decoder)
    orc = OD.DAFNetOracle(P, dict(decoder_type=decoder))
    d = Hh.to_torch(Hh.make_step_data(B, H, H, seed=99), torch.float32)
    t0 = time.time()
    orc.train_batch(d, supervised=True)
    dt = time.time() - t0
    return {'value': B / dt, 'unit': 'paired slices/s', 'cores': cores, 'kind': 'port',
            'sample': 'one full DAFNet-%s iteration at %dx%d with batch %d (torch-CPU oracle, fp32, %d threads): %.1f s'
                      % (decoder, H, H, B, cores, dt)}


def cpu_baseline(H, decoder, B, limit_s=240):
    """Time the oracle in a CPU-only child process with a hard limit, so that a slow or oversubscribed host can never
    stall the benchmark line; on a timeout the baseline is reported as unmeasured."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', OMP_NUM_THREADS=str(host_cores()))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--size', str(H), '--decoder', decoder,
           '--batch', str(B)]
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=limit_s,
                             cwd=os.path.dirname(os.path.abspath(__file__)))
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except Exception as exc:       # timeout, non-zero exit, unparsable output
        return {'value': None, 'unit': 'paired slices/s', 'cores': host_cores(), 'kind': 'port',
                'sample': 'unmeasured: oracle iteration at %dx%d batch %d did not finish within %d s (%s)'
                          % (H, H, B, limit_s, type(exc).__name__)}


def launch_ranks(n, dry_run=False):
    """start `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a CHILD process (this parent has not
    touched the GPU and never replaces itself); -> the child's return code.  `dry_run`: print the command and the environment it
    would get as one JSON line instead of starting it (tests/test_bench_cli.py)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    argv = [a for a in sys.argv[1:] if a != '--dry-run']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):          # the launcher sets these for its ranks
        env.pop(k, None)
    if dry_run:
        print(json.dumps({'dry_run': True, 'cmd': cmd, 'cwd': ROOT,
                          'env': {k: env[k] for k in ('HSA_ENABLE_IPC_MODE_LEGACY',) if k in env}}))
        return 0
    return subprocess.call(cmd, env=env, cwd=ROOT)


_T0 = time.perf_counter()


def _progress(msg):
    """stage marker on stderr (stdout carries only the JSON line)"""
    sys.stderr.write('[bench %7.1fs] %s\n' % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--decoder', default='film', choices=['film', 'spade'])
    ap.add_argument('--model', default='dafnet', choices=['dafnet', 'mmsdnet'],
                    help="mmsdnet: the MMSDNet iteration (mmsdnet_executor.py:238-331) instead of the headline DAFNet one")
    ap.add_argument('--modalities', type=int, default=2, choices=[2, 3],
                    help='3 (with --model mmsdnet): the build-defined 3-modality MMSDNet of BASELINE config #5')
    ap.add_argument('--l_mix', type=float, default=1.0)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f16'],
                    help='bf16: bf16 MFMA operands (fp32 accumulation, fp32 tensors in HBM, fp32 weight gradients) -- configs #3/#5')
    ap.add_argument('--act16', action='store_true',
                    help='with --dtype bf16 | f16: the activations / gradients of the MFMA trunk live in HBM in the 16-bit type (conf.act_storage = half)')
    ap.add_argument('--conv16', type=int, default=1, choices=[0, 1, 2],
                    help='16-bit runs: which kernel multiplies the 64-channel-multiple layers (mmseg_conv16_mode): 0 register-staged 128-wide '
                         'kernel only, 1 (default) the 256-pixel direct-to-LDS kernels where they pay, 2 wherever they apply')
    ap.add_argument('--graphs', action='store_true',
                    help='conf.hip_graphs: every trainer step is recorded into a hipGraph after two eager steps and replayed (single GPU)')
    ap.add_argument('--multi-stream', action='store_true',
                    help='conf.multi_stream: the mask- and image-discriminator phases of an iteration on concurrent HIP streams (bit-identical '
                         'results, ~2.5 %% faster iteration; off by default because the elapsed time of a kernel that shares the GPU with '
                         "another stream's kernels no longer measures that kernel: the per-kernel roofline entries would read low)")
    ap.add_argument('--no-multi-stream', action='store_true',
                    help='the 16-bit modes (--dtype bf16 | f16) run with conf.multi_stream by default (the product default for reduced precision); this keeps them on one stream')
    ap.add_argument('--no-multi-stream-region', action='store_true', help='skip the informational second timed region with conf.multi_stream')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-conv-timer', action='store_true')
    ap.add_argument('--conv-breakdown', action='store_true', help='per-shape convolution table on stderr')
    ap.add_argument('--conv-timer-stride', type=int, default=7, help='time every n-th convolution launch with HIP events')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--dry-run', action='store_true', help='with --gpus N > 1: print the launcher command instead of starting the ranks')
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_measure(args.size, args.decoder, args.batch)))
        return

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` on its own: this parent has not touched the GPU (no torch.cuda call so far) and starts the
        # N ranks as a CHILD process (never exec from a process that may own a GPU context), relays their output and exits with
        # the child's return code
        sys.exit(launch_ranks(args.gpus, args.dry_run))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    # the data-parallel branch runs whenever there is more than one rank; MMSEG_BENCH_FORCE_DIST=1 sends a ONE-rank run (under the same
    # launcher) through it too -- RCCL process group, broadcast, gradient all-reduces, replica check -- which is how the 1-GPU test box
    # rehearses the multi-GPU run (tests/test_bench_cli.py)
    distributed = world > 1 or os.environ.get('MMSEG_BENCH_FORCE_DIST') == '1'
    if distributed:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from multimodal_segmentation_amd import nn, _native
    from multimodal_segmentation_amd.configuration import dafnet_config_chaos, dafnet_spade_config_chaos
    from multimodal_segmentation_amd.models.dafnet import DAFNet
    from multimodal_segmentation_amd.model_executors.dafnet_executor import DAFNetExecutor
    from multimodal_segmentation_amd.parallel import dp
    from multimodal_segmentation_amd.utils.config import EasyDict

    _native.load()
    nn.set_default_device('cuda:%d' % local_rank)
    if args.model == 'mmsdnet':
        from multimodal_segmentation_amd.configuration import mmsdnet_config_chaos, mmsdnet3_config_chaos
        cfg = (mmsdnet3_config_chaos if args.modalities == 3 else mmsdnet_config_chaos).get()
    else:
        cfg = (dafnet_config_chaos if args.decoder == 'film' else dafnet_spade_config_chaos).get()
    H = args.size
    cfg['input_shape'] = (H, H, 1)
    cfg['anatomy_encoder']['input_shape'] = (H, H, 1)
    cfg['anatomy_encoder']['output_shape'] = (H, H, 8)
    cfg['d_mask_params']['input_shape'] = (H, H, cfg['num_masks'])
    if 'd_image_params' in cfg:
        cfg['d_image_params']['input_shape'] = (H, H, 1)
    cfg['batch_size'] = args.batch
    cfg['l_mix'] = args.l_mix
    cfg['n_pairs'] = 1
    cfg['compute_dtype'] = {'f32': 'fp32', 'bf16': 'bf16', 'f16': 'fp16'}[args.dtype]
    if args.conv16 != 1:
        from multimodal_segmentation_amd import _native as _N
        _N.call('mmseg_conv16_mode', args.conv16)
    if args.act16:
        if args.dtype == 'f32':
            raise SystemExit('--act16 needs --dtype bf16 or f16')
        cfg['act_storage'] = 'half'
        DTYPE_NAME[args.dtype] = DTYPE_NAME[args.dtype].replace('(fp32 accumulate)', '(fp32 accumulate), 16-bit trunk activations in HBM')
    # conf.multi_stream: opt-in for fp32 (the headline keeps per-kernel times meaningful), the product's default in the
    # reduced-precision modes (their iterations are short enough for the discriminators' small launches to leave the GPU half empty)
    args.multi_stream = bool(args.multi_stream or (args.dtype != 'f32' and not args.no_multi_stream))
    cfg['multi_stream'] = args.multi_stream
    if args.multi_stream:
        DTYPE_NAME[args.dtype] += ', discriminator phases on concurrent streams'
    if args.graphs:
        cfg['hip_graphs'] = True
        # replayed launches do not pass through the Python call the timer hooks: the hooks only COUNT the launches and their algorithmic
        # work during the first (eager) warm-up iteration; the durations come from the committed rocprofv3 trace of this command
        args.conv_timer_stride = 1 << 30
        DTYPE_NAME[args.dtype] += ', trainer steps replayed from hipGraphs'
    cfg['folder'] = '/tmp/mmseg_bench'
    conf = EasyDict(cfg)

    _progress('building model')
    if args.model == 'mmsdnet':
        from multimodal_segmentation_amd.models.mmsdnet import MMSDNet
        from multimodal_segmentation_amd.model_executors.mmsdnet_executor import MMSDNetExecutor
        DAFNet, DAFNetExecutor = MMSDNet, MMSDNetExecutor
    model = DAFNet(conf)
    model.build()
    dp.enable(distributed, force=distributed and world == 1)        # one forced rank: the all-reduces still run (identity)
    if distributed:
        all_models = model._generator_models() + [d for d in (model.D_Mask, getattr(model, 'D_Image1', None),
                                                             getattr(model, 'D_Image2', None)) if d is not None]
        dp.broadcast_models(all_models)
    _progress('building executor + data')
    ex = DAFNetExecutor(conf, model)
    ex.keep_losses_on_device = True
    # synthetic volumes: the smallest number of slices per volume for which every generator (14 labelled volumes; 2 x 14 for
    # the real masks) yields FULL batches only -- a short last batch of a pass would make some discriminator steps cheaper
    # than the workload's
    n_lab = int(round(args.l_mix * 14))
    n_ul = 14 - n_lab if args.l_mix < 1 else 0
    sizes = [n for n in (n_lab, n_ul, 2 * n_lab + n_ul, 14) if n > 0]      # labelled, unlabelled, real masks, all images
    spv = next(s for s in range(2, 2 + 8 * args.batch) if all((n * s) % args.batch == 0 for n in sizes))
    ex.init_train_data(device_resident=True, slices_per_volume=spv)

    timer = ConvTimer(1 if args.conv_breakdown else args.conv_timer_stride)
    timer.prec = {'f32': 0, 'bf16': 1, 'f16': 2}[args.dtype]
    if not args.no_conv_timer:
        timer.install()

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    _progress('warmup')
    losses = {n: [] for n in ex.get_loss_names()}
    for i in range(args.warmup):
        timer.enabled = bool(args.graphs) and i == 0          # --graphs: count one eager iteration's launches (see above)
        ex.train_batch(losses)
    sync()
    graph_counts = (dict(timer.kcounts), {k: list(v) for k, v in timer.kwork.items()}) if args.graphs else None
    _progress('timed region')
    timer.enabled = not args.graphs
    dp.counters(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ex.train_batch(losses)
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False
    _progress('timed region done: %.1f ms/step' % (1000.0 * dt / args.steps))
    if args.conv_breakdown and rank == 0:
        torch.cuda.synchronize()
        timer.breakdown(args.steps)
    dp_counters = dp.counters()
    per_rank_ms = [1000.0 * dt / args.steps]
    replicas_ok = None
    if distributed:
        mine = torch.tensor([dt], device='cuda')
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [1000.0 * float(t.item()) / args.steps for t in allt]
        tmax = torch.tensor([dt], device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # self-verification of a multi-GPU run: after the timed iterations every rank must hold bit-identical weights (the gradient
        # all-reduce is the only thing that keeps replicas together; a wrong or missing collective shows here, not in the rate)
        replicas_ok, checksum = dp.replicas_identical(all_models)
        if not replicas_ok:          # reported in the JSON line (`dp.replicas_identical_after_timed_region`), loudly, but the measurement is kept
            sys.stderr.write('bench.py: rank %d: WARNING -- the weight replicas DIVERGED during the timed region (checksums %s)\n' % (rank, checksum[:6]))

    passes = (1 if args.l_mix > 0 else 0) + (1 if args.l_mix < 1 else 0)
    pairs = world * args.batch * args.steps * passes
    value = pairs / dt
    # informational second region (not `value`): the same iterations with conf.multi_stream -- the discriminator phases on concurrent
    # HIP streams, bit-identical results.  Kept out of the headline because per-kernel elapsed times of time-shared kernels would
    # misstate the `roofline` entries.
    multi = None
    if args.model == 'dafnet' and not args.multi_stream and not args.graphs and not args.no_multi_stream_region and not distributed:
        conf['multi_stream'] = True
        for _ in range(2):
            ex.train_batch(losses)
        sync()
        k2 = min(args.steps, 10)
        t1 = time.perf_counter()
        for _ in range(k2):
            ex.train_batch(losses)
        sync()
        dt2 = time.perf_counter() - t1
        conf['multi_stream'] = False
        if world > 1:
            tmax = torch.tensor([dt2], device='cuda')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt2 = float(tmax.item())
        multi = {'value': world * args.batch * k2 * passes / dt2, 'unit': 'paired slices/s', 'ms_per_step': 1000.0 * dt2 / k2, 'steps': k2,
                 'note': 'conf.multi_stream: mask- and image-discriminator phases on concurrent HIP streams; same results bit for bit'}
    line = {
        'metric': '2D slices/sec %s train step, %dx%dx2-modality bs=%d/GPU' % ('DAFNet' if args.model == 'dafnet' else 'MMSDNet',
                                                                                H, H, args.batch),
        'value': value, 'unit': 'paired slices/s', 'n_gpus': (dist.get_world_size() if distributed else 1),
        'rccl_ranks': (dist.get_world_size() if distributed else 0), 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1000.0 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'DAFNet-%s (dafnet%s_config_chaos) %dx%d 2-modality training iteration: generator fit + '
                               '2 mask-D fits + 2 image-D fits incl. fake pools, bs=%d/GPU, %s, l_mix=%g'
                               % (args.decoder, '' if args.decoder == 'film' else '_spade', H, H, args.batch, DTYPE_NAME[args.dtype], args.l_mix),
                   'global_batch': world * args.batch, 'parallelism': 'dp%d' % world},
    }
    if args.model == 'mmsdnet':
        line['config']['workload'] = ('MMSDNet (mmsdnet%s_config_chaos) %dx%d %d-modality training iteration: generator fit + '
                                      'Z-regressor fit + mask-D fit incl. the fake pool, bs=%d/GPU, %s, l_mix=%g'
                                      % ('3' if args.modalities == 3 else '', H, H, args.modalities, args.batch,
                                         DTYPE_NAME[args.dtype], args.l_mix))
        line['metric'] = line['metric'].replace('x2-modality', 'x%d-modality' % args.modalities)
    key = (args.decoder if args.model == 'dafnet' else ('mmsdnet' if args.modalities == 2 else 'mmsdnet3'), H)
    if key in TFLOP_PER_PAIR:
        line['conv_tflops_whole_step'] = TFLOP_PER_PAIR[key] * value / world
        # against the MFMA peak of the dtype the convolutions multiply in (fp32: 157.3; bf16 / fp16: 2500 TFLOP/s)
        line['conv_roofline_frac_whole_step'] = TFLOP_PER_PAIR[key] * value / world / (
            BF16_MFMA_PEAK_TFLOPS if args.dtype in ('bf16', 'f16') else FP32_MFMA_PEAK_TFLOPS)
    line['per_rank_ms_per_step'] = per_rank_ms
    if distributed:
        st = max(dp_counters['steps'], 1)
        line['dp'] = {'trainer_steps': dp_counters['steps'], 'collectives_per_iteration': dp_counters['collectives'] / float(args.steps),
                      'overlapped_with_backward_per_iteration': dp_counters['overlapped'] / float(args.steps),
                      'replicas_identical_after_timed_region': replicas_ok,
                      'note': 'gradient all-reduces per iteration (one per component arena and trainer step) and how many of them were '
                              'issued while the backward pass was still being queued; %d trainer steps timed' % st}
    if rank == 0:
        summ = timer.summary() if not args.no_conv_timer else {}
        # HBM bytes per launch from rocprofv3 PMC passes of THIS workload (tools/pmc_traffic.py), keyed by workload and kernel
        wkey = '%s-%s-%d-bs%d-%s-lmix%g' % (args.model if args.model != 'mmsdnet' or args.modalities == 2 else 'mmsdnet3',
                                          args.decoder, H, args.batch, args.dtype + ('-act16' if args.act16 else ''), args.l_mix)
        traffic = {}
        tname = 'r02_conv_traffic.json'
        for cand in ('r04_conv_traffic.json', 'r03_conv_traffic.json', 'r02_conv_traffic.json'):      # the latest round's PMC passes that cover this workload
            tpath = os.path.join(ROOT, 'profiles', cand)
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get('workloads', {}).get(wkey, {})
                if traffic:
                    tname = cand
                    break
        peak = BF16_MFMA_PEAK_TFLOPS if args.dtype in ('bf16', 'f16') else FP32_MFMA_PEAK_TFLOPS
        prec_note = '%s MFMA operands (%s, fp32 accumulation)' % (args.dtype, '16-bit trunk activations + fp32 elsewhere in HBM' if args.act16
                                                                     else 'fp32 tensors in HBM') if args.dtype != 'f32' else 'fp32 MFMA'

        def entry(k, launches, label):
            ach = k['flops'] / (k['ms'] * 1e-3) / 1e12
            avg = k['ms'] / k['launches']
            if label.split('<')[0] in HBM_BOUND_FAMILIES:
                # the small-channel layers are bandwidth kernels: algorithmic bytes (every input, weight and output element once)
                # over the launch time against the HBM peak
                gbs = k['bytes'] / (k['ms'] * 1e-3) / 1e9
                return {'bound': 'hbm', 'kernel': label, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                        'traffic': None, 'algorithmic_bytes_per_launch': k['bytes'] / k['launches'],
                        'flops_per_launch': k['flops'] / k['launches'], 'launches': launches, 'timed_launches': k['launches'],
                        'avg_launch_ms': avg, 'gpu_ms_per_step': avg * launches / args.steps}
            return {'bound': 'mfma', 'kernel': label, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                    'traffic': None, 'algorithmic_bytes_per_launch': k['bytes'] / k['launches'],
                    'flops_per_launch': k['flops'] / k['launches'], 'launches': launches, 'timed_launches': k['launches'],
                    'avg_launch_ms': avg, 'gpu_ms_per_step': avg * launches / args.steps}
        per_kernel = []
        for key, k in summ.items():
            if isinstance(key, tuple) and k['ms'] > 0:
                ent = entry(k, timer.kcounts.get(key[1], k['launches']), key[1])
                ent['traffic'] = traffic.get('kernels', {}).get(key[1], {}).get('hbm_bytes_per_launch')
                per_kernel.append(ent)
        per_kernel.sort(key=lambda e: -e['gpu_ms_per_step'])
        if per_kernel:
            # the dominant kernel instance of the step (largest GPU time), named as rocprofv3 names it
            line['roofline'] = dict(per_kernel[0], precision=prec_note,
                                    sampling='every %d-th convolution launch of the timed region bracketed by HIP events' % timer.stride,
                                    traffic_source=('profiles/%s[%s]' % (tname, wkey)) if per_kernel[0]['traffic'] else None)
            line['roofline_kernels'] = [e for e in per_kernel if e['gpu_ms_per_step'] >= 0.5 or e['bound'] == 'hbm']
        fam = {}
        for kind, label in (('conv_fwd_kernel', 'forward + data-gradient convolution launches (all template instances)'),
                            ('conv_wgrad_kernel', 'weight-gradient launches incl. the slab reduction (all template instances)')):
            k = summ.get(kind)
            if k:
                fam[kind] = entry(k, timer.counts.get(kind, k['launches']), label)
                fam[kind]['traffic'] = traffic.get('families', {}).get('conv_fwd' if kind == 'conv_fwd_kernel' else 'conv_wgrad', {}).get('hbm_bytes_per_launch')
        if fam:
            line['roofline_family'] = fam
        if args.graphs and graph_counts and graph_counts[0]:
            # --graphs: no per-launch events inside a replayed graph.  Per kernel instance: the algorithmic work of its launches in one
            # (eager, counted) iteration / the average duration of that instance in the committed rocprofv3 kernel trace of THIS command
            gname = 'r03_graphs_kernel_stats_bench_%s_%s_%d_bs%d_%s.txt' % (args.model, args.decoder, H, args.batch, args.dtype + ('_act16' if args.act16 else ''))
            gpath = os.path.join(ROOT, 'profiles', gname)
            if os.path.exists(gpath):
                import re
                avg_us = {}
                for ln in open(gpath):
                    m = re.match(r'(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', ln)
                    if m:
                        avg_us[m.group(1).strip()] = float(m.group(4))
                ents = []
                for kname, n in graph_counts[0].items():
                    if kname in avg_us and n > 0:
                        fl, by = graph_counts[1][kname]
                        k = {'flops': fl, 'bytes': by, 'ms': avg_us[kname] * 1e-3 * n, 'launches': n}
                        ent = entry(k, n * args.steps, kname)
                        ent['traffic'] = traffic.get('kernels', {}).get(kname, {}).get('hbm_bytes_per_launch')
                        ents.append(ent)
                ents.sort(key=lambda e: -e['gpu_ms_per_step'])
                if ents:
                    line['roofline'] = dict(ents[0], precision=prec_note, sampling='launches and algorithmic work counted in one eager iteration; '
                                            'average launch duration from profiles/%s (rocprofv3 --kernel-trace of this command: replayed graphs)' % gname)
                    line['roofline_kernels'] = [e for e in ents if e['gpu_ms_per_step'] >= 0.5 or e['bound'] == 'hbm']
        if multi is not None:
            line['multi_stream'] = multi
        # share of the timed window with a kernel running, from the rocprofv3 kernel trace of THIS command committed under profiles/
        # (tools/gpu_busy.py; like `traffic` it cannot be measured from inside the process without timing every launch)
        # Numbers that are NOT measured by this run but read from rocprofv3 summaries committed under profiles/ (a kernel trace / PMC
        # passes of this same command, taken by the builder): kept apart from the live measurements under one key, with their files
        # (advisor, round 3).  `roofline.traffic` is the one such number the line's contract places inside `roofline`: its file is
        # named in `roofline.traffic_source` and repeated here.
        committed = {}
        for rnd in ('r04', 'r03'):
            bpath = os.path.join(ROOT, 'profiles', '%s_gpu_busy_bench_%s_%s_%d_bs%d.txt' % (rnd, args.model, args.decoder, H, args.batch))
            if args.dtype == 'f32' and args.l_mix == 1.0 and not args.graphs and not args.multi_stream and os.path.exists(bpath):
                import re
                m = re.search(r'= ([0-9.]+) % GPU-busy', open(bpath).read())
                if m:
                    committed['gpu_busy_frac'] = {'value': float(m.group(1)) / 100.0, 'source': os.path.relpath(bpath, ROOT)}
                    break
        if line.get('roofline', {}).get('traffic'):
            committed['hbm_traffic_per_launch'] = {'source': 'profiles/%s[%s]' % (tname, wkey),
                                                   'note': 'rocprofv3 --pmc FETCH_SIZE (x 2 on gfx950) and WRITE_SIZE passes, tools/pmc_traffic.py'}
        if committed:
            committed['note'] = 'from committed profiles of this command, not measured in this run'
            line['from_committed_profile'] = committed
        if world == 1 and not args.no_cpu_baseline and args.model == 'dafnet':
            _progress('cpu baseline (oracle, bounded sample)')
            line['cpu_baseline'] = cpu_baseline(H, args.decoder, args.batch)
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
