#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM convolution kernels on the UNet layer shapes of the benchmark workload
(B=8, 256x256).  Prints TFLOP/s per shape for forward, data-gradient and weight-gradient launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N

SHAPES = [  # H, Cin, Cout, k
    (256, 64, 64, 3), (256, 128, 64, 3), (128, 64, 128, 3), (128, 128, 128, 3), (128, 256, 128, 3),
    (64, 128, 256, 3), (64, 256, 256, 3), (64, 512, 256, 3), (32, 256, 512, 3), (32, 512, 512, 3),
    (32, 1024, 512, 3), (16, 512, 1024, 3), (16, 1024, 1024, 3), (256, 8, 64, 3), (256, 8, 8, 3), (256, 64, 8, 1),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(os.environ.get('B', 8))
    dev = torch.device('cuda')
    N.load()
    print('%-24s %10s %10s %10s   (TFLOP/s; ms)' % ('shape', 'fwd', 'dgrad', 'wgrad'))
    for H, Cin, Cout, k in SHAPES:
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        y = torch.empty(B, H, H, Cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        wt = torch.empty(k, k, Cout, Cin, device=dev)
        need = N.call('mmseg_conv2d_wgrad_workspace', B, H, H, Cin, Cout, k, k)
        ws = torch.empty(max(need, 1), device=dev)
        p = k // 2
        flops = 2.0 * B * H * H * Cin * Cout * k * k
        wp = torch.empty_like(wt)
        N.call('mmseg_conv2d_wprep', w, wp, k, k, Cin, Cout, 0)
        t_f = timeit(lambda: N.call('mmseg_conv2d_fwd', x, None, w, wp, b, y, None, B, H, H, Cin, 0, H, H, Cout, k, k, 1, p, p, 0, 0, 1, 0.0, 0))
        N.call('mmseg_conv2d_wflip', w, wt, k, k, Cin, Cout)
        wp2 = torch.empty_like(wt)
        N.call('mmseg_conv2d_wprep', w, wp2, k, k, Cin, Cout, 1)
        t_d = timeit(lambda: N.call('mmseg_conv2d_fwd', y, None, wt, wp2, None, dx, None, B, H, H, Cout, 0, H, H, Cin, k, k, 1, p, p, 0, 0, 0, 0.0, 0))
        t_w = timeit(lambda: N.call('mmseg_conv2d_wgrad', x, None, y, dw, ws, ws.numel(), B, H, H, Cin, 0, H, H, Cout, k, k, 1, p, p, 0, 0))
        if os.environ.get('CONV_BENCH_COLS') == 'wgrad':
            print('%d^2_%d->%d_k%d %6.1f %6.3f' % (H, Cin, Cout, k, flops / t_w / 1e9, t_w))
            continue
        print('%4d^2 %4d->%4d k%d      %6.1f %5.2f %6.1f %5.2f %6.1f %5.2f' %
              (H, Cin, Cout, k, flops / t_f / 1e9, t_f, flops / t_d / 1e9, t_d, flops / t_w / 1e9, t_w))


if __name__ == '__main__':
    main()
