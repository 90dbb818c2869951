#!/usr/bin/env python
"""Isolated timings of the HBM-bound small-channel convolution kernels (csrc/smallconv.hpp) at the BASELINE geometry: algorithmic
bytes (inputs + weights + outputs once) / time, forward and weight gradient, beside the generic kernels (MMSEG_SMALLCONV=0 in a
second process).

    python tools/smallconv_bench.py [B=8] [H=256]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3      # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = 'cuda'
    N.load()
    cases = [  # (name, B mult, C1, Cout, k, stride, pad)
        ('anatomy head 64->8 1x1', 1, 64, 8, 1, 1, 0), ('segmentor head 64->5 1x1', 1, 64, 5, 1, 1, 0),
        ('decoder head 8->1 1x1 (6 calls batched)', 6, 8, 1, 1, 1, 0), ('dgrad 5->64', 1, 5, 64, 1, 1, 0), ('dgrad 8->64', 1, 8, 64, 1, 1, 0),
        ('dgrad 1->8', 6, 1, 8, 1, 1, 0), ('UNet first 1->64 3x3', 1, 1, 64, 3, 1, 1), ('D_Image first 1->64 4x4 s2', 2, 1, 64, 4, 2, 0)]
    print('%-42s %10s %10s %10s %10s' % ('layer', 'fwd us', 'fwd GB/s', 'wgrad us', 'wgrad GB/s'))
    for name, bm, C1, Cout, k, stride, pad in cases:
        b = B * bm
        Ho = (H + 2 * pad - k) // stride + 1
        x = torch.randn(b, H, H, C1, device=dev)
        w = torch.randn(k, k, C1, Cout, device=dev) * 0.1
        bias = torch.randn(Cout, device=dev)
        y = torch.empty(b, Ho, Ho, Cout, device=dev)
        f = lambda: N.call('mmseg_conv2d_fwd', x, None, w, None, bias, y, None, b, H, H, C1, 0, Ho, Ho, Cout, k, k, stride, pad, pad, 0, 0, 1, 0.0, 0)
        tf = bench(f)
        nbytes = 4.0 * (x.numel() + w.numel() + y.numel())
        need = N.call('mmseg_conv2d_wgrad_workspace', b, Ho, Ho, C1, Cout, k, k)
        ws = torch.empty(max(need, 1024), device=dev)
        dw = torch.zeros_like(w)
        g = lambda: N.call('mmseg_conv2d_wgrad', x, None, y, dw.view(-1), ws, ws.numel(), b, H, H, C1, 0, Ho, Ho, Cout, k, k, stride, pad, pad, 0, 1)
        tg = bench(g)
        print('%-42s %10.1f %10.0f %10.1f %10.0f   (kernel id %d)' % (name, tf, nbytes / tf / 1e3, tg, nbytes / tg / 1e3, N.call('mmseg_conv2d_last_kernel')))


if __name__ == '__main__':
    main()
