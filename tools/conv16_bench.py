#!/usr/bin/env python
"""A/B of the two 16-bit convolution kernels on the layer shapes of BASELINE configs #3 / #5 (16-bit tensors in HBM): conv_fast_kernel
(mmseg_conv16_mode 0) vs conv16_kernel (mode 2), interleaved in ONE process on random data.  Prints TFLOP/s and the fraction of the
2.5 PFLOP/s dense bf16 peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N, ops as P

SHAPES = [  # B, H, C1, C2, Cout, k, ups
    (8, 256, 64, 0, 64, 3, 0), (8, 256, 64, 64, 64, 3, 0), (8, 128, 64, 0, 128, 3, 0), (8, 128, 128, 0, 128, 3, 0),
    (8, 128, 128, 128, 128, 3, 0), (8, 64, 128, 0, 256, 3, 0), (8, 64, 256, 0, 256, 3, 0), (8, 64, 256, 256, 256, 3, 0),
    (8, 32, 256, 0, 512, 3, 0), (8, 32, 512, 0, 512, 3, 0), (8, 32, 1024, 0, 512, 3, 1), (8, 16, 512, 0, 1024, 3, 0),
    (8, 16, 1024, 0, 1024, 3, 0), (8, 256, 128, 0, 64, 3, 1),
    (48, 256, 128, 0, 32, 3, 0), (48, 128, 128, 0, 64, 3, 0), (48, 64, 128, 0, 128, 3, 0), (48, 32, 128, 0, 256, 3, 0),
    (48, 64, 128, 0, 128, 3, 0), (48, 128, 64, 0, 128, 3, 0),
]


def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    f32 = os.environ.get('DTYPE') == 'f32'          # DTYPE=f32: the fp32 instances of the patch-resident kernel vs conv_fast_kernel (fp32 MFMA)
    dt = torch.float32 if f32 else torch.bfloat16
    P.set_conv_precision('fp32' if f32 else 'bf16')
    dev = torch.device('cuda')
    only = os.environ.get('ONLY')
    print('%-34s %9s %9s %7s %7s' % ('B,H,C1,C2,Cout,k,ups', 'fast TF', 'conv16 TF', 'frac', 'speedup'))
    for (B, H, C1, C2, Cout, k, ups) in SHAPES:
        if only and str(Cout) != only:
            continue
        H1 = H // 2 if ups else H
        x1 = torch.randn(B, H1, H1, C1, device=dev).to(dt)
        x2 = torch.randn(B, H, H, C2, device=dev).to(dt) if C2 else None
        Cin = C1 + C2
        w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        wp = torch.empty(w.numel(), device=dev)
        N.call('mmseg_conv2d_wprep', w, wp, k, k, Cin, Cout, 0)
        y = torch.empty(B, H, H, Cout, device=dev, dtype=dt)
        p = k // 2
        io = 1 | (2 if C2 else 0) | 4
        if f32:
            fn = lambda: N.call('mmseg_conv2d_fwd', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, p, p, ups, 0, 1, 0.0, 0)
        else:
            fn = lambda: N.call('mmseg_conv2d_fwd_t', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, p, p, ups, 0, 1, 0.0, 0, io)
        flops = 2.0 * B * H * H * Cin * Cout * k * k
        res = {0: [], 2: []}
        for m in (0, 2):
            N.call('mmseg_conv16_mode', m)
            fn()
        torch.cuda.synchronize()
        for rnd_ in range(3):
            for m in (0, 2):
                N.call('mmseg_conv16_mode', m)
                res[m].append(timeit(fn, 10))
        N.call('mmseg_conv16_mode', 1)
        fn()
        fam = N.call('mmseg_conv2d_last_kernel')
        t0, t2 = min(res[0]), min(res[2])
        print('%-34s %9.1f %9.1f %7.3f %7.2fx   auto->%d' % (str((B, H, C1, C2, Cout, k, ups)), flops / t0 / 1e9, flops / t2 / 1e9,
                                                         flops / t2 / 1e9 / (157.3 if f32 else 2500.0), t0 / t2, fam))


if __name__ == '__main__':
    main()
