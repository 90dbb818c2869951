#!/usr/bin/env python
"""A/B of the fp32 weight-gradient kernels on the UNet's 3x3 layer shapes: conv_wgrad_tr_kernel (mmseg_conv16_mode 0) vs wgrad32h_kernel (mode 2),
interleaved in one process on random data; TFLOP/s incl. the slab reduction, fraction of the 157.3 TFLOP/s fp32 MFMA peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N, ops as P

SHAPES = [  # B, H, C1, C2, Cout, ups
    (8, 256, 64, 0, 64, 0), (8, 256, 64, 64, 64, 0), (8, 256, 128, 0, 64, 1), (8, 128, 64, 0, 128, 0), (8, 128, 128, 0, 128, 0),
    (8, 128, 128, 128, 128, 0), (8, 128, 256, 0, 128, 1), (8, 64, 128, 0, 256, 0), (8, 64, 256, 0, 256, 0), (8, 64, 256, 256, 256, 0),
    (8, 64, 512, 0, 256, 1), (8, 32, 256, 0, 512, 0), (8, 32, 512, 0, 512, 0), (8, 32, 512, 512, 512, 0), (32, 256, 64, 0, 64, 0),
    (48, 128, 128, 0, 64, 0), (48, 64, 128, 0, 128, 0), (48, 32, 128, 0, 256, 0),
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
    h16 = os.environ.get('DTYPE') == 'bf16'        # DTYPE=bf16: wgrad16h_kernel vs the 16-bit instance of conv_wgrad_tr_kernel (16-bit tensors)
    P.set_conv_precision('bf16' if h16 else 'fp32')
    dev = torch.device('cuda')
    print('%-30s %9s %9s %7s %7s' % ('B,H,C1,C2,Cout,ups', 'tr TF', 'h TF', 'frac', 'speedup'))
    for (B, H, C1, C2, Cout, ups) in SHAPES:
        H1 = H // 2 if ups else H
        x1 = torch.randn(B, H1, H1, C1, device=dev)
        x2 = torch.randn(B, H, H, C2, device=dev) if C2 else None
        dy = torch.randn(B, H, H, Cout, device=dev)
        if h16:
            x1, dy = x1.to(torch.bfloat16), dy.to(torch.bfloat16)
            x2 = x2.to(torch.bfloat16) if C2 else None
        Cin = C1 + C2
        dw = torch.zeros(3, 3, Cin, Cout, device=dev)
        need = N.call('mmseg_conv2d_wgrad_workspace', B, H, H, Cin, Cout, 3, 3)
        ws = torch.empty(max(need, 1), device=dev)
        if h16:
            fn = lambda: N.call('mmseg_conv2d_wgrad_t', x1, x2, dy, dw.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout, 3, 3, 1, 1, 1, ups, 1, 5)
        else:
            fn = lambda: N.call('mmseg_conv2d_wgrad', x1, x2, dy, dw.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout, 3, 3, 1, 1, 1, ups, 1)
        flops = 2.0 * B * H * H * Cin * Cout * 9
        res = {0: [], 2: []}
        for m in (0, 2):
            N.call('mmseg_conv16_mode', m)
            fn()
        torch.cuda.synchronize()
        for _ in range(3):
            for m in (0, 2):
                N.call('mmseg_conv16_mode', m)
                res[m].append(timeit(fn, 10))
        N.call('mmseg_conv16_mode', 1)
        fn()
        fam = N.call('mmseg_conv2d_last_kernel')
        t0, t2 = min(res[0]), min(res[2])
        print('%-30s %9.1f %9.1f %7.3f %7.2fx   auto->%d' % (str((B, H, C1, C2, Cout, ups)), flops / t0 / 1e9, flops / t2 / 1e9, flops / t2 / 1e9 / (2500.0 if h16 else 157.3),
                                                         t0 / t2, fam))


if __name__ == '__main__':
    main()
