#!/usr/bin/env python
"""Forward / data-gradient convolution launches on the 64-output-channel layers at 256 x 256 (the 4096-tile launches) and a few
others: time and TFLOP/s; CHECK=1 compares with torch's convolution."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N

SHAPES = [  # name, H, C1, C2, Cout, ups
    ('d0b/u0cb/seg.c1', 256, 64, 0, 64, 0), ('u0ca', 256, 64, 64, 64, 0), ('u0', 256, 128, 0, 64, 1), ('dgrad d1a', 128, 128, 0, 64, 0),
    ('d1b', 128, 128, 0, 128, 0), ('d2b', 64, 256, 0, 256, 0), ('d3b', 32, 512, 0, 512, 0), ('bott.b', 16, 1024, 0, 1024, 0),
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
    B = 8
    check = os.environ.get('CHECK', '0') == '1'
    dev = torch.device('cuda')
    N.load()
    # DTYPE=bf16|fp16 selects the reduced-precision mode; IO=<bits> stores x (1 | 2) and / or y (4) as 16-bit tensors (mmseg_conv2d_fwd_t)
    dt = os.environ.get('DTYPE', 'fp32')
    io = int(os.environ.get('IO', '0'))
    from multimodal_segmentation_amd import ops as P
    P.set_conv_precision(dt)
    half = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(dt)
    for name, H, C1, C2, Cout, ups in SHAPES:
        g = torch.Generator().manual_seed(2)
        H1 = H // 2 if ups else H
        x1 = torch.randn(B, H1, H1, C1, generator=g).to(dev)
        x2 = torch.randn(B, H, H, C2, generator=g).to(dev) if C2 else None
        Cin = C1 + C2
        w = (torch.randn(3, 3, Cin, Cout, generator=g) * 0.05).to(dev)
        b = torch.randn(Cout, generator=g).to(dev)
        y = torch.empty(B, H, H, Cout, device=dev)
        wp = torch.empty(w.numel(), device=dev)
        N.call('mmseg_conv2d_wprep', w, wp, 3, 3, Cin, Cout, 0)
        run = lambda: N.call('mmseg_conv2d_fwd', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, 3, 3, 1, 1, 1, ups, 0, 1, 0.0, 0)
        if io:
            xa = x1.to(half) if io & 1 else x1
            xb = (x2.to(half) if io & 1 else x2) if C2 else None
            ya = torch.empty(B, H, H, Cout, device=dev, dtype=half if io & 4 else torch.float32)
            run = lambda: N.call('mmseg_conv2d_fwd_t', xa, xb, w, wp, b, ya, None, B, H, H, C1, C2, H, H, Cout, 3, 3, 1, 1, 1, ups, 0, 1, 0.0, 0,
                                 (1 if io & 1 else 0) | (2 if (io & 1 and C2) else 0) | (io & 4))
        t = timeit(run)
        flops = 2.0 * B * H * H * Cin * Cout * 9
        err = ''
        if check:
            xin = x1.repeat_interleave(2, 1).repeat_interleave(2, 2) if ups else x1
            if C2:
                xin = torch.cat([xin, x2], -1)
            ref = torch.relu(torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), b.double(),
                                                        padding=1)).permute(0, 2, 3, 1)
            err = '%.2e' % float((y.double() - ref).abs().max() / ref.abs().max())
        print('%-18s %4d %12s %8.3f ms %7.1f TFLOP/s %s' % (name, H, '%d+%d->%d%s' % (C1, C2, Cout, ' ups' if ups else ''), t, flops / t / 1e9, err))


if __name__ == '__main__':
    main()
