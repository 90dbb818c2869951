#!/usr/bin/env python
"""Weight-gradient kernels on every 3x3 layer shape of the DAFNet UNet / segmentor at the benchmark size (B = 8, 256 x 256):
time, TFLOP/s and -- with CHECK=1 -- the deviation from torch's own convolution weight gradient.

    python tools/wgrad_bench.py                 # table
    MMSEG_WGRAD_TR=0 python tools/wgrad_bench.py   # the round-1 kernel (pixel-major LDS tiles, many slabs)
    MMSEG_WGRAD_TR_S=<n> ...                    # force the number of slabs of the transposed-staging kernel
    DTYPE=bf16|fp16 [IO=<bits>] ...             # reduced-precision products; IO: x (1|2) and / or dy (4) stored 16-bit (mmseg_conv2d_wgrad_t)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N

# name, H (conv resolution), C1, C2, Cout, ups
SHAPES = [
    ('d0b/u0cb/seg.c1', 256, 64, 0, 64, 0), ('u0ca', 256, 64, 64, 64, 0), ('u0', 256, 128, 0, 64, 1),
    ('d1a', 128, 64, 0, 128, 0), ('d1b/u1cb', 128, 128, 0, 128, 0), ('u1ca', 128, 128, 128, 128, 0), ('u1', 128, 256, 0, 128, 1),
    ('d2a', 64, 128, 0, 256, 0), ('d2b/u2cb', 64, 256, 0, 256, 0), ('u2ca', 64, 256, 256, 256, 0), ('u2', 64, 512, 0, 256, 1),
    ('d3a', 32, 256, 0, 512, 0), ('d3b/u3cb', 32, 512, 0, 512, 0), ('u3ca', 32, 512, 512, 512, 0), ('u3', 32, 1024, 0, 512, 1),
    ('bott.a', 16, 512, 0, 1024, 0), ('bott.b', 16, 1024, 0, 1024, 0),
    ('spade gamma', 256, 128, 0, 32, 0), ('spade c0', 256, 32, 0, 16, 0),
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
    check = os.environ.get('CHECK', '0') == '1'
    only = os.environ.get('ONLY')
    dev = torch.device('cuda')
    N.load()
    dt = os.environ.get('DTYPE', 'fp32')
    io = int(os.environ.get('IO', '0'))
    from multimodal_segmentation_amd import ops as P
    P.set_conv_precision(dt)
    half = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(dt)
    tot_t = tot_f = 0.0
    print('%-18s %6s %10s %9s %9s %s' % ('layer', 'H', 'C1+C2->N', 'ms', 'TFLOP/s', 'max rel err' if check else ''))
    for name, H, C1, C2, Cout, ups in SHAPES:
        if only and only not in name:
            continue
        g = torch.Generator(device='cpu').manual_seed(1)
        H1 = H // 2 if ups else H
        x1 = torch.randn(B, H1, H1, C1, generator=g).to(dev)
        x2 = torch.randn(B, H, H, C2, generator=g).to(dev) if C2 else None
        dy = torch.randn(B, H, H, Cout, generator=g).to(dev)
        Cin = C1 + C2
        dw = torch.zeros(3, 3, Cin, Cout, device=dev)
        need = N.call('mmseg_conv2d_wgrad_workspace', B, H, H, Cin, Cout, 3, 3)
        ws = torch.empty(max(need, 1), device=dev)
        run = lambda acc=0: N.call('mmseg_conv2d_wgrad', x1, x2, dy, dw.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout, 3, 3,
                                   1, 1, 1, ups, acc)
        if io and N.call('mmseg_conv2d_wgrad_t_supported', H, H, 1, C1, C2, Cout):
            xa = x1.to(half) if io & 1 else x1
            xb = (x2.to(half) if io & 1 else x2) if C2 else None
            da = dy.to(half) if io & 4 else dy
            bits = (1 if io & 1 else 0) | (2 if (io & 1 and C2) else 0) | (io & 4)
            run = lambda acc=0: N.call('mmseg_conv2d_wgrad_t', xa, xb, da, dw.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout, 3, 3,
                                       1, 1, 1, ups, acc, bits)
        t = timeit(run)
        flops = 2.0 * B * H * H * Cin * Cout * 9
        err = ''
        if check:
            run(0)
            xin = x1
            if ups:
                xin = x1.repeat_interleave(2, 1).repeat_interleave(2, 2)
            if C2:
                xin = torch.cat([xin, x2], -1)
            ref = torch.nn.grad.conv2d_weight(xin.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(),
                                              padding=1).permute(2, 3, 1, 0)
            err = '%.2e' % float((dw.double() - ref).abs().max() / ref.abs().max())
            got1 = dw.clone()
            run(1)                                            # accumulate on top: 2x
            err += ' acc %.2e' % float((dw.double() - 2 * ref).abs().max() / ref.abs().max())
            run(0)
            err += ' bitwise' if torch.equal(dw, got1) else ' NOT-REPRODUCIBLE'
        tot_t += t
        tot_f += flops
        print('%-18s %6d %10s %9.3f %9.1f %s' % (name, H, '%d+%d->%d%s' % (C1, C2, Cout, ' ups' if ups else ''), t, flops / t / 1e9, err))
    print('%-18s %6s %10s %9.3f %9.1f' % ('sum', '', '', tot_t, tot_f / tot_t / 1e9))


if __name__ == '__main__':
    main()
