#!/usr/bin/env python
"""The SPADE units' shared convolution (8 -> 128 channels, 3x3, ReLU, 16-bit output) at the decoder's resolutions: conv8h_kernel (one
launch) against im2col rows + 1x1 product (mmseg_conv16_mode 0), through ops.conv2d.   python tools/conv8_bench.py [B] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N, ops as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.float16 if os.environ.get('DTYPE') == 'fp16' else torch.bfloat16
P.set_conv_precision('fp16' if dt == torch.float16 else 'bf16')
dev = torch.device('cuda')
w = torch.randn(3, 3, 8, 128, device=dev) * 0.1
b = torch.randn(128, device=dev)
for H in (32, 64, 128, 256):
    x = torch.randn(B, H, H, 8, device=dev)
    line = []
    for m16 in (0, 1):
        N.call('mmseg_conv16_mode', m16)
        with torch.no_grad():
            for _ in range(3):
                y = P.conv2d(x, w, b, 1, 'same', 'relu', 0.0, out_dtype=dt)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                y = P.conv2d(x, w, b, 1, 'same', 'relu', 0.0, out_dtype=dt)
            e.record()
            torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / iters
        line.append('%s %.1f us (%.2f TB/s of output)' % ('conv8h' if m16 else 'im2col+1x1', us, B * H * H * 128 * 2 / us / 1e6))
    print('B %d, %d x %d: %s' % (B, H, H, '; '.join(line)), flush=True)
