#!/usr/bin/env python
"""conv_fast_kernel tile A/B on the small-plane fp32 layers (measurement build: MMSEG_FAST_TILE 1 = 128x128, 2 = 128x64, 3 = 64x64, 0 = the picker):
   MMSEG_HIP_LIB=.../libmmseg_hip_ab.so MMSEG_FAST_TILE=2 python tools/fast_tile_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N

dev = torch.device('cuda')
N.load()
for B, H, Cin, Cout in ((8, 16, 1024, 1024), (8, 16, 512, 1024), (8, 16, 1024, 512), (8, 32, 512, 512), (8, 32, 1024, 512), (8, 32, 512, 1024)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.02
    wp = torch.empty(w.numel(), device=dev)
    N.call('mmseg_conv2d_wprep', w, wp, 3, 3, Cin, Cout, 0)
    y = torch.empty(B, H, H, Cout, device=dev)
    run = lambda: N.call('mmseg_conv2d_fwd', x, None, w, wp, None, y, None, B, H, H, Cin, 0, H, H, Cout, 3, 3, 1, 1, 1, 0, 0, 1, 0.0, 0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 10
    print('tile %s  %d x %d^2 %d -> %d: %.1f us, %.1f TFLOP/s (kernel %d)' % (os.environ.get('MMSEG_FAST_TILE', '0'), B, H, Cin, Cout, us,
                                                                          2.0 * B * H * H * 9 * Cin * Cout / us / 1e6, N.call('mmseg_conv2d_last_kernel')), flush=True)
