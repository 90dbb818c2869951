#!/usr/bin/env python
"""Weight gradient of the FiLM decoder's 8 -> 8 3x3 layers (conv_wgrad_c8m_kernel + slab reduction) at the BASELINE geometries:
   python tools/wgrad_c8_bench.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda')
N.load()
for B, H in ((8, 256), (48, 256), (16, 320), (240, 320)):
    x = torch.randn(B, H, H, 8, device=dev)
    dy = torch.randn(B, H, H, 8, device=dev)
    dw = torch.zeros(3, 3, 8, 8, device=dev)
    need = N.call('mmseg_conv2d_wgrad_workspace', B, H, H, 8, 8, 3, 3)
    ws = torch.empty(max(need, 1), device=dev)
    run = lambda: N.call('mmseg_conv2d_wgrad', x, None, dy, dw.view(-1), ws, ws.numel(), B, H, H, 8, 0, H, H, 8, 3, 3, 1, 1, 1, 0, 0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run()
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    print('B %d, %d x %d: %.1f us incl. the slab reduction, %.2f TB/s of x + dy (last kernel %d)' % (B, H, H, us, 2.0 * B * H * H * 32 / us / 1e6,
                                                                                                  N.call('mmseg_conv2d_last_kernel')), flush=True)
