#!/usr/bin/env python
"""One 16-bit convolution shape, one kernel, N launches (for rocprofv3 --pmc / --kernel-trace):
   python tools/conv16_one.py MODE B H C1 C2 Cout k ups [iters]      MODE: 0 conv_fast_kernel, 2 conv16_kernel"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_segmentation_amd import _native as N, ops as P

mode, B, H, C1, C2, Cout, k, ups = [int(v) for v in sys.argv[1:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20
f32 = os.environ.get('DTYPE') == 'f32'
dt = torch.float32 if f32 else torch.bfloat16
P.set_conv_precision('fp32' if f32 else 'bf16')
dev = torch.device('cuda')
H1 = H // 2 if ups else H
x1 = torch.randn(B, H1, H1, C1, device=dev).to(dt)
x2 = torch.randn(B, H, H, C2, device=dev).to(dt) if C2 else None
Cin = C1 + C2
w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
b = torch.randn(Cout, device=dev)
wp = torch.empty(w.numel(), device=dev)
N.call('mmseg_conv2d_wprep', w, wp, k, k, Cin, Cout, 0)
y = torch.empty(B, H, H, Cout, device=dev, dtype=dt)
N.call('mmseg_conv16_mode', mode)
for _ in range(3):
    (N.call('mmseg_conv2d_fwd', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, k // 2, k // 2, ups, 0, 1, 0.0, 0) if f32 else N.call('mmseg_conv2d_fwd_t', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, k // 2, k // 2, ups, 0, 1, 0.0, 0, 1 | (2 if C2 else 0) | 4))
torch.cuda.synchronize()
_s, _e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_s.record()
for _ in range(iters):
    (N.call('mmseg_conv2d_fwd', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, k // 2, k // 2, ups, 0, 1, 0.0, 0) if f32 else N.call('mmseg_conv2d_fwd_t', x1, x2, w, wp, b, y, None, B, H, H, C1, C2, H, H, Cout, k, k, 1, k // 2, k // 2, ups, 0, 1, 0.0, 0, 1 | (2 if C2 else 0) | 4))
_e.record()
torch.cuda.synchronize()
ms = _s.elapsed_time(_e) / iters
print('mode %d abl %s shape %s: %.1f us, %.1f TFLOP/s, last kernel %d' % (mode, os.environ.get('MMSEG_CONV16_ABL', '-'), sys.argv[2:9], 1e3 * ms,
                                                                       2.0 * B * H * H * Cin * Cout * k * k / ms / 1e9, N.call('mmseg_conv2d_last_kernel')))
