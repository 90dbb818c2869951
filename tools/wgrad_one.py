#!/usr/bin/env python
"""One weight-gradient shape a few times (for rocprofv3 --pmc): [DTYPE=bf16|fp16 [IO=<bits>]] python tools/wgrad_one.py H C1 C2 Cout ups [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_segmentation_amd import _native as N
H, C1, C2, Cout, ups = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
B = 8
dev = torch.device('cuda')
N.load()
H1 = H // 2 if ups else H
x1 = torch.randn(B, H1, H1, C1, device=dev)
x2 = torch.randn(B, H, H, C2, device=dev) if C2 else None
dy = torch.randn(B, H, H, Cout, device=dev)
dw = torch.zeros(3, 3, C1 + C2, Cout, device=dev)
need = N.call('mmseg_conv2d_wgrad_workspace', B, H, H, C1 + C2, Cout, 3, 3)
ws = torch.empty(max(need, 1), device=dev)
dt, io = os.environ.get('DTYPE', 'fp32'), int(os.environ.get('IO', '0'))
from multimodal_segmentation_amd import ops as P
P.set_conv_precision(dt)
half = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(dt)
if io:
    x1 = x1.to(half) if io & 1 else x1
    x2 = x2.to(half) if (io & 1 and C2) else x2
    dy = dy.to(half) if io & 4 else dy
bits = (1 if io & 1 else 0) | (2 if (io & 1 and C2) else 0) | (io & 4)
for _ in range(iters):
    if io:
        N.call('mmseg_conv2d_wgrad_t', x1, x2, dy, dw.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout, 3, 3, 1, 1, 1, ups, 0, bits)
    else:
        N.call('mmseg_conv2d_wgrad', x1, x2, dy, dw.view(-1), ws, ws.numel(), B, H, H, C1, C2, H, H, Cout, 3, 3, 1, 1, 1, ups, 0)
torch.cuda.synchronize()
