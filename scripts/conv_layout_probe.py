#!/usr/bin/env python3
"""Diagnostic: MIOpen fp32 3x3 convolution time per VGG layer shape (B = 32, 512^2 pass), NCHW vs channels_last.
Decides whether the (PyTorch-side) codec should run channels_last.  Not part of the library."""
import time

import torch

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B = 32
shapes = [(3, 64, 512), (64, 64, 512), (64, 128, 256), (128, 128, 256), (128, 256, 128),   # encoder relu3_1
          (256, 128, 128), (128, 128, 256), (128, 64, 256), (64, 64, 512), (64, 3, 512)]   # decoder
for cin, cout, hw in shapes:
    x = torch.randn(B, cin, hw + 2, hw + 2, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    res = {}
    for name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        xx, ww = x.contiguous(memory_format=fmt), w.contiguous(memory_format=fmt)
        with torch.inference_mode():
            for _ in range(3):
                y = torch.nn.functional.conv2d(xx, ww)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                y = torch.nn.functional.conv2d(xx, ww)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 10
    flops = 2.0 * B * cout * cin * 9 * hw * hw
    print(f"conv {cin:3d}->{cout:3d} @{hw}^2: nchw {res['nchw']*1e3:7.3f} ms ({flops/res['nchw']/1e12:6.1f} TF/s)   "
          f"nhwc {res['nhwc']*1e3:7.3f} ms ({flops/res['nhwc']/1e12:6.1f} TF/s)", flush=True)
