#!/usr/bin/env python3
"""Diagnostic: time the layout variants of optex_vgg_glue_layout on the largest codec tensor (B x 64 x 512 x 512, bias +
ReLU + reflection pad).  Not part of the library."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimaltextures_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 64, 512, 512, device=dev); b = torch.randn(64, device=dev)
x_cl = x.contiguous(memory_format=torch.channels_last)
gb = 4 * (x.numel() + B * 64 * 514 * 514) / 1e9
for name, src, out_cl in [("planar->planar", x, False), ("planar->cl", x, True), ("cl->planar", x_cl, False), ("cl->cl", x_cl, True)]:
    for _ in range(2):
        y = ops.vgg_glue(src, b, relu=True, pad=1, out_nhwc=out_cl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        y = ops.vgg_glue(src, b, relu=True, pad=1, out_nhwc=out_cl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:16s} {dt*1e3:7.3f} ms  {gb/dt/1e3:5.2f} TB/s", flush=True)
