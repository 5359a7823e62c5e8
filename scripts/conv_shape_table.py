#!/usr/bin/env python3
"""Per-shape table of the VGG codec's convolutions at a given batch: every distinct conv2d call of one
bench step — shape, how often it runs per step, and its time (MIOpen find mode, `cudnn.benchmark`) in BOTH memory layouts
— and what a per-shape layout choice would return against the codec's policy (vgg.CODEC_LAYOUT = "mixed": channels-last
for every 3x3 convolution with >= 64 output channels).
    python scripts/conv_shape_table.py [B ...]        (default: 8 64)"""
import collections
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optimaltextures_amd import dist as otdist  # noqa: E402


def time_conv(x, w, reps=20):
    for _ in range(3):
        F.conv2d(x, w, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        F.conv2d(x, w, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    style = bench.synthetic_style(dev)
    tex = bench.make_texturizer("cdf", dev)
    for B in [int(a) for a in sys.argv[1:]] or [8, 64]:
        calls = collections.Counter()
        real = F.conv2d

        def spy(x, w, b=None, *a, **k):
            cl = (not x.is_contiguous()) and x.permute(0, 2, 3, 1).is_contiguous()
            calls[(tuple(x.shape), tuple(w.shape), bool(cl))] += 1
            return real(x, w, b, *a, **k)

        with torch.inference_mode():
            tex.rng = otdist.rotation_stream(0, 0, dev)
            tex.forward(otdist.texture_noise(0, B, (3, 512, 512), dev), [style])   # warm-up (find)
            torch.nn.functional.conv2d = spy
            try:
                tex.rng = otdist.rotation_stream(0, 1, dev)
                tex.forward(otdist.texture_noise(B, B, (3, 512, 512), dev), [style])
            finally:
                torch.nn.functional.conv2d = real
            torch.cuda.synchronize()
            rows, tot_policy, tot_best = [], 0.0, 0.0
            for (xs, ws, cl), cnt in sorted(calls.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3] * kv[0][1][0]):
                n, cin, h, w_ = xs
                cout, _, kh, kw = ws
                x = torch.randn(xs, device=dev)
                wt = torch.randn(ws, device=dev) * 0.05
                t = {}
                t[False] = time_conv(x, wt)
                t[True] = time_conv(x.contiguous(memory_format=torch.channels_last), wt.contiguous(memory_format=torch.channels_last))
                flop = 2.0 * n * cout * (h - kh + 1) * (w_ - kw + 1) * cin * kh * kw
                rows.append((xs, ws, cl, cnt, t[cl], t[not cl], flop / t[cl] * 1e-6))
                tot_policy += cnt * t[cl]
                tot_best += cnt * min(t[True], t[False])
            print(f"\n## {B} textures per step: {sum(calls.values())} conv2d calls, {len(calls)} distinct shapes\n")
            print("| input [N, C, H, W] | weight | policy layout | calls | policy us | other layout us | TFLOP/s (policy) | better |")
            print("|---|---|---|---:|---:|---:|---:|---|")
            for xs, ws, cl, cnt, tp, to, tf in rows:
                print(f"| {list(xs)} | {list(ws)} | {'NHWC' if cl else 'NCHW'} | {cnt} | {tp:.1f} | {to:.1f} | {tf:.1f} | "
                      f"{'other by %.0f %%' % (100 * (tp - to) / tp) if to < 0.97 * tp else ''} |")
            print(f"\nsum over the step (isolated timings, back-to-back launches): policy {tot_policy / 1e3:.2f} ms, best layout per shape "
                  f"{tot_best / 1e3:.2f} ms ({100 * (tot_policy - tot_best) / tot_policy:.1f} % less; a layout change costs nothing extra: "
                  f"the glue pass between two convolutions writes either layout)")


if __name__ == "__main__":
    main()
