#!/usr/bin/env python3
"""bench.py's step at B textures per step with the two codec layouts (vgg.CODEC_LAYOUTS): does the channels-last policy that was
chosen at 64 textures per step also win at 8?      python scripts/b8_layout_probe.py [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optimaltextures_amd import dist as otdist  # noqa: E402
from optimaltextures_amd.driver import OptimalTexture  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
style = bench.synthetic_style(dev)
for B in [int(a) for a in sys.argv[1:]] or [8]:
    for layout in ("mixed", "nchw"):
        tex = OptimalTexture(size=512, iters=500, passes=5, hist_mode="cdf", no_pca=True, layers=(3,), independent=True,
                             codec_layout=layout).to(dev).eval()
        sched = tex.rotation_schedule()
        steps = max(3, 64 // B)

        def run(n, q0):
            rng = otdist.rotation_stream(0, q0, dev)
            rng.prefetch(sched)
            for q in range(q0, q0 + n):
                nxt = otdist.rotation_stream(0, q + 1, dev)
                nxt.begin_feed(sched)
                tex.rng, tex.rng_next = rng, nxt
                tex.forward(otdist.texture_noise(q * B, B, (3, 512, 512), dev), [style])
                rng = nxt

        with torch.inference_mode():
            run(2, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps, 2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"B = {B:3d} codec_layout = {layout:6s}: {B * steps / dt:7.1f} textures/s ({1e3 * dt / steps:6.1f} ms/step)", flush=True)
