#!/usr/bin/env python3
"""Latency of ONE 512^2 texture with the reference's default command line (B = 1, relu5_1..relu1_1, PCA, chol, 493 OT
iterations; synthetic weights) — for rocprofv3 --kernel-trace: where a launch-bound single-image run spends its time.
    python scripts/single_latency.py [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_style  # noqa: E402
from optimaltextures_amd import dist as otdist  # noqa: E402
from optimaltextures_amd.driver import OptimalTexture  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
style = synthetic_style(dev)
m = OptimalTexture(size=512, iters=500, passes=5, hist_mode=sys.argv[2] if len(sys.argv) > 2 else "chol", layers=(5, 4, 3, 2, 1)).to(dev).eval()
torch.backends.cudnn.benchmark = True   # like bench.py
with torch.inference_mode():
    for rep in range(reps):
        m.rng = otdist.rotation_rng(0, rep)
        x = otdist.texture_noise(rep, 1, (3, 512, 512), dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.forward(x, [style])
        torch.cuda.synchronize()
        print(f"call {rep}: {time.perf_counter() - t0:.3f} s", flush=True)
