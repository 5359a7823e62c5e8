#!/usr/bin/env python3
"""bench.py's un-shared-rotations step on its own (one rotation sequence per texture, 64 numpy streams advanced on the GPU),
for a kernel trace:   rocprofv3 --kernel-trace ... -- python scripts/ownrot_step.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optimaltextures_amd import dist as otdist  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = 64
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
style = bench.synthetic_style(dev)
tex = bench.make_texturizer("cdf", dev)
with torch.inference_mode():
    for q in range(steps):
        tex.rng = otdist.rotation_stream(0, [q * B + j for j in range(B)], dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tex.forward(otdist.texture_noise(q * B, B, (3, 512, 512), dev), [style])
        torch.cuda.synchronize()
        print(f"step {q}: {B / (time.perf_counter() - t0):.1f} textures/s")
