#!/usr/bin/env python3
"""Where does the HOST spend a bench step?  cProfile of bench.py's step at B textures per step (default 8: BASELINE config 4's
per-GPU shard, where the step is short enough for the host to matter), top functions by own time.
    python scripts/host_profile.py [B] [steps]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optimaltextures_amd import dist as otdist  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    style = bench.synthetic_style(dev)
    tex = bench.make_texturizer("cdf", dev)

    def run(n, q0=0):
        for q in range(q0, q0 + n):
            tex.rng = otdist.rotation_rng(0, q)
            tex.forward(otdist.texture_noise(q * B, B, (3, 512, 512), dev), [style])

    with torch.inference_mode():
        run(2)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        run(steps, 2)
        pr.disable()
        torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
