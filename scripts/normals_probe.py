#!/usr/bin/env python3
"""optex_legacy_normals alone on an idle GPU: time per draw of one bench pass's normals (13 rotations of 256^2), one stream
and 64 streams.   python scripts/normals_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimaltextures_amd.rotation import DeviceNormals  # noqa: E402

dev = torch.device("cuda:0")
for n_streams in (1, 64):
    dn = DeviceNormals(list(range(1000, 1000 + n_streams)), dev, side_stream=False)
    for count in (32895, 13 * 32895, 52 * 32895):
        dn.draw(count)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            dn.draw(count)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{n_streams:3d} stream(s) x {count:8d} normals: {1e3 * dt:8.3f} ms  ({1e9 * dt / count:6.2f} ns per normal per stream, "
              f"{count * n_streams / dt / 1e6:8.1f} M normals/s)")
