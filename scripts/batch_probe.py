#!/usr/bin/env python3
"""Where does a SMALL batch lose throughput?  bench.py's step at B textures per step (BASELINE config 4 shards 8 per GPU),
four ways: rotations drawn from the numpy stream on the HOST inside the step (bench.py --host_rng), the same stream advanced
on the DEVICE (bench.py's default, rotation.DeviceNormals), the rotations of every (pass, layer) cached on the device
beforehand (no RNG at all inside the step: the GPU-side bound), and the host cost of drawing one step's normals alone.
    python scripts/batch_probe.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optimaltextures_amd import dist as otdist, rotation  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    style = bench.synthetic_style(dev)
    tex = bench.make_texturizer("cdf", dev)
    sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 64]
    for B in sizes:
        steps = max(3, 96 // B)

        def run(n):
            for q in range(n):
                tex.rng = otdist.rotation_rng(0, q)
                tex.forward(otdist.texture_noise(q * B, B, (3, 512, 512), dev), [style])

        with torch.inference_mode():
            run(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps)
            host = time.perf_counter() - t0
            torch.cuda.synchronize()
            live = time.perf_counter() - t0

            def run_dev(n):
                for q in range(n):
                    tex.rng = otdist.rotation_stream(0, q, dev)
                    tex.forward(otdist.texture_noise(q * B, B, (3, 512, 512), dev), [style])

            run_dev(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_dev(steps)
            host_d = time.perf_counter() - t0
            torch.cuda.synchronize()
            devt = time.perf_counter() - t0
            # device stream, group q + 1 drawn on the side stream while group q runs (bench.py's default since round 5)
            def run_ahead(n, q0):
                sched = tex.rotation_schedule()
                rng = otdist.rotation_stream(0, q0, dev)
                rng.prefetch(sched)
                for q in range(q0, q0 + n):
                    nxt = otdist.rotation_stream(0, q + 1, dev)
                    nxt.prefetch(sched)
                    tex.rng = rng
                    tex.forward(otdist.texture_noise(q * B, B, (3, 512, 512), dev), [style])
                    rng = nxt

            run_ahead(2, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_ahead(steps, 2)
            host_a = time.perf_counter() - t0
            torch.cuda.synchronize()
            aht = time.perf_counter() - t0

            # ... the next group's draws FED during this step's decode phases instead (bench.py's default from round 5's second half)
            def run_fed(n, q0):
                sched = tex.rotation_schedule()
                rng = otdist.rotation_stream(0, q0, dev)
                rng.prefetch(sched)
                for q in range(q0, q0 + n):
                    nxt = otdist.rotation_stream(0, q + 1, dev)
                    nxt.begin_feed(sched)
                    tex.rng, tex.rng_next = rng, nxt
                    tex.forward(otdist.texture_noise(q * B, B, (3, 512, 512), dev), [style])
                    rng = nxt

            run_fed(2, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_fed(steps, 2)
            torch.cuda.synchronize()
            fedt = time.perf_counter() - t0
            # device stream with one workgroup of the persistent GEMM on EVERY CU (spare = 0: what rounds 3-4 did)
            from optimaltextures_amd import ops as _ops
            tex.gemm_spare_cus = 0
            run_dev(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_dev(steps)
            torch.cuda.synchronize()
            dev0 = time.perf_counter() - t0
            tex.gemm_spare_cus = "auto"
            # the same steps with every rotation batch of the call served from a device-side cache
            cache, own = {}, rotation.rotations

            def cached(N, count, device, rng=None, want64=False):
                key = (N, count)
                if key not in cache:
                    cache[key] = own(N, count, device, rng=np.random.RandomState(1))
                return cache[key]

            rotation.rotations = cached
            import optimaltextures_amd.driver as drv
            drv.rotation.rotations = cached
            run(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps)
            host_c = time.perf_counter() - t0
            torch.cuda.synchronize()
            gpu = time.perf_counter() - t0
            rotation.rotations = own
            drv.rotation.rotations = own
        t0 = time.perf_counter()
        r = np.random.RandomState(0)
        for it in (13, 12, 10, 9, 8):
            r.normal(size=(it, 256 * 257 // 2 - 1))
        draw = time.perf_counter() - t0
        print(f"B = {B:3d}: host stream {B * steps / live:7.1f} textures/s ({1e3 * live / steps:6.1f} ms/step, host enqueue {1e3 * host / steps:6.1f} ms) | "
              f"device stream {B * steps / devt:7.1f} textures/s ({1e3 * devt / steps:6.1f} ms/step, host enqueue {1e3 * host_d / steps:6.1f} ms) | "
              f"device stream, no spare CU in the GEMM grid {B * steps / dev0:7.1f} textures/s ({1e3 * dev0 / steps:6.1f} ms/step) | "
              f"device stream a step ahead {B * steps / aht:7.1f} textures/s ({1e3 * aht / steps:6.1f} ms/step, host enqueue {1e3 * host_a / steps:6.1f} ms) | "
              f"device stream fed during the previous step's decode phases {B * steps / fedt:7.1f} textures/s ({1e3 * fedt / steps:6.1f} ms/step) | "
              f"cached rotations {B * steps / gpu:7.1f} textures/s ({1e3 * gpu / steps:6.1f} ms/step, host enqueue {1e3 * host_c / steps:6.1f} ms) | "
              f"drawing one step's 1.71 M normals on the host: {1e3 * draw:.1f} ms")


if __name__ == "__main__":
    main()
