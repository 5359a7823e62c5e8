#!/usr/bin/env python3
"""optex_vgg_glue_layout, channels-last -> planar (bias + ReLU; the encoder outputs the OT loop consumes), at the shapes of a
64-texture bench step, timed with the library's HIP events (class vgg_glue).  Run twice for an A/B of a probe build.
    python scripts/glue_planar_probe.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimaltextures_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for C, H, pool, pad, out_nhwc in ((64, 512, False, 0, False), (128, 256, False, 0, False), (256, 128, False, 0, False),
                                      (64, 512, True, 0, False), (64, 512, False, 1, False), (64, 448, False, 1, False),
                                      (64, 384, False, 1, False), (64, 320, False, 1, False), (64, 256, False, 1, False),
                                      (64, 512, False, 1, True), (64, 256, False, 1, True), (128, 128, False, 1, True)):
        x = torch.randn(B, H, H, C, device=dev).permute(0, 3, 1, 2) if not out_nhwc else torch.randn(B, C, H, H, device=dev)
        b = torch.randn(C, device=dev)
        want = torch.relu(x + b[None, :, None, None])
        if pool:
            want = torch.nn.functional.max_pool2d(want, 2, ceil_mode=True)
        if pad:
            want = torch.nn.functional.pad(want, (1, 1, 1, 1), mode="reflect")
        for _ in range(2):
            y = ops.vgg_glue(x, b, relu=True, pool=pool, pad=pad, out_nhwc=out_nhwc)
        ok = bool((y == want).all())
        del want
        torch.cuda.synchronize()
        ops.profile_collect()
        ops.profile_enable(True)
        for _ in range(6):
            y = ops.vgg_glue(x, b, relu=True, pool=pool, pad=pad, out_nhwc=out_nhwc)
        torch.cuda.synchronize()
        ops.profile_enable(False)
        p = ops.profile_collect()["vgg_glue"]
        us = 1e3 * p["ms"] / p["launches"]
        gb = 4.0 * (x.numel() + y.numel()) * 1e-9
        print(f"[{B}, {C}, {H}, {H}] {'cl -> planar' if not out_nhwc else 'planar -> cl'}{' pool' if pool else ''}{' pad' if pad else ''}: "
              f"{us:8.1f} us  {gb / us * 1e3:5.2f} TB/s  {'bit-equal to torch' if ok else 'WRONG'}", flush=True)
        del x, y


if __name__ == "__main__":
    main()
