#!/usr/bin/env python3
"""Why is cdf_fused_kernel slower inside the OT loop than alone (0.55 of HBM peak in the bench, 0.63 in
scripts/cdf_probe)?  The same launch at [64, 256, n], timed with the library's HIP events (class cdf_match), in four
surroundings: alone back to back; right behind a rotation GEMM that wrote its input; behind the GEMM and an idle gap (does the
clock the GEMM leaves recover?); behind a GEMM that wrote ANOTHER buffer (is it the data, or the neighbour?).
    python scripts/cdf_inloop_probe.py [n ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimaltextures_amd import ops  # noqa: E402
from optimaltextures_amd.ops import Seg  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    S, C = 64, 256
    g = torch.Generator(device="cpu").manual_seed(0)
    q, _ = torch.linalg.qr(torch.randn(C, C, generator=g))
    R = q.contiguous().to(dev)
    for n in [int(a) for a in sys.argv[1:]] or [16384, 12544, 9216]:
        ns = n * 3 // 4
        x = torch.relu(torch.randn(S, C, n, generator=g) * 2 + 0.3).to(dev)
        other = torch.empty_like(x)
        y = torch.empty_like(x)
        sty = torch.relu(torch.randn(1, C, ns, generator=g) * 1.5 + 0.5).to(dev)
        ys = ops.rotate_seg(sty, R)
        reps = 8

        def timed(label, body):
            for _ in range(2):
                body()
            torch.cuda.synchronize()
            ops.profile_enable(True)
            ops.profile_collect()
            for _ in range(reps):
                body()
            torch.cuda.synchronize()
            ops.profile_enable(False)
            p = ops.profile_collect()
            c = p["cdf_match"]
            us = 1e3 * c["ms"] / c["launches"]
            gm = p.get("gemm_tn")
            extra = f"   (GEMM {1e3 * gm['ms'] / gm['launches']:.1f} us)" if gm else ""
            print(f"n = {n:5d}  {label:<58s} {us:7.1f} us   {8.0 * S * C * n / us * 1e-6:5.2f} TB/s = {8.0 * S * C * n / us * 1e-6 / 8.0:.3f} of peak{extra}")

        def cdf():
            ops.cdf_match_seg(Seg.of(y), Seg.of(ys), out=Seg.of(y))

        ops.rotate_seg(x, R, out=y)
        timed("alone, back to back (its own output as input)", cdf)
        timed("behind the GEMM that wrote its input", lambda: (ops.rotate_seg(x, R, out=y), cdf()))
        timed("behind that GEMM and 400 us of idle", lambda: (ops.rotate_seg(x, R, out=y), torch.cuda._sleep(800_000), cdf()))
        timed("behind a GEMM that wrote another buffer", lambda: (ops.rotate_seg(x, R, out=other), cdf()))
        timed("behind two GEMMs (the loop's neighbours)", lambda: (ops.rotate_seg(y, R, out=other), ops.rotate_seg(x, R, out=y), cdf()))
        del x, y, other


if __name__ == "__main__":
    main()
