#!/usr/bin/env python3
"""The linear modes' apply GEMM at [64, 256, n] with per-segment matrices (at_ss = C*C): plain, with a per-row bias (badd), and
with the centring inside the k-loop (bsub + badd) — the library's HIP events, class gemm_tn.
    python scripts/apply_gemm_probe.py [n ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimaltextures_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    S, C = 64, 256
    g = torch.Generator(device=dev).manual_seed(0)
    for n in [int(a) for a in sys.argv[1:]] or [16384, 9216, 4096]:
        x = torch.randn((S, C, n), device=dev, generator=g).clamp_min_(0) * 2
        out = torch.empty_like(x)
        M = torch.randn((S, C, C), device=dev, generator=g) / 16
        mu = torch.randn((S, C), device=dev, generator=g)
        b = torch.randn((S, C), device=dev, generator=g)
        for label, kw in (("plain", {}), ("badd", dict(badd=b, badd_ss=C)), ("bsub + badd", dict(bsub=mu, bsub_ss=C, badd=b, badd_ss=C)),
                          ("plain, one shared matrix", dict(shared=True))):
            shared = kw.pop("shared", False)

            def run():
                ops.gemm_tn(M[0] if shared else M, x, out, C, C, n, S, lda=C, at_ss=0 if shared else C * C, ldb=n, b_ss=C * n, ldo=n, o_ss=C * n, **kw)

            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ops.profile_collect()
            ops.profile_enable(True)
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            ops.profile_enable(False)
            p = ops.profile_collect()["gemm_tn"]
            us = 1e3 * p["ms"] / p["launches"]
            print(f"n = {n:5d}  {label:<26s} {us:7.1f} us  {2.0 * S * C * C * n / us * 1e-6:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
