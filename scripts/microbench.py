#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the headline shapes (SURVEY 8d "kernel micro-inputs"): every kernel class of
liboptex_hip.so on [S, 256, n] channel-major segments, timed with the library's own HIP-event profiler.

    python scripts/microbench.py [--S 32] [--n 16384] [--C 256] [--reps 20] [--only gemm,sort,...]

Prints one JSON line per kernel class: achieved TFLOP/s or GB/s (algorithmic work / event time) and the roofline
fraction.  Used under rocprofv3 (--kernel-trace --stats, or --pmc passes) to produce profiles/*.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from optimaltextures_amd import ops, rotation  # noqa: E402
from optimaltextures_amd.ops import Seg  # noqa: E402

PEAK_HBM, PEAK_MFMA = 8000.0, 157.3
MFMA = ("gemm_tn", "gram")


def report(tag, prof):
    for name, r in prof.items():
        if r["ms"] <= 0:
            continue
        if name in MFMA:
            ach, peak, unit = r["flops"] / (r["ms"] * 1e9), PEAK_MFMA, "TFLOP/s"
        else:
            ach, peak, unit = r["bytes"] / (r["ms"] * 1e6), PEAK_HBM, "GB/s"
        print(json.dumps({"case": tag, "kernel": name, "achieved": round(ach, 2), "unit": unit,
                          "frac": round(ach / peak, 4), "avg_us": round(1e3 * r["ms"] / r["launches"], 2),
                          "launches": r["launches"]}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=32)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--ns", type=int, default=12288)
    ap.add_argument("--C", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=str, default="gemm,cdf,sort,linear,glue,loop")
    ap.add_argument("--cdf_fused", type=int, default=1, help="0: the two-kernel cdf pipeline (optex_cdf_fused)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.cdf_fused(bool(args.cdf_fused))
    S, C, n, ns = args.S, args.C, args.n, args.ns
    only = set(args.only.split(","))
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((S, C, n), device=dev, generator=g).clamp_min_(0) * 2  # ReLU-like features
    style = torch.randn((1, C, ns), device=dev, generator=g).clamp_min_(0) * 1.5
    R32, Rt32 = rotation.rotations(C, 4, dev, rng=np.random.RandomState(0))
    y = torch.empty_like(x)
    tag = f"S{S}_C{C}_n{n}" + ("" if args.cdf_fused else "_twokernel")

    def timed(fn, reps=args.reps, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ops.profile_collect()
        ops.profile_enable(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        return ops.profile_collect()

    if "gemm" in only:
        report(tag + "_rotate", timed(lambda: ops.rotate_seg(x, R32[0], out=y)))
        report(tag + "_unrotate_blend", timed(lambda: ops.unrotate_seg(y, Rt32[0], out=y.new_empty(y.shape), content=x,
                                                                       strength=0.05), reps=5))
    ops.rotate_seg(x, R32[0], out=y)
    ys = ops.rotate_seg(style, R32[0])
    if "cdf" in only:
        out = torch.empty_like(y)
        report(tag + "_cdf", timed(lambda: ops.cdf_match_seg(Seg.of(y), Seg.of(ys), out=Seg.of(out))))
    if "sort" in only and n <= 16384:
        report(tag + "_sort_kv", timed(lambda: ops.sort_columns(y), reps=5))
        out = torch.empty_like(y)
        report(tag + "_sort_match", timed(lambda: ops.sort_match_seg(Seg.of(y), Seg.of(ys), out=Seg.of(out)), reps=5))
        # tie-heavy variant: un-rotated ReLU features (about half the keys are exactly 0)
        report(tag + "_sort_kv_ties", timed(lambda: ops.sort_columns(x), reps=5))
    if "sortmatch" in only and n <= 16384:  # the match on the hot path alone (PMC passes: one kernel class per run)
        out = torch.empty_like(y)
        report(tag + "_sort_match", timed(lambda: ops.sort_match_seg(Seg.of(y), Seg.of(ys), out=Seg.of(out))))
    if "linalg" in only:
        _, cov = ops.linear_stats(Seg.of(y), pool=False)
        _, cov_s = ops.linear_stats(Seg.of(ys), pool=False)
        report(tag + "_chol_inv", timed(lambda: ops.chol_inv(cov), reps=10))
        report(tag + "_spd_sqrt", timed(lambda: ops.spd_sqrt(cov), reps=5))
        for mode in ("chol", "pca", "sym"):
            report(tag + "_transfer_" + mode, timed(lambda: ops.transfer_operator_t(cov, cov_s, mode), reps=5))
            xx = x.clone()
            report(tag + "_loop_" + mode, timed(lambda: ops.ot_loop(mode, xx, style, R32, Rt32), reps=2, warm=1))
            xx = x.clone()
            report(tag + "_loop_fused_" + mode, timed(lambda: ops.ot_loop(mode, xx, style, R32, Rt32, fuse_rotations=True), reps=2, warm=1))
    if "linear" in only:
        report(tag + "_linear", timed(lambda: ops.linear_stats(Seg.of(y), pool=False), reps=5))
    if "glue" in only:
        img = torch.randn((S, 64, 512, 512), device=dev, generator=g)
        bias = torch.randn(64, device=dev, generator=g)
        report("S%d_64x512x512_glue_relu_pad" % S, timed(lambda: ops.vgg_glue(img, bias, relu=True, pad=1), reps=10))
        report("S%d_64x512x512_glue_relu_pool_pad" % S, timed(lambda: ops.vgg_glue(img, bias, relu=True, pool=True, pad=1), reps=10))
        half = img[:, :, :256, :256].contiguous()
        report("S%d_64x256x256_glue_relu_up_pad" % S, timed(lambda: ops.vgg_glue(half, bias, relu=True, up=True, pad=1), reps=10))
    if "loop" in only:
        xx = x.clone()
        report(tag + "_loop_cdf", timed(lambda: ops.ot_loop("cdf", xx, style, R32, Rt32), reps=3, warm=1))
    if "loopsort" in only:  # the sort matcher as the hot loop runs it: range from the rotation GEMM's epilogue, hoisted style sort
        xx = x.clone()
        report(tag + "_loop_sort", timed(lambda: ops.ot_loop("sort", xx, style, R32, Rt32), reps=3, warm=1))


if __name__ == "__main__":
    main()
