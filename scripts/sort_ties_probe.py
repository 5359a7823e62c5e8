#!/usr/bin/env python3
"""optex_sort_columns (default kernel chain: rank_match5w_kernel -> rank_match4_kernel on its flagged columns -> radix kernel) and
optex_sort_match (that chain, and OPTEX_F_SORT_RANK4 = the round-5 chain) on rotated columns, on quantised ones and on tie-heavy ones
(un-rotated ReLU features: about half the keys are exactly 0).  Times are the library's HIP events over classes sort_columns + sort_match + sort_radix_sweep.
    python scripts/sort_ties_probe.py [n ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimaltextures_amd import _lib, ops, rotation  # noqa: E402
from optimaltextures_amd.ops import Seg  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    S, C = 64, 256
    g = torch.Generator(device=dev).manual_seed(0)
    R32, _ = rotation.rotations(C, 1, dev, rng=np.random.RandomState(0))
    for n in [int(a) for a in sys.argv[1:]] or [16384, 9216, 4096]:
        x = torch.randn((S, C, n), device=dev, generator=g).clamp_min_(0) * 2
        y = ops.rotate_seg(x, R32[0])
        q = torch.round(y * 64) / 64       # a few hundred distinct values per column
        cases = (("rotated (no ties)", y), ("quantised to 1/64", q), ("ReLU, half zeros", x))
        if os.environ.get("TIES_ONLY"):   # (one case per rocprofv3 --kernel-trace run)
            cases = cases[int(os.environ["TIES_ONLY"]):][:1]
        for label, t in cases:
            def timed(fn):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                ops.profile_collect()
                ops.profile_enable(True)
                for _ in range(5):
                    r = fn()
                torch.cuda.synchronize()
                ops.profile_enable(False)
                p = ops.profile_collect()
                return r, 1e3 * sum(p[c]["ms"] for c in ("sort_columns", "sort_match", "sort_radix_sweep") if c in p) / 5

            (k, i), us = timed(lambda: ops.sort_columns(t))
            ks, _ = torch.sort(t[:2], dim=-1, stable=True)
            ok = bool((ks == k[:2]).all()) and bool((torch.gather(t[:2], -1, i[:2].long()) == k[:2]).all())
            print(f"n = {n:5d}  {label:<20s} sort_columns               {us:8.1f} us  {12.0 * S * C * n / us * 1e-6 / 8:.3f} of 8 TB/s at 12 B per key  "
                  f"{'ok' if ok else 'WRONG'}")
            outs = []
            for name, fl in (("default chain", 0), ("F_SORT_RANK4", _lib.F_SORT_RANK4)):
                o, us = timed(lambda: ops.sort_match_seg(Seg.of(t), Seg.of(t[:1]), flags=fl))
                outs.append(o)
                print(f"n = {n:5d}  {label:<20s} sort_match, {name:<14s} {us:8.1f} us  {8.0 * S * C * n / us * 1e-6 / 8:.3f} of 8 TB/s at  8 B per key")
            print(f"           the two chains agree bit for bit: {bool((outs[0] == outs[1]).all())}")


if __name__ == "__main__":
    main()
