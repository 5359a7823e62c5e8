#!/usr/bin/env python3
"""Randomised stress of optex_sort_match and optex_sort_columns against the oracle (not part of the test suite: minutes of CPU oracle time).
Column lengths around every workgroup-shape boundary of csrc/sort_rank4.hip, source lengths below / equal / above, mixed
edge distributions per column, unaligned views.   python scripts/sort_stress.py [n_cases] [seed]
With `--loop` as the first argument: the match as optex_ot_loop runs it (round 6: rank_match5w_kernel, csrc/sort_rank5.hip — it needs the
column range of the rotation GEMM's epilogue) — ONE iteration with R = I (the rotation is then exact up to the sign of zero; the
oracle's fma chain does the same), C = 128, column lengths = multiples of 64 around every workgroup-shape boundary, mixed edge
distributions per column, against the oracle and against the same call with OPTEX_F_SORT_RANK4.
    python scripts/sort_stress.py --loop [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import oracle as orc  # noqa: E402
from optimaltextures_amd import ops  # noqa: E402
from optimaltextures_amd.ops import Seg  # noqa: E402


def column(kind, n, rng):
    if kind == 0:
        return rng.standard_normal(n)
    if kind == 1:
        return np.maximum(rng.standard_normal(n), 0)
    if kind == 2:
        return rng.standard_normal(n) * np.exp(rng.standard_normal(n))          # heavy tail
    if kind == 3:
        return np.round(rng.standard_normal(n) * rng.choice([4, 32, 256])) / 8   # tie groups of every size
    if kind == 4:
        x = rng.standard_normal(n)
        x[rng.random(n) < 0.3] = rng.choice([0.0, -0.0, 1.5])
        return x
    if kind == 5:
        return np.sort(rng.standard_normal(n))[:: rng.choice([1, -1])]
    if kind == 6:
        b = rng.standard_normal((n + 2) // 3).astype(np.float32)
        return rng.permutation(np.stack([b, np.nextafter(b, np.float32(9)), np.nextafter(b, np.float32(-9))], 1).reshape(-1)[:n])
    if kind == 7:
        return rng.standard_normal(n) * 10.0 ** rng.integers(-30, 30)
    return rng.uniform(-1, 1, n) + rng.choice([0, 1000.0])


def loop_main(argv):
    cases = int(argv[0]) if argv else 60
    rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 0)
    dev = torch.device("cuda:0")
    C = 128
    eye = np.eye(C, dtype=np.float32)[None]
    Rd = torch.from_numpy(eye).to(dev)
    edges = [2112, 2560, 3072, 4096, 4160, 5120, 6400, 7168, 7232, 8192, 9216, 9280, 10240, 11264, 12288, 12544, 13312, 13376, 14336,
             15360, 16320, 16384]
    bad = flagged_like = 0
    for it in range(cases):
        n = int(rng.choice(edges)) if rng.random() < 0.7 else 64 * int(rng.integers(33, 257))
        ns = int(rng.choice([n, max(4, 3 * n // 4 // 4 * 4), max(4, (n // 2 + 8) // 4 * 4), min(16384, n + 1024), 4 * int(rng.integers(1, 4097))]))
        S = int(rng.integers(1, 3))
        t = np.stack([[column(int(rng.integers(0, 9)), n, rng) for _ in range(C)] for _ in range(S)]).astype(np.float32)
        s = np.stack([column(int(rng.integers(0, 9)), ns, rng) for _ in range(C)]).astype(np.float32)[None]
        s = np.nan_to_num(s, nan=0.0, posinf=3e38, neginf=-3e38)
        t = np.nan_to_num(t, nan=0.0, posinf=3e38, neginf=-3e38)
        outs = []
        for flags in (0, ops.F_SORT_RANK4):
            xd = torch.from_numpy(t).to(dev)
            ops.ot_loop("sort", xd, torch.from_numpy(s).to(dev), Rd, Rd, flags=flags)
            outs.append(xd.cpu().numpy())
        if not np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)):
            bad += 1
            print(f"MISMATCH (rank5w vs rank4) case {it}: n={n} ns={ns} S={S}", flush=True)
        for k in range(S):
            want = orc.unrotate_cm(orc.sort_match(orc.rotate_cm(t[k], eye[0]), orc.rotate_cm(s[0], eye[0])), eye[0])
            if not np.array_equal(outs[0][k].view(np.uint32), want.view(np.uint32)):
                bad += 1
                print(f"MISMATCH (vs oracle) case {it}: n={n} ns={ns} S={S} segment {k}", flush=True)
    print(f"--loop: {cases} cases ({cases * C} .. {2 * cases * C} columns), {bad} mismatches")
    return 1 if bad else 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--loop":
        return loop_main(sys.argv[2:])
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    edges = [512, 1024, 2048, 2049, 2304, 2305, 2560, 2561, 3072, 4096, 5120, 5121, 6144, 6400, 7168, 8192, 9216, 10240, 10241,
             11264, 12288, 12544, 13312, 14336, 15360, 16383, 16384]
    bad = 0
    for it in range(cases):
        n = int(rng.choice(edges)) + int(rng.choice([0, 0, -1, -3, 4, 7])) if rng.random() < 0.8 else int(rng.integers(512, 16385))
        n = max(512, min(16384, n))
        ns = int(rng.choice([n, n, max(1, 3 * n // 4), max(1, n // 2 + 5), min(16384, n + 1000), int(rng.integers(1, 16385))]))
        S, C = int(rng.integers(1, 3)), int(rng.integers(1, 6))
        off = int(rng.choice([0, 0, 1, 2, 3]))
        t = np.stack([[column(int(rng.integers(0, 9)), n + off, rng) for _ in range(C)] for _ in range(S)]).astype(np.float32)
        s = np.stack([column(int(rng.integers(0, 9)), ns, rng) for _ in range(C)]).astype(np.float32)[None]
        s = np.nan_to_num(s, nan=0.0, posinf=3e38, neginf=-3e38)
        t = np.nan_to_num(t, nan=0.0, posinf=3e38, neginf=-3e38)
        td = torch.from_numpy(t).to(dev)
        out = ops.sort_match_seg(Seg(td[:, :, off:], n + off, C * (n + off), n, C, S), Seg.of(torch.from_numpy(s).to(dev))).cpu().numpy()
        for k in range(S):
            want = orc.sort_match(np.ascontiguousarray(t[k][:, off:]), s[0])
            if not np.array_equal(out[k], want):
                bad += 1
                print(f"MISMATCH case {it}: n={n} ns={ns} S={S} C={C} off={off} segment {k}", flush=True)
        # optex_sort_columns (keys + stable indices) on the same columns
        tc = np.ascontiguousarray(t[:, :, off:])
        keys, idx = ops.sort_columns(torch.from_numpy(tc).to(dev))
        keys, idx = keys.cpu().numpy(), idx.cpu().numpy().view(np.uint32)
        for k in range(S):
            ok, oi = orc.sort_columns(tc[k])
            if not (np.array_equal(idx[k], oi) and np.array_equal(keys[k].view(np.uint32), ok.view(np.uint32))):
                bad += 1
                print(f"MISMATCH (sort_columns) case {it}: n={n} S={S} C={C} segment {k}", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
