#!/usr/bin/env python3
"""Diagnostic: which columns of a full-size sort_match differ from a torch stable-sort restatement? (not part of the library)"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimaltextures_amd import ops
from optimaltextures_amd.ops import Seg

dev = torch.device("cuda:0")
for nt, ns in [(16384, 16384), (12544, 16384), (4096, 3072)]:
    g = torch.Generator(device=dev).manual_seed(nt + ns)
    t = torch.randn((32, 256, nt), device=dev, generator=g) * 3 + 1
    t[:, ::9].clamp_min_(0)
    t[:, 5::31] = (t[:, 5::31] * 8).round() / 8 + 0.0
    s = torch.randn((1, 256, ns), device=dev, generator=g) * 2 - 1
    out = ops.sort_match_seg(Seg.of(t), Seg.of(s))
    order = torch.sort(t, dim=2, stable=True).indices
    q = ((2 * torch.arange(nt, device=dev, dtype=torch.int64) + 1) * ns) // (2 * nt)
    want = torch.sort(s, dim=2).values[0][:, q]
    bad = (torch.gather(out, 2, order) != want[None]).sum(2)          # [32, 256]
    cols = bad.nonzero().tolist()
    print(f"nt={nt} ns={ns}: {len(cols)} bad columns of {32*256}; by channel%9==0: {sum(1 for a,c in cols if c%9==0)}, c%31==5: {sum(1 for a,c in cols if c%31==5)}, other: {sum(1 for a,c in cols if c%9 and c%31!=5)}")
    for a, c in cols[:8]:
        print("   seg", a, "col", c, "bad elements", int(bad[a, c]), "distinct values", int(torch.unique(t[a, c]).numel()), "has -0:", bool(((t[a, c] == 0) & torch.signbit(t[a, c])).any()))
