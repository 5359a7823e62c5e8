"""The two Gram kernels of linear.hip against each other at the bench's shapes (64 segments of [256, n]): the whole-triangle
kernel (one workgroup per split and segment, 36 tile products per pixel pair) and the 128 x 128 tile-pair kernel.
GPU box only:  python scripts/gram_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimaltextures_amd import _lib, ops
from optimaltextures_amd.ops import Seg
dev = torch.device("cuda:0")
flag = ctypes.c_bool.in_dll(_lib.lib(), "_ZN5optex16gram_tri_enabledE")
print("| C | n | segments | whole-triangle us | tile-pair us | TFLOP/s on C (C + 1) n flops (whole-triangle / tile-pair) |")
print("|---:|---:|---:|---:|---:|---|")
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]  # optional "C,n,S" triples instead of the default list
for C, n, S in shapes or ((256, 16384, 64), (256, 12544, 64), (256, 9216, 64), (256, 6400, 64), (256, 4096, 64), (256, 16384, 8), (256, 16384, 1),
                (224, 16384, 64), (200, 4096, 64), (184, 16384, 64), (184, 4096, 64), (168, 9216, 64), (136, 16384, 64)):
    x = torch.randn((S, C, n), device=dev).clamp_min(0)
    res = {}
    for tri in (True, False):
        flag.value = tri
        for _ in range(3):
            ops.linear_stats(Seg.of(x), pool=False)
        ops.profile_collect()
        ops.profile_enable(True)
        for _ in range(10):
            ops.linear_stats(Seg.of(x), pool=False)
        ops.profile_enable(False)
        p = ops.profile_collect()["gram"]
        res[tri] = p["ms"] * 1e3 / p["launches"]
    flag.value = True
    fl = C * (C + 1.0) * n * S
    print(f"| {C} | {n} | {S} | {res[True]:.1f} | {res[False]:.1f} | {fl / res[True] / 1e6:.1f} / {fl / res[False] / 1e6:.1f} |")
