"""optex_ot_loop in the linear modes at the bench's batch (64 textures of [256, n]) without the codec: ms per OT iteration.
GPU box only:  python scripts/sym_loop_probe.py [n]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimaltextures_amd import ops, rotation
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S, C, iters = 64, 256, 8
x0 = torch.randn((S, C, n), device=dev).clamp_min(0) * 2
sty = torch.randn((1, C, n * 3 // 4), device=dev).clamp_min(0) * 3
R, Rt = rotation.rotations(C, iters, dev, rng=np.random.RandomState(1))
for mode in ("chol", "pca", "sym"):
    for fused in (0, 3):
        x = x0.clone()
        ops.ot_loop(mode, x, sty, R, Rt, fuse_rotations=fused)
        torch.cuda.synchronize()
        x = x0.clone()
        t0 = time.perf_counter()
        ops.ot_loop(mode, x, sty, R, Rt, fuse_rotations=fused)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / iters
        print(f"{mode} fuse_rotations={fused} n={n}: {ms:.3f} ms per iteration, finite={bool(torch.isfinite(x).all())}", flush=True)
