#!/usr/bin/env python3
"""Experiment: 16x16x4 vs 32x32x2 fp32 MFMA in the rotation GEMM (OPTEX_GEMM_MFMA16=0|1|2).  Prints a bit-level checksum of
the result (must be identical across variants: same k-ordered fma chains) and the sustained TFLOP/s over 600 launches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimaltextures_amd import ops, rotation  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 256, 16384), device=dev, generator=g).clamp_min_(0) * 2
R32, _ = rotation.rotations(256, 1, dev, rng=np.random.RandomState(0))
y = torch.empty_like(x)
ops.rotate_seg(x, R32[0], out=y)
chk = int(y.view(torch.int32).to(torch.int64).sum().item())
for _ in range(50):
    ops.rotate_seg(x, R32[0], out=y)
torch.cuda.synchronize()
ops.profile_collect()
ops.profile_enable(True)
for _ in range(600):
    ops.rotate_seg(x, R32[0], out=y)
torch.cuda.synchronize()
ops.profile_enable(False)
p = ops.profile_collect()["gemm_tn"]
print(f"MFMA16={os.environ.get('OPTEX_GEMM_MFMA16', '0')} checksum={chk} TFLOP/s={p['flops'] / (p['ms'] * 1e9):.2f} "
      f"avg_us={1e3 * p['ms'] / p['launches']:.1f}", flush=True)
