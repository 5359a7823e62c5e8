"""How many Newton-Schulz iterations run (ns_init_kernel decides on the device) and what a batch of 64 square roots costs
with the count decided on the device vs all 12 enqueued iterations.  GPU box only:  python scripts/ns_count_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimaltextures_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
flag = ctypes.c_bool.in_dll(_lib.lib(), "_ZN5optex11ns_adaptiveE")
rng = np.random.default_rng(0)
C, B = 256, 64
q, _ = np.linalg.qr(rng.standard_normal((C, C)))
print("| lambda_max (lambda_min = 1) | |A|_F | device-decided us | all 12 us | max rel. difference of the roots |")
print("|---:|---:|---:|---:|---:|")
for top in (1.7, 20.0, 200.0, 2e3, 2e4, 1e6):
    w = np.concatenate([[1.0], np.geomspace(1.0, top, C - 1)])
    A = torch.from_numpy(np.repeat(((q * w) @ q.T).astype(np.float32)[None], B, 0)).to(dev)
    res = {}
    for adaptive in (True, False):
        flag.value = adaptive
        for _ in range(3):
            Y, Z = ops.spd_sqrt(A, lambda_min=1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            Y, Z = ops.spd_sqrt(A, lambda_min=1.0)
        torch.cuda.synchronize()
        res[adaptive] = ((time.perf_counter() - t0) / 20 * 1e6, Y.clone(), Z.clone())
    flag.value = True
    d = max(((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max()).item(),
            ((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max()).item())
    print(f"| {top:g} | {np.linalg.norm(w):.1f} | {res[True][0]:.0f} | {res[False][0]:.0f} | {d:.1e} |")
