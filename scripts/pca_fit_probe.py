#!/usr/bin/env python3
"""Where a PCA fit (optex.py:180-190) spends its time on the GPU box, by route: torch.linalg.svd of the [N, C] matrix
(rocSOLVER gesvdj) against the Gram matrix + a symmetric eigensolver (host LAPACK at several thread counts, rocSOLVER on the
device in fp64 / fp32).   python scripts/pca_fit_probe.py"""
import os
import sys
import time

import torch
from threadpoolctl import threadpool_limits

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


print("host cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
for n, c in [(12288, 256), (49152, 128), (196608, 64)]:
    a = torch.randn(n, c, device=dev).clamp_min(0)
    a = a - a.mean()
    print(f"[{n}, {c}]")
    print("  torch.linalg.svd (gpu)        %8.2f ms" % timed(lambda: torch.linalg.svd(a, full_matrices=False)))
    print("  fp64 gram (gpu)               %8.2f ms" % timed(lambda: (a.double().t() @ a.double())))

    def split_gram():
        a64 = a.double()
        a3 = a64.view(n // 4096, 4096, c)
        return torch.bmm(a3.transpose(1, 2), a3).sum(0)

    print("  fp64 gram, batched split-K    %8.2f ms" % timed(split_gram))
    from optimaltextures_amd import driver
    cm = a.t().contiguous()[None]
    for route in ("svd", "gram"):
        driver.PCA_FIT = route
        print("  fit_pca_cm route %-4s         %8.2f ms" % (route, timed(lambda: driver.fit_pca_cm(cm))))
    g = (a.double().t() @ a.double())
    print("  gram -> host copy             %8.2f ms" % timed(lambda: g.cpu()))
    gc = g.cpu()
    print("  eigh host, default threads    %8.2f ms" % timed(lambda: torch.linalg.eigh(gc)))
    for k in (1, 4, 16):
        with threadpool_limits(limits=k):
            print("  eigh host, %2d thread(s)       %8.2f ms" % (k, timed(lambda: torch.linalg.eigh(gc))))
    print("  eigh device fp64              %8.2f ms" % timed(lambda: torch.linalg.eigh(g)))
    print("  eigh device fp32              %8.2f ms" % timed(lambda: torch.linalg.eigh(g.float())))

# all 25 fits of a five-layer, five-pass run at once: device eigh one after the other against host LAPACK, one thread per
# matrix on a thread pool (what OptimalTexture.prefetch_style_sides can do, since the style side does not depend on the pastiche)
from concurrent.futures import ThreadPoolExecutor
grams = []
for cdim in (64, 128, 256, 512, 512):
    x = torch.randn(4096, cdim, device=dev, dtype=torch.float64)
    grams += [(x.t() @ x) for _ in range(5)]
print("25 device eigh, sequential        %8.2f ms" % timed(lambda: [torch.linalg.eigh(g) for g in grams], reps=3))
host = [g.cpu() for g in grams]
pool = ThreadPoolExecutor(max_workers=25)
for lim in (1, 2, 4):
    with threadpool_limits(limits=lim):
        print("25 host eigh on a pool, %d BLAS thread(s) each %8.2f ms" % (lim, timed(lambda: list(pool.map(torch.linalg.eigh, host)), reps=3)))
with threadpool_limits(limits=1):
    print("one 512 x 512 host eigh, 1 thread  %8.2f ms" % timed(lambda: torch.linalg.eigh(host[-1]), reps=3))
print("one 512 x 512 device eigh          %8.2f ms" % timed(lambda: torch.linalg.eigh(grams[-1]), reps=3))
