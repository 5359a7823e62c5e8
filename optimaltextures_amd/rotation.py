"""Host side of the SO(N) generator (optex.py:142-149): the numpy-legacy gaussian stream that scipy's
special_ortho_group.rvs consumes.  Drawing from numpy's GLOBAL RandomState (the default) reproduces the reference's
matrices after np.random.seed(s); the O(N^3) Householder accumulation runs on the GPU (csrc/rotation.hip)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import ops

_pool = None


def draw_normals(N: int, count: int = 1, rng=None) -> np.ndarray:
    """[count, N(N+1)/2 - 1] float64: per rotation, the concatenated normal(size=N-n) draws for n = 0..N-2.
    RandomState.normal is a sequential stream (the one-value gaussian cache carries over), so one big draw equals
    scipy's N-1 small ones."""
    if N is None or not np.isscalar(N) or N <= 1 or N != int(N):
        raise ValueError("Dimension of rotation must be specified,\n and must be a scalar greater than 1.")
    per = ops.rotation_normals(int(N))
    src = np.random if rng is None else rng
    return src.normal(size=(count, per))


def rotations(N: int, count: int, device, rng=None, want64=False):
    """count Haar-random rotations as device tensors: (R32 [count,N,N], Rt32 transposes[, R64])"""
    return ops.rotations_from_normals(draw_normals(N, count, rng), int(N), count, device, want64=want64)


def rotations_per_segment(N: int, count: int, device, rngs):
    """One rotation sequence PER SEGMENT (the reference run once per image draws its own rotations, optex.py:149,168):
    rngs = one numpy RandomState per segment, segment s gets the `count` rotations its stream yields — exactly what a
    B = 1 run seeded like rngs[s] would draw.  Returns (R32 [S, count, N, N], Rt32).  The streams are sequential by
    construction (MT19937 + polar method with a one-value cache), so the host draw is the cost of this mode: S * count *
    (N (N + 1) / 2 - 1) normals, drawn on a thread pool (numpy's legacy generator releases the GIL)."""
    global _pool
    S = len(rngs)
    per = ops.rotation_normals(int(N))
    if _pool is None:
        _pool = ThreadPoolExecutor(max_workers=max(1, min(64, os.cpu_count() or 1)))
    normals = np.empty((S, count, per), dtype=np.float64)

    def draw(i):
        normals[i] = rngs[i].normal(size=(count, per))

    list(_pool.map(draw, range(S)))
    R32, Rt32 = ops.rotations_from_normals(normals.reshape(S * count, per), int(N), S * count, device)
    return R32.view(S, count, N, N), Rt32.view(S, count, N, N)
