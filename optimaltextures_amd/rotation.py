"""Host side of the SO(N) generator (optex.py:142-149): the numpy-legacy gaussian stream that scipy's
special_ortho_group.rvs consumes.  Drawing from numpy's GLOBAL RandomState (the default) reproduces the reference's
matrices after np.random.seed(s); the O(N^3) Householder accumulation runs on the GPU (csrc/rotation.hip)."""
import numpy as np

from . import ops


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
