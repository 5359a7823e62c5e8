"""MI355X counterpart of the reference's optex.py hot path:

    random_rotation(N, device="cpu", impl="scipy")                 optex.py:142-164
    optimal_transport(pastiche_feature, style_feature, hist_mode)  optex.py:167-177

Same names, argument order and defaults.  The orchestration around them (OptimalTexture, fit_pca,
mix_style_features) lives in driver.py."""
import torch
from torch import Tensor

from . import ops, rotation
from ._lib import CHANNEL_MAJOR, PIXEL_MAJOR
from .histmatch import LINEAR_MODES, linear_match_pooled
from .ops import Seg


def _device_of(device):
    if device is None or str(device) == "cpu":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def random_rotation(N: int, device: str = "cpu", impl: str = "scipy"):
    """Haar-random SO(N) matrix, fp64 [N, N] on `device` like the reference (optex.py:149 returns a CPU fp64 tensor by
    default).  The gaussian stream comes from numpy's global RandomState exactly as scipy draws it, so
    np.random.seed(s) reproduces the reference's matrices (to fp64 round-off); the Householder chain runs on the GPU.
    impl="torch" (never used by the reference, optex.py:168) draws the normals from torch's CPU generator instead."""
    dev = _device_of(device)
    if impl == "scipy":
        normals = rotation.draw_normals(N, 1)
    else:
        per = ops.rotation_normals(int(N))
        normals = torch.randn(1, per, dtype=torch.float64).numpy()
    _, _, r64 = ops.rotations_from_normals(normals, int(N), 1, dev, want64=True)
    return r64[0].to(device)


def _as_pixel_major(x: Tensor):
    """NHWC tensor -> ([B*n, C] contiguous memory, True) or its NCHW backing ([B, C, n], False) without copying"""
    b, h, w, c = x.shape
    if x.is_contiguous():
        return x.view(b * h * w, c), True
    nchw = x.permute(0, 3, 1, 2)
    if nchw.is_contiguous():
        return nchw.reshape(b, c, h * w), False
    return x.contiguous().view(b * h * w, c), True


def _rotate_to_pooled(x: Tensor, R: Tensor) -> Tensor:
    """x NHWC [B,H,W,C] -> (x @ R) as pooled channel-major rows [C, B*n] (optex.py:170-171 + histmatch.py:6-8)"""
    b, h, w, c = x.shape
    n = h * w
    out = torch.empty((c, b * n), dtype=torch.float32, device=x.device)
    mem, pixel_major = _as_pixel_major(x)
    if pixel_major:  # one segment of B*n pixels, rows of C channels
        ops.gemm_tn(R, mem, out, c, c, b * n, 1, lda=c, ldb=c, b_ss=0, b_layout=PIXEL_MAJOR, ldo=b * n, o_ss=0)
    else:            # NCHW memory: B segments of [C, n]; scatter them side by side into the pooled rows
        ops.gemm_tn(R, mem, out, c, c, n, b, lda=c, ldb=n, b_ss=c * n, b_layout=CHANNEL_MAJOR, ldo=b * n, o_ss=n)
    return out


def optimal_transport(pastiche_feature: Tensor, style_feature: Tensor, hist_mode: str):
    """One sliced-OT step: rotate both feature sets by a fresh random rotation, match the marginals, rotate back.
    NHWC fp32 in, NHWC-contiguous fp32 out (new tensor), batch items pooled like the reference."""
    b, h, w, c = pastiche_feature.shape
    bs = style_feature.shape[0]
    dev = pastiche_feature.device
    R32, Rt32 = rotation.rotations(c, 1, dev)
    R, Rt = R32[0], Rt32[0]
    rp = _rotate_to_pooled(pastiche_feature, R)
    rs = _rotate_to_pooled(style_feature, R)
    if hist_mode == "cdf":
        m = ops.cdf_match_seg(Seg.of(rp[None]), Seg.of(rs[None]), out=Seg.of(rp[None]))[0]
    elif hist_mode == "sort":
        m = ops.sort_match_seg(Seg.of(rp[None]), Seg.of(rs[None]), out=Seg.of(rp[None]))[0]
    elif hist_mode in LINEAR_MODES:
        m = linear_match_pooled(rp, b, rs, bs, hist_mode)
        if m.dim() == 3:  # B_t = 1 broadcast against B_s > 1
            b = m.shape[1]
            m = m.reshape(c, -1)
    else:
        raise ValueError(f"hist_mode must be one of chol|pca|sym|cdf|sort, got {hist_mode!r}")
    n_all = m.shape[1]
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=dev)
    ops.gemm_tn(Rt, m, out, c, c, n_all, 1, lda=c, ldb=n_all, b_ss=0, ldo=c, o_ss=0, o_layout=PIXEL_MAJOR)
    return out
