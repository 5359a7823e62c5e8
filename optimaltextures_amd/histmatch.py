"""MI355X counterpart of the reference's histmatch.py — same three entry points, same argument meaning:

    hist_match(target, source, mode="chol", eps=1)   histmatch.py:5-46   (modes chol | pca | sym | cdf, + "sort")
    cdf_match(target, source, bins=256)              histmatch.py:49-69
    interp(x, xp, fp)                                histmatch.py:72-92

Tensors are CUDA fp32.  All work runs in the HIP kernels of liboptex_hip.so, the C x C factorizations of the linear modes
included (csrc/linalg.hip: batched Cholesky + inverse, Newton-Schulz square roots); matrices wider than 512 channels
fall back to torch.linalg (rocSOLVER).
"""
import torch
from torch import Tensor

from . import ops
from .ops import Seg

LINEAR_MODES = ("chol", "pca", "sym")


def _pooled_cm(x: Tensor) -> Tensor:
    """NHWC [B,H,W,C] -> contiguous [C, B*H*W]: the reference's `x.permute(3,0,1,2).view(c,-1)` (histmatch.py:6-8,11).
    Zero-copy when x is an NHWC view of NCHW memory with B == 1 (what the VGG encoder returns, vgg.py:153)."""
    c = x.shape[-1]
    return x.permute(3, 0, 1, 2).reshape(c, -1).contiguous()


def _spd_sqrt_pair(cov: Tensor):
    """Q = V sqrt(L) V^T and Q^-1 = V L^-1/2 V^T of a symmetric PD matrix (histmatch.py:30-31)"""
    w, v = torch.linalg.eigh(cov, UPLO="U")
    r = w.sqrt()
    return (v * r.unsqueeze(-2)) @ v.mT, (v / r.unsqueeze(-2)) @ v.mT


def transfer_operator(cov_t: Tensor, cov_s: Tensor, mode: str, eps: float = 0.0) -> Tensor:
    """T with matched = T @ hist_t (histmatch.py:24-42); cov_* are [..., C, C] symmetric positive definite (batched over
    independent segments).  eps is NOT added here: it is a known LOWER BOUND of both spectra — the eps of a
    `cov + eps * I` the caller formed — and only tunes the scaling of the Newton-Schulz square roots (pca / sym).  The
    default 0 means "unknown" (the bound is then taken from the matrix norm, slower convergence but always safe); a bound
    larger than the true smallest eigenvalue would give a wrong root, so pass it only where eps * I was really added."""
    if mode not in LINEAR_MODES:
        raise ValueError(f"unknown linear mode {mode!r}")
    c = cov_t.shape[-1]
    if c <= ops.LINEAR_MAX_C:
        ct, cs = cov_t.reshape(-1, c, c), cov_s.reshape(-1, c, c)
        return ops.transfer_operator_t(ct, cs, mode, eps).mT.reshape(cov_t.shape)
    return _transfer_operator_torch(cov_t, cov_s, mode)


def _transfer_operator_torch(cov_t: Tensor, cov_s: Tensor, mode: str) -> Tensor:
    """the same operators on torch.linalg (rocSOLVER) for C > 512"""
    if mode == "chol":
        lt, ls = torch.linalg.cholesky(cov_t), torch.linalg.cholesky(cov_s)
        return torch.linalg.solve_triangular(lt, ls, upper=False, left=False)  # L_s @ L_t^-1
    if mode == "pca":
        _, qt_inv = _spd_sqrt_pair(cov_t)
        qs, _ = _spd_sqrt_pair(cov_s)
        return qs @ qt_inv
    if mode == "sym":
        qt, qt_inv = _spd_sqrt_pair(cov_t)
        mid, _ = _spd_sqrt_pair(qt @ cov_s @ qt)
        return qt_inv @ mid @ qt_inv
    raise ValueError(f"unknown linear mode {mode!r}")


def linear_match_pooled(t_cm: Tensor, bt: int, s_cm: Tensor, bs: int, mode: str, eps: float = 1.0):
    """histmatch.py:16-44 on pooled channel-major rows: t_cm [C, bt*n], s_cm [C, bs*ns].  Returns [C, bt*n] (or the
    broadcast [C, bs, n] result when bt == 1 < bs, like the reference)."""
    c, nt = t_cm.shape
    n = nt // bt
    mu_t, cov_t = ops.linear_stats(Seg.pooled(t_cm, bt), pool=True, eps=eps)
    mu_s, cov_s = ops.linear_stats(Seg.pooled(s_cm, bs), pool=True, eps=eps)
    if bs != bt and bs != 1 and bt != 1:
        raise RuntimeError(f"The size of tensor a ({bt}) must match the size of tensor b ({bs}) at non-singleton dimension 1")
    if c <= ops.LINEAR_MAX_C:  # At[k][m] = T[m][k], straight from the device factorization
        Tt = ops.transfer_operator_t(cov_t[None], cov_s[None], mode, eps)[0]
    else:
        Tt = _transfer_operator_torch(cov_t, cov_s, mode).mT.contiguous()
    out = torch.empty_like(t_cm)
    late_bias = bt == 1 and bs > 1
    ops.gemm_tn(Tt, t_cm, out, c, c, n, bt, lda=c, ldb=nt, b_ss=n, ldo=nt, o_ss=n, bsub=mu_t, bsub_ss=c,
                badd=None if late_bias else mu_s, badd_ss=c if bs == bt else 0)
    if late_bias:  # B_t = 1, B_s = 2 silently broadcasts to a B = 2 output in the reference
        return out.view(c, 1, n) + mu_s.t().reshape(c, bs, 1)
    return out


def hist_match(target: Tensor, source: Tensor, mode: str = "chol", eps: float = 1):
    """Match the per-channel distribution of `target` ([B,H,W,C]) to `source`; batch items are POOLED exactly like the
    reference (histmatch.py:11,17-18).  Returns an NHWC view of channel-major memory, as the reference does."""
    b, h, w, c = target.shape
    bs = source.shape[0]
    t_cm, s_cm = _pooled_cm(target), _pooled_cm(source)
    if mode == "cdf":
        out = ops.cdf_match_seg(Seg.of(t_cm[None]), Seg.of(s_cm[None]))[0]
    elif mode == "sort":
        out = ops.sort_match_seg(Seg.of(t_cm[None]), Seg.of(s_cm[None]))[0]
    elif mode in LINEAR_MODES:
        out = linear_match_pooled(t_cm, b, s_cm, bs, mode, float(eps))
        if out.dim() == 3:
            b = out.shape[1]
    else:
        raise ValueError(f"hist_mode must be one of chol|pca|sym|cdf|sort, got {mode!r}")
    return out.reshape(c, b, h, w).permute(1, 2, 3, 0)


def cdf_match(target: Tensor, source: Tensor, bins: int = 256):
    """target [C, Nt], source [C, Ns] -> [C, Nt]  (histmatch.py:49-69)"""
    if bins != ops.BINS:  # no caller inside the reference; one workgroup per column (optex_cdf_match_bins)
        if int(bins) != bins or bins < 1:
            raise ValueError(f"bins must be a positive integer, got {bins!r}")
        return ops.cdf_match_bins_seg(Seg.of(target.contiguous()[None]), Seg.of(source.contiguous()[None]), int(bins))[0]
    return ops.cdf_match_seg(Seg.of(target.contiguous()[None]), Seg.of(source.contiguous()[None]))[0]


def interp(x: Tensor, xp: Tensor, fp: Tensor):
    """The reference's right-anchored interpolation with its non-finite fallback (histmatch.py:72-92), any 1-D sizes."""
    return ops.interp(x.reshape(-1), xp, fp).reshape(x.shape)
