"""Thin typed wrappers over the C ABI (include/optex.h) working on torch CUDA tensors.

Layout vocabulary: a *segment tensor* is fp32 [S, C, n] — S independent textures (segments), channel-major, n pixels
per channel contiguous: exactly NCHW memory.  Strided views of it are passed through (ld, seg_stride) without copies.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib
from ._lib import CHANNEL_MAJOR, PIXEL_MAJOR, F_CDF_TWO_KERNEL, F_DEFAULT, F_SORT_RANK4, check, f_spare_cus, ptr, stream_ptr, workspace

_tls = threading.local()


class call_flags:
    """`with ops.call_flags(f_spare_cus(0) | F_CDF_TWO_KERNEL): ...` — the `flags` word (include/optex.h, ABI 10) of every
    library call made inside the block BY THIS THREAD that does not pass `flags=` itself.  Nested blocks combine: the spare-CU
    byte of the inner one replaces the outer one's if it gives one, the other bits add up.  Nothing process-wide is touched:
    two OptimalTexture objects on two threads (two devices) run with their own choices."""

    def __init__(self, flags: int):
        self.flags = int(flags)

    def __enter__(self):
        self.prev = getattr(_tls, "flags", F_DEFAULT)
        spare = (self.flags & 0xff) or (self.prev & 0xff)
        _tls.flags = ((self.flags | self.prev) & ~0xff) | spare
        return self

    def __exit__(self, *exc):
        _tls.flags = self.prev
        return False


def _flags(flags):
    return ctypes.c_uint(getattr(_tls, "flags", F_DEFAULT) if flags is None else int(flags))

BINS = 256
LOOP_MODES = {"cdf": 0, "sort": 1, "chol": 2, "pca": 3, "sym": 4}  # optex_ot_loop / optex_transfer_operator mode codes
LINEAR_MAX_C = 512


def _f32c(t):
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype} (the reference's cdf path is fp32-only too, histmatch.py:59)")
    return t


class Seg:
    """(tensor, ld, seg_stride, n, C, n_seg) description of channel-major segments inside `base`."""

    __slots__ = ("t", "ld", "ss", "n", "C", "S")

    def __init__(self, t, ld, ss, n, C, S):
        self.t, self.ld, self.ss, self.n, self.C, self.S = t, int(ld), int(ss), int(n), int(C), int(S)

    @staticmethod
    def of(x):
        """x: contiguous [S, C, n]"""
        x = _f32c(x)
        assert x.dim() == 3 and x.is_contiguous()
        S, C, n = x.shape
        return Seg(x, n, C * n, n, C, S)

    @staticmethod
    def pooled(x, n_items):
        """x: contiguous [C, n_items * n] — the reference's hist.view(c, -1); items are sub-ranges of each row"""
        x = _f32c(x)
        assert x.dim() == 2 and x.is_contiguous()
        C, N = x.shape
        assert N % n_items == 0
        return Seg(x, N, N // n_items, N // n_items, C, n_items)


def gemm_tn(At, B, out, M, K, n, n_seg, *, lda, at_ss=0, ldb, b_ss, b_layout=CHANNEL_MAJOR, ldo, o_ss,
            o_layout=CHANNEL_MAJOR, bsub=None, bsub_ss=0, badd=None, badd_ss=0, content=None, strength=0.0, flags=None):
    check(_lib.lib().optex_gemm_tn(ptr(At), lda, at_ss, ptr(B), ldb, b_ss, b_layout, ptr(out), ldo, o_ss, o_layout,
                                   M, K, n, n_seg, ptr(bsub), bsub_ss, ptr(badd), badd_ss, ptr(content),
                                   ctypes.c_float(strength), _flags(flags), stream_ptr()))
    return out


def rotate_seg(x, R, out=None):
    """[S, C, n] channel-major:  out[s] = R^T @ x[s]   == (x_nhwc @ R) in channel-major form (optex.py:170-171)"""
    S, C, n = x.shape
    out = torch.empty_like(x) if out is None else out
    return gemm_tn(R, x, out, C, C, n, S, lda=C, ldb=n, b_ss=C * n, ldo=n, o_ss=C * n)


def unrotate_seg(m, Rt, out=None, content=None, strength=0.0):
    """out[s] = R @ m[s] == (m_nhwc @ R.T) (optex.py:175), optional content blend (optex.py:115-117).  Rt = R.T contiguous."""
    S, C, n = m.shape
    out = torch.empty_like(m) if out is None else out
    return gemm_tn(Rt, m, out, C, C, n, S, lda=C, ldb=n, b_ss=C * n, ldo=n, o_ss=C * n, content=content,
                   strength=strength)


def col_minmax(x):
    S, C, n = x.shape
    mn = torch.empty((S, C), dtype=torch.float32, device=x.device)
    mx = torch.empty_like(mn)
    check(_lib.lib().optex_col_minmax(ptr(_f32c(x)), n, C * n, n, C, S, ptr(mn), ptr(mx), stream_ptr()))
    return mn, mx


def col_histc(x, lo, hi):
    S, C, n = x.shape
    hist = torch.empty((S, C, BINS), dtype=torch.int32, device=x.device)
    check(_lib.lib().optex_col_histc(ptr(_f32c(x)), n, C * n, n, C, S, ptr(_f32c(lo).contiguous()),
                                     ptr(_f32c(hi).contiguous()), ptr(hist), stream_ptr()))
    return hist


def interp(x, xp, fp):
    x, xp, fp = _f32c(x).contiguous(), _f32c(xp).contiguous(), _f32c(fp).contiguous()
    out = torch.empty_like(x)
    check(_lib.lib().optex_interp(ptr(x), x.numel(), ptr(xp), ptr(fp), xp.numel(), ptr(out), stream_ptr()))
    return out


def cdf_match_seg(t: Seg, s: Seg, out: Seg = None, debug=False, flags=None):
    lib = _lib.lib()
    assert t.C == s.C
    if out is None:
        o = torch.empty((t.S, t.C, t.n), dtype=torch.float32, device=t.t.device)
        out = Seg.of(o)
    ws = workspace(lib.optex_cdf_ws_bytes(t.C, t.S), t.t.device)
    dbg = torch.empty((t.S, t.C, 2 + 4 * BINS), dtype=torch.float32, device=t.t.device) if debug else None
    check(lib.optex_cdf_match(ptr(t.t), t.ld, t.ss, t.n, ptr(s.t), s.ld, s.ss, s.n, s.S, t.C, t.S, ptr(out.t), out.ld,
                              out.ss, ptr(ws), ws.numel(), ptr(dbg), _flags(flags), stream_ptr()))
    if debug:
        d = dict(lo=dbg[..., 0], hi=dbg[..., 1], hist_t=dbg[..., 2:2 + BINS], hist_s=dbg[..., 2 + BINS:2 + 2 * BINS],
                 bin_edges=dbg[..., 2 + 2 * BINS:2 + 3 * BINS], remapped=dbg[..., 2 + 3 * BINS:])
        return out.t, d
    return out.t


def cdf_match_bins_seg(t: Seg, s: Seg, bins: int, out: Seg = None):
    """histmatch.py:49-69 with the reference's `bins` argument free (optex_cdf_match_bins)"""
    lib = _lib.lib()
    assert t.C == s.C
    if out is None:
        out = Seg.of(torch.empty((t.S, t.C, t.n), dtype=torch.float32, device=t.t.device))
    ws = workspace(lib.optex_cdf_bins_ws_bytes(t.C, t.S, int(bins)), t.t.device)
    check(lib.optex_cdf_match_bins(ptr(t.t), t.ld, t.ss, t.n, ptr(s.t), s.ld, s.ss, s.n, s.S, t.C, t.S, int(bins), ptr(out.t),
                                   out.ld, out.ss, ptr(ws), ws.numel(), stream_ptr()))
    return out.t


def sort_columns(x, want_keys=True, want_idx=True):
    lib = _lib.lib()
    S, C, n = x.shape
    ok = torch.empty((S, C, n), dtype=torch.float32, device=x.device) if want_keys else None
    oi = torch.empty((S, C, n), dtype=torch.int32, device=x.device) if want_idx else None
    ws = workspace(lib.optex_sort_ws_bytes(n, C, S), x.device)
    check(lib.optex_sort_columns(ptr(_f32c(x)), n, C * n, n, C, S, ptr(ok), ptr(oi), ptr(ws), ws.numel(), stream_ptr()))
    return ok, oi


def sort_match_seg(t: Seg, s: Seg, out: Seg = None, flags=None):
    lib = _lib.lib()
    assert t.C == s.C
    if out is None:
        out = Seg.of(torch.empty((t.S, t.C, t.n), dtype=torch.float32, device=t.t.device))
    ws = workspace(lib.optex_sort_match_ws_bytes(t.n, s.n, t.C, t.S, s.S), t.t.device)
    check(lib.optex_sort_match(ptr(t.t), t.ld, t.ss, t.n, ptr(s.t), s.ld, s.ss, s.n, s.S, t.C, t.S, ptr(out.t), out.ld,
                               out.ss, ptr(ws), ws.numel(), _flags(flags), stream_ptr()))
    return out.t


def linear_stats(x: Seg, pool, eps=1.0):
    """histmatch.py:16-22.  returns mu [S, C], cov [S, C, C] (pool=False) or [C, C] (pool=True)"""
    lib = _lib.lib()
    dev = x.t.device
    mu = torch.empty((x.S, x.C), dtype=torch.float32, device=dev)
    cov = torch.empty((x.C, x.C) if pool else (x.S, x.C, x.C), dtype=torch.float32, device=dev)
    ws = workspace(lib.optex_linear_stats_ws_bytes(x.n, x.C, x.S), dev)
    check(lib.optex_linear_stats(ptr(x.t), x.ld, x.ss, x.n, x.C, x.S, int(bool(pool)), ctypes.c_float(eps), ptr(mu),
                                 ptr(cov), ptr(ws), ws.numel(), stream_ptr()))
    return mu, cov


def chol_inv(a):
    """a [B, C, C] symmetric positive definite -> (U = L^T [B, C, C], Linv = L^-1 [B, C, C]) with a = L L^T
    (histmatch.py:25-27: torch.linalg.cholesky + torch.inverse of the factor), one workgroup per matrix"""
    lib = _lib.lib()
    a = _f32c(a).contiguous()
    b, c, _ = a.shape
    ld = lib.optex_chol_ld(c)
    u = torch.empty((b, ld, ld), dtype=torch.float32, device=a.device)
    li = torch.empty_like(u)
    check(lib.optex_chol_inv(ptr(a), c * c, c, b, ptr(u), ptr(li), stream_ptr()))
    return u[:, :c, :c], li[:, :c, :c]


def spd_sqrt(a, lambda_min=0.0):
    """a [B, C, C] symmetric positive definite -> (a^1/2, a^-1/2): the reference's eve @ sqrt(diag(eva)) @ eve.T
    (histmatch.py:30-31) and its inverse, by scaled Newton-Schulz iterations on the MFMA GEMM; lambda_min = a lower
    bound of the spectrum when one is known (the eps of cov + eps * I)"""
    lib = _lib.lib()
    a = _f32c(a).contiguous()
    b, c, _ = a.shape
    y, z = torch.empty_like(a), torch.empty_like(a)
    ws = workspace(lib.optex_spd_sqrt_ws_bytes(c, b), a.device)
    check(lib.optex_spd_sqrt(ptr(a), c * c, c, b, float(lambda_min), ptr(y), ptr(z), ptr(ws), ws.numel(), stream_ptr()))
    return y, z


def transfer_operator_t(cov_t, cov_s, mode, eps=0.0):
    """T^T per segment (the `At` operand of the apply GEMM): cov_t [S, C, C], cov_s [1 or S, C, C] symmetric positive
    definite (histmatch.py:24-42).  eps: a known lower bound of both spectra (the eps of `cov + eps * I` when the caller
    added it, as linear_stats does); 0 = unknown, always safe"""
    lib = _lib.lib()
    cov_t, cov_s = _f32c(cov_t).contiguous(), _f32c(cov_s).contiguous()
    s, c, _ = cov_t.shape
    ss = cov_s.shape[0]
    m = LOOP_MODES[mode]
    out = torch.empty_like(cov_t)
    ws = workspace(lib.optex_transfer_operator_ws_bytes(m, c, s, ss), cov_t.device)
    check(lib.optex_transfer_operator(m, ptr(cov_t), ptr(cov_s), c, s, ss, float(eps), ptr(out), ptr(ws), ws.numel(), stream_ptr()))
    return out


def rotation_normals(N):
    return int(_lib.load().optex_rotation_normals(int(N)))


def rotations_from_normals(normals, N, count, device, want64=False):
    """normals: host float64 array [count, N(N+1)/2-1] (numpy) or device tensor.  Returns (R32, Rt32[, R64])."""
    lib = _lib.lib()
    per = rotation_normals(N)
    if not torch.is_tensor(normals):
        normals = torch.from_numpy(np.ascontiguousarray(normals, dtype=np.float64).reshape(count, per))
    nd = normals.to(device=device, dtype=torch.float64, non_blocking=True).contiguous()
    R32 = torch.empty((count, N, N), dtype=torch.float32, device=device)
    Rt32 = torch.empty_like(R32)
    R64 = torch.empty((count, N, N), dtype=torch.float64, device=device) if want64 else None
    ws = workspace(lib.optex_rotation_ws_bytes(N, count), device)
    check(lib.optex_rotations_from_normals(ptr(nd), N, count, ptr(R64), ptr(R32), ptr(Rt32), ptr(ws), ws.numel(), stream_ptr()))
    return (R32, Rt32, R64) if want64 else (R32, Rt32)


def ot_loop(mode, x, style, R32, Rt32, content=None, strength=0.0, fuse_rotations=False, flags=None):
    """optex.py:112-117, all iterations enqueued by one C call, for every hist_mode; x [S, C, n] (independent segments) is
    updated IN PLACE.  R32 / Rt32: [iters, C, C] shared by all segments (the reference shares R across its batch), or
    [S, iters, C, C]: one rotation set per segment.  fuse_rotations = True / 1 (labelled fast paths, fp32
    round-off differences only): cdf / sort evaluate (m @ R_i^T) @ R_{i+1} as m @ (R_i^T R_{i+1}) (needs content=None);
    the linear modes run the whole step as one affine map in un-rotated space.  fuse_rotations = 3 (labelled
    too): the linear modes without a content blend run the whole CHAIN in C x C algebra — cov(x') = M cov(x) M^T follows every
    step analytically — and touch the feature map twice per call; with a content blend, or for cdf / sort, 3
    means 1."""
    lib = _lib.lib()
    S, C, n = x.shape
    Ss, Cs, ns = style.shape
    assert Cs == C and x.is_contiguous() and style.is_contiguous()
    per_seg = R32.dim() == 4
    iters = R32.shape[1] if per_seg else R32.shape[0]
    shape = (S, iters, C, C) if per_seg else (iters, C, C)
    assert R32.shape == shape and Rt32.shape == shape and R32.is_contiguous() and Rt32.is_contiguous()
    r_ss = iters * C * C if per_seg else 0
    if content is not None:
        assert content.shape == x.shape and content.is_contiguous()
    m = LOOP_MODES[mode]
    # 0 = default, 1 / True = labelled fast path, 2 = linear modes with the apply and the rotation back as separate GEMMs
    fuse = int(fuse_rotations)
    if fuse == 3 and (content is not None or m < 2):
        fuse = 1
    if content is not None and m < 2 and fuse == 1:
        fuse = 0
    ws = workspace(lib.optex_ot_loop_ws_bytes(m, n, ns, C, S, Ss, iters, fuse, r_ss), x.device)
    check(lib.optex_ot_loop(m, ptr(_f32c(x)), n, S, ptr(_f32c(style)), ns, Ss, C, ptr(R32), ptr(Rt32), r_ss, iters,
                            ptr(content), ctypes.c_float(strength), fuse, ptr(ws), ws.numel(), _flags(flags), stream_ptr()))
    return x


def ot_loop_pca(mode, x_full, eigvecs, eigvecs_t, style, R32, Rt32, content=None, strength=0.0, flags=None):
    """optex.py:110 -> 112-117 -> 120 in one C call: x_full [S, C_full, n] un-projected features, updated IN PLACE; eigvecs
    [C_full, k] and its transpose; style [Ss, k, ns] and content (None or [S, k, n]) projected already; R32 / Rt32 [iters, k, k].
    The projection rides in the first rotation and (cdf / sort, no content) the unprojection in the last (optex_ot_loop_pca)."""
    lib = _lib.lib()
    S, Cf, n = x_full.shape
    Ss, C, ns = style.shape
    iters = R32.shape[0]
    assert eigvecs.shape == (Cf, C) and eigvecs_t.shape == (C, Cf) and eigvecs.is_contiguous() and eigvecs_t.is_contiguous()
    assert x_full.is_contiguous() and style.is_contiguous() and R32.shape == (iters, C, C) and Rt32.shape == (iters, C, C)
    if content is not None:
        assert content.shape == (S, C, n) and content.is_contiguous()
    m = LOOP_MODES[mode]
    ws = workspace(lib.optex_ot_loop_pca_ws_bytes(m, n, ns, C, Cf, S, Ss, iters), x_full.device)
    check(lib.optex_ot_loop_pca(m, ptr(_f32c(x_full)), Cf, ptr(_f32c(eigvecs)), ptr(_f32c(eigvecs_t)), n, S, ptr(_f32c(style)), ns, Ss,
                                C, ptr(R32), ptr(Rt32), iters, ptr(content), ctypes.c_float(strength), ptr(ws), ws.numel(),
                                _flags(flags), stream_ptr()))
    return x_full


def vgg_glue(x, bias=None, relu=False, pool=False, up=False, pad=0, out_nhwc=False):
    """pad(up(pool(relu(x + bias)))) in one pass over a fp32 tensor of logical shape [N, C, H, W] (vgg.py's module glue,
    see include/optex.h).  x is either NCHW-contiguous or channels-last (a permuted view of [N, H, W, C] memory, what
    MIOpen's convolutions return for channels-last inputs); out_nhwc picks the layout of the result, again returned
    with the logical shape [N, C, Ho, Wo].  A layout change rides along for free."""
    x = _f32c(x)
    n, c, h, w = x.shape
    if x.is_contiguous():
        in_nhwc = False
    elif x.permute(0, 2, 3, 1).is_contiguous():
        in_nhwc = True
    else:
        x, in_nhwc = x.contiguous(), False
    hm = (h + 1) // 2 if pool else (2 * h if up else h)
    wm = (w + 1) // 2 if pool else (2 * w if up else w)
    ho, wo = hm + 2 * pad, wm + 2 * pad
    if in_nhwc and out_nhwc and c % 4:
        raise ValueError("channels-last on both sides needs C % 4 == 0")
    if out_nhwc:
        buf = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
        out = buf.permute(0, 3, 1, 2)
    else:
        buf = out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = _f32c(bias).contiguous()
        assert bias.numel() == c
    check(_lib.lib().optex_vgg_glue_layout(ptr(x), ptr(bias), ptr(buf), n, c, h, w, int(relu), int(pool), int(up),
                                           int(pad), int(in_nhwc), int(out_nhwc), stream_ptr()))
    return out


def profile_enable(on=True):
    check(_lib.lib().optex_prof_enable(int(bool(on))))


def profile_collect():
    """{class name: dict(ms, launches, flops, bytes)} measured with HIP events on the launch stream; resets the tallies"""
    lib = _lib.lib()
    k = lib.optex_prof_num_classes()
    ms, fl, by = (ctypes.c_double * k)(), (ctypes.c_double * k)(), (ctypes.c_double * k)()
    ln = (ctypes.c_longlong * k)()
    check(lib.optex_prof_collect(k, ms, ln, fl, by))
    return {lib.optex_prof_class_name(i).decode(): dict(ms=ms[i], launches=int(ln[i]), flops=fl[i], bytes=by[i])
            for i in range(k) if ln[i]}


def gemm_spare_cus(spare: int) -> int:
    """DEPRECATED (ABI 10): the process-wide DEFAULT of the CUs the persistent rotation GEMM leaves out of its grid — what a call
    without f_spare_cus(n) in its flags gets (include/optex.h, optex_gemm_spare_cus).  Use `flags=` / `ops.call_flags`.
    Returns the previous value."""
    return int(_lib.load().optex_gemm_spare_cus(int(spare)))


def cdf_fused(on: bool) -> bool:
    """DEPRECATED (ABI 10): the process-wide default of calls without F_CDF_TWO_KERNEL (include/optex.h, optex_cdf_fused).  Use
    `flags=F_CDF_TWO_KERNEL` / `ops.call_flags`.  Returns the previous setting."""
    return bool(_lib.load().optex_cdf_fused(1 if on else 0))
