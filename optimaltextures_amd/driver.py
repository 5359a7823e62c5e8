"""Orchestration around the hot path — own counterpart of the reference's OptimalTexture (optex.py:15-139), fit_pca
(optex.py:180-190) and mix_style_features (optex.py:193-206), restructured for MI355X:

  * features never leave channel-major NCHW memory ([segment][channel][pixel]); the NHWC views the reference shuffles
    around (vgg.py:153, histmatch.py:6-8,46) do not exist here, so no kernel in the loop transposes;
  * PCA project / unproject (optex.py:110,120) run on the same MFMA GEMM as the rotations;
  * all rotations of a (pass, layer) are generated in one batched device launch from one host draw of the numpy stream;
  * all iterations of a (pass, layer) are enqueued by one C call (optex_ot_loop) in every hist_mode — the linear modes'
    C x C factorizations run on the device too (csrc/linalg.hip);
  * `independent=True` turns a batch into independent textures (one segment each, shared rotations): the reference's
    --batch pools all images into ONE distribution (histmatch.py:11,17-18), which is kept as the default.
"""
from typing import List, Optional

import torch
from torch import Tensor
from torch.nn.functional import interpolate

from . import ops, rotation
from .histmatch import LINEAR_MODES, hist_match, transfer_operator
from .ops import Seg
from .util import get_iters_and_sizes, get_size, layer_iters, resize, to_nchw, to_nhwc
from .vgg import Decoder, Encoder

LOOP_MODES = ("cdf", "sort", "chol", "pca", "sym")


# ------------------------------------------------------------------------------------------------ PCA (optex.py:180-190)
# How the basis is found (both stay on PyTorch, as the north star scopes the PCA fit):
#   "gram" (default): right singular vectors = eigenvectors of the C x C Gram matrix a^T a, singular values = the square
#       roots of its eigenvalues.  The Gram matrix is taken in fp64 (a batched split-K product: rocBLAS has no split-K for a
#       64 x 64 x 196608 dgemm) and torch.linalg.eigh solves the 64..512-wide symmetric problem on the device in fp64.
#       torch.linalg.svd on ROCm runs rocSOLVER's Jacobi gesvdj on the whole [N, C] matrix — thousands of tiny launches,
#       50 / 36 / 78 ms per fit at relu3_1 / 2_1 / 1_1 (scripts/pca_fit_probe.py), 84 % of the GPU time of a single-image
#       run with the reference's default flags (profiles/r03_single_texture_kernel_summary.md); this route takes 8 / 4 / 6 ms.
#       (Host LAPACK would take 4 ms on one thread but 150 ms on the 128 threads torch picks on the GPU box's 256-core host.)
#   "svd": torch.linalg.svd of the [N, C] matrix, the literal counterpart of optex.py:183.
# Singular vectors are defined up to sign (and up to a rotation inside equal singular values) in either route and in the
# reference's LAPACK alike; the parity tests align signs before comparing (tests/test_gpu_configs.py).
PCA_FIT = "gram"


def _centred(style_cm: Tensor) -> Tensor:
    """[B, C, n] -> [B n, C] minus the GLOBAL scalar mean (optex.py:182)"""
    return style_cm.permute(0, 2, 1).reshape(-1, style_cm.shape[1]) - style_cm.mean()


def _gram64(a: Tensor) -> Tensor:
    """a^T a in fp64 as a batched split-K product (rocBLAS has no split-K for a 64 x 64 x 196608 dgemm: 21 ms against 0.3)"""
    a64 = a.double()
    rows, c = a64.shape
    chunk = 4096
    full = rows // chunk
    gram = torch.zeros((c, c), dtype=torch.float64, device=a.device)
    if full:
        a3 = a64[:full * chunk].view(full, chunk, c)
        gram += torch.bmm(a3.transpose(1, 2), a3).sum(0)
    if rows > full * chunk:
        tail = a64[full * chunk:]
        gram += tail.t() @ tail
    return gram


# The kept rank k is where the cumulative singular-VALUE share crosses 0.9 (optex.py:184-185): a discontinuous function of the
# spectrum.  When the crossing is closer than this to 0.9 the Gram route's round-off could decide k differently from the
# reference's SVD (and k sets C for the whole layer: rotation sizes, the RNG stream consumed, everything after it), so such
# a fit is redone with the literal torch.linalg.svd call.
PCA_RANK_MARGIN = 2e-6


def _kept_rank(sing: Tensor):
    """optex.py:184-185: first index whose cumulative singular-VALUE share exceeds 0.9, and the distance of the crossing from
    0.9 on either side (device scalars; batched over dim 0)"""
    share = torch.cumsum(sing / torch.sum(sing, dim=-1, keepdim=True), dim=-1)
    k = (share > 0.9).to(torch.int32).argmax(dim=-1)
    above = torch.gather(share, -1, k.long().unsqueeze(-1)).squeeze(-1) - 0.9
    below = 0.9 - torch.gather(share, -1, (k.long() - 1).clamp_min(0).unsqueeze(-1)).squeeze(-1)
    return k, torch.minimum(above, torch.where(k > 0, below, above)).to(torch.float32)


def _check_rank(k: int) -> int:
    if k < 2:
        raise ValueError(f"PCA kept {k} component(s); the rotation needs a dimension greater than 1 "
                         "(the reference fails the same way in special_ortho_group.rvs)")
    return k


def _route(route: Optional[str]) -> str:
    route = PCA_FIT if route is None else route
    if route not in ("gram", "svd"):
        raise ValueError(f"pca_fit must be 'gram' or 'svd', got {route!r}")
    return route


def fit_pca_cm(style_cm: Tensor, route: Optional[str] = None):
    """style_cm [B, C, n] -> (projected [B, k, n], eigvecs [C, k]).  Reference quirks kept: centring by the GLOBAL scalar
    mean, projecting the UNCENTRED tensor, k = first index whose cumulative *singular-value* share exceeds 0.9.
    route: "gram" / "svd" (None = the module default PCA_FIT)."""
    a = _centred(style_cm)
    gram = _route(route) == "gram" and style_cm.is_cuda
    if gram:
        lam, vec = torch.linalg.eigh(_gram64(a))                       # ascending eigenvalues, columns = eigenvectors
        sing = lam.clamp_min(0).sqrt().flip(0).to(torch.float32)       # singular values, descending (optex.py:183)
        vh = vec.flip(1).t().to(torch.float32)                         # rows = right singular vectors
    else:
        _, sing, vh = torch.linalg.svd(a, full_matrices=False)
    k, margin = _kept_rank(sing)
    k, margin = int(k.item()), float(margin.item())
    if gram and margin < PCA_RANK_MARGIN:
        return fit_pca_cm(style_cm, "svd")  # the rank is decided inside round-off: take the reference's own route
    eigvecs = vh[:_check_rank(k)].t().contiguous().to(style_cm.device)  # [C, k]
    return project_cm(style_cm, eigvecs), eigvecs


def fit_pca_many(feats: List[Tensor], route: Optional[str] = None):
    """fit_pca_cm for several feature sets at once (the style at every pass size and every layer of a forward call: the
    style side does not depend on the pastiche).  The symmetric eigenproblems of equal width are solved as ONE batched
    torch.linalg.eigh — 25 device calls one after the other take 187 ms for a five-layer run, rocSOLVER's batched solver
    works on them side by side — and all ranks k come back in one host transfer.  Returns [(projected, eigvecs), ...]."""
    if not (_route(route) == "gram" and feats and feats[0].is_cuda):
        return [fit_pca_cm(f, route) for f in feats]
    by_width = {}
    for i, f in enumerate(feats):
        by_width.setdefault(int(f.shape[1]), []).append(i)
    sing, vh = [None] * len(feats), [None] * len(feats)
    for c, idx in by_width.items():
        lam, vec = torch.linalg.eigh(torch.stack([_gram64(_centred(feats[i])) for i in idx]))
        s_all = lam.clamp_min(0).sqrt().flip(-1).to(torch.float32)
        v_all = vec.flip(-1).transpose(-1, -2).to(torch.float32)
        for j, i in enumerate(idx):
            sing[i], vh[i] = s_all[j], v_all[j]
    km = [_kept_rank(sv) for sv in sing]
    packed = torch.stack([torch.stack([k.to(torch.float32), m]) for k, m in km]).tolist()   # the one host synchronisation of all fits
    out = []
    for f, v, (k, margin) in zip(feats, vh, packed):
        if margin < PCA_RANK_MARGIN:
            out.append(fit_pca_cm(f, "svd"))
            continue
        eigvecs = v[:_check_rank(int(k))].t().contiguous()
        out.append((project_cm(f, eigvecs), eigvecs))
    return out


def fit_pca(tensor: Tensor):
    """reference signature: NHWC features -> (features @ eigvecs, eigvecs)"""
    b, h, w, c = tensor.shape
    proj, eigvecs = fit_pca_cm(to_nchw(tensor).reshape(b, c, h * w).contiguous())
    return proj.view(b, -1, h, w).permute(0, 2, 3, 1), eigvecs


def project_cm(x_cm: Tensor, eigvecs: Tensor) -> Tensor:
    """[S, C, n] -> [S, k, n]  (== x_nhwc @ eigvecs, optex.py:110)"""
    s, c, n = x_cm.shape
    k = eigvecs.shape[1]
    out = torch.empty((s, k, n), dtype=torch.float32, device=x_cm.device)
    return ops.gemm_tn(eigvecs, x_cm, out, k, c, n, s, lda=k, ldb=n, b_ss=c * n, ldo=n, o_ss=k * n)


def unproject_cm(x_cm: Tensor, eigvecs_t: Tensor) -> Tensor:
    """[S, k, n] -> [S, C, n]  (== x_nhwc @ eigvecs.T, optex.py:120); eigvecs_t = eigvecs.T contiguous [k, C]"""
    s, k, n = x_cm.shape
    c = eigvecs_t.shape[1]
    out = torch.empty((s, c, n), dtype=torch.float32, device=x_cm.device)
    return ops.gemm_tn(eigvecs_t, x_cm, out, c, k, n, s, lda=c, ldb=n, b_ss=k * n, ldo=n, o_ss=c * n)


# ------------------------------------------------------------------------------------------------ style mixing (optex.py:193-206)
def mix_style_features(style_features: List[Tensor], mixing_mask: Tensor, mixing_alpha: float, hist_mode: str):
    """NHWC style feature pairs [2,H,W,C] -> one blended target [1,H,W,C] per layer"""
    i = mixing_alpha
    for l, sf in enumerate(style_features):
        mix = to_nhwc(interpolate(mixing_mask, size=sf.shape[1:3], mode="nearest"))
        a, b = sf[[0]], sf[[1]]
        a_to_b = hist_match(a, b, mode=hist_mode)
        b_to_a = hist_match(b, a, mode=hist_mode)
        style_features[l] = (a * (1 - i) + a_to_b * i) * mix + (b_to_a * (1 - i) + b * i) * (1 - mix)
    return style_features


# ------------------------------------------------------------------------------------------------ colour helpers (kornia absent)
def rgb_to_hls(img: Tensor) -> Tensor:
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    mx, mn = img.max(1).values, img.min(1).values
    l = (mx + mn) / 2
    d = mx - mn
    s = torch.where(l < 0.5, d / (mx + mn).clamp_min(1e-12), d / (2 - mx - mn).clamp_min(1e-12))
    s = torch.where(d == 0, torch.zeros_like(s), s)
    dz = torch.where(d == 0, torch.ones_like(d), d)
    h = torch.where(mx == r, ((g - b) / dz) % 6, torch.where(mx == g, (b - r) / dz + 2, (r - g) / dz + 4))
    h = torch.where(d == 0, torch.zeros_like(h), h) * (torch.pi / 3)
    return torch.stack([h, l, s], 1)


def hls_to_rgb(img: Tensor) -> Tensor:
    h, l, s = img[:, 0] * (6 / (2 * torch.pi)), img[:, 1], img[:, 2]
    a = s * torch.minimum(l, 1 - l)

    def f(n):
        k = (n + h * 2) % 12
        return l - a * torch.clamp(torch.minimum(k - 3, 9 - k), -1, 1)

    return torch.stack([f(0), f(8), f(4)], 1)


# ------------------------------------------------------------------------------------------------ the hot loop
def ot_iterations(x: Tensor, style: Tensor, hist_mode: str, iters: int, content: Optional[Tensor] = None,
                  strength: float = 0.0, pooled: bool = False, rng=None, fuse_rotations: bool = False) -> Tensor:
    """`iters` sliced-OT steps (optex.py:112-117) on channel-major segments.  x [S, C, n] is updated in place and
    returned; style [1 or S, C, ns]; content None or [S, C, n].  pooled=True reproduces the reference's batch
    semantics (all S images form ONE distribution per channel); otherwise segments are independent textures."""
    s, c, n = x.shape
    if iters <= 0:
        return x
    if isinstance(rng, rotation.DeviceNormals):
        # the numpy stream(s) advanced on the GPU: one stream = one sequence shared by the batch, S streams = one per texture
        if rng.n != 1 and (pooled or rng.n != s):
            raise ValueError(f"per-texture rotation streams need independent textures and one stream per texture (got {rng.n} for {s})")
        if hist_mode not in LOOP_MODES:
            raise ValueError(f"hist_mode must be one of chol|pca|sym|cdf|sort, got {hist_mode!r}")
        if rng.n != 1 and hist_mode in LINEAR_MODES and c > ops.LINEAR_MAX_C:
            raise NotImplementedError(f"per-texture rotation sequences need C <= {ops.LINEAR_MAX_C} in the linear modes")
        R32, Rt32 = rng.rotations(c, iters)
        if content is not None and content.shape[0] != s:
            content = content.expand(s, c, n).contiguous()
        if rng.n != 1:
            return ops.ot_loop(hist_mode, x, style, R32, Rt32, content=content, strength=strength)
        return _shared_rotation_iterations(x, style, hist_mode, R32, Rt32, content, strength, pooled, fuse_rotations)
    if isinstance(rng, (list, tuple)):
        # one numpy stream per texture: every segment draws its own rotations, like the reference run once per image
        if pooled or len(rng) != s:
            raise ValueError(f"per-texture rotation streams need independent textures and one stream per texture (got {len(rng)} for {s})")
        if hist_mode not in LOOP_MODES or (hist_mode in LINEAR_MODES and c > ops.LINEAR_MAX_C):
            raise NotImplementedError(f"per-texture rotation sequences need a hist_mode of the fused loop and C <= {ops.LINEAR_MAX_C}")
        R32, Rt32 = rotation.rotations_per_segment(c, iters, x.device, rng)
        if content is not None and content.shape[0] != s:
            content = content.expand(s, c, n).contiguous()
        return ops.ot_loop(hist_mode, x, style, R32, Rt32, content=content, strength=strength)
    R32, Rt32 = rotation.rotations(c, iters, x.device, rng=rng)
    if content is not None and content.shape[0] != s:
        content = content.expand(s, c, n).contiguous()
    return _shared_rotation_iterations(x, style, hist_mode, R32, Rt32, content, strength, pooled, fuse_rotations)


def _shared_rotation_iterations(x, style, hist_mode, R32, Rt32, content, strength, pooled, fuse_rotations):
    """the iterations of one (pass, layer) with ONE rotation sequence for the whole batch (optex.py:168-170)"""
    s, c, n = x.shape
    iters = R32.shape[0]
    if pooled and s > 1:
        return _pooled_iterations(x, style, hist_mode, R32, Rt32, content, strength)
    if hist_mode not in LOOP_MODES:
        raise ValueError(f"hist_mode must be one of chol|pca|sym|cdf|sort, got {hist_mode!r}")
    if hist_mode not in LINEAR_MODES or c <= ops.LINEAR_MAX_C:
        return ops.ot_loop(hist_mode, x, style, R32, Rt32, content=content, strength=strength,
                           fuse_rotations=fuse_rotations)
    # linear modes wider than 512 channels: host loop around torch.linalg's factorizations
    ss = style.shape[0]
    y, ys, m = torch.empty_like(x), torch.empty_like(style), torch.empty_like(x)
    for it in range(iters):
        ops.rotate_seg(x, R32[it], out=y)           # optex.py:170
        ops.rotate_seg(style, R32[it], out=ys)      # optex.py:171
        mu_t, cov_t = ops.linear_stats(Seg.of(y), pool=False)    # histmatch.py:16-18
        mu_s, cov_s = ops.linear_stats(Seg.of(ys), pool=False)   # histmatch.py:20-22
        tt = transfer_operator(cov_t, cov_s, hist_mode, eps=1.0).mT.contiguous()  # histmatch.py:24-42 (eps * I added above)
        ops.gemm_tn(tt, y, m, c, c, n, s, lda=c, at_ss=c * c, ldb=n, b_ss=c * n, ldo=n, o_ss=c * n, bsub=mu_t, bsub_ss=c,
                    badd=mu_s, badd_ss=c if ss == s else 0)          # histmatch.py:27/34/42,44
        ops.unrotate_seg(m, Rt32[it], out=x, content=content, strength=strength)  # optex.py:175 + 115-117
    return x


def _pooled_iterations(x, style, hist_mode, R32, Rt32, content, strength):
    """reference --batch semantics: rotate NCHW segments straight into pooled rows [C, S*n], match, rotate back"""
    from .histmatch import linear_match_pooled
    s, c, n = x.shape
    ss, _, ns = style.shape
    y = torch.empty((c, s * n), dtype=torch.float32, device=x.device)
    ys = torch.empty((c, ss * ns), dtype=torch.float32, device=x.device)
    for it in range(R32.shape[0]):
        ops.gemm_tn(R32[it], x, y, c, c, n, s, lda=c, ldb=n, b_ss=c * n, ldo=s * n, o_ss=n)
        ops.gemm_tn(R32[it], style, ys, c, c, ns, ss, lda=c, ldb=ns, b_ss=c * ns, ldo=ss * ns, o_ss=ns)
        if hist_mode == "cdf":
            m = ops.cdf_match_seg(Seg.of(y[None]), Seg.of(ys[None]), out=Seg.of(y[None]))[0]
        elif hist_mode == "sort":
            m = ops.sort_match_seg(Seg.of(y[None]), Seg.of(ys[None]), out=Seg.of(y[None]))[0]
        else:
            m = linear_match_pooled(y, s, ys, ss, hist_mode)
        ops.gemm_tn(Rt32[it], m, x, c, c, n, s, lda=c, ldb=s * n, b_ss=n, ldo=n, o_ss=c * n, content=content,
                    strength=strength)
    return x


# ------------------------------------------------------------------------------------------------ OptimalTexture
class OptimalTexture(torch.nn.Module):
    """Same constructor arguments and defaults as the reference (optex.py:16-28) plus extensions:
    layers (VGG depths to run, deepest first; the reference hard-codes 5..1), models_dir (pretrained weights; None =
    seeded synthetic weights), allow_synthetic (depths whose .pth file is missing get synthetic weights instead of
    raising), independent (batch = independent textures instead of one pooled distribution), codec_layout (vgg.py), pca_fit (how fit_pca finds its basis: "gram" or the literal "svd")."""

    def __init__(self, size: int = 512, iters: int = 500, passes: int = 5, hist_mode: str = "chol",
                 color_transfer: Optional[str] = None, content_strength: float = 0.1, style_scale: float = 1,
                 mixing_alpha: float = 0.5, no_pca: bool = False, no_multires: bool = False,
                 layers=(5, 4, 3, 2, 1), models_dir: Optional[str] = None, independent: bool = False,
                 fuse_rotations: bool = False, allow_synthetic: bool = False, index_by_position: bool = False,
                 codec_layout: Optional[str] = None, pca_fit: Optional[str] = None, fold_pca: bool = False):
        super().__init__()
        self.hist_mode = hist_mode
        self.color_transfer = color_transfer
        self.content_strength = content_strength
        self.style_scale = style_scale
        self.mixing_alpha = mixing_alpha
        self.use_pca = not no_pca
        self.pca_fit = None if pca_fit is None else _route(pca_fit)  # "gram" / "svd" (optex.py:183's literal call); None = PCA_FIT
        self.independent = independent
        # optional re-association: PCA project / unproject folded into the first / last rotation of every
        # (pass, layer) — (feat @ E) @ R_0 = feat @ (E R_0) — two feature-map GEMMs fewer; independent textures with one
        # shared rotation sequence only; never the default (fp32 round-off differences)
        self.fold_pca = fold_pca
        self.fuse_rotations = fuse_rotations  # optional re-association (m @ R^T) @ R' -> m @ (R^T R'), cdf / sort only
        self.passes = passes
        self.iters_per_pass_and_layer, self.sizes = get_iters_and_sizes(size, iters, passes, not no_multires)
        self.layers = tuple(sorted({int(l) for l in layers}, reverse=True))
        # codec_layout: memory layout of the convolutions inside the fused codec path ("mixed" / "nchw", vgg.py); None = default
        self.encoders = torch.nn.ModuleList([Encoder(l, models_dir, allow_synthetic, codec_layout) for l in self.layers])
        self.decoders = torch.nn.ModuleList([Decoder(l, models_dir, allow_synthetic, codec_layout) for l in self.layers])
        # The reference indexes its schedule and its content blend by the POSITION l of an encoder in its list
        # (optex.py:112-117); with its hard-coded five-encoder list position == 5 - depth, which is what a subset of
        # layers keeps by default.  index_by_position=True reproduces a reference whose list holds only `layers`.
        self.index_by_position = index_by_position
        self.style_sync = None  # multi-GPU hook: callable(list of tensors or None) -> list of tensors (dist.py)
        # numpy RandomState for the rotations (None = numpy's global state, like the reference: ONE sequence shared by the
        # whole batch, optex.py:168-170), or a list of one RandomState per texture (independent=True): every
        # texture then draws its own rotations — the batch equals B separate runs of the reference, seed for seed
        self.rng = None
        self.rng_next = None          # the NEXT call's DeviceNormals (begin_feed() done): fed during this call's codec phases
        self.call_flags = 0            # further per-call flags of include/optex.h (ops.F_CDF_TWO_KERNEL, ops.F_SORT_RANK4), OR-ed in
        self.gemm_spare_cus = "auto"   # CUs the persistent rotation GEMM leaves free: "auto" (0 unless generator kernels may run beside an OT loop) or an int

    # -- optex.py:45-79, channel-major
    def _needs_resize(self, hw, size: int) -> bool:
        return hw[0] != size and hw[1] != size  # the reference resizes only if BOTH sides differ (optex.py:47)

    def _style_tensors(self, styles: List[Tensor], size: int, resized: bool):
        if not resized:
            return styles
        return [resize(s, size=get_size(size, self.style_scale, s.shape[2], s.shape[3])) for s in styles]

    def _compute_style_sides(self, style_tens_per_pass: List[List[Tensor]]):
        """for every pass given: per encoder the style features [n_styles, k, Hs*Ws] (channel-major), the PCA basis [C, k]
        (empty without PCA) and the feature-map size — local work, no communication.  All PCA fits of the call go through
        fit_pca_many together."""
        feats, hws = [], []
        for style_tens in style_tens_per_pass:
            for encoder in self.encoders:
                sf = torch.cat([encoder.features(s) for s in style_tens])  # [n_styles, C, Hs, Ws]
                hws.append((int(sf.shape[2]), int(sf.shape[3])))
                feats.append(sf.reshape(sf.shape[0], sf.shape[1], -1))
        if self.use_pca:
            fitted = fit_pca_many(feats, self.pca_fit)
        else:
            fitted = [(sf, torch.empty((0, 0), device=sf.device)) for sf in feats]
        n_enc, out = len(self.encoders), []
        for p in range(len(style_tens_per_pass)):
            part = fitted[p * n_enc:(p + 1) * n_enc]
            out.append(([sf.contiguous() for sf, _ in part], [e for _, e in part], hws[p * n_enc:(p + 1) * n_enc]))
        return out

    def _compute_style_side(self, style_tens: List[Tensor]):
        return self._compute_style_sides([style_tens])[0]

    def _sync_style_sides(self, sides, n_sides: int):
        """sides: list of n_sides (resized, features, eigvecs, hw) known on the source rank (None elsewhere) -> the same
        list on every rank.  ONE packed broadcast for all of them (dist.StyleSync.broadcast_packed): two messages and one
        host synchronisation per call, whatever the number of passes and layers — the header is sized from the tensor
        count, which every rank knows (2 per (pass, layer))."""
        if self.style_sync is None:
            return sides
        n_enc = len(self.encoders)
        counts = (2 * n_sides * n_enc, 1 + n_sides * (1 + 2 * n_enc))
        tensors, ints = None, None
        if self.style_sync.is_source:
            tensors, ints = [], [len(sides)]
            for resized, sf, eig, hw in sides:
                ints.append(int(resized))
                for f, e, (h, w) in zip(sf, eig, hw):
                    tensors += [f, e]
                    ints += [h, w]
        tensors, ints = self.style_sync.broadcast_packed(tensors, ints, counts=counts)
        out, ti, ii = [], 0, 1
        for _ in range(ints[0]):
            resized = bool(ints[ii])
            ii += 1
            sf, eig, hw = [], [], []
            for _ in range(n_enc):
                sf.append(tensors[ti])
                eig.append(tensors[ti + 1])
                hw.append((int(ints[ii]), int(ints[ii + 1])))
                ti += 2
                ii += 2
            out.append((resized, sf, eig, hw))
        return out

    def _style_side(self, style_tens: Optional[List[Tensor]]):
        """the style side of ONE pass (the fallback when nothing was prefetched); with a style_sync hook only its source
        rank encodes / fits, everyone receives the result"""
        need = self.style_sync is None or self.style_sync.is_source
        side = [(False,) + self._compute_style_side(style_tens)] if need else None
        return self._sync_style_sides(side, 1)[0][1:]

    def prefetch_style_sides(self, pastiche_hw, styles: List[Tensor], content: Optional[Tensor]):
        """The style side of EVERY pass, before the first one starts.  It depends on the pastiche only through its
        size, which is known in advance (each pass leaves the pastiche at its content size).  With a style_sync hook the
        source rank encodes them all and ONE exchange carries them.
        Without PCA every shape is computable on every rank (Encoder.out_shape of the — globally known — style image size):
        the exchange is one asynchronous payload broadcast per pass with NO host synchronisation on any rank
        (StyleSync.broadcast_known); `styles` must then have the true shapes on every rank.  Their CONTENT matters on the
        source rank only with the hook's default (StyleSync(spread=False)); a hook built with spread=True lets group rank
        (src + p) mod world encode pass p from ITS `styles`, so every rank must hold the real images (bench.py and the CLI
        do).  With PCA the rank k is data dependent: one int64 header (the one host synchronisation of a
        forward call on the receiving ranks) precedes the payload (StyleSync.broadcast_packed)."""
        hw, plan = (int(pastiche_hw[0]), int(pastiche_hw[1])), []
        need = self.style_sync is None or self.style_sync.is_source
        for p in range(self.passes):
            size = self.sizes[p]
            resized = self._needs_resize(hw, size)
            plan.append((size, resized))
            if resized:
                hw = (get_size(size, 1.0, content.shape[2], content.shape[3], oversize=True) if content is not None
                      else (size, size))
        if self.style_sync is None or self.use_pca:
            sides = None
            if need:
                computed = self._compute_style_sides([self._style_tensors(styles, size, resized) for size, resized in plan])
                sides = [(resized,) + side for (size, resized), side in zip(plan, computed)]
            return self._sync_style_sides(sides, self.passes)
        # Shapes from the layer lists alone: every rank knows what every exchange carries, so the style sides of the passes are
        # SPREAD over the ranks — pass p is encoded by rank p mod world and broadcast from there (round 5).  One rank encoding
        # all of them sat on the critical path of every rank at the start of a call (five one-image encodes, 2-3 ms of a
        # 47 ms step at 8 textures per GPU: BASELINE config 4); now no rank encodes more than ceil(passes / world) of them and
        # they run side by side.  Every rank issues the same exchanges in the same order (pass 0 first: it is needed first).
        sync = self.style_sync
        out = []
        for p, (size, resized) in enumerate(plan):
            shapes, hws = [], []
            for encoder in self.encoders:
                dims = [get_size(size, self.style_scale, st.shape[2], st.shape[3]) if resized else (int(st.shape[2]), int(st.shape[3]))
                        for st in styles]
                c, h, w = encoder.out_shape(*dims[0])
                assert all(encoder.out_shape(*d) == (c, h, w) for d in dims), "style images must have the same shape"
                shapes.append((len(styles), c, h * w))
                hws.append((h, w))
            src = (sync.src + p) % sync.world if sync.spread else sync.src
            flat = None
            if sync.rank == src:
                flat = self._compute_style_sides([self._style_tensors(styles, size, resized)])[0][0]
            feats = sync.broadcast_known(flat, shapes, src=src, defer=True)
            out.append((resized, feats, [torch.empty((0, 0), device=f.device) for f in feats], hws))
        sync.raise_deferred()  # a source with bad tensors has joined every exchange of the call before it raises
        return out

    def rotation_schedule(self, sides=None):
        """[(C, iterations), ...] of a forward() call in the order the loops ask for their rotations (pass-major, encoder-
        minor, the colour-transfer draw last).  Without PCA it follows from the layer lists alone; with PCA the kept ranks
        are those of `sides` (prefetch_style_sides), which must then be given."""
        schedule = []
        for p in range(self.passes):
            for li, encoder in enumerate(self.encoders):
                enc_index = li if self.index_by_position else 5 - encoder.depth
                c = int(sides[p][2][li].shape[1]) if self.use_pca else encoder.out_shape(16, 16)[0]
                schedule.append((c, layer_iters(self.iters_per_pass_and_layer, p, enc_index)))
        if self.color_transfer == "opt":
            schedule.append((3, 3))
        return schedule

    def encode_inputs(self, pastiche: Tensor, styles: List[Tensor], content: Optional[Tensor], size: int,
                      style_side=None):
        resized = self._needs_resize(pastiche.shape[-2:], size)
        if resized:
            if content is not None:
                cont_size = get_size(size, 1.0, content.shape[2], content.shape[3], oversize=True)
                cont_tens = resize(content, size=cont_size)
            else:
                cont_size, cont_tens = (size, size), None
            pastiche = resize(pastiche, size=cont_size)
        else:
            cont_tens = content
        if style_side is not None and style_side[0] == resized:
            style_side = style_side[1:]
        else:  # no prefetch, or the pastiche did not have the predicted size (same on every rank: shapes are global)
            need = self.style_sync is None or self.style_sync.is_source
            style_side = self._style_side(self._style_tensors(styles, size, resized) if need else None)
        style_features, style_eigvs, style_hw = style_side

        content_features = []
        if cont_tens is not None:
            for encoder, sf, eigvecs in zip(self.encoders, style_features, style_eigvs):
                cf = encoder.features(cont_tens)
                cf = cf.reshape(cf.shape[0], cf.shape[1], -1)
                if self.use_pca:
                    cf = project_cm(cf.contiguous(), eigvecs)
                cf = cf - cf.mean() + torch.mean(sf)  # scalar re-centring (optex.py:76)
                content_features.append(cf.contiguous())
        return pastiche, style_features, style_eigvs, content_features, style_hw

    def forward(self, pastiche: Tensor, styles: List[Tensor], content: Optional[Tensor] = None, verbose: bool = False,
                on_layer=None):
        """optex.py:81-139.  on_layer (extension): callable(pass, layer_position, image) invoked after every decoder; a
        tensor it returns replaces the image (progress previews; teacher-forced parity tests against recorded references)."""
        # multi-GPU: all style broadcasts of this call up front (see prefetch_style_sides); single GPU: pass by pass
        # with PCA: all fits of the call up front too (one batched eigensolve per layer width instead of one call per fit)
        sides = (self.prefetch_style_sides(pastiche.shape[-2:], styles, content)
                 if (self.style_sync is not None or self.use_pca) else None)
        ungated = False
        if isinstance(self.rng, rotation.DeviceNormals):
            # device-side numpy stream(s): the draws of the whole call go out now, on the generator's side stream — they
            # depend on nothing but the stream state and the (known) sizes, so they run beside the convolutions.  A caller
            # that knows its next job (bench.py, a sharded CLI run) may have enqueued them already, a call ahead
            # (rotation_schedule + DeviceNormals.prefetch): then they ran beside the PREVIOUS call and nothing is drawn here.
            schedule = self.rotation_schedule(sides)
            if not self.rng.covers(schedule):
                self.rng.prefetch(schedule)
                ungated = True   # ~12 ms of generator kernels start now, beside whatever this call does first
        # a caller that knows its NEXT job hands over that job's generator (`rng_next`, begin_feed() done): its draws are released
        # one (pass, layer) at a time at the start of this call's codec phases — beside convolutions, never beside an OT loop
        nxt = getattr(self, "rng_next", None)
        nxt = nxt if isinstance(nxt, rotation.DeviceNormals) and nxt.feeding() else None

        def feed_next():
            if nxt is not None and nxt.feeding():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(pastiche.device))
                nxt.feed_one(after=ev)

        # The persistent rotation GEMM wants every CU whole (include/optex.h, OPTEX_F_SPARE_CUS): one CU is left out of its grid
        # only when the generator's one-workgroup kernels may run beside an OT loop of this call — the un-gated prefetch above.
        # Draws that were fed during the previous call's codec phases (and this call's feeding of the next one) run beside
        # convolutions, which share a CU without harm.
        # (with a style_sync hook the later passes' RCCL broadcasts may still be in flight during the first loops: one CU stays free)
        crowded = ungated or (self.style_sync is not None and sides is not None)
        spare = (1 if crowded else 0) if self.gemm_spare_cus == "auto" else int(self.gemm_spare_cus)
        # per CALL, through the flags word of every library call of this forward() on this thread (ABI 10) — not a process-wide
        # setter any more: two OptimalTexture objects on two threads / devices do not see each other's choice
        with ops.call_flags(ops.f_spare_cus(spare) | int(self.call_flags)):
            return self._forward_passes(pastiche, styles, content, verbose, on_layer, sides, nxt, feed_next)

    def _forward_passes(self, pastiche, styles, content, verbose, on_layer, sides, nxt, feed_next):
        for p in range(self.passes):
            if verbose:
                print(f"Pass {p}, size {self.sizes[p]}")
            pastiche, style_features, style_eigvs, content_features, style_hw = self.encode_inputs(
                pastiche, styles, content, self.sizes[p], style_side=sides[p] if sides is not None else None)

            if len(styles) > 1:
                # the reference sizes the mask on the relu4_1 grid (style_features[1], optex.py:98-99)
                ref = min(1, len(style_hw) - 1)
                # drawn from torch's CPU generator (then moved): torch.manual_seed reproduces the reference CPU path's mask
                mask = torch.ceil(torch.rand(style_hw[ref]) - self.mixing_alpha)[None, None].to(pastiche.device)
                nhwc = [sf.view(sf.shape[0], sf.shape[1], *hw).permute(0, 2, 3, 1) for sf, hw in zip(style_features, style_hw)]
                mixed = mix_style_features(nhwc, mask, self.mixing_alpha, self.hist_mode)
                style_features = [to_nchw(m).reshape(1, m.shape[-1], -1).contiguous() for m in mixed]

            for li, (encoder, decoder) in enumerate(zip(self.encoders, self.decoders)):
                # position in the reference's encoder list (0 = relu5_1 in the full list)
                enc_index = li if self.index_by_position else 5 - encoder.depth
                if verbose:
                    print(f"Layer: relu{encoder.depth}_1")
                feat = encoder.features(pastiche)
                b, c, h, w = feat.shape
                x = feat.reshape(b, c, h * w)
                blend = len(content_features) > 0 and enc_index <= 2
                strength = self.content_strength / 2 ** (4 - enc_index) if blend else 0.0
                n_it = layer_iters(self.iters_per_pass_and_layer, p, enc_index)
                folded = (self.use_pca and self.fold_pca and self.independent and not self.fuse_rotations and n_it > 0 and
                          not isinstance(self.rng, (list, tuple)) and not (isinstance(self.rng, rotation.DeviceNormals) and self.rng.n != 1)
                          and self.hist_mode in LOOP_MODES and style_features[li].shape[1] <= ops.LINEAR_MAX_C)
                if folded:
                    # optex.py:110 + 112-117 + 120 in one call, projection and unprojection inside the rotations
                    k = int(style_eigvs[li].shape[1])
                    R32, Rt32 = (self.rng.rotations(k, n_it) if isinstance(self.rng, rotation.DeviceNormals)
                                 else rotation.rotations(k, n_it, x.device, rng=self.rng))
                    cf = content_features[li] if blend else None
                    if cf is not None and cf.shape[0] != b:
                        cf = cf.expand(b, k, h * w).contiguous()
                    x = ops.ot_loop_pca(self.hist_mode, x.contiguous(), style_eigvs[li], style_eigvs[li].t().contiguous(),
                                        style_features[li], R32, Rt32, content=cf, strength=strength)
                    feed_next()
                    pastiche = decoder.decode(x.view(b, -1, h, w))
                    if on_layer is not None:
                        replaced = on_layer(p, li, pastiche)
                        if replaced is not None:
                            pastiche = replaced
                    continue
                if self.use_pca:
                    x = project_cm(x.contiguous(), style_eigvs[li])
                elif not x.is_contiguous():
                    x = x.contiguous()
                x = ot_iterations(x, style_features[li], self.hist_mode, n_it,
                                  content=content_features[li] if blend else None, strength=strength,
                                  pooled=not self.independent, rng=self.rng, fuse_rotations=self.fuse_rotations)
                if self.use_pca:
                    x = unproject_cm(x, style_eigvs[li].t().contiguous())
                feed_next()
                pastiche = decoder.decode(x.view(b, -1, h, w))
                if on_layer is not None:
                    replaced = on_layer(p, li, pastiche)
                    if replaced is not None:
                        pastiche = replaced

        if nxt is not None:
            nxt.finish_feed()
            self.rng_next = None
        if self.color_transfer is not None:
            assert content is not None, "Color transfer requires content image"
            target_hls = rgb_to_hls(content)
            target_hls[:, 1] = rgb_to_hls(pastiche)[:, 1]  # swap lightness channel
            target = hls_to_rgb(target_hls)
            if self.color_transfer == "opt":
                b, _, h, w = pastiche.shape
                x = pastiche.reshape(b, 3, h * w).contiguous()
                t = target.reshape(target.shape[0], 3, -1).contiguous()
                x = ot_iterations(x, t, "cdf", 3, pooled=not self.independent, rng=self.rng)
                pastiche = x.view(b, 3, h, w)
            elif self.color_transfer == "lum":
                pastiche = target
        return pastiche
