"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU oracle and against the
golden vectors captured from the reference.

Bars:
  * integer / index / elementwise work (min-max, histogram counts, bin edges, LUT, interp, cdf_match, sort indices):
    BIT-EXACT against the oracle and the reference goldens on identical inputs;
  * the rotation GEMM is an fp32 fma chain in k order on both sides (MFMA == fmaf), so HIP == oracle bit-exactly, and
    within 2e-5 * max|ref| of the reference's BLAS GEMM;
  * linear modes: <= 1e-4 * max|ref| per step, <= 1e-3 over a 13-step chain (fp32 LAPACK vs rocSOLVER vs fp64 oracle);
  * cdf through our own GEMM vs the reference output: >= 99.5 % of elements within 1e-4 * range (the map is discontinuous).
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

GEMM_TOL, LIN_TOL, CHAIN_TOL = 2e-5, 1e-4, 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def maxrel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def biteq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def relu_feat(rng, *shape, scale=1.0, shift=0.0):
    return np.maximum(rng.standard_normal(shape) * scale + shift, 0).astype(np.float32)


# ================================================================================================ K1 GEMM
@pytest.mark.parametrize("C,n,S", [(3, 240, 1), (16, 576, 2), (23, 225, 1), (64, 4096, 1), (181, 1000, 2),
                                   (256, 4096, 1), (256, 16384, 2), (512, 1024, 1)])
def test_gemm_channel_major_bit_exact(dev, C, n, S):
    from optimaltextures_amd import ops
    rng = np.random.default_rng(C * 7 + n)
    x = (rng.standard_normal((S, C, n)) * 3).astype(np.float32)
    R = orc.random_rotation(C, orc.LegacyRNG(C)).astype(np.float32)
    y = ops.rotate_seg(cu(x, dev), cu(R, dev)).cpu().numpy()
    for s in range(S):
        assert biteq(y[s], orc.rotate_cm(x[s], R)), f"segment {s}"
    back = ops.unrotate_seg(cu(y, dev), cu(np.ascontiguousarray(R.T), dev)).cpu().numpy()
    for s in range(S):
        assert biteq(back[s], orc.unrotate_cm(y[s], R))
    assert maxrel(back, x) < 1e-5  # rotation round trip
    ref = np.einsum("skn,kc->scn", x.astype(np.float64), R.astype(np.float64))
    assert maxrel(y, ref) < GEMM_TOL


@pytest.mark.parametrize("C,n", [(3, 77), (16, 576), (23, 1024), (181, 900), (256, 4096)])
def test_gemm_pixel_major_layouts_bit_exact(dev, C, n):
    """the boundary variants: NHWC-contiguous in (optex.py:170) and NHWC-contiguous out (optex.py:175)"""
    from optimaltextures_amd import ops
    from optimaltextures_amd._lib import PIXEL_MAJOR
    rng = np.random.default_rng(C + n)
    xp = (rng.standard_normal((n, C)) * 2).astype(np.float32)  # pixel-major
    R = orc.random_rotation(C, orc.LegacyRNG(C + 1)).astype(np.float32)
    want = orc.rotate_cm(np.ascontiguousarray(xp.T), R)
    out = torch.empty((C, n), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(R, dev), cu(xp, dev), out, C, C, n, 1, lda=C, ldb=C, b_ss=0, b_layout=PIXEL_MAJOR, ldo=n, o_ss=0)
    assert biteq(out.cpu().numpy(), want)
    outp = torch.empty((n, C), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(R, dev), cu(want, dev), outp, C, C, n, 1, lda=C, ldb=n, b_ss=0, ldo=C, o_ss=0, o_layout=PIXEL_MAJOR)
    assert biteq(outp.cpu().numpy().T, orc.gemm_tn(R, want))
    outpp = torch.empty((n, C), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(R, dev), cu(xp, dev), outpp, C, C, n, 1, lda=C, ldb=C, b_ss=0, b_layout=PIXEL_MAJOR, ldo=C, o_ss=0,
                o_layout=PIXEL_MAJOR)
    assert biteq(outpp.cpu().numpy().T, want)


def test_gemm_bias_and_content_blend_bit_exact(dev):
    from optimaltextures_amd import ops
    rng = np.random.default_rng(5)
    S, C, n = 2, 48, 640
    x = rng.standard_normal((S, C, n)).astype(np.float32)
    At = rng.standard_normal((S, C, C)).astype(np.float32)  # per-segment operator (linear modes)
    bsub = rng.standard_normal((S, C)).astype(np.float32)
    badd = rng.standard_normal((S, C)).astype(np.float32)
    content = rng.standard_normal((S, C, n)).astype(np.float32)
    out = torch.empty((S, C, n), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(At, dev), cu(x, dev), out, C, C, n, S, lda=C, at_ss=C * C, ldb=n, b_ss=C * n, ldo=n, o_ss=C * n,
                bsub=cu(bsub, dev), bsub_ss=C, badd=cu(badd, dev), badd_ss=C, content=cu(content, dev), strength=0.05)
    got = out.cpu().numpy()
    for s in range(S):
        want = orc.content_blend(orc.gemm_tn(At[s], x[s], bsub[s], badd[s]), content[s], 0.05)
        assert biteq(got[s], want)


def test_gemm_hot_loop_tile_with_bias_and_blend_bit_exact(dev):
    """the 16x16x4-MFMA hot-loop kernel (M = K = 256, whole pixel tiles, >= 2 tiles per CU) also carries the centring, the
    bias and the content blend: per-segment operators as in the linear modes' apply step, bit-exact vs the oracle"""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(6)
    S, C, n = 4, 256, 16384
    x = rng.standard_normal((S, C, n)).astype(np.float32)
    At = (rng.standard_normal((S, C, C)) / 16).astype(np.float32)
    bsub = rng.standard_normal((S, C)).astype(np.float32)
    badd = rng.standard_normal((S, C)).astype(np.float32)
    content = rng.standard_normal((S, C, n)).astype(np.float32)
    out = torch.empty((S, C, n), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(At, dev), cu(x, dev), out, C, C, n, S, lda=C, at_ss=C * C, ldb=n, b_ss=C * n, ldo=n, o_ss=C * n,
                bsub=cu(bsub, dev), bsub_ss=C, badd=cu(badd, dev), badd_ss=C, content=cu(content, dev), strength=0.05)
    got = out.cpu().numpy()
    for s in (0, 3):
        want = orc.content_blend(orc.gemm_tn(At[s], x[s], bsub[s], badd[s]), content[s], 0.05)
        assert biteq(got[s], want)
    out2 = torch.empty_like(out)  # blend only (the inverse rotation of style transfer)
    ops.gemm_tn(cu(At[0], dev), cu(x, dev), out2, C, C, n, S, lda=C, ldb=n, b_ss=C * n, ldo=n, o_ss=C * n,
                content=cu(content, dev), strength=0.0125)
    assert biteq(out2.cpu().numpy()[2], orc.content_blend(orc.gemm_tn(At[0], x[2]), content[2], 0.0125))


# ================================================================================================ K2a/K2b stages
def test_minmax_and_histc_stage_known_answers(dev, golden):
    from optimaltextures_amd import ops
    g = golden("cdf_match.npz")
    t, s = g["target"], g["source"]
    mn_t, mx_t = ops.col_minmax(cu(t[None], dev))
    mn_s, mx_s = ops.col_minmax(cu(s[None], dev))
    lo, hi = torch.minimum(mn_t, mn_s), torch.maximum(mx_t, mx_s)
    assert biteq(lo[0].cpu().numpy(), g["lo"]) and biteq(hi[0].cpu().numpy(), g["hi"])
    ht = ops.col_histc(cu(t[None], dev), lo, hi)[0].cpu().numpy()
    hs = ops.col_histc(cu(s[None], dev), lo, hi)[0].cpu().numpy()
    assert biteq(ht.astype(np.float32), g["hist_t"]) and biteq(hs.astype(np.float32), g["hist_s"])
    assert ht.sum(1).tolist() == [t.shape[1]] * t.shape[0]  # checksum of counts


def test_histc_raw_semantics_vs_torch_golden(dev, golden):
    from optimaltextures_amd import ops
    g = golden("histc_linspace.npz")
    x = cu(g["x"][None], dev)  # [1, 12, 5000]
    lo, hi = cu(g["lo"][None], dev), cu(g["hi"][None], dev)
    h = ops.col_histc(x, lo, hi)[0].cpu().numpy().astype(np.float32)
    assert biteq(h, g["hist"])


def test_minmax_long_columns_atomic_path(dev):
    from optimaltextures_amd import ops
    rng = np.random.default_rng(9)
    x = rng.standard_normal((1, 3, 300_001)).astype(np.float32)  # forces chunks > 1 and the scalar tail
    x[0, 1, 299_999] = -77.5
    x[0, 2, 4] = 91.25
    mn, mx = ops.col_minmax(cu(x, dev))
    assert biteq(mn.cpu().numpy(), x.min(-1)) and biteq(mx.cpu().numpy(), x.max(-1))
    lo, hi = mn, mx
    h = ops.col_histc(cu(x, dev), lo, hi)[0].cpu().numpy()
    for c in range(3):
        assert biteq(h[c].astype(np.float32), orc.histc(x[0, c], x[0, c].min(), x[0, c].max()))


# ================================================================================================ A8 interp
def test_interp_known_answers_and_random(dev, golden):
    import optimaltextures_amd as ot
    g = golden("interp.npz")
    for k in ("ka1", "ka2", "rand0", "rand1", "rand2", "rand3", "rand4", "rand5"):
        out = ot.interp(cu(g[f"{k}_x"], dev), cu(g[f"{k}_xp"], dev), cu(g[f"{k}_fp"], dev)).cpu().numpy()
        assert biteq(out, g[f"{k}_out"]), k


# ================================================================================================ A6 cdf_match
def test_cdf_match_golden_bit_exact_with_intermediates(dev, golden):
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    g = golden("cdf_match.npz")
    out, d = ops.cdf_match_seg(Seg.of(cu(g["target"][None], dev)), Seg.of(cu(g["source"][None], dev)), debug=True)
    for k in ("lo", "hi", "hist_t", "hist_s", "bin_edges", "remapped"):
        assert biteq(d[k][0].cpu().numpy(), g[k]), k
    assert biteq(out[0].cpu().numpy(), g["out"])


def test_cdf_match_public_api_golden(dev, golden):
    import optimaltextures_amd as ot
    g = golden("cdf_match.npz")
    for a, b, o in (("target", "source", "out"), ("target2", "source2", "out2"), ("target3", "source3", "out3")):
        assert biteq(ot.cdf_match(cu(g[a], dev), cu(g[b], dev)).cpu().numpy(), g[o]), o
    lin = torch.linspace(0, 4, 80)[None]
    assert biteq(ot.cdf_match(torch.full((1, 64), 2.0, device=dev), lin.to(dev)).cpu().numpy(), g["deg_const_t_out"])
    assert biteq(ot.cdf_match(torch.linspace(0, 1, 64)[None].to(dev), torch.full((1, 80), 0.5, device=dev)).cpu().numpy(),
                 g["deg_const_s_out"])
    assert biteq(ot.cdf_match(torch.full((1, 64), 3.0, device=dev), torch.full((1, 80), 3.0, device=dev)).cpu().numpy(),
                 g["deg_both_out"])
    with pytest.raises(ValueError):
        ot.cdf_match(cu(g["target"], dev), cu(g["source"], dev), bins=0)


def test_cdf_match_other_bin_counts_golden(dev, golden):
    """histmatch.py:49 with `bins` free (optex_cdf_match_bins): the reference's outputs for 11 bin counts, LDS- and
    workspace-resident tables both"""
    import optimaltextures_amd as ot
    g = golden("cdf_match_bins.npz")
    for b in g["bins"]:
        out = ot.cdf_match(cu(g["target"], dev), cu(g["source"], dev), bins=int(b)).cpu().numpy()
        assert biteq(out, g[f"out_{b}"]), int(b)
    both = ot.cdf_match(torch.full((1, 64), 3.0, device=dev), torch.full((1, 80), 3.0, device=dev), bins=7)
    assert biteq(both.cpu().numpy(), g["deg_both_out_7"])


@pytest.mark.parametrize("bins", [5, 64, 256, 777, 4096])
@pytest.mark.parametrize("S,Ss,C,nt,ns", [(1, 1, 16, 4096, 3000), (3, 1, 8, 1000, 1500), (2, 2, 5, 777, 640)])
def test_cdf_match_bins_segments_vs_oracle_bit_exact(dev, bins, S, Ss, C, nt, ns):
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(S * 100 + C + bins)
    t = (rng.standard_normal((S, C, nt)) * rng.uniform(0.5, 4, (S, C, 1)) + rng.uniform(-2, 2, (S, C, 1))).astype(np.float32)
    s = relu_feat(rng, Ss, C, ns, scale=2.0, shift=0.5)
    t[0, 0] = np.maximum(t[0, 0], 0)  # ties
    # strided views: rows longer than the columns
    tp = torch.zeros((S, C, nt + 12), device=dev)
    tp[..., :nt] = cu(t, dev)
    out = ops.cdf_match_bins_seg(Seg(tp, nt + 12, C * (nt + 12), nt, C, S), Seg.of(cu(s, dev)), bins).cpu().numpy()
    for k in range(S):
        assert biteq(out[k], orc.cdf_match_bins(t[k], s[k if Ss > 1 else 0], bins)), (k, bins)
    if bins == 256:
        assert biteq(out, ops.cdf_match_seg(Seg.of(cu(t, dev)), Seg.of(cu(s, dev))).cpu().numpy())


@pytest.mark.parametrize("S,Ss,C,nt,ns", [(1, 1, 16, 4096, 3000), (3, 1, 8, 1000, 1500), (2, 2, 5, 777, 640),
                                           (1, 1, 4, 70001, 1234)])
def test_cdf_match_segments_vs_oracle_bit_exact(dev, S, Ss, C, nt, ns):
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(S * 100 + C)
    t = (rng.standard_normal((S, C, nt)) * rng.uniform(0.5, 4, (S, C, 1)) + rng.uniform(-2, 2, (S, C, 1))).astype(np.float32)
    s = relu_feat(rng, Ss, C, ns, scale=2.0, shift=0.5)
    t[0, 0] = np.maximum(t[0, 0], 0)  # ties
    out = ops.cdf_match_seg(Seg.of(cu(t, dev)), Seg.of(cu(s, dev))).cpu().numpy()
    for k in range(S):
        assert biteq(out[k], orc.cdf_match(t[k], s[k if Ss > 1 else 0])), f"segment {k}"


@pytest.mark.parametrize("nt,ns", [(1024, 800), (2044, 3000), (4096, 3072), (6400, 4800), (9216, 6912), (12544, 9408), (16384, 12288),
                                    (16380, 20000), (8, 4)])
def test_cdf_match_fused_kernel_equals_two_kernel_pipeline_and_oracle(dev, nt, ns):
    """the one-launch matcher with the column in registers (ABI 9) against the histogram + apply pipeline it
    replaces (every NV instantiation, ragged last vectors, in place, shared and per-segment sources), and against the oracle:
    non-finite values, constant columns, ties, values on bin edges included"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(nt + ns)
    S, C = 3, 6
    t = (rng.standard_normal((S, C, nt)) * rng.uniform(0.5, 4, (S, C, 1)) + rng.uniform(-2, 2, (S, C, 1))).astype(np.float32)
    t[0, 0] = np.maximum(t[0, 0], 0)                     # ties
    t[0, 1] = 1.25                                       # constant column
    t[1, 2, ::7] = np.nan                                # histc skips them, interp propagates them
    t[1, 3, 5] = np.inf
    t[2, 4] = np.round(t[2, 4] * 8) / 8                  # many values exactly on bin edges
    for Ss in (1, S):
        s = relu_feat(rng, Ss, C, ns, scale=2.0, shift=0.5)
        s[0, 1] = 1.25 if Ss == 1 else 3.0
        td, sd = cu(t, dev), cu(s, dev)
        # per call (ABI 10): OPTEX_F_CDF_TWO_KERNEL in the flags word, directly and through the thread's call_flags block
        two, d2 = ops.cdf_match_seg(Seg.of(td), Seg.of(sd), debug=True, flags=ops.F_CDF_TWO_KERNEL)
        with ops.call_flags(ops.F_CDF_TWO_KERNEL):
            two_b = ops.cdf_match_seg(Seg.of(td), Seg.of(sd))
        assert biteq(two_b.cpu().numpy(), two.cpu().numpy())
        one, d1 = ops.cdf_match_seg(Seg.of(td), Seg.of(sd), debug=True)
        inplace = td.clone()
        ops.cdf_match_seg(Seg.of(inplace), Seg.of(sd), out=Seg.of(inplace))
        assert biteq(one.cpu().numpy(), two.cpu().numpy())
        assert biteq(inplace.cpu().numpy(), two.cpu().numpy())
        for k in d1:
            assert biteq(d1[k].cpu().numpy(), d2[k].cpu().numpy()), k
        o = one.cpu().numpy()
        for k in range(S):
            want = orc.cdf_match(t[k], s[k if Ss > 1 else 0])
            if k != 1:
                assert biteq(o[k], want), f"segment {k}"
                continue
            # Segment 1 holds the stated deviation, asserted explicitly: the oracle takes torch.min / max literally, so
            # a NaN in a column makes its joint range NaN and the whole column NaN (histmatch.py:52-53), an inf makes it inf; the
            # kernels drop NaN from the range.  Both kernels agree with each other everywhere (above) and with the oracle in
            # every OTHER column of the segment, bit for bit; in the NaN column the finite values are matched as if the NaN values
            # were not there; what a NaN VALUE itself maps to is unspecified (both kernels give the same bits, above).
            for c in range(C):
                if c not in (2, 3):
                    assert biteq(o[k, c], want[c]), f"segment 1 column {c}"
            assert np.isnan(want[2]).all() and np.isinf(want[3]).all()
            nanpos = np.isnan(t[1, 2])
            src = s[1 if Ss > 1 else 0, 2]
            # ... i.e. they are what the reference's arithmetic gives for the column WITHOUT its NaN values (same range, same
            # counts, same normalisation by the number of counted values), bit for bit
            assert biteq(o[1, 2][~nanpos], orc.cdf_match(t[1, 2][~nanpos][None], src[None])[0])


# ================================================================================================ K6 sort mode
@pytest.mark.parametrize("S,C,n", [(1, 4, 240), (2, 3, 2048), (1, 5, 4096), (1, 3, 5000), (1, 2, 9000), (2, 2, 16384)])
def test_sort_columns_indices_bit_exact(dev, S, C, n):
    from optimaltextures_amd import ops
    rng = np.random.default_rng(n)
    x = rng.standard_normal((S, C, n)).astype(np.float32)
    x[0, 0] = np.maximum(x[0, 0], 0)  # ~50 % ties: stability is visible in the indices
    x[0, 1, :8] = [-0.0, 0.0, 0.0, -0.0, np.inf, -np.inf, 1e-42, -1e-42]
    keys, idx = ops.sort_columns(cu(x, dev))
    keys, idx = keys.cpu().numpy(), idx.cpu().numpy().view(np.uint32)
    for s in range(S):
        ok, oi = orc.sort_columns(x[s])
        assert biteq(idx[s], oi), "sort indices must be bit-exact (stable)"
        assert biteq(keys[s].view(np.uint32), ok.view(np.uint32))


@pytest.mark.parametrize("S,Ss,C,nt,ns", [(1, 1, 6, 1024, 1024), (2, 1, 4, 4096, 3072), (2, 2, 3, 900, 2000),
                                           (1, 1, 2, 16384, 12288)])
def test_sort_match_vs_oracle_bit_exact(dev, S, Ss, C, nt, ns):
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(nt + ns)
    t = rng.standard_normal((S, C, nt)).astype(np.float32)
    s = (rng.standard_normal((Ss, C, ns)) * 2 + 1).astype(np.float32)
    t[0, 0] = np.maximum(t[0, 0], 0)
    out = ops.sort_match_seg(Seg.of(cu(t, dev)), Seg.of(cu(s, dev))).cpu().numpy()
    for k in range(S):
        assert biteq(out[k], orc.sort_match(t[k], s[k if Ss > 1 else 0]))


@pytest.mark.parametrize("nt_threads,k", [(1024, 2), (256, 9), (256, 10), (640, 10)] + [(512, k) for k in range(6, 11)] +
                         [(1024, k) for k in range(6, 17)])
def test_sort_match_every_kernel_variant_bit_exact(dev, nt_threads, k):
    """rank_match4_kernel is instantiated for 9 / 10 keys per thread on 256 threads, 6..10 on 512, 6..16 on 1024 and 10 on
    640 (a column takes the workgroup that gives it 6..10 keys per thread, csrc/sort_rank4.hip); columns a little shorter
    than k * threads keys — ragged last register, ties and a ReLU-like half-zero column included — an n % 4 == 0 column
    (16-byte load variants where they exist) and a column that fills the registers exactly, against the oracle"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(100 + k + nt_threads)
    nt, ns = nt_threads * k - 3 - (k % 3), nt_threads * k // 2 + 7
    t = rng.standard_normal((2, 3, nt)).astype(np.float32)
    s = (rng.standard_normal((1, 3, ns)) * 2 + 1).astype(np.float32)
    t[0, 0] = np.maximum(t[0, 0], 0)
    t[1, 1, ::7] = t[1, 1, 3]                      # a tie group spread over the column
    out = ops.sort_match_seg(Seg.of(cu(t, dev)), Seg.of(cu(s, dev))).cpu().numpy()
    for i in range(2):
        assert biteq(out[i], orc.sort_match(t[i], s[0]))
    t4 = np.ascontiguousarray(t[:, :, : nt_threads * k - 8])
    out = ops.sort_match_seg(Seg.of(cu(t4, dev)), Seg.of(cu(s, dev))).cpu().numpy()
    for i in range(2):
        assert biteq(out[i], orc.sort_match(t4[i], s[0]))
    tf = np.concatenate([t, t[:, :, : nt_threads * k - nt]], axis=2)   # exactly k * threads keys
    out = ops.sort_match_seg(Seg.of(cu(tf, dev)), Seg.of(cu(s, dev))).cpu().numpy()
    for i in range(2):
        assert biteq(out[i], orc.sort_match(tf[i], s[0]))


@pytest.mark.parametrize("nt,ns", [(9216, 9216), (9216, 6912), (12544, 12544), (12544, 9408), (6400, 6400), (6400, 4801)])
def test_sort_match_pass_sizes_bit_exact(dev, nt, ns):
    """the pass sizes of the 512^2 schedule that are no power of two (6400 keys run on 640 threads x 10 keys, which they
    fill exactly — csrc/sort_rank4.hip): ties, a half-zero column, signed zeros, a tie group spread over the column,
    against the oracle"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(nt + 3 * ns)
    t = rng.standard_normal((2, 4, nt)).astype(np.float32)
    s = (rng.standard_normal((1, 4, ns)) * 2 + 1).astype(np.float32)
    t[0, 0] = np.maximum(t[0, 0], 0)
    t[1, 1, ::7] = t[1, 1, 3]
    t[0, 2, ::5] = np.where(rng.random(t[0, 2, ::5].shape) < 0.5, -0.0, 0.0)
    t[1, 3, :40] = t[1, 3, 100]
    out = ops.sort_match_seg(Seg.of(cu(t, dev)), Seg.of(cu(s, dev))).cpu().numpy()
    for i in range(2):
        assert biteq(out[i], orc.sort_match(t[i], s[0]))


@pytest.mark.parametrize("nt,ns", [(300, 200), (2000, 1500), (4000, 4096), (8000, 5000), (12000, 12288), (16384, 12288)])
def test_sort_radix_sweep_every_instantiation(dev, nt, ns):
    """the LSD radix kernel behind the ranking kernel (csrc/sort.hip) at each of its keys-per-thread instantiations: every
    column carries a NaN / an infinity, so the fast path flags it and the sweep sorts it (below RK_MIN_N = 512 keys the
    radix kernel is the only path).  Match and emit, against the oracle, NaNs included (IEEE totalOrder)."""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(nt * 3 + ns)
    names, t = _sort_edge_columns(nt, rng)
    t = t.copy()
    t[:, nt // 3] = np.float32(np.inf)
    t[::2, nt // 2] = np.float32(np.nan)
    t[1::2, nt // 5] = -np.float32(np.inf)
    s = (rng.standard_normal((1, t.shape[0], ns)) * 2 + 1).astype(np.float32)
    out = ops.sort_match_seg(Seg.of(cu(t[None], dev)), Seg.of(cu(s, dev))).cpu().numpy()[0]
    want = orc.sort_match(t, s[0])
    for c, name in enumerate(names):
        assert biteq(out[c], want[c]), (nt, name)
    keys, idx = ops.sort_columns(cu(t[None], dev))
    wk, wi = orc.sort_columns(t)
    assert biteq(keys.cpu().numpy()[0], wk)
    assert np.array_equal(idx.cpu().numpy()[0].astype(np.int64), wi.astype(np.int64))


def test_sort_stress_slice(dev):
    """200 cases of scripts/sort_stress.py (the round-2 stress ran 5300 outside the suite): column lengths around every
    workgroup-shape boundary of csrc/sort_rank4.hip, source lengths below / equal / above, mixed edge distributions,
    unaligned views — optex_sort_match and optex_sort_columns against the oracle, 0 mismatches"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sort_stress.py"), "200", "31"], capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0 and "200 cases, 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _ulp_clusters(n, rng):
    base = rng.standard_normal((n + 2) // 3).astype(np.float32)
    c = np.stack([base, np.nextafter(base, np.float32(np.inf)), np.nextafter(base, np.float32(-np.inf))], 1).reshape(-1)[:n]
    return rng.permutation(c).astype(np.float64)


def _sort_edge_columns(n, rng):
    """columns that exercise every path of the rank kernel and its radix sweep"""
    cols = {
        "gaussian": rng.standard_normal(n),
        "relu_half_zeros": np.maximum(rng.standard_normal(n), 0),                    # one oversized all-equal bucket
        "constant": np.full(n, 3.25),                                                # klo == khi shortcut
        "two_values": rng.integers(0, 2, n).astype(np.float64),                      # two oversized all-equal buckets
        "quantised_256": rng.integers(0, 256, n) / 255.0,                            # many big tie groups -> radix sweep
        "outlier": np.concatenate([rng.standard_normal(n - 1) * 1e-3, [1e6]]),       # equalisation must absorb it
        "heavy_tail": rng.standard_cauchy(n),
        "signed_zeros": np.where(rng.random(n) < 0.5, -0.0, 0.0),                    # totalOrder: -0 < +0, zero range
        "with_inf_nan": np.concatenate([rng.standard_normal(n - 3), [np.inf, -np.inf, np.nan]]),
        "ascending": np.arange(n, dtype=np.float64),
        "descending": -np.arange(n, dtype=np.float64),
        "tiny_range": 1.0 + rng.integers(0, 3, n) * np.float64(np.finfo(np.float32).eps),
        "denormals": rng.standard_normal(n) * 1e-41,
        # the two-columns-per-CU match kernel (csrc/sort_rank2.hip): bucket runs of 17..48 slots (per-lane loop), ties
        # inside short runs and distinct keys one ulp apart (equal `sub`, decided by re-reading the real keys)
        "groups_of_24": rng.permutation(np.arange(n) // 24).astype(np.float64) + 0.5,
        "groups_of_40_gauss": np.sort(rng.standard_normal(n))[(np.arange(n) // 40) * 40][rng.permutation(n)],
        "pairs": rng.permutation(np.repeat(rng.standard_normal((n + 1) // 2), 2)[:n]),
        "ulp_clusters": _ulp_clusters(n, rng),
    }
    names = list(cols)
    return names, np.stack([cols[k] for k in names]).astype(np.float32)


# (2049 .. 16384 keys: rank_match5w_kernel's EMIT epilogue and own range reduction first — every workgroup shape of
# rank5w_threads: 256 / 448 / 1024 threads, whole and ragged columns — then rank_match4_kernel on what it flagged, then the radix sweep)
@pytest.mark.parametrize("n", [600, 2048, 2560, 4096, 5000, 6400, 7170, 9216, 12544, 15041, 16384])
def test_sort_edge_distributions_indices_bit_exact(dev, n):
    from optimaltextures_amd import ops
    names, x = _sort_edge_columns(n, np.random.default_rng(n + 1))
    keys, idx = ops.sort_columns(cu(x[None], dev))
    keys, idx = keys.cpu().numpy()[0], idx.cpu().numpy().view(np.uint32)[0]
    ok, oi = orc.sort_columns(x)
    for c, name in enumerate(names):
        assert biteq(idx[c], oi[c]), f"{name}: indices differ"
        assert biteq(keys[c].view(np.uint32), ok[c].view(np.uint32)), f"{name}: keys differ"


@pytest.mark.parametrize("nt,ns", [(2048, 2048), (5000, 3100), (16384, 12288), (12544, 16384), (4096, 4096), (6400, 4800), (9216, 6912)])
def test_sort_match_edge_distributions_bit_exact(dev, nt, ns):
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(nt + 7 * ns)
    names, t = _sort_edge_columns(nt, rng)
    keep = [i for i, k in enumerate(names) if k != "with_inf_nan"]  # nan in the target has no defined match value order
    t = t[keep]
    s = (rng.standard_normal((1, len(keep), ns)) * 2 + 1).astype(np.float32)
    s[0, 1] = np.maximum(s[0, 1], 0)
    out = ops.sort_match_seg(Seg.of(cu(t[None], dev)), Seg.of(cu(s, dev))).cpu().numpy()[0]
    want = orc.sort_match(t, s[0])
    for c, i in enumerate(keep):
        assert biteq(out[c], want[c]), names[i]


def test_sort_full_size_properties(dev):
    """BASELINE size ([32, 256, 16384]): permutation / sortedness / stability, and the oracle on a column sample"""
    from optimaltextures_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((32, 256, 16384), device=dev, generator=g)
    x[:, ::7].clamp_min_(0)  # tie-heavy columns in between
    keys, idx = ops.sort_columns(x)
    assert bool((keys[..., 1:] >= keys[..., :-1]).all()), "sorted"
    idx = idx.long()
    assert bool((torch.gather(x, 2, idx) == keys).all()), "keys are the gathered input"
    assert bool((idx.sort(dim=2).values == torch.arange(16384, device=dev)).all()), "indices are a permutation"
    ties = keys[..., 1:] == keys[..., :-1]
    assert bool((idx[..., 1:][ties] > idx[..., :-1][ties]).all()), "stable: ties keep pixel order"
    xs = x[3, 5:9].cpu().numpy()
    ok, oi = orc.sort_columns(xs)
    assert biteq(idx[3, 5:9].cpu().numpy().astype(np.uint32), oi)


@pytest.mark.parametrize("nt,ns", [(16384, 16384), (12544, 16384), (9216, 12288), (4096, 3072)])
def test_sort_match_full_size_properties(dev, nt, ns):
    """BASELINE batch (32 textures x 256 channels) through the match kernel: the matched column, read in the stable order
    of the target, is exactly the sequence of source order statistics floor((2 i + 1) ns / (2 nt)) — checked with torch's
    own stable sort, no oracle involved; a column sample against the oracle; same input -> same bits"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    g = torch.Generator(device=dev).manual_seed(nt + ns)
    t = torch.randn((32, 256, nt), device=dev, generator=g) * 3 + 1
    t[:, ::9].clamp_min_(0)        # tie-heavy columns in between
    t[:, 5::31] = (t[:, 5::31] * 8).round() / 8 + 0.0   # many medium tie groups (+ 0.0: no -0, torch.sort has no totalOrder)
    s = torch.randn((1, 256, ns), device=dev, generator=g) * 2 - 1
    out = ops.sort_match_seg(Seg.of(t), Seg.of(s))
    order = torch.sort(t, dim=2, stable=True).indices
    q = ((2 * torch.arange(nt, device=dev, dtype=torch.int64) + 1) * ns) // (2 * nt)
    want = torch.sort(s, dim=2).values[0][:, q]                       # [256, nt]
    assert bool((torch.gather(out, 2, order) == want[None]).all())
    assert bool((ops.sort_match_seg(Seg.of(t), Seg.of(s)) == out).all()), "deterministic"
    tc, sc = t[7, 3:12].cpu().numpy(), s[0, 3:12].cpu().numpy()
    assert biteq(out[7, 3:12].cpu().numpy(), orc.sort_match(tc, sc))


@pytest.mark.parametrize("n,off", [(4099, 1), (8190, 2), (16383, 3), (1001, 1)])
def test_sort_match_unaligned_columns_bit_exact(dev, n, off):
    """columns whose length / base address rule out 16-byte loads and stores take the scalar variant of the kernel"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(n)
    t = rng.standard_normal((2, 3, n + off)).astype(np.float32)
    s = (rng.standard_normal((1, 3, n)) * 2).astype(np.float32)
    td = cu(t, dev)
    out = ops.sort_match_seg(Seg(td[:, :, off:], n + off, 3 * (n + off), n, 3, 2), Seg.of(cu(s, dev))).cpu().numpy()
    for k in range(2):
        assert biteq(out[k], orc.sort_match(np.ascontiguousarray(t[k][:, off:]), s[0]))


@pytest.mark.parametrize("S,C,n", [(1, 3, 16385), (2, 2, 40000), (1, 2, 262144)])
def test_sort_long_columns_global_radix_bit_exact(dev, S, C, n):
    """columns longer than one LDS (1024^2 / 2048^2 feature maps): csrc/sort_large.hip, same specification"""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(n)
    x = rng.standard_normal((S, C, n)).astype(np.float32)
    x[0, 0] = np.maximum(x[0, 0], 0)                       # ~50 % ties
    x[0, 1, :6] = [-0.0, 0.0, np.inf, -np.inf, 1e-42, np.nan]
    keys, idx = ops.sort_columns(cu(x, dev))
    keys, idx = keys.cpu().numpy(), idx.cpu().numpy().view(np.uint32)
    for s in range(S):
        ok, oi = orc.sort_columns(x[s])
        assert biteq(idx[s], oi), "sort indices must be bit-exact (stable)"
        assert biteq(keys[s].view(np.uint32), ok.view(np.uint32))


@pytest.mark.parametrize("nt,ns", [(20000, 16384), (16384, 30000), (65536, 49152)])
def test_sort_match_long_columns_bit_exact(dev, nt, ns):
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(nt + ns)
    t = rng.standard_normal((2, 3, nt)).astype(np.float32)
    s = (rng.standard_normal((1, 3, ns)) * 2 + 1).astype(np.float32)
    t[0, 0] = np.maximum(t[0, 0], 0)
    out = ops.sort_match_seg(Seg.of(cu(t, dev)), Seg.of(cu(s, dev))).cpu().numpy()
    for k in range(2):
        assert biteq(out[k], orc.sort_match(t[k], s[0]))


# ================================================================================================ K4 linear stats
# (the last four: few long columns — one texture at relu1_1 / relu2_1 — take more than 64 Gram splits and the multi-workgroup
#  column mean, linear.hip: gram_split_cap / col_sum_parts_kernel; 262144 = the 512^2 map, 70002: ragged, rows not 16-byte aligned)
@pytest.mark.parametrize("S,C,n,pool", [(1, 8, 256, False), (2, 8, 144, True), (3, 70, 1000, False), (1, 256, 4096, False),
                                         (1, 48, 262144, False), (2, 64, 65536, True), (1, 33, 70002, False), (1, 128, 147456, False)])
def test_linear_stats_vs_fp64(dev, S, C, n, pool):
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(C + n)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.4)
    mu, cov = ops.linear_stats(Seg.of(cu(x, dev)), pool=pool, eps=1.0)
    mu, cov = mu.cpu().numpy(), cov.cpu().numpy()
    x64 = x.astype(np.float64)
    m64 = x64.mean(-1)
    assert np.abs(mu - m64).max() < 1e-6 * max(1.0, np.abs(m64).max())
    h = x64 - m64[..., None]
    if pool:
        hp = np.concatenate(list(h), axis=1)
        ref = hp @ hp.T / hp.shape[1] + np.eye(C)
    else:
        ref = np.einsum("sin,sjn->sij", h, h) / n + np.eye(C)
    assert maxrel(cov, ref) < 5e-6
    assert np.array_equal(cov, np.swapaxes(cov, -1, -2))  # exactly symmetric (mirrored tiles)


@pytest.mark.parametrize("S,C,n,pool", [(3, 256, 4096, False), (2, 256, 16384, True), (2, 200, 1000, False), (1, 224, 6400, False),
                                         (5, 193, 260, False), (1, 256, 36, False), (2, 184, 4096, False), (3, 160, 1000, True),
                                         (1, 132, 260, False), (2, 192, 16384, False)])
def test_linear_stats_whole_triangle_kernel(dev, S, C, n, pool):
    """128 < C <= 256 (rows 16-byte aligned): one workgroup computes the 36 (C > 192) or 21 upper 32 x 32 tiles of a segment's
    Gram matrix (gram_tri_kernel).
    Against numpy fp64, and against the tile-pair kernel it replaces (`optex::gram_tri_enabled`, an internal switch of the
    library, not ABI) — different split-K partitions, so equal to round-off, not bit for bit."""
    import ctypes
    from optimaltextures_amd import _lib, ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(C + n)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.4)
    flag = ctypes.c_bool.in_dll(_lib.lib(), "_ZN5optex16gram_tri_enabledE")
    assert flag.value
    ops.profile_collect()
    ops.profile_enable(True)
    mu, cov = ops.linear_stats(Seg.of(cu(x, dev)), pool=pool, eps=1.0)
    ops.profile_enable(False)
    assert ops.profile_collect()["gram"]["launches"] == 1
    try:
        flag.value = False
        mu2, cov2 = ops.linear_stats(Seg.of(cu(x, dev)), pool=pool, eps=1.0)
    finally:
        flag.value = True
    assert torch.equal(mu, mu2)
    assert (cov - cov2).abs().max().item() <= 1e-5 * cov2.abs().max().item()
    cov = cov.cpu().numpy()
    h = x.astype(np.float64)
    h = h - h.mean(-1, keepdims=True)
    if pool:
        hp = np.concatenate(list(h), axis=1)
        ref = hp @ hp.T / hp.shape[1] + np.eye(C)
    else:
        ref = np.einsum("sin,sjn->sij", h, h) / n + np.eye(C)
    assert maxrel(cov, ref) < 5e-6
    assert np.array_equal(cov, np.swapaxes(cov, -1, -2))


# ================================================================================================ A5 hist_match
@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_hist_match_linear_golden(dev, golden, mode):
    import optimaltextures_amd as ot
    g = golden("hist_match.npz")
    for t, s, o in (("target", "source", f"out_{mode}"), ("target_b2", "source", f"out_b2_{mode}"),
                    ("target_b2", "source_b2", f"out_b2s2_{mode}"), ("target_w", "source_w", f"out_w_{mode}")):
        out = ot.hist_match(cu(g[t], dev), cu(g[s], dev), mode)
        assert out.shape == g[o].shape
        assert maxrel(out.cpu().numpy(), g[o]) < LIN_TOL, (mode, o)


def test_hist_match_cdf_and_constant_channel_golden(dev, golden):
    import optimaltextures_amd as ot
    g = golden("hist_match.npz")
    assert biteq(ot.hist_match(cu(g["target"], dev), cu(g["source"], dev), "cdf").cpu().numpy(), g["out_cdf"])
    assert biteq(ot.hist_match(cu(g["target_b2"], dev), cu(g["source"], dev), "cdf").cpu().numpy(), g["out_b2_cdf"])
    out = ot.hist_match(cu(g["target_const"], dev), cu(g["source"], dev), "chol").cpu().numpy()
    assert np.isfinite(out).all() and maxrel(out, g["out_const_chol"]) < LIN_TOL


def test_hist_match_output_is_channel_major_view_like_reference(dev, golden):
    import optimaltextures_amd as ot
    g = golden("hist_match.npz")
    t = cu(g["target"], dev)
    out = ot.hist_match(t, cu(g["source"], dev), "chol")
    b, h, w, c = t.shape
    assert out.shape == t.shape and out.stride() == (h * w, w, 1, b * h * w)
    t3 = torch.cat([cu(g["target_b2"], dev), cu(g["target_b2"], dev)[:1]])
    with pytest.raises(RuntimeError):
        ot.hist_match(t3, cu(g["source_b2"], dev), "chol")
    with pytest.raises(ValueError):
        ot.hist_match(t, t, "nope")


def test_hist_match_accepts_nchw_backed_views(dev, golden):
    """the encoder hands an NHWC *view* of NCHW memory (vgg.py:153): same numbers, no copy needed"""
    import optimaltextures_amd as ot
    g = golden("hist_match.npz")
    t = cu(g["target"], dev)
    t_view = t.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    assert not t_view.is_contiguous()
    a = ot.hist_match(t, cu(g["source"], dev), "cdf")
    b = ot.hist_match(t_view, cu(g["source"], dev), "cdf")
    assert torch.equal(a, b)


# ================================================================================================ A1 rotation
@pytest.mark.parametrize("n,seed", [(2, 1), (3, 2), (4, 0), (23, 3), (64, 4)])
def test_random_rotation_golden(dev, golden, n, seed):
    import optimaltextures_amd as ot
    g = golden("rotation.npz")
    np.random.seed(seed)
    R = ot.random_rotation(n)
    assert R.dtype == torch.float64 and R.device.type == "cpu" and R.shape == (n, n)
    assert np.abs(R.numpy() - g[f"R_{n}_seed{seed}"]).max() < 1e-13


@pytest.mark.parametrize("n,seed", [(170, 5), (256, 6), (512, 7)])
def test_random_rotation_large_golden(dev, golden, n, seed):
    import optimaltextures_amd as ot
    g = golden("rotation.npz")
    np.random.seed(seed)
    R = ot.random_rotation(n, device="cuda").cpu().numpy()
    R32 = g[f"R32_{n}_seed{seed}"]
    assert np.abs(R.astype(np.float32) - R32).max() <= 6e-8
    assert (R.astype(np.float32) != R32).mean() < 1e-4
    assert np.abs(R @ R.T - np.eye(n)).max() < 1e-13 and abs(np.linalg.det(R) - 1) < 1e-10


def test_random_rotation_stream_and_errors(dev, golden):
    import optimaltextures_amd as ot
    from optimaltextures_amd import rotation
    g = golden("rotation.npz")
    np.random.seed(11)
    a, b = ot.random_rotation(5), ot.random_rotation(5)
    assert np.abs(a.numpy() - g["R_5_seed11_first"]).max() < 1e-14
    assert np.abs(b.numpy() - g["R_5_seed11_second"]).max() < 1e-14
    with pytest.raises(ValueError):
        ot.random_rotation(1)
    # batched generation == one at a time (same stream)
    np.random.seed(3)
    R32, Rt32 = rotation.rotations(23, 3, dev)
    assert np.abs(R32[0].cpu().numpy() - g["R_23_seed3"].astype(np.float32)).max() < 1e-7
    assert torch.equal(Rt32, R32.transpose(1, 2))


# ================================================================================================ A2/A10 optimal_transport
@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_optimal_transport_linear_golden(dev, golden, mode):
    import optimaltextures_amd as ot
    g = golden("optimal_transport.npz")
    np.random.seed(42)  # same numpy stream the reference consumed -> same rotation
    out = ot.optimal_transport(cu(g["pastiche"], dev), cu(g["style"], dev), mode)
    assert out.is_contiguous() and out.shape == g["pastiche"].shape
    assert maxrel(out.cpu().numpy(), g[f"out_{mode}"]) < LIN_TOL
    np.random.seed(43)
    x = cu(g["pastiche"], dev)
    for _ in range(13):
        x = ot.optimal_transport(x, cu(g["style"], dev), mode)
    assert maxrel(x.cpu().numpy(), g[f"chain13_out_{mode}"]) < CHAIN_TOL


def test_optimal_transport_cdf_golden(dev, golden):
    import optimaltextures_amd as ot
    g = golden("optimal_transport.npz")
    np.random.seed(42)
    out = ot.optimal_transport(cu(g["pastiche"], dev), cu(g["style"], dev), "cdf").cpu().numpy()
    ref = g["out_cdf"]
    d, rng_ = np.abs(out - ref), float(ref.max() - ref.min())
    assert (d <= 1e-4 * rng_).mean() >= 0.995 and d.max() <= rng_ / 256 * 4
    # and on the reference's own rotated tensors the match itself is bit-exact
    m = ot.hist_match(cu(g["rotated_pastiche_cdf"], dev), cu(g["rotated_style_cdf"], dev), "cdf")
    assert biteq(m.cpu().numpy(), g["matched_cdf"])
    np.random.seed(46)
    out = ot.optimal_transport(cu(g["pastiche_c3"], dev), cu(g["style_c3"], dev), "cdf").cpu().numpy()
    ref = g["out_c3_cdf"]
    assert (np.abs(out - ref) <= 1e-4 * float(ref.max() - ref.min())).mean() >= 0.99


def test_optimal_transport_pooled_batch_and_blend_golden(dev, golden):
    import optimaltextures_amd as ot
    g = golden("optimal_transport.npz")
    np.random.seed(45)
    out = ot.optimal_transport(cu(g["pastiche_b2"], dev), cu(g["style"], dev), "chol")
    assert maxrel(out.cpu().numpy(), g["out_b2_chol"]) < LIN_TOL
    np.random.seed(44)
    x = cu(g["pastiche"], dev)
    content = cu(g["blend_content"], dev)
    for _ in range(3):
        x = ot.optimal_transport(x, cu(g["style"], dev), "chol")
        x += (0.2 / 2 ** (4 - 2)) * (content - x)
    assert maxrel(x.cpu().numpy(), g["blend_out"]) < 3e-4


@pytest.mark.parametrize("mode", ["cdf", "sort"])
def test_optimal_transport_hip_equals_oracle_bit_exact(dev, golden, mode):
    """The reference-API call with the rotation drawn on the device from numpy's global stream: the rotation the device
    Householder chain produced is read back (same seed, same stream), and with THAT fp32 matrix on both sides the whole
    step (GEMM, match, GEMM) must equal the oracle bit for bit, including the discontinuous cdf map.  The device
    rotation itself is within 1 ulp (fp32) of the reference's golden matrix."""
    import optimaltextures_amd as ot
    from optimaltextures_amd import rotation
    g = golden("optimal_transport.npz")
    C = g["pastiche"].shape[-1]
    np.random.seed(42)
    R32 = rotation.rotations(C, 1, dev)[0][0].cpu().numpy()
    ref32 = g["R_cdf"].astype(np.float32)
    assert np.abs(R32 - ref32).max() <= np.spacing(np.float32(1.0))  # the golden rotation of np.random.seed(42), <= 1 ulp at 1
    np.random.seed(42)
    out = ot.optimal_transport(cu(g["pastiche"], dev), cu(g["style"], dev), mode).cpu().numpy()
    want = orc.optimal_transport(g["pastiche"], g["style"], mode, R32)
    assert biteq(out, want)


@pytest.mark.parametrize("mode", ["cdf", "sort"])
@pytest.mark.parametrize("S,Ss,C,n,ns,blend", [(2, 1, 32, 1024, 768, False), (1, 1, 16, 576, 560, True),
                                                (3, 3, 8, 400, 300, True),
                                                # one style per segment: nothing is hoisted out of the loop, the style is rotated
                                                # (and sorted) per iteration while the ranges still come from the GEMM epilogue
                                                (2, 2, 256, 1024, 768, False)])
def test_ot_loop_vs_oracle_bit_exact(dev, mode, S, Ss, C, n, ns, blend):
    """the fused hot loop (optex.py:112-117) over 4 iterations with explicit rotations"""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(S + C + n)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, Ss, C, ns, scale=1.5, shift=0.5)
    content = relu_feat(rng, S, C, n, scale=2.0) if blend else None
    lr = orc.LegacyRNG(77)
    R = np.stack([orc.random_rotation(C, lr) for _ in range(4)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    xd = cu(x, dev)
    ops.ot_loop(mode, xd, cu(sty, dev), cu(R, dev), cu(Rt, dev), content=cu(content, dev) if blend else None,
                strength=0.05 if blend else 0.0)
    got = xd.cpu().numpy()
    for s in range(S):
        w = x[s]
        for it in range(4):
            rp, rs = orc.rotate_cm(w, R[it]), orc.rotate_cm(sty[s if Ss > 1 else 0], R[it])
            m = orc.cdf_match(rp, rs) if mode == "cdf" else orc.sort_match(rp, rs)
            w = orc.unrotate_cm(m, R[it])
            if blend:
                w = orc.content_blend(w, content[s], 0.05)
        assert biteq(got[s], w), f"segment {s}"


@pytest.mark.parametrize("mode,C,blend", [("cdf", 256, False), ("sort", 256, False), ("cdf", 256, True), ("chol", 256, True),
                                           ("cdf", 181, False), ("sort", 181, True), ("chol", 181, False), ("pca", 256, False)])
def test_ot_loop_at_the_bench_shape_vs_oracle(dev, mode, C, blend):
    """optex_ot_loop at the shape bench.py times: 8 independent 128 x 128 segments, 256 channels
    (and the ragged PCA rank 181), style 128 x 96, 2 iterations — the launch that selects the hot-loop GEMM with the
    row-statistics epilogue and the matcher fed from its partials — against the oracle chain on two sampled segments:
    bit-exact for cdf / sort, by tolerance for chol."""
    from optimaltextures_amd import ops
    S, n, ns, iters = 8, 16384, 12288, 2
    if mode in ("chol", "pca"):
        S = 16   # 262144 pixels per launch: the apply GEMM takes its centring as a folded per-row bias (ot_loop.hip, affine_bias_kernel)
    rng = np.random.default_rng(C + len(mode) + blend)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.5)
    content = relu_feat(rng, S, C, n, scale=2.0) if blend else None
    lr = orc.LegacyRNG(C)
    R = np.stack([orc.random_rotation(C, lr) for _ in range(iters)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    xd = cu(x, dev)
    ops.ot_loop(mode, xd, cu(sty, dev), cu(R, dev), cu(Rt, dev), content=cu(content, dev) if blend else None,
                strength=0.05 if blend else 0.0)
    got = xd.cpu().numpy()
    assert np.isfinite(got).all()
    for s in (1, S - 2):
        w = x[s]
        for it in range(iters):
            rp, rs = orc.rotate_cm(w, R[it]), orc.rotate_cm(sty[0], R[it])
            if mode == "cdf":
                m = orc.cdf_match(rp, rs)
            elif mode == "sort":
                m = orc.sort_match(rp, rs)
            else:
                m = orc.linear_match(rp, 1, rs, 1, mode)
            w = orc.unrotate_cm(m, R[it])
            if blend:
                w = orc.content_blend(w, content[s], 0.05)
        if mode in ("cdf", "sort"):
            assert biteq(got[s], w), f"{mode} C={C} segment {s}: {np.count_nonzero(got[s] != w)} elements differ"
        else:
            assert maxrel(got[s], w) <= 2 * LIN_TOL, f"{mode} C={C} segment {s}: {maxrel(got[s], w):.2e}"


@pytest.mark.parametrize("n,ns", [(9216, 13248), (12544, 18032), (6400, 9200)])
@pytest.mark.parametrize("mode", ["cdf", "sort", "chol"])
def test_ot_loop_at_the_small_batch_shapes_vs_oracle(dev, mode, n, ns):
    """optex_ot_loop at BASELINE config 4's per-GPU shard: 8 independent segments of 96^2 / 112^2 / 80^2
    pixels, 256 channels, the style at that pass's size — the launches where gemm.hip's small-batch routing hands the forward
    rotation to the R-stationary kernel WITH the row-statistics epilogue (gemm_rs_kernel<4, 64, rowstat>: min / max for cdf
    and sort, sums for chol) and the tail tiles of its balanced partition — against the oracle chain on two sampled
    segments: bit-exact for cdf / sort, by tolerance for chol."""
    from optimaltextures_amd import ops
    S, C, iters = 8, 256, 2
    rng = np.random.default_rng(n + len(mode))
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.5)
    lr = orc.LegacyRNG(n)
    R = np.stack([orc.random_rotation(C, lr) for _ in range(iters)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    xd = cu(x, dev)
    ops.ot_loop(mode, xd, cu(sty, dev), cu(R, dev), cu(Rt, dev))
    got = xd.cpu().numpy()
    assert np.isfinite(got).all()
    for s in (0, 7):  # the first and the last segment: the head and the tail of the persistent workgroups' ranges
        w = x[s]
        for it in range(iters):
            rp, rs = orc.rotate_cm(w, R[it]), orc.rotate_cm(sty[0], R[it])
            if mode == "cdf":
                m = orc.cdf_match(rp, rs)
            elif mode == "sort":
                m = orc.sort_match(rp, rs)
            else:
                m = orc.linear_match(rp, 1, rs, 1, mode)
            w = orc.unrotate_cm(m, R[it])
        if mode in ("cdf", "sort"):
            assert biteq(got[s], w), f"{mode} n={n} segment {s}: {np.count_nonzero(got[s] != w)} elements differ"
        else:
            assert maxrel(got[s], w) <= 2 * LIN_TOL, f"{mode} n={n} segment {s}: {maxrel(got[s], w):.2e}"


@pytest.mark.parametrize("S,C,n,ns,style_scale", [(4, 64, 4096, 3072, 6.0), (3, 256, 1024, 2048, 0.2), (2, 32, 4096, 4096, 2.0)])
def test_ot_loop_cdf_shared_style_histogram_vs_oracle(dev, S, C, n, ns, style_scale):
    """round 5: with a shared style and shared rotations the style's histogram over its OWN range is taken once per (iteration,
    channel) and used by every texture whose joint range (histmatch.py:52-53) is the style's range; the others bin the style
    with their own range as before.  A style much wider than the pastiche (every column reuses), much narrower (none does)
    and alike (mixed): all bit-exact against the oracle chain, every segment."""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(S * C + int(10 * style_scale))
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, 1, C, ns, scale=style_scale, shift=0.5)
    lr = orc.LegacyRNG(C + S)
    R = np.stack([orc.random_rotation(C, lr) for _ in range(3)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    xd = cu(x, dev)
    ops.ot_loop("cdf", xd, cu(sty, dev), cu(R, dev), cu(Rt, dev))
    got = xd.cpu().numpy()
    for s in range(S):
        w = x[s]
        for it in range(3):
            w = orc.unrotate_cm(orc.cdf_match(orc.rotate_cm(w, R[it]), orc.rotate_cm(sty[0], R[it])), R[it])
        assert biteq(got[s], w), f"segment {s}: {np.count_nonzero(got[s] != w)} elements differ"


@pytest.mark.parametrize("mode,S,C,n,ns,blend", [("cdf", 3, 32, 1024, 768, False), ("sort", 2, 16, 576, 560, True),
                                                  ("cdf", 8, 256, 4096, 3072, False), ("cdf", 4, 181, 4096, 3072, True)])
def test_per_texture_rotation_streams_equal_separate_runs(dev, mode, S, C, n, ns, blend):
    """un-shared rotations (optex.py:168: the reference run once per image draws its own R every
    iteration): texture i of a batch driven by one numpy stream per texture equals the B = 1 run with stream i, bit for
    bit, and that run equals the oracle chain.  (A shared-rotation batch differs: the control.)"""
    from optimaltextures_amd import driver
    rng = np.random.default_rng(S * C + n)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.5)
    content = relu_feat(rng, S, C, n, scale=2.0) if blend else None
    iters, seeds = 3, [1000 + 7 * i for i in range(S)]
    xd = cu(x, dev)
    driver.ot_iterations(xd, cu(sty, dev), mode, iters, content=cu(content, dev) if blend else None,
                         strength=0.05 if blend else 0.0, rng=[np.random.RandomState(sd) for sd in seeds])
    batch = xd.cpu().numpy()
    for i in range(S):
        xi = cu(x[i:i + 1], dev)
        driver.ot_iterations(xi, cu(sty, dev), mode, iters, content=cu(content[i:i + 1], dev) if blend else None,
                             strength=0.05 if blend else 0.0, rng=np.random.RandomState(seeds[i]))
        assert biteq(batch[i], xi.cpu().numpy()[0]), f"texture {i}"
    # the oracle chain on one texture, with the rotations its stream yields (device Householder chain, <= 1 ulp of scipy's)
    from optimaltextures_amd import rotation
    i = S - 1
    Rs = rotation.rotations(C, iters, dev, rng=np.random.RandomState(seeds[i]))[0].cpu().numpy()
    w = x[i]
    for R in Rs:
        rp, rs = orc.rotate_cm(w, R), orc.rotate_cm(sty[0], R)
        w = orc.unrotate_cm(orc.cdf_match(rp, rs) if mode == "cdf" else orc.sort_match(rp, rs), R)
        if blend:
            w = orc.content_blend(w, content[i], 0.05)
    assert biteq(batch[i], w)
    shared = cu(x, dev)
    driver.ot_iterations(shared, cu(sty, dev), mode, iters, content=cu(content, dev) if blend else None,
                         strength=0.05 if blend else 0.0, rng=np.random.RandomState(seeds[0]))
    assert biteq(shared.cpu().numpy()[0], batch[0]) and not biteq(shared.cpu().numpy()[1], batch[1])


@pytest.mark.parametrize("mode,C,blend", [("cdf", 256, False), ("sort", 256, False), ("chol", 256, True), ("sym", 64, False),
                                           ("pca", 181, False), ("cdf", 181, True)])
def test_ot_loop_is_hipgraph_capturable(dev, mode, C, blend):
    """include/optex.h promises stream-ordered, hipGraph-capturable calls: a whole (pass, layer)
    optex_ot_loop is captured with torch.cuda.graph (nothing runs during capture), replayed, and gives the eager call's
    bits; replayed again on restored inputs it gives them again (no state left behind in the scratch)."""
    from optimaltextures_amd import ops
    S, n, ns, iters = 4, 4096, 3072, 3
    rng = np.random.default_rng(C + len(mode))
    x0 = cu(relu_feat(rng, S, C, n, scale=2.0, shift=0.3), dev)
    sty = cu(relu_feat(rng, 1, C, ns, scale=1.5, shift=0.5), dev)
    content = cu(relu_feat(rng, S, C, n, scale=2.0), dev) if blend else None
    lr = orc.LegacyRNG(C + 1)
    R = np.stack([orc.random_rotation(C, lr) for _ in range(iters)]).astype(np.float32)
    Rd, Rtd = cu(R, dev), cu(np.ascontiguousarray(R.transpose(0, 2, 1)), dev)
    kw = dict(content=content, strength=0.05 if blend else 0.0)
    eager = x0.clone()
    ops.ot_loop(mode, eager, sty, Rd, Rtd, **kw)
    torch.cuda.synchronize()
    static_x = x0.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.ot_loop(mode, static_x, sty, Rd, Rtd, **kw)
    torch.cuda.synchronize()
    assert torch.equal(static_x, x0), "capture must not execute anything"
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_x, eager), f"{mode}: the replayed graph differs from the eager call"
        static_x.copy_(x0)


def test_ot_loop_sort_with_non_finite_features_takes_the_fallback(dev):
    """Inside optex_ot_loop the sort matcher gets its column range from the rotation GEMM's epilogue, whose min / max drop NaN:
    a column with a NaN key (here: every column of one texture, after the rotation has spread one NaN pixel over all
    channels) is still recognised per key, flagged before anything is stored, and swept by the radix kernel — the result is
    the oracle's (totalOrder: NaN sorts last), the other texture is untouched by it."""
    from optimaltextures_amd import ops
    S, C, n, ns = 2, 256, 4096, 3072
    rng = np.random.default_rng(77)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    x[1, 5, 100] = np.nan
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.5)
    lr = orc.LegacyRNG(9)
    R = np.stack([orc.random_rotation(C, lr)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    xd = cu(x, dev)
    ops.ot_loop("sort", xd, cu(sty, dev), cu(R, dev), cu(Rt, dev))
    got = xd.cpu().numpy()
    for s in range(S):
        rp, rs = orc.rotate_cm(x[s], R[0]), orc.rotate_cm(sty[0], R[0])
        want = orc.unrotate_cm(orc.sort_match(rp, rs), R[0])
        assert np.array_equal(np.isnan(got[s]), np.isnan(want)), f"segment {s}: NaN pattern"
        ok = ~np.isnan(want)
        assert np.array_equal(got[s][ok], want[ok]), f"segment {s}"
    assert np.isfinite(got).all()   # the match replaces every key — a NaN ranks last and takes the style's largest values


def test_ot_loop_pca_without_iterations_is_the_projector(dev):
    """optex_ot_loop_pca with iters = 0: optex.py:110 then :120 with nothing in between, x <- (x @ E) @ E^T"""
    from optimaltextures_amd import ops
    S, Cf, k, n = 2, 64, 23, 1024
    rng = np.random.default_rng(1)
    feat = relu_feat(rng, S, Cf, n, scale=2.0)
    E = np.linalg.qr(rng.standard_normal((Cf, k)))[0].astype(np.float32)
    Et = np.ascontiguousarray(E.T)
    xd = cu(feat, dev)
    empty = torch.empty((0, k, k), dtype=torch.float32, device=dev)
    ops.ot_loop_pca("cdf", xd, cu(E, dev), cu(Et, dev), cu(np.zeros((1, k, 8), np.float32), dev), empty, empty)
    for s in range(S):
        assert biteq(xd[s].cpu().numpy(), orc.gemm_tn(Et, orc.gemm_tn(E, feat[s])))


# ================================================================================================ N1: PCA folded into the rotations
@pytest.mark.parametrize("mode,blend,Cf,k,n,ns,iters", [("cdf", False, 64, 23, 1024, 768, 3), ("sort", False, 64, 23, 1024, 768, 2),
                                                        ("cdf", True, 64, 23, 1024, 768, 3), ("cdf", False, 256, 181, 4096, 3072, 2),
                                                        ("sort", False, 128, 84, 2048, 1536, 1)])
def test_ot_loop_pca_folded_vs_oracle_bit_exact(dev, mode, blend, Cf, k, n, ns, iters):
    """optex_ot_loop_pca (optex.py:110, 112-117, 120): the PCA projection folded into the first rotation,
    (feat @ E) @ R_0 = feat @ (E R_0), and — without a content blend — the unprojection into the last one.  Bit-exact against
    the oracle's restatement of exactly that association (every product a k-ordered fma chain), cdf and sort."""
    from optimaltextures_amd import ops
    S = 2
    rng = np.random.default_rng(Cf + k + iters)
    feat = relu_feat(rng, S, Cf, n, scale=2.0, shift=0.3)
    E = np.linalg.qr(rng.standard_normal((Cf, k)))[0].astype(np.float32)      # an orthonormal basis like fit_pca's
    sty = (rng.standard_normal((1, k, ns)) * 1.5).astype(np.float32)
    content = (rng.standard_normal((S, k, n)) * 2).astype(np.float32) if blend else None
    lr = orc.LegacyRNG(k)
    R = np.stack([orc.random_rotation(k, lr) for _ in range(iters)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    Et = np.ascontiguousarray(E.T)
    xd = cu(feat, dev)
    ops.ot_loop_pca(mode, xd, cu(E, dev), cu(Et, dev), cu(sty, dev), cu(R, dev), cu(Rt, dev),
                    content=cu(content, dev) if blend else None, strength=0.05 if blend else 0.0)
    got = xd.cpu().numpy()
    match = orc.cdf_match if mode == "cdf" else orc.sort_match
    ER0 = np.ascontiguousarray(orc.gemm_tn(R[0], Et).T)                         # [Cf, k] = E R_0
    G = orc.gemm_tn(R[-1], Et)                                                  # [k, Cf] = (E R_l)^T
    for s in range(S):
        w = None
        for it in range(iters):
            rp = orc.gemm_tn(ER0, feat[s]) if it == 0 else orc.rotate_cm(w, R[it])
            m = match(rp, orc.rotate_cm(sty[0], R[it]))
            if it == iters - 1 and not blend:
                out = orc.gemm_tn(G, m)                                         # the unprojection inside the last rotation back
            else:
                w = orc.unrotate_cm(m, R[it])
                if blend:
                    w = orc.content_blend(w, content[s], 0.05)
        if blend:
            out = orc.gemm_tn(Et, w)                                            # optex.py:120 on its own
        assert biteq(got[s], out), f"{mode} segment {s}: {np.count_nonzero(got[s] != out)} of {out.size} elements differ"


@pytest.mark.parametrize("mode", ["chol", "pca", "cdf"])
def test_ot_loop_pca_folded_vs_unfolded(dev, mode):
    """the folded call against project -> optex_ot_loop -> unproject on the device: the same map to fp32 round-off in the
    smooth modes (three iterations); in cdf mode (a discontinuous map: one iteration, chains cannot be compared)
    99 % of the elements within 1e-4 of the range, the rest within 1 % of it (a value that changes its bin moves by up to a bin in
    ONE of the 181 kept coordinates; the unprojection spreads that over all 256 output channels of the pixel — measured:
    99.4 % within 1e-4, largest difference 0.26 % of the range)"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.driver import project_cm, unproject_cm
    S, Cf, k, n, ns, iters = 3, 256, 181, 4096, 3072, (1 if mode == "cdf" else 3)
    rng = np.random.default_rng(len(mode))
    feat = cu(relu_feat(rng, S, Cf, n, scale=2.0, shift=0.3), dev)
    E = cu(np.linalg.qr(rng.standard_normal((Cf, k)))[0].astype(np.float32), dev)
    Et = E.t().contiguous()
    sty = cu((rng.standard_normal((1, k, ns)) * 1.5).astype(np.float32), dev)
    lr = orc.LegacyRNG(5)
    R = np.stack([orc.random_rotation(k, lr) for _ in range(iters)]).astype(np.float32)
    Rd, Rtd = cu(R, dev), cu(np.ascontiguousarray(R.transpose(0, 2, 1)), dev)
    folded = ops.ot_loop_pca(mode, feat.clone(), E, Et, sty, Rd, Rtd).cpu().numpy()
    xk = project_cm(feat, E)
    ops.ot_loop(mode, xk, sty, Rd, Rtd)
    plain = unproject_cm(xk, Et).cpu().numpy()
    err = np.abs(folded - plain)
    scale = np.abs(plain).max()
    if mode == "cdf":
        assert np.mean(err <= 1e-4 * scale) >= 0.99 and err.max() <= 0.01 * scale, (np.mean(err <= 1e-4 * scale), err.max() / scale)
    else:
        assert err.max() <= 2e-4 * scale, err.max() / scale


def test_forward_with_folded_pca_equals_unfolded(dev):
    """OptimalTexture(fold_pca=True) against the default driver path on the same rotations (chol: a smooth map)"""
    from optimaltextures_amd.driver import OptimalTexture
    kw = dict(size=128, iters=60, passes=2, hist_mode="chol", layers=(3, 2), independent=True)
    g = torch.Generator().manual_seed(1)
    style = torch.rand(1, 3, 96, 128, generator=g).to(dev)
    noise = torch.rand(2, 3, 128, 128, generator=g).to(dev)
    outs = []
    for fold in (False, True):
        tex = OptimalTexture(fold_pca=fold, **kw).to(dev).eval()
        tex.rng = np.random.RandomState(11)
        with torch.inference_mode():
            outs.append(tex.forward(noise.clone(), [style]).cpu().numpy())
    assert np.abs(outs[0] - outs[1]).max() <= 2e-3 * np.abs(outs[0]).max(), np.abs(outs[0] - outs[1]).max()


# ================================================================================================ the numpy stream on the device
def _ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


@pytest.mark.parametrize("count", [1, 2, 7, 311, 312, 313, 1000, 32895, 200001])
def test_device_normals_follow_numpy_stream(dev, count):
    """optex_legacy_normals (ABI 7): numpy's RandomState.normal — MT19937 + legacy polar method with its cache, what scipy's
    special_ortho_group.rvs draws from (optex.py:149) — advanced on the GPU from numpy's own state tuple.  Same words, same
    accept / reject decisions, same cache.  Every operation is the host's IEEE operation except log(): the device evaluates a
    correctly rounded logarithm (double-double), glibc's is within 0.52 ulp — they disagree on about one argument in 2000,
    and f * x then differs from numpy's by a few units in the last place of the double.  Bar: bit-equal in >= 99.8 % of the
    draws, never more than 4 ulp apart.  The state handed back is numpy's state after the same draws (key words, position,
    cache flag: exact), so host and device can take turns on one stream."""
    from optimaltextures_amd.rotation import DeviceNormals
    seeds = [123, 7, 2 ** 31 + 5]
    dn = DeviceNormals([np.random.RandomState(sd) for sd in seeds], dev, side_stream=False)
    got = dn.draw(count)[0].cpu().numpy()
    refs = [np.random.RandomState(sd) for sd in seeds]
    want = np.stack([r.normal(size=count) for r in refs])
    u = _ulps(got, want)
    print(f"count {count}: {np.count_nonzero(u)} of {u.size} values differ from numpy's, by at most {u.max()} ulp")
    assert u.max() <= 4, f"{np.count_nonzero(u > 4)} values differ by more than four ulp (max {u.max()})"
    assert np.count_nonzero(u) <= max(1, 0.002 * u.size)
    for i, r in enumerate(refs):
        _, key, pos, has, cached = r.get_state()
        _, dkey, dpos, dhas, dcached = dn.state(i)
        assert np.array_equal(key, dkey) and (pos, has) == (dpos, dhas)
        assert _ulps(np.array([cached]), np.array([dcached])).max() <= 4
    # the stream goes on: a second draw continues exactly where numpy continues (cache, block boundary, leftover words)
    more = dn.draw(777)[0].cpu().numpy()
    u2 = _ulps(more, np.stack([r.normal(size=777) for r in refs]))
    assert u2.max() <= 4 and np.count_nonzero(u2) <= 5


def test_device_seeding_equals_numpy_seeding(dev):
    """optex_mt19937_seed: RandomState(seed) for integer seeds, seeded on the device (bench.py's streams never cross the bus):
    the key words, the position and the first values are numpy's"""
    from optimaltextures_amd.rotation import DeviceNormals
    seeds = [2 ** 32 - 2, 2 ** 32 - 1, 0, 1]           # consecutive mod 2^32: the strided seeding path
    dn = DeviceNormals(seeds, dev, side_stream=False)
    for i, sd in enumerate(seeds):
        _, key, pos, has, cached = np.random.RandomState(sd).get_state()
        _, dkey, dpos, dhas, dcached = dn.state(i)
        assert np.array_equal(key, dkey) and (pos, has, cached) == (dpos, dhas, dcached)
    got = dn.draw(501)[0].cpu().numpy()
    want = np.stack([np.random.RandomState(sd).normal(size=501) for sd in seeds])
    assert _ulps(got, want).max() <= 4 and np.count_nonzero(_ulps(got, want)) <= 4
    scattered = DeviceNormals([5, 1000, 17], dev, side_stream=False)   # no common stride: states built on the host, pinned copy
    assert np.array_equal(scattered.state(1)[1], np.random.RandomState(1000).get_state()[1])


def test_device_normals_take_over_a_used_host_stream(dev):
    """a stream handed over in the middle of a block, at a position that is not a multiple of an attempt's four words, with
    a value in the cache: the device continues it like numpy would"""
    from optimaltextures_amd.rotation import DeviceNormals
    r = np.random.RandomState(99)
    r.random_sample(3)          # 6 words: the position is now 2 mod 4
    r.normal(size=5)            # odd count: one value sits in the cache
    twin = np.random.RandomState(99)
    twin.random_sample(3)
    twin.normal(size=5)
    dn = DeviceNormals(r, dev, side_stream=False)
    got = np.concatenate([dn.draw(c)[0].cpu().numpy()[0] for c in (1, 4, 623, 2000)])
    want = twin.normal(size=1 + 4 + 623 + 2000)
    assert _ulps(got, want).max() <= 4 and np.count_nonzero(_ulps(got, want)) <= 6
    _, key, pos, has, _ = twin.get_state()
    _, dkey, dpos, dhas, _ = dn.state(0)
    assert np.array_equal(key, dkey) and (pos, has) == (dpos, dhas)


@pytest.mark.parametrize("N,count", [(3, 5), (23, 4), (64, 3), (181, 2), (256, 2)])
def test_device_stream_rotations_equal_host_stream_rotations(dev, N, count):
    """rotations from the device-side stream vs the host-side numpy stream of the same seed: <= 1 ulp of fp32 apart (the bar of
    the scipy goldens), one stream shared and one stream per texture; prefetched draws (side stream) equal on-demand ones"""
    from optimaltextures_amd import rotation
    from optimaltextures_amd.rotation import DeviceNormals
    host = rotation.rotations(N, count, dev, rng=np.random.RandomState(31))[0].cpu().numpy()
    dn = DeviceNormals(np.random.RandomState(31), dev)
    dn.prefetch([(N, count), (N, 1)])
    R32, Rt32 = dn.rotations(N, count)
    torch.cuda.synchronize()
    got = R32.cpu().numpy()
    assert got.shape == host.shape and np.abs(got - host).max() <= 1.2e-7
    assert np.array_equal(Rt32.cpu().numpy(), got.transpose(0, 2, 1))
    assert np.mean(got == host) > 0.99
    # a request that leaves the prefetched schedule (the prefetched draw was for ONE rotation): what was drawn ahead is dropped —
    # consumed — and the request is drawn from where the stream stands (no raise, no stale hand-out)
    twin = np.random.RandomState(31)
    twin.normal(size=(count + 1) * (N * (N + 1) // 2 - 1))
    want2 = rotation.rotations(N, 2, dev, rng=twin)[0].cpu().numpy()
    with pytest.warns(RuntimeWarning, match="drawn ahead were discarded"):   # ... said once, and counted
        got2 = dn.rotations(N, 2)[0].cpu().numpy()
    assert not dn.pending() and np.abs(got2 - want2).max() <= 1.2e-7
    assert dn.dropped_rotations == 1
    many = DeviceNormals([np.random.RandomState(31), np.random.RandomState(32)], dev)
    R2 = many.rotations(N, count)[0].cpu().numpy()
    assert R2.shape == (2, count, N, N) and np.abs(R2[0] - host).max() <= 1.2e-7
    other = rotation.rotations(N, count, dev, rng=np.random.RandomState(32))[0].cpu().numpy()
    assert np.abs(R2[1] - other).max() <= 1.2e-7


def test_device_stream_drawn_a_step_ahead_is_the_same_stream(dev):
    """bench.py draws rotation group q + 1 while group q is synthesised: a stream whose whole schedule was prefetched before
    the call (DeviceNormals.covers -> forward() draws nothing) hands out the same rotations as one that draws at the call's
    start and as one that draws on demand; a prefetch over PREFETCH_BYTES keeps only the head and draws the rest on demand"""
    from optimaltextures_amd.driver import OptimalTexture
    from optimaltextures_amd.rotation import DeviceNormals
    tex = OptimalTexture(size=128, iters=60, passes=2, hist_mode="cdf", layers=(3, 2), no_pca=True, independent=True).to(dev).eval()
    sched = tex.rotation_schedule()
    assert [c for c, _ in sched] == [256, 128, 256, 128] and all(n > 0 for _, n in sched)
    a, b, c = (DeviceNormals(1234, dev) for _ in range(3))
    a.prefetch(sched)
    assert a.covers(sched) and a.pending() == [(int(n), int(k)) for n, k in sched] and not b.covers(sched)
    old = DeviceNormals.PREFETCH_BYTES
    try:
        DeviceNormals.PREFETCH_BYTES = 8 * sched[0][1] * 256 * 256 + 1   # room for the first entry only
        c.prefetch(sched)
        assert c.pending() == [(256, sched[0][1])] and c.covers(sched)
    finally:
        DeviceNormals.PREFETCH_BYTES = old
    for n, k in sched:
        ra, rb, rc = a.rotations(n, k), b.rotations(n, k), c.rotations(n, k)
        torch.cuda.synchronize()
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[0], rc[0])
    assert not a.pending() and not a.covers(sched)
    # and forward() leaves a covered stream alone: afterwards the three streams stand at the same position
    d = DeviceNormals(1234, dev)
    d.prefetch(sched)
    tex.rng = d
    g = torch.Generator().manual_seed(0)
    with torch.inference_mode():
        out = tex.forward(torch.rand(1, 3, 128, 128, generator=g).to(dev), [torch.rand(1, 3, 96, 128, generator=g).to(dev)])
    assert bool(torch.isfinite(out).all()) and not d.pending()
    sa, sd = a.state(0), d.state(0)
    assert np.array_equal(sa[1], sd[1]) and sa[2:4] == sd[2:4]


@pytest.mark.parametrize("n", [16384, 12544, 9216, 6400, 4096, 15040, 8256, 5184, 2560])
def test_sort_match_rank5_kernel_through_the_loop_edge_distributions(dev, n):
    """rank_match5w_kernel (csrc/sort_rank5.hip, round 6: 8-bit buckets, keys alone in their bucket ranked without a window) is
    reached through optex_ot_loop — it needs the column range the rotation GEMM's epilogue leaves (column lengths here are
    multiples of 64, what that epilogue asks for).  With R = I the rotation is exact up to the sign of zero (x * 1 + 0 * ...: the
    oracle's fma chain does the same), so ONE iteration is sort_match of the input itself: every workgroup shape (full, ragged
    last row, scalar rows), ns below / equal / above n, and the distributions that send a column to its rare paths or to the radix sweep —
    gaussian, a tie group, pairs of exact ties, quantised values (massive ties: flagged), one outlier that sets the range, a
    heavy tail, half zeros, signed zeros, three values — against the oracle bit for bit, and against OPTEX_F_SORT_RANK4 (the
    round-2-5 kernel) on the same call."""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(n)
    C, S = 128, 2
    eye = np.eye(C, dtype=np.float32)[None]
    Rd = cu(eye, dev)
    x = rng.standard_normal((S, C, n)).astype(np.float32)
    x[0, 1, rng.random(n) < 0.02] = 0.5                                   # one tie group of ~2 % of the keys
    x[0, 2, 1::2] = x[0, 2, 0:(n - 1) if n % 2 else n:2][:len(x[0, 2, 1::2])]   # every key twice: pairs of exact ties
    x[0, 3] = np.floor(x[0, 3] * 64) / 64                                 # ~500 distinct values
    x[0, 4, 7] = 1.0e6                                                    # one outlier sets the range
    x[0, 5] = np.exp(3 * x[0, 5])                                         # heavy tail
    x[0, 6] = np.maximum(x[0, 6], 0)                                      # half zeros
    x[0, 7] = np.where(rng.random(n) < 0.5, -0.0, 0.0).astype(np.float32)
    x[0, 7, ::97] = rng.standard_normal(len(x[0, 7, ::97]))               # signed zeros + a few values
    x[0, 8] = np.floor(rng.random(n) * 3)                                 # three values
    x[0, 9] = 2.5                                                         # constant
    x[0, 10] = 0.0
    x[0, 10, ::2] = -0.0                                                  # zero range, both signs
    x[0, 11, :3] = x[0, 11, 5]                                            # a handful of ties
    x[0, 12] = x[0, 12] * 1e-30                                           # tiny keys (2^-100 and below in part)
    for ns in sorted({n, (n * 3 // 4 + 3) // 4 * 4, n * 23 // 16 // 4 * 4}):
        sty = rng.standard_normal((1, C, ns)).astype(np.float32)
        outs = []
        for flags in (0, ops.F_SORT_RANK4):
            xd = cu(x, dev)
            ops.ot_loop("sort", xd, cu(sty, dev), Rd, Rd, flags=flags)
            outs.append(xd.cpu().numpy())
        assert biteq(outs[0], outs[1]), f"ns = {ns}: the two ranking kernels disagree"
        for sgm in range(S):
            want = orc.unrotate_cm(orc.sort_match(orc.rotate_cm(x[sgm], eye[0]), orc.rotate_cm(sty[0], eye[0])), eye[0])
            assert biteq(outs[0][sgm], want), f"ns = {ns} segment {sgm}"


def test_per_call_flags_two_threads_with_different_settings(dev):
    """ABI 10: the spare-CU count of the persistent GEMM, the two-kernel cdf pipeline and the older sort kernel are
    chosen per CALL.  Two threads run OT loops at the same time with different choices, each on its own stream; each gets the
    bits of the same call made alone (every choice is bit-identical by construction, so any cross-talk that changed a LAUNCH
    would still pass — what is asserted besides is which kernel classes each thread's calls recorded while the other ran)."""
    import threading
    from optimaltextures_amd import ops
    rng = np.random.default_rng(3)
    S, C, n, ns, iters = 4, 256, 4096, 3072, 3
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.5)
    lr = orc.LegacyRNG(11)
    R = np.stack([orc.random_rotation(C, lr) for _ in range(iters)]).astype(np.float32)
    Rd, Rtd, sd = cu(R, dev), cu(np.ascontiguousarray(R.transpose(0, 2, 1)), dev), cu(sty, dev)
    alone = {}
    for mode in ("cdf", "sort"):
        xd = cu(x, dev)
        ops.ot_loop(mode, xd, sd, Rd, Rtd)
        alone[mode] = xd.cpu().numpy()
    results, errors = {}, []
    gate = threading.Barrier(2)

    def job(name, mode, flags):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st), ops.call_flags(flags):
                outs = []
                gate.wait()
                for _ in range(6):
                    xd = cu(x, dev)
                    ops.ot_loop(mode, xd, sd, Rd, Rtd)
                    outs.append(xd)
                st.synchronize()
            assert getattr(ops._tls, "flags", 0) == 0
            results[name] = [o.cpu().numpy() for o in outs]
        except Exception as e:   # noqa: BLE001
            errors.append((name, repr(e)))

    for mode, fa, fb in (("cdf", ops.F_CDF_TWO_KERNEL | ops.f_spare_cus(0), ops.f_spare_cus(2)),
                         ("sort", ops.F_SORT_RANK4 | ops.f_spare_cus(1), ops.f_spare_cus(0))):
        ta = threading.Thread(target=job, args=("a", mode, fa))
        tb = threading.Thread(target=job, args=("b", mode, fb))
        ta.start(); tb.start(); ta.join(); tb.join()
        assert not errors, errors
        for name in ("a", "b"):
            for o in results[name]:
                assert biteq(o, alone[mode]), (mode, name)
    # the choice really reaches the launch, and only the call that carries it: kernel classes recorded by single calls
    ops.profile_enable(True)
    try:
        ops.profile_collect()
        xd = cu(x, dev)
        ops.ot_loop("cdf", xd, sd, Rd, Rtd, flags=ops.F_CDF_TWO_KERNEL)
        two = ops.profile_collect()
        xd = cu(x, dev)
        ops.ot_loop("cdf", xd, sd, Rd, Rtd)
        one = ops.profile_collect()
    finally:
        ops.profile_enable(False)
    assert "cdf_apply" in two and "cdf_match" not in two
    assert "cdf_match" in one and "cdf_apply" not in one


def test_device_stream_fed_during_the_previous_call_is_the_same_stream(dev):
    """DeviceNormals.begin_feed / feed_one / finish_feed + OptimalTexture.rng_next: the NEXT call's draws released one (pass, layer)
    at a time at the start of THIS call's decode phases.  Same stream, same rotations, same images as two calls that each
    prefetch their own schedule; a fed stream that is asked before its release point draws on demand; auto spare-CU policy"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.driver import OptimalTexture
    from optimaltextures_amd.rotation import DeviceNormals
    tex = OptimalTexture(size=128, iters=60, passes=2, hist_mode="cdf", layers=(3, 2), no_pca=True, independent=True).to(dev).eval()
    sched = tex.rotation_schedule()
    want = [(int(n), int(k)) for n, k in sched]
    a, b = DeviceNormals(77, dev), DeviceNormals(77, dev)
    a.begin_feed(sched)
    assert a.feeding() and a.covers(sched) and a.pending() == []
    ev = torch.cuda.Event()
    ev.record()
    assert a.feed_one(after=ev) and a.pending() == want[:1] and a.covers(sched)
    r0 = a.rotations(*want[0])          # released: taken from the queue
    r1 = a.rotations(*want[1])          # not released yet: drawn on demand, same stream position
    a.finish_feed()
    assert a.pending() == want[2:] and not a.feeding()
    rest = [a.rotations(n, k) for n, k in want[2:]]
    for got, (n, k) in zip([r0, r1] + rest, want):
        ref = b.rotations(n, k)
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    # two calls back to back, the second one's generator fed during the first (MIOpen's convolutions are not reproducible run
    # to run — scripts/miopen_determinism_probe.py — so the images are not compared: the streams' positions are)
    g = torch.Generator().manual_seed(0)
    x0, x1 = (torch.rand(1, 3, 128, 128, generator=g).to(dev) for _ in range(2))
    style = torch.rand(1, 3, 96, 128, generator=g).to(dev)
    with torch.inference_mode():
        first, second = DeviceNormals(5, dev), DeviceNormals(6, dev)
        first.prefetch(sched)
        second.begin_feed(sched)
        tex.rng, tex.rng_next = first, second
        prev = ops.gemm_spare_cus(1)
        out0 = tex.forward(x0.clone(), [style])
        assert ops.gemm_spare_cus(prev) == 1            # forward() never touches the (deprecated) process-wide default: its choice
        assert getattr(ops._tls, "flags", 0) == 0       # travels in the flags word of its calls and is gone when it returns
        assert tex.rng_next is None and not second.feeding() and second.pending() == want and second.covers(sched)
        tex.rng = second
        out1 = tex.forward(x1.clone(), [style])
    assert bool(torch.isfinite(out0).all()) and bool(torch.isfinite(out1).all()) and not second.pending()
    for seed, used in ((5, first), (6, second)):
        ref = DeviceNormals(seed, dev)
        for n, k in want:
            ref.rotations(n, k)
        sa, sb = ref.state(0), used.state(0)
        assert np.array_equal(sa[1], sb[1]) and sa[2:4] == sb[2:4]


def test_forward_with_device_rotation_stream_equals_host_stream(dev):
    """OptimalTexture.forward with its rotations drawn on the GPU (prefetched for the whole call on a side stream) against the
    same call with the numpy stream on the host: same seeds -> same rotations to 1 ulp of fp32 -> the same image to fp32
    round-off in a smooth mode"""
    from optimaltextures_amd.driver import OptimalTexture
    from optimaltextures_amd.rotation import DeviceNormals
    tex = OptimalTexture(size=128, iters=60, passes=2, hist_mode="chol", layers=(3, 2), independent=True).to(dev).eval()
    g = torch.Generator().manual_seed(0)
    style = torch.rand(1, 3, 96, 128, generator=g).to(dev)
    noise = torch.rand(2, 3, 128, 128, generator=g).to(dev)
    with torch.inference_mode():
        tex.rng = np.random.RandomState(77)
        a = tex.forward(noise.clone(), [style]).cpu().numpy()
        tex.rng = DeviceNormals(np.random.RandomState(77), dev)
        b = tex.forward(noise.clone(), [style]).cpu().numpy()
        tex.rng = DeviceNormals([np.random.RandomState(77), np.random.RandomState(78)], dev)
        c = tex.forward(noise.clone(), [style]).cpu().numpy()
    assert np.abs(a - b).max() <= 2e-3 * np.abs(a).max(), np.abs(a - b).max()
    # own streams: texture 0 keeps stream 77 (the same image), texture 1 follows stream 78 (another one)
    assert np.abs(c[0] - a[0]).max() <= 2e-3 * np.abs(a).max() and np.abs(c[1] - a[1]).max() > 1e-2


# ================================================================================================ full-size (BASELINE) properties
def test_full_size_relu3_1_step_vs_oracle(dev):
    """BASELINE config shape: relu3_1 at the 512 pass, C = 256, n = 128*128, style 128x96 — one whole cdf step bit-exact
    against the oracle, plus size-independent properties."""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(2024)
    C, n, ns = 256, 16384, 12288
    x = relu_feat(rng, 1, C, n, scale=2.0, shift=0.2)
    sty = relu_feat(rng, 1, C, ns, scale=1.7, shift=0.4)
    R = orc.random_rotation(C, orc.LegacyRNG(0)).astype(np.float32)
    Rd, Rtd = cu(R, dev), cu(np.ascontiguousarray(R.T), dev)
    y = ops.rotate_seg(cu(x, dev), Rd)
    ys = ops.rotate_seg(cu(sty, dev), Rd)
    y_np, ys_np = y.cpu().numpy()[0], ys.cpu().numpy()[0]
    assert biteq(y_np, orc.rotate_cm(x[0], R)) and biteq(ys_np, orc.rotate_cm(sty[0], R))
    m, d = ops.cdf_match_seg(Seg.of(y), Seg.of(ys), debug=True)
    assert int(d["hist_t"].sum().item()) == C * n and int(d["hist_s"].sum().item()) == C * ns  # checksum of counts
    assert biteq(m.cpu().numpy()[0], orc.cdf_match(y_np, ys_np))
    out = ops.unrotate_seg(m, Rtd).cpu().numpy()[0]
    assert biteq(out, orc.unrotate_cm(m.cpu().numpy()[0], R))
    # sort mode at full size: sortedness, permutation validity, self-match is the identity
    keys, idx = ops.sort_columns(y)
    k = keys[0].cpu().numpy()
    assert (np.diff(k, axis=1) >= 0).all()
    ii = np.sort(idx[0].cpu().numpy().view(np.uint32), axis=1)
    assert (ii == np.arange(n, dtype=np.uint32)).all()
    ident = ops.sort_match_seg(Seg.of(y), Seg.of(y))
    assert torch.equal(ident, y)
    sm = ops.sort_match_seg(Seg.of(y), Seg.of(ys)).cpu().numpy()[0]
    assert biteq(sm[:8], orc.sort_match(y_np[:8], ys_np[:8]))
    # sliced OT must move the feature distribution toward the style's: per-channel means approach after one exact step
    gap0 = np.abs(y_np.mean(1) - ys_np.mean(1)).mean()
    gap1 = np.abs(sm.mean(1) - ys_np.mean(1)).mean()
    assert gap1 < 0.05 * gap0


def test_determinism_same_input_same_bits(dev):
    """integer-atomic histograms and fixed-order reductions: two runs give identical bits"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(3)
    x = cu(relu_feat(rng, 2, 64, 40000, scale=2.0), dev)
    s = cu(relu_feat(rng, 1, 64, 30000, scale=1.0), dev)
    a = ops.cdf_match_seg(Seg.of(x), Seg.of(s))
    b = ops.cdf_match_seg(Seg.of(x), Seg.of(s))
    assert torch.equal(a, b)
    m1 = ops.linear_stats(Seg.of(x), pool=False)
    m2 = ops.linear_stats(Seg.of(x), pool=False)
    assert torch.equal(m1[0], m2[0]) and torch.equal(m1[1], m2[1])


# ================================================================================================ N3 VGG glue
@pytest.mark.parametrize("N,C,H,W", [(1, 3, 8, 8), (2, 5, 17, 9), (1, 64, 64, 48), (2, 8, 33, 31)])
@pytest.mark.parametrize("relu,pool,up,pad", [(False, False, False, 1), (True, False, False, 1), (True, True, False, 1),
                                               (True, False, True, 1), (True, False, False, 0), (True, True, False, 0),
                                               (False, False, True, 0)])
def test_vgg_glue_matches_torch_modules_bit_exact(dev, N, C, H, W, relu, pool, up, pad):
    """csrc/glue.hip against the nn module sequence it replaces (vgg.py:14-135), odd sizes included (ceil_mode pool)"""
    from optimaltextures_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + C * 100 + H + W)
    x = torch.randn(N, C, H, W, generator=g)
    b = torch.randn(C, generator=g)
    want = x + b.view(1, -1, 1, 1)
    if relu:
        want = torch.relu(want)
    if pool:
        want = torch.nn.functional.max_pool2d(want, 2, 2, 0, ceil_mode=True)
    if up:
        want = torch.nn.functional.interpolate(want, scale_factor=2, mode="nearest")
    if pad:
        want = torch.nn.functional.pad(want, (1, 1, 1, 1), mode="reflect")
    got = ops.vgg_glue(x.to(dev), b.to(dev), relu=relu, pool=pool, up=up, pad=pad).cpu()
    assert got.shape == want.shape
    assert torch.equal(got, want)
    if not (relu or pool or up):  # bias-free pad-only form (decoder input)
        assert torch.equal(ops.vgg_glue(x.to(dev), None, pad=pad).cpu(),
                           torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect") if pad else x)
    # the same pass with either side channels-last (what MIOpen's implicit-GEMM convolutions take and return)
    x_cl = x.to(dev).contiguous(memory_format=torch.channels_last)
    for in_cl, out_cl in ((False, True), (True, False), (True, True)):
        if in_cl and out_cl and C % 4:
            continue
        got = ops.vgg_glue(x_cl if in_cl else x.to(dev), b.to(dev), relu=relu, pool=pool, up=up, pad=pad, out_nhwc=out_cl)
        assert got.shape == want.shape
        assert got.permute(0, 2, 3, 1).is_contiguous() if out_cl else got.is_contiguous()
        assert torch.equal(got.cpu(), want), f"in_nhwc={in_cl} out_nhwc={out_cl}"


@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_vgg_codec_fused_path_equals_module_path(dev, depth):
    """Encoder.features / Decoder.decode through the fused glue == the plain nn.Sequential on the same device, in both
    layout policies ("mixed": the wide convolutions run channels-last through other MIOpen kernels)"""
    from optimaltextures_amd.vgg import Decoder, Encoder
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(depth)).to(dev)
    for layout, tol in (("nchw", 1e-5), ("mixed", 5e-5), (None, 5e-5)):   # None = the module default ("mixed")
        enc, dec = Encoder(depth, codec_layout=layout).to(dev).eval(), Decoder(depth, codec_layout=layout).to(dev).eval()
        with torch.inference_mode():
            f_fused, f_plain = enc.features(x), enc.model(x)
            # bias-free conv + our add vs MIOpen's own bias handling (and, for "mixed", another convolution kernel):
            # fp32 summation-order noise only
            assert f_fused.is_contiguous() and f_fused.shape == f_plain.shape
            assert torch.allclose(f_fused, f_plain, rtol=0, atol=tol * float(f_plain.abs().max())), layout
            d_fused, d_plain = dec.decode(f_plain), dec.model(f_plain)
            assert d_fused.shape == d_plain.shape == x.shape and d_fused.is_contiguous()
            assert torch.allclose(d_fused, d_plain, rtol=0, atol=tol * float(d_plain.abs().max())), layout
    with pytest.raises(ValueError):
        Encoder(depth, codec_layout="nhwc")


# ================================================================================================ N1 / N2 "next" rows
@pytest.mark.parametrize("route", ["gram", "svd"])
def test_fit_pca_golden(dev, golden, route, monkeypatch):
    """optex.py:180-190 (quirks: scalar-mean centring, uncentred projection, off-by-one k): same k, same subspace — by
    either route to the basis (eigenvectors of the fp64 Gram matrix, the default; torch.linalg.svd of the [N, C] matrix)"""
    from optimaltextures_amd import driver
    from optimaltextures_amd.driver import fit_pca
    monkeypatch.setattr(driver, "PCA_FIT", route)
    g = golden("next_rows.npz")
    feats, eig = fit_pca(cu(g["pca_in"], dev))
    assert eig.shape[1] == int(g["pca_k"])
    e, want_e = eig.cpu().numpy().astype(np.float64), g["pca_eigvecs"].astype(np.float64)
    # singular vectors are defined up to sign: compare the projectors onto the kept subspace, and |cos| per component
    assert np.abs(e @ e.T - want_e @ want_e.T).max() < 2e-4
    assert np.all(np.abs(np.sum(e * want_e, axis=0)) > 1 - 1e-4)
    signs = np.sign(np.sum(e * want_e, axis=0)).astype(np.float32)
    assert maxrel(feats.cpu().numpy() * signs, g["pca_features"]) < 1e-4


@pytest.mark.parametrize("mode,tol", [("chol", LIN_TOL), ("cdf", 0.0)])
def test_mix_style_features_golden(dev, golden, mode, tol):
    """optex.py:193-206 through the boundary (two hist_match calls, un-rotated features with ReLU ties)"""
    from optimaltextures_amd.driver import mix_style_features
    g = golden("next_rows.npz")
    out = mix_style_features([cu(g["mix_style"], dev)], cu(g["mix_mask"], dev), 0.5, mode)[0].cpu().numpy()
    want = g[f"mix_out_{mode}"]
    if mode == "cdf":   # identical un-rotated inputs: the cdf pipeline is bit-exact, the blend arithmetic is plain fp32
        assert np.abs(out - want).max() <= 1e-6 * np.abs(want).max()
    else:
        assert maxrel(out, want) < tol


def test_driver_loop_equals_oracle_chain(dev):
    """driver.ot_iterations (independent segments, shared rotation stream, content blend) == the same chain of oracle
    calls, bit for bit in cdf mode"""
    from optimaltextures_amd.driver import ot_iterations
    rng = np.random.default_rng(3)
    S, C, n, ns, iters = 3, 32, 1024, 640, 4
    x = relu_feat(rng, S, C, n, scale=2.0)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.3)
    content = relu_feat(rng, S, C, n, scale=2.0)
    got = ot_iterations(cu(x, dev), cu(sty, dev), "cdf", iters, content=cu(content, dev), strength=0.025,
                        rng=np.random.RandomState(11)).cpu().numpy()
    lr = orc.LegacyRNG(11)
    Rs = [orc.random_rotation(C, lr).astype(np.float32) for _ in range(iters)]
    for s in range(S):
        w = x[s]
        for R in Rs:
            w = orc.content_blend(orc.unrotate_cm(orc.cdf_match(orc.rotate_cm(w, R), orc.rotate_cm(sty[0], R)), R),
                                  content[s], 0.025)
        assert biteq(got[s], w), f"segment {s}"


# ================================================================================================ optional fused rotations
@pytest.mark.parametrize("mode", ["cdf", "sort"])
def test_ot_loop_fused_rotations(dev, mode):
    """fuse_rotations re-associates (m @ R_i^T) @ R_{i+1} = m @ (R_i^T R_{i+1}).  (1) the HIP loop equals the oracle's
    restatement of exactly that algorithm bit for bit; (2) the re-associated product itself differs from the literal
    two-GEMM product only by fp32 round-off (<= 2e-6 * max), which is the whole numerical content of the option."""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(17)
    S, C, n, ns, iters = 2, 32, 1024, 768, 4
    x = relu_feat(rng, S, C, n, scale=2.0)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.2)
    lr = orc.LegacyRNG(5)
    R = np.stack([orc.random_rotation(C, lr) for _ in range(iters)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    xd = cu(x, dev)
    ops.ot_loop(mode, xd, cu(sty, dev), cu(R, dev), cu(Rt, dev), fuse_rotations=True)
    got = xd.cpu().numpy()
    match = orc.cdf_match if mode == "cdf" else orc.sort_match
    P = [orc.gemm_tn(R[i], R[i + 1]) for i in range(iters - 1)]
    for s in range(S):
        y = orc.rotate_cm(x[s], R[0])
        for i in range(iters):
            m = match(y, orc.rotate_cm(sty[0], R[i]))
            if i + 1 < iters:
                y = orc.gemm_tn(P[i], m)
            else:
                out = orc.unrotate_cm(m, R[i])
        assert biteq(got[s], out), f"segment {s}"
    m = relu_feat(rng, C, 4096, scale=3.0)
    literal = orc.rotate_cm(orc.unrotate_cm(m, R[0]), R[1])
    fused = orc.gemm_tn(P[0], m)
    assert np.abs(literal - fused).max() <= 2e-6 * np.abs(literal).max()


# ================================================================================================ end to end (driver)
def _cpu_forward(tex_cpu, pastiche, styles, content, np_seed, mode):
    """OptimalTexture.forward restated on the host: torch-CPU VGG (same weights), the ORACLE for every hot-path call,
    the same rotation stream.  Mirrors optex.py:81-139 for the no_pca / single-style case."""
    from optimaltextures_amd.util import get_size, layer_iters, resize
    rng = orc.LegacyRNG(np_seed)
    with torch.inference_mode():
        for p in range(tex_cpu.passes):
            size = tex_cpu.sizes[p]
            if pastiche.shape[-2] != size and pastiche.shape[-1] != size:
                sty = [resize(s, size=get_size(size, tex_cpu.style_scale, s.shape[2], s.shape[3])) for s in styles]
                cont = None
                cont_size = (size, size)
                if content is not None:
                    cont_size = get_size(size, 1.0, content.shape[2], content.shape[3], oversize=True)
                    cont = resize(content, size=cont_size)
                pastiche = resize(pastiche, size=cont_size)
            else:
                sty, cont = styles, content
            for enc, dec in zip(tex_cpu.encoders, tex_cpu.decoders):
                enc_index = 5 - enc.depth
                sf = enc.model(sty[0])[0]
                sf = sf.reshape(sf.shape[0], -1).numpy()
                cf = None
                if cont is not None:
                    cf = enc.model(cont)[0]
                    cf = cf.reshape(cf.shape[0], -1)
                    cf = (cf - cf.mean() + float(sf.mean())).numpy()
                feat = enc.model(pastiche)
                _, c, h, w = feat.shape
                x = feat[0].reshape(c, h * w).numpy()
                blend = cf is not None and enc_index <= 2
                strength = tex_cpu.content_strength / 2 ** (4 - enc_index) if blend else 0.0
                for _ in range(layer_iters(tex_cpu.iters_per_pass_and_layer, p, enc_index)):
                    R = orc.random_rotation(c, rng).astype(np.float32)
                    x = orc.unrotate_cm(orc.hist_match_cm(orc.rotate_cm(x, R), 1, orc.rotate_cm(sf, R), 1, mode), R)
                    if blend:
                        x = orc.content_blend(x, cf, strength)
                pastiche = dec.model(torch.from_numpy(x).view(1, c, h, w))
    return pastiche


@pytest.mark.parametrize("mode,with_content", [("chol", False), ("chol", True), ("sort", False)])
def test_forward_end_to_end_vs_host_restatement(dev, mode, with_content):
    """The whole driver (multi-pass schedule, [l-1] table indexing, resize rules, scalar content re-centring, blend
    strengths /2^(4-l), glue-fused codec, hot loop) against a host restatement built from the oracle.  Smooth modes
    agree to fp32 noise; `sort` is checked on image statistics (a rank flip moves single pixels, not the texture)."""
    from optimaltextures_amd.driver import OptimalTexture
    kw = dict(size=256, iters=60, passes=2, hist_mode=mode, no_pca=True, layers=(3, 2), content_strength=0.2)
    tex_cpu = OptimalTexture(**kw).eval()
    tex_gpu = OptimalTexture(**kw).to(dev).eval()   # same seeded synthetic weights
    g = torch.Generator().manual_seed(11)
    low = torch.rand(1, 3, 12, 16, generator=g)
    style = torch.nn.functional.interpolate(low, size=(160, 224), mode="bicubic", align_corners=False).clamp(0, 1)
    content = torch.rand(1, 3, 256, 256, generator=g) if with_content else None
    pastiche = torch.rand(1, 3, 256, 256, generator=g)
    want = _cpu_forward(tex_cpu, pastiche.clone(), [style], content, 77, mode).numpy()
    tex_gpu.rng = np.random.RandomState(77)
    with torch.inference_mode():
        got = tex_gpu.forward(pastiche.to(dev), [style.to(dev)], None if content is None else content.to(dev)).cpu().numpy()
    assert got.shape == want.shape and np.isfinite(got).all()
    if mode == "sort":
        assert abs(got.mean() - want.mean()) < 2e-3 * np.abs(want).max()
        assert abs(got.std() - want.std()) < 5e-3 * want.std()
        assert np.mean(np.abs(got - want) < 1e-2 * np.abs(want).max()) > 0.98
    else:
        assert np.abs(got - want).max() < 2e-3 * np.abs(want).max()


# ================================================================================================ BASELINE configs 3 / 5 (shape coverage)
def _smooth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, max(h // 32, 2), max(w // 32, 2), generator=g)
    return torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)


def test_config3_style_transfer_1024_runs(dev):
    """BASELINE config 3 in miniature iterations: 1024^2 style transfer, content blend on relu3_1 (strength/4), pooled
    B = 1, default chol, colour transfer 'opt' (3 cdf iterations on the RGB image, optex.py:131-134)"""
    from optimaltextures_amd.driver import OptimalTexture
    tex = OptimalTexture(size=1024, iters=40, passes=2, hist_mode="chol", content_strength=0.2, layers=(3, 2),
                         color_transfer="opt").to(dev).eval()
    assert tex.sizes == [256, 1024]
    style, content = _smooth_image(416, 416, 1).to(dev), _smooth_image(1024, 1024, 2).to(dev)
    noise = torch.rand(content.shape, generator=torch.Generator().manual_seed(4)).to(dev)
    tex.rng = np.random.RandomState(3)
    with torch.inference_mode():
        out = tex.forward(noise.clone(), [style], content)
    assert out.shape == (1, 3, 1024, 1024) and bool(torch.isfinite(out).all())
    # the blend is live (the codec has random weights here, so nothing can be said about what the image looks like):
    # the same rotations with content_strength 0 give a different, equally finite result
    tex0 = OptimalTexture(size=1024, iters=40, passes=2, hist_mode="chol", content_strength=0.0, layers=(3, 2),
                          color_transfer="opt").to(dev).eval()
    tex0.rng = np.random.RandomState(3)
    with torch.inference_mode():
        plain = tex0.forward(noise.clone(), [style], content)
    assert bool(torch.isfinite(plain).all()) and float((out - plain).abs().max()) > 1e-3


@pytest.mark.parametrize("no_pca,with_content", [(True, False), (False, True)])
def test_driver_style_prefetch_equals_pass_by_pass(dev, no_pca, with_content):
    """Multi-GPU runs compute / broadcast the style side of every pass at the start of forward() (driver.py
    prefetch_style_sides).  With a pass-through sync hook (world size 1): the prefetched style sides are the ones the
    pass-by-pass path computes (same resize decisions, same feature-map sizes, same PCA rank; values up to the fp32
    noise of MIOpen's convolutions, whose kernel choice and summation order are not fixed across calls), and a forward
    call through the hook runs to a finite image of the right shape.  (A whole-image comparison is meaningless: the cdf
    map is discontinuous, so convolution round-off noise grows to O(1) over the iterations.)"""
    from optimaltextures_amd import dist as otdist
    from optimaltextures_amd.driver import OptimalTexture
    tex = OptimalTexture(size=320, iters=30, passes=3, hist_mode="cdf", no_pca=no_pca, layers=(3, 2),
                         content_strength=0.2, independent=not with_content).to(dev).eval()
    style = _smooth_image(200, 264, 11).to(dev)
    content = _smooth_image(288, 352, 12).to(dev) if with_content else None
    noise = torch.rand((1 if with_content else 3, 3, 320, 320), generator=torch.Generator().manual_seed(13)).to(dev)
    with torch.inference_mode():
        tex.style_sync = otdist.StyleSync(dev)
        sides = tex.prefetch_style_sides(noise.shape[-2:], [style], content)
        tex.style_sync = None
        pastiche = noise.clone()
        for p in range(tex.passes):
            resized = tex._needs_resize(pastiche.shape[-2:], tex.sizes[p])
            assert sides[p][0] == resized
            pastiche, sf, eig, cf, hw = tex.encode_inputs(pastiche, [style], content, tex.sizes[p])
            assert hw == sides[p][3]
            for got, want in zip(sides[p][1], sf):
                assert got.shape == want.shape
                if no_pca:  # with PCA the basis vectors are only defined up to sign / rotation inside near-equal values
                    assert torch.allclose(got, want, rtol=0, atol=1e-4 * float(want.abs().max()))
            for got, want in zip(sides[p][2], eig):
                assert got.shape == want.shape
        tex.style_sync = otdist.StyleSync(dev)
        tex.rng = np.random.RandomState(14)
        out = tex.forward(noise.clone(), [style], content)
    assert out.shape[-2:] == pastiche.shape[-2:] and out.shape[0] == noise.shape[0] and bool(torch.isfinite(out).all())


def test_config5_two_style_mixing_runs(dev):
    """BASELINE config 5 at reduced size: two-style mixing (optex.py:97-101,193-206) over relu3_1..relu1_1 with PCA,
    hist_mode cdf — the un-rotated hist_match calls of mix_style_features go through the same boundary"""
    from optimaltextures_amd.driver import OptimalTexture
    tex = OptimalTexture(size=512, iters=60, passes=2, hist_mode="cdf", mixing_alpha=0.5, layers=(3, 2, 1)).to(dev).eval()
    a, b = _smooth_image(256, 320, 5).to(dev), _smooth_image(256, 320, 6).to(dev)
    tex.rng = np.random.RandomState(8)
    torch.manual_seed(9)
    with torch.inference_mode():
        out = tex.forward(torch.rand(1, 3, 512, 512, device=dev), [a, b])
    assert out.shape == (1, 3, 512, 512) and bool(torch.isfinite(out).all())
    assert 0.0 < float(out.std()) < 1.0


# ================================================================================================ GEMM epilogue statistics
def test_ot_loop_cdf_epilogue_minmax_equals_separate_kernels(dev):
    """At hot-loop shapes the forward rotation GEMM takes the per-channel min / max of its output in its epilogue
    (GemmArgs::rowstat) and cdf_match skips its own pass over the rotated map (histmatch.py:52-53): the loop must equal
    the same steps made of the separate C-ABI calls (rotate, cdf_match with its own min / max kernel, rotate back), bit
    for bit — min and max do not depend on the order they are taken in."""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    g = torch.Generator(device=dev).manual_seed(11)
    S, C, n, ns, iters = 32, 256, 2048, 1536, 2          # 16 pixel tiles x 32 segments = 512 blocks: the hot-loop kernel
    x = torch.randn((S, C, n), device=dev, generator=g).clamp_min_(0) * 2
    sty = torch.randn((1, C, ns), device=dev, generator=g).clamp_min_(0) * 1.5
    R32, Rt32 = __import__("optimaltextures_amd.rotation", fromlist=["rotations"]).rotations(C, iters, dev, rng=np.random.RandomState(3))
    got = x.clone()
    ops.ot_loop("cdf", got, sty, R32, Rt32)
    want = x.clone()
    for it in range(iters):
        y, ys = ops.rotate_seg(want, R32[it]), ops.rotate_seg(sty, R32[it])
        m = ops.cdf_match_seg(Seg.of(y), Seg.of(ys))
        want = ops.unrotate_seg(m, Rt32[it])
    assert torch.equal(got, want)


def test_ot_loop_chol_epilogue_sums_match_separate_mean_kernel(dev):
    """linear modes at a hot-loop shape: the means come from row sums taken in the rotation GEMM's epilogue (fp32 per tile,
    double across tiles) instead of col_mean_kernel's double sum over the map; fuse_rotations=2 (literal sequence, separate
    mean kernel) must agree to fp32 round-off"""
    from optimaltextures_amd import ops
    g = torch.Generator(device=dev).manual_seed(12)
    S, C, n, ns, iters = 32, 256, 2048, 1536, 2
    x = torch.randn((S, C, n), device=dev, generator=g).clamp_min_(0) * 2
    sty = torch.randn((1, C, ns), device=dev, generator=g).clamp_min_(0) * 1.5
    R32, Rt32 = __import__("optimaltextures_amd.rotation", fromlist=["rotations"]).rotations(C, iters, dev, rng=np.random.RandomState(4))
    a, b = x.clone(), x.clone()
    ops.ot_loop("chol", a, sty, R32, Rt32)
    ops.ot_loop("chol", b, sty, R32, Rt32, fuse_rotations=2)
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
