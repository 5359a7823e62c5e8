"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/gen_golden.py).  Bit-exact where the arithmetic is integer/index/elementwise (cdf path, RNG
stream); stated tolerances where the reference goes through BLAS/LAPACK (GEMM order, factorizations)."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as orc


def maxrel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ------------------------------------------------------------------------------------------------ A1 rotation
def test_legacy_rng_matches_numpy_randomstate():
    for seed in (0, 1, 42, 2**32 - 1):
        a = orc.LegacyRNG(seed)
        b = np.random.RandomState(seed)
        for n in (1, 2, 7, 1000, 3):  # odd sizes exercise the one-value gaussian cache
            assert np.array_equal(a.normal(n), b.normal(size=n))


@pytest.mark.parametrize("n,seed", [(2, 1), (3, 2), (4, 0), (23, 3), (64, 4)])
def test_random_rotation_small(golden, n, seed):
    g = golden("rotation.npz")
    np.random.seed(seed)
    R = orc.random_rotation(n)
    ref = g[f"R_{n}_seed{seed}"]
    assert R.dtype == np.float64
    assert np.abs(R - ref).max() < 1e-13  # same algorithm, BLAS vs plain-C summation order only
    assert np.array_equal(orc.random_rotation(n, orc.LegacyRNG(seed)), R)  # C RNG == numpy global RNG


@pytest.mark.parametrize("n,seed", [(170, 5), (256, 6), (512, 7)])
def test_random_rotation_large(golden, n, seed):
    g = golden("rotation.npz")
    R = orc.random_rotation(n, orc.LegacyRNG(seed))
    R32 = g[f"R32_{n}_seed{seed}"]
    assert np.abs(R.astype(np.float32) - R32).max() <= 6e-8  # <= 1 fp32 ulp at |r| < 1
    assert (R.astype(np.float32) != R32).mean() < 1e-4
    assert np.abs(R @ R.T - np.eye(n)).max() < 1e-13
    assert abs(np.linalg.det(R) - 1.0) < 1e-10
    assert abs(float(g[f"det_{n}"]) - 1.0) < 1e-10


def test_rotation_stream_continuity(golden):
    g = golden("rotation.npz")
    np.random.seed(11)
    a, b = orc.random_rotation(5), orc.random_rotation(5)
    assert np.abs(a - g["R_5_seed11_first"]).max() < 1e-14
    assert np.abs(b - g["R_5_seed11_second"]).max() < 1e-14


def test_random_rotation_rejects_dim1():
    with pytest.raises(ValueError):
        orc.random_rotation(1)


# ------------------------------------------------------------------------------------------------ A7 histc/linspace
def test_histc_bit_exact(golden):
    g = golden("histc_linspace.npz")
    for x, lo, hi, h in zip(g["x"], g["lo"], g["hi"], g["hist"]):
        assert np.array_equal(orc.histc(x, lo, hi), h)
    assert np.array_equal(orc.histc(np.full(100, 3.0, np.float32), 3.0, 3.0), g["const_hist"])


def test_linspace_bit_exact(golden):
    g = golden("histc_linspace.npz")
    for lo, hi, e in zip(g["lo"], g["hi"], g["edges257"]):
        assert np.array_equal(orc.linspace257(lo, hi), e)
    for lo, hi, e in zip(g["ls_lo"], g["ls_hi"], g["ls_edges257"]):
        assert np.array_equal(orc.linspace257(lo, hi), e)


# ------------------------------------------------------------------------------------------------ A8 interp
def test_interp_known_answers(golden):
    g = golden("interp.npz")
    assert np.array_equal(orc.interp(g["ka1_x"], g["ka1_xp"], g["ka1_fp"]), g["ka1_out"])
    assert list(g["ka1_out"]) == [-5, 0, -5, 10, 15, 90, 90]
    assert np.array_equal(orc.interp(g["ka2_x"], g["ka2_xp"], g["ka2_fp"]), g["ka2_out"])
    assert list(g["ka2_out"]) == [1, 1, 4, 4]


@pytest.mark.parametrize("case", range(6))
def test_interp_random_bit_exact(golden, case):
    g = golden("interp.npz")
    out = orc.interp(g[f"rand{case}_x"], g[f"rand{case}_xp"], g[f"rand{case}_fp"])
    assert np.array_equal(out, g[f"rand{case}_out"], equal_nan=True)


# ------------------------------------------------------------------------------------------------ A6 cdf_match
def test_cdf_match_stages_bit_exact(golden):
    g = golden("cdf_match.npz")
    out, d = orc.cdf_match(g["target"], g["source"], debug=True)
    for k in ("lo", "hi", "hist_t", "hist_s", "bin_edges", "remapped"):
        assert np.array_equal(d[k], g[k], equal_nan=True), k
    assert np.array_equal(out, g["out"])


def test_cdf_match_dense_and_ties_bit_exact(golden):
    g = golden("cdf_match.npz")
    assert np.array_equal(orc.cdf_match(g["target2"], g["source2"]), g["out2"])
    assert np.array_equal(orc.cdf_match(g["target3"], g["source3"]), g["out3"])


def test_cdf_match_degenerate_channels(golden):
    g = golden("cdf_match.npz")
    a = orc.cdf_match(np.full((1, 64), 2.0, np.float32), np.linspace(0, 4, 80, dtype=np.float32)[None])
    assert np.array_equal(a, g["deg_const_t_out"]) and np.all(a == np.float32(0.015625))
    t01 = g["deg_const_s_out"]
    import torch  # linspace(0,1,64) must be torch's, not numpy's (different rounding)
    b = orc.cdf_match(torch.linspace(0, 1, 64)[None].numpy(), np.full((1, 80), 0.5, np.float32))
    assert np.array_equal(b, t01) and np.all(b == np.float32(0.50390625))
    c = orc.cdf_match(np.full((1, 64), 3.0, np.float32), np.full((1, 80), 3.0, np.float32))
    assert np.array_equal(c, g["deg_both_out"]) and np.all(c == 3.0)


def test_cdf_match_other_bin_counts_bit_exact(golden):
    """histmatch.py:49 `bins` free: the generalised restatement against the reference's outputs (gen_cdf_bins_golden.py)"""
    g = golden("cdf_match_bins.npz")
    for b in g["bins"]:
        assert np.array_equal(orc.cdf_match_bins(g["target"], g["source"], int(b)), g[f"out_{b}"], equal_nan=True), int(b)
    both = orc.cdf_match_bins(np.full((1, 64), 3.0, np.float32), np.full((1, 80), 3.0, np.float32), 7)
    assert np.array_equal(both, g["deg_both_out_7"])
    assert np.array_equal(orc.cdf_match_bins(g["target"], g["source"], 256), orc.cdf_match(g["target"], g["source"]))


# ------------------------------------------------------------------------------------------------ A4/A5 linear modes
LIN_TOL = 1e-4  # single linear step <= 1e-4 * max|ref| (fp32 LAPACK vs fp64 factorization)


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_hist_match_linear(golden, mode):
    g = golden("hist_match.npz")
    assert maxrel(orc.hist_match(g["target"], g["source"], mode), g[f"out_{mode}"]) < LIN_TOL
    assert maxrel(orc.hist_match(g["target_b2"], g["source"], mode), g[f"out_b2_{mode}"]) < LIN_TOL
    assert maxrel(orc.hist_match(g["target_b2"], g["source_b2"], mode), g[f"out_b2s2_{mode}"]) < LIN_TOL
    assert maxrel(orc.hist_match(g["target_w"], g["source_w"], mode), g[f"out_w_{mode}"]) < LIN_TOL


def test_hist_match_cdf_pooled_batch(golden):
    g = golden("hist_match.npz")
    assert np.array_equal(orc.hist_match(g["target"], g["source"], "cdf"), g["out_cdf"])
    assert np.array_equal(orc.hist_match(g["target_b2"], g["source"], "cdf"), g["out_b2_cdf"])


def test_hist_match_constant_channel_finite(golden):
    g = golden("hist_match.npz")
    out = orc.hist_match(g["target_const"], g["source"], "chol")
    assert np.isfinite(out).all() and maxrel(out, g["out_const_chol"]) < LIN_TOL


def test_hist_match_batch_mismatch_raises(golden):
    g = golden("hist_match.npz")
    t3 = np.concatenate([g["target_b2"], g["target_b2"][:1]])
    with pytest.raises(RuntimeError):
        orc.hist_match(t3, g["source_b2"], "chol")


# ------------------------------------------------------------------------------------------------ A2/A10 optimal_transport
GEMM_TOL = 2e-5  # rotation GEMM <= 2e-5 * max|ref|


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_optimal_transport_linear_single_step(golden, mode):
    g = golden("optimal_transport.npz")
    out, mid = orc.optimal_transport(g["pastiche"], g["style"], mode, g[f"R_{mode}"], return_intermediates=True)
    assert maxrel(mid["rotated_pastiche"], g[f"rotated_pastiche_{mode}"]) < GEMM_TOL
    assert maxrel(mid["rotated_style"], g[f"rotated_style_{mode}"]) < GEMM_TOL
    assert maxrel(mid["matched"], g[f"matched_{mode}"]) < LIN_TOL
    assert maxrel(out, g[f"out_{mode}"]) < LIN_TOL


def test_optimal_transport_cdf_single_step(golden):
    """cdf is discontinuous: on the reference's own rotated inputs the match is bit-exact; through our
    own GEMM (different summation order than MKL) >= 99.5 % of outputs agree to 1e-4*range, the rest within a bin."""
    g = golden("optimal_transport.npz")
    rp, rs = g["rotated_pastiche_cdf"], g["rotated_style_cdf"]
    assert np.array_equal(orc.hist_match(rp, rs, "cdf"), g["matched_cdf"])
    out = orc.optimal_transport(g["pastiche"], g["style"], "cdf", g["R_cdf"])
    ref = g["out_cdf"]
    rng_ = float(ref.max() - ref.min())
    d = np.abs(out - ref)
    assert (d <= 1e-4 * rng_).mean() >= 0.995
    assert d.max() <= rng_ / 256 * 4


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_optimal_transport_chain13(golden, mode):
    g = golden("optimal_transport.npz")
    x = g["pastiche"]
    for R in g[f"chain13_R_{mode}"]:
        x = orc.optimal_transport(x, g["style"], mode, R)
    assert maxrel(x, g[f"chain13_out_{mode}"]) < 1e-3  # SURVEY 8c chain tolerance


def test_optimal_transport_content_blend(golden):
    g = golden("optimal_transport.npz")
    x = g["pastiche"]
    for R in g["blend_R"]:
        x = orc.optimal_transport(x, g["style"], "chol", R)
        x = orc.content_blend(x, g["blend_content"], 0.2 / 2 ** (4 - 2))
    assert maxrel(x, g["blend_out"]) < 3e-4


def test_optimal_transport_pooled_batch_and_c3(golden):
    g = golden("optimal_transport.npz")
    out = orc.optimal_transport(g["pastiche_b2"], g["style"], "chol", g["R_b2_chol"])
    assert maxrel(out, g["out_b2_chol"]) < LIN_TOL
    out = orc.optimal_transport(g["pastiche_c3"], g["style_c3"], "cdf", g["R_c3_cdf"])
    ref = g["out_c3_cdf"]
    d = np.abs(out - ref)
    assert (d <= 1e-4 * float(ref.max() - ref.min())).mean() >= 0.99


# ------------------------------------------------------------------------------------------------ A9 sort mode (own spec)
def test_sort_columns_is_stable_argsort():
    rng = np.random.default_rng(0)
    k = np.maximum(rng.standard_normal((5, 3000)), 0).astype(np.float32)  # ~50 % ties
    k[0, :10] = [-0.0, 0.0, -0.0, 0.0, 1, 1, -1, -1, np.inf, -np.inf]
    sk, si = orc.sort_columns(k)
    for c in range(k.shape[0]):
        # np stable argsort treats -0 == +0; the spec orders -0 < +0, so compare on the integer key
        u = k[c].view(np.uint32)
        key = np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)
        ref = np.argsort(key, kind="stable")
        assert np.array_equal(si[c], ref.astype(np.uint32))
        assert np.array_equal(sk[c].view(np.uint32), k[c][ref].view(np.uint32))


def test_sort_match_is_exact_1d_transport():
    rng = np.random.default_rng(1)
    t = rng.standard_normal((4, 512)).astype(np.float32)
    s = (rng.standard_normal((4, 512)) * 3 + 1).astype(np.float32)
    out = orc.sort_match(t, s)
    for c in range(4):  # equal sizes: output is a permutation of the source with the target's ranks
        assert np.array_equal(np.sort(out[c]), np.sort(s[c]))
        assert np.array_equal(np.argsort(out[c], kind="stable"), np.argsort(t[c], kind="stable"))
    s2 = (rng.standard_normal((4, 300))).astype(np.float32)
    out2 = orc.sort_match(t, s2)
    for c in range(4):
        ss = np.sort(s2[c])
        r = np.argsort(np.argsort(t[c], kind="stable"), kind="stable")
        assert np.array_equal(out2[c], ss[((2 * r + 1) * 300) // (2 * 512)])


# ------------------------------------------------------------------------------------------------ SURVEY 7.4-3 on the oracle
@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_collapsed_linear_chain_is_the_literal_chain(mode):
    """The algebra behind optex_ot_loop(fuse_rotations = 3), checked on the CPU restatement alone: a chain of linear-mode
    steps x' = hist_match(x R, s R) R^T equals ONE affine map M_k ... M_1 (x - mean(x)) + mean(s) with M_i = R_i T_i R_i^T,
    where the covariance every later step sees follows analytically, cov(x') = M_i cov(x) M_i^T (no pass over the map)."""
    rng = np.random.default_rng(11)
    C, n, ns, iters = 12, 900, 700, 6
    x = np.maximum(rng.standard_normal((C, n)) * 2 + 0.3, 0).astype(np.float32)
    s = np.maximum(rng.standard_normal((C, ns)) * 1.5 + 0.5, 0).astype(np.float32)
    lr = orc.LegacyRNG(5)
    Rs = [orc.random_rotation(C, lr) for _ in range(iters)]
    # literal chain (optex.py:112-117 without a content blend)
    w = x
    for R in Rs:
        w = orc.unrotate_cm(orc.linear_match(orc.rotate_cm(w, R), 1, orc.rotate_cm(s, R), 1, mode), R)
    # collapsed chain in fp64: statistics only
    x64, s64 = x.astype(np.float64), s.astype(np.float64)
    mu_x, mu_s = x64.mean(1, keepdims=True), s64.mean(1, keepdims=True)
    cov = (x64 - mu_x) @ (x64 - mu_x).T / n
    cov_s = (s64 - mu_s) @ (s64 - mu_s).T / ns
    eye = np.eye(C)

    def sqrtm(a):
        ev, v = np.linalg.eigh(a)
        return (v * np.sqrt(ev)) @ v.T

    acc = eye
    for R in Rs:
        ct, cs = R.T @ cov @ R + eye, R.T @ cov_s @ R + eye       # cov(x R) = R^T cov(x) R; eps * I is rotation-invariant
        if mode == "chol":
            T = np.linalg.cholesky(cs) @ np.linalg.inv(np.linalg.cholesky(ct))
        elif mode == "pca":
            T = sqrtm(cs) @ np.linalg.inv(sqrtm(ct))
        else:
            qt = sqrtm(ct)
            qi = np.linalg.inv(qt)
            T = qi @ sqrtm(qt @ cs @ qt) @ qi
        M = R @ T @ R.T
        cov = M @ cov @ M.T
        acc = M @ acc
    got = acc @ (x64 - mu_x) + mu_s
    assert np.abs(got - w).max() <= 2e-4 * np.abs(w).max()
