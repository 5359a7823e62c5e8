"""GPU parity tests (pytest -m gpu) of the R-stationary hot-loop GEMM (csrc/gemm_rs.hip) through optex_gemm_tn: every
(row tiles, k-steps) instantiation, ragged M and K (PCA ranks), per-segment matrices, strided operands, bias and content
blend — bit-exact against the oracle's k-ordered fmaf chain (oracle/optex_oracle.c orc_gemm_tn)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def biteq(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


# (M, K): the four instantiations (rows (128,192] / (192,256]  x  depth (128,192] / (192,256]) at their edges and inside
SHAPES = [(256, 256), (181, 181), (192, 192), (193, 193), (129, 129), (181, 256), (256, 181), (165, 200), (250, 133),
          (256, 131), (130, 256), (255, 255), (224, 225),
          # relu2_1 with PCA (C = 128, k ~ 71-86) and small ranks at relu3_1: rows (64, 128], k-steps [64, 128]
          (84, 84), (84, 128), (128, 84), (128, 128), (65, 64), (100, 256), (256, 100), (71, 190), (190, 71), (127, 65)]


@pytest.mark.parametrize("M,K", SHAPES)
def test_rs_gemm_shared_matrix_bit_exact(dev, M, K):
    """one matrix for all segments (the rotations of the hot loop): S x n / 64 >= 2 tiles per CU selects the kernel"""
    from optimaltextures_amd import ops
    S, n = 5, 7040  # 110 tiles per segment: the workgroups' tile ranges straddle segment boundaries
    rng = np.random.default_rng(M * 1000 + K)
    x = (rng.standard_normal((S, K, n)) * 2).astype(np.float32)
    At = (rng.standard_normal((K, M)) / 8).astype(np.float32)
    out = torch.full((S, M, n), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(At, dev), cu(x, dev), out, M, K, n, S, lda=M, ldb=n, b_ss=K * n, ldo=n, o_ss=M * n)
    got = out.cpu().numpy()
    for s in (0, 2, 4):
        assert biteq(got[s], orc.gemm_tn(At, x[s])), f"M={M} K={K} segment {s}"
    assert np.isfinite(got).all()


@pytest.mark.parametrize("M,K", [(256, 256), (181, 181), (181, 256), (256, 181)])
def test_rs_gemm_per_segment_matrices_bias_blend_strided(dev, M, K):
    """per-segment matrices (the apply step of the linear modes; un-shared rotations), bias and content blend in the
    epilogue, operands that are strided views (leading dimensions larger than the rows); the pad columns hold NaN"""
    from optimaltextures_amd import ops
    S, n, ld = 8, 4096, 4096 + 64
    rng = np.random.default_rng(M + K)
    xb = np.full((S, K, ld), np.nan, dtype=np.float32)
    xb[:, :, :n] = rng.standard_normal((S, K, n)).astype(np.float32)
    At = (rng.standard_normal((S, K, M)) / 8).astype(np.float32)
    badd = rng.standard_normal((S, M)).astype(np.float32)
    content = rng.standard_normal((S, M, ld)).astype(np.float32)
    x_d, c_d = cu(xb, dev), cu(content, dev)
    out = torch.full((S, M, ld), -7.0, dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(At, dev), x_d, out, M, K, n, S, lda=M, at_ss=K * M, ldb=ld, b_ss=K * ld, ldo=ld, o_ss=M * ld,
                badd=cu(badd, dev), badd_ss=M, content=c_d, strength=0.05)
    got = out.cpu().numpy()
    assert (got[:, :, n:] == -7.0).all()  # nothing written into the pad
    for s in (0, 5, 7):
        want = orc.content_blend(orc.gemm_tn(At[s], np.ascontiguousarray(xb[s, :, :n]), None, badd[s]),
                                 np.ascontiguousarray(content[s, :, :n]), 0.05)
        assert biteq(got[s, :, :n], want), f"segment {s}"
    # bias only, shared bias
    out2 = torch.empty((S, M, n), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(At, dev), x_d, out2, M, K, n, S, lda=M, at_ss=K * M, ldb=ld, b_ss=K * ld, ldo=n, o_ss=M * n,
                badd=cu(badd[3], dev), badd_ss=0)
    assert biteq(out2.cpu().numpy()[6], orc.gemm_tn(At[6], np.ascontiguousarray(xb[6, :, :n]), None, badd[3]))


def test_rs_gemm_nonfinite_rows_beyond_k_do_not_leak(dev):
    """K = 181 stops inside a k-step: the three rows behind it exist in memory (the next segment's first rows) and are
    made non-finite here — they must enter as exact zeros, not as 0 * inf"""
    from optimaltextures_amd import ops
    S, M, K, n = 8, 181, 181, 4096
    rng = np.random.default_rng(9)
    x = rng.standard_normal((S, K, n)).astype(np.float32)
    x[1:, 0:3, :] = np.inf  # rows 181..183 of segment s are rows 0..2 of segment s + 1
    At = (rng.standard_normal((K, M)) / 8).astype(np.float32)
    out = torch.empty((S, M, n), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(At, dev), cu(x, dev), out, M, K, n, S, lda=M, ldb=n, b_ss=K * n, ldo=n, o_ss=M * n)
    got = out.cpu().numpy()
    assert np.isfinite(got[0]).all() and biteq(got[0], orc.gemm_tn(At, x[0]))


@pytest.mark.parametrize("M,K", [(181, 181), (165, 165), (181, 256), (192, 192), (84, 84), (100, 128), (256, 256), (256, 181), (224, 160),
                                 (256, 224)])
def test_rs_gemm_centred_operand_bit_exact(dev, M, K):
    """the apply step of the linear modes at PCA ranks (histmatch.py:27/34/42,44: T @ (hist_t - mu_t) + mu_s): per-segment
    operators, per-segment centring vector and bias, content blend — one subtraction per operand element, one rounding"""
    from optimaltextures_amd import ops
    S, n = 8, 4096
    rng = np.random.default_rng(M * 3 + K)
    x = rng.standard_normal((S, K, n)).astype(np.float32)
    At = (rng.standard_normal((S, K, M)) / 8).astype(np.float32)
    bsub = rng.standard_normal((S, K)).astype(np.float32)
    badd = rng.standard_normal((S, M)).astype(np.float32)
    content = rng.standard_normal((S, M, n)).astype(np.float32)
    out = torch.empty((S, M, n), dtype=torch.float32, device=dev)
    ops.gemm_tn(cu(At, dev), cu(x, dev), out, M, K, n, S, lda=M, at_ss=K * M, ldb=n, b_ss=K * n, ldo=n, o_ss=M * n,
                bsub=cu(bsub, dev), bsub_ss=K, badd=cu(badd, dev), badd_ss=M, content=cu(content, dev), strength=0.05)
    got = out.cpu().numpy()
    for s_ in (0, 4, 7):
        want = orc.content_blend(orc.gemm_tn(At[s_], x[s_], bsub[s_], badd[s_]), content[s_], 0.05)
        assert biteq(got[s_], want), f"segment {s_}"
    out2 = torch.empty_like(out)  # centring only, one shared vector and one shared matrix
    ops.gemm_tn(cu(At[1], dev), cu(x, dev), out2, M, K, n, S, lda=M, ldb=n, b_ss=K * n, ldo=n, o_ss=M * n, bsub=cu(bsub[2], dev))
    assert biteq(out2.cpu().numpy()[5], orc.gemm_tn(At[1], x[5], bsub[2], None))
