"""GPU tests (pytest -m gpu) of the C x C algebra of the linear modes on the device (csrc/linalg.hip, include/optex.h K5)
and of the linear modes inside the fused hot loop (optex_ot_loop modes 2-4).  Bars: the reference computes
these in fp32 LAPACK, the oracle in fp64; a single hist_match step must agree within 1e-4 * max|ref|, a chain of steps
within 1e-3."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

LIN_TOL, CHAIN_TOL = 1e-4, 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def cu(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def relu_feat(rng, *shape, scale=1.0, shift=0.0):
    return np.maximum(rng.standard_normal(shape) * scale + shift, 0).astype(np.float32)


def feature_cov(rng, C, n, rank=None, scale=3.0):
    """cov + I of ReLU-like features with a few strong directions (VGG-like: |A|_F / lambda_min in the hundreds)"""
    base = rng.standard_normal((rank or max(C // 6, 2), n))
    mix = rng.standard_normal((C, base.shape[0])) * scale
    x = np.maximum(mix @ base + rng.standard_normal((C, n)), 0)
    x -= x.mean(1, keepdims=True)
    return (x @ x.T / n + np.eye(C)).astype(np.float32)


@pytest.mark.parametrize("C,batch", [(2, 1), (3, 2), (23, 3), (32, 1), (64, 2), (181, 2), (256, 3), (384, 1), (512, 2)])
def test_chol_inv_vs_fp64(dev, C, batch):
    """histmatch.py:25-27: A = L L^T and L^-1 — residuals and the factor itself against numpy fp64"""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(C * 3 + batch)
    A = np.stack([feature_cov(rng, C, 4 * C + 50) for _ in range(batch)])
    U, Li = ops.chol_inv(cu(A, dev))
    U, Li = U.cpu().numpy().astype(np.float64), Li.cpu().numpy().astype(np.float64)
    for b in range(batch):
        L = np.linalg.cholesky(A[b].astype(np.float64))
        assert np.allclose(np.triu(U[b]), U[b]) and np.allclose(np.tril(Li[b]), Li[b])   # triangles are clean
        assert np.abs(U[b].T - L).max() <= 2e-6 * np.abs(L).max() * max(1, C / 64)
        assert np.abs(U[b].T @ U[b] - A[b]).max() <= 1e-5 * np.abs(A[b]).max()
        assert np.abs(Li[b] @ U[b].T - np.eye(C)).max() <= 2e-5
        assert np.abs(Li[b] - np.linalg.inv(L)).max() <= 2e-5 * np.abs(np.linalg.inv(L)).max() * max(1, C / 64)


@pytest.mark.parametrize("C,batch", [(3, 2), (23, 2), (64, 3), (181, 2), (256, 4), (512, 1)])
def test_spd_sqrt_vs_eigh_fp64(dev, C, batch):
    """histmatch.py:30-31: Q = V sqrt(L) V^T and its inverse, Newton-Schulz on the device vs numpy eigh in fp64"""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(C + batch)
    A = np.stack([feature_cov(rng, C, 4 * C + 50) for _ in range(batch)])
    Y, Z = ops.spd_sqrt(cu(A, dev), lambda_min=1.0)
    Y, Z = Y.cpu().numpy().astype(np.float64), Z.cpu().numpy().astype(np.float64)
    for b in range(batch):
        w, V = np.linalg.eigh(A[b].astype(np.float64))
        assert w[0] >= 0.99
        q, qi = (V * np.sqrt(w)) @ V.T, (V / np.sqrt(w)) @ V.T
        assert np.abs(Y[b] - q).max() <= 1e-5 * np.abs(q).max()
        assert np.abs(Z[b] - qi).max() <= 1e-5 * np.abs(qi).max()
        assert np.abs(Y[b] @ Z[b] - np.eye(C)).max() <= 2e-5


@pytest.mark.parametrize("C,top,bound", [(96, 4e3, 1.0), (96, 4e3, 0.0), (256, 1e5, 1.0), (181, 1e6, 1.0), (256, 1e5, 0.0)])
def test_spd_sqrt_ill_conditioned(dev, C, top, bound):
    """|A|_F / lambda_min up to 1e6 (far beyond VGG features with eps = 1): the scaled iteration converges with the
    default count, with the caller's spectrum bound and without one, and stays on its fixed point (no late divergence:
    the true-product form of the coupled iteration is the stable one)"""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((C, C)))
    w = np.concatenate([[1.0, 1.5, 2.0], np.geomspace(3.0, top, C - 3)])
    A = ((q * w) @ q.T).astype(np.float32)
    Y, Z = ops.spd_sqrt(cu(A[None], dev), lambda_min=bound)
    Y, Z = Y[0].cpu().numpy().astype(np.float64), Z[0].cpu().numpy().astype(np.float64)
    ref = (q * np.sqrt(w)) @ q.T
    refi = (q / np.sqrt(w)) @ q.T
    # fp32 round-off of the inverse root grows with the conditioning: ~ 1e-7 * sqrt(kappa)
    assert np.abs(Y - ref).max() <= 1e-4 * np.abs(ref).max()
    assert np.abs(Z - refi).max() <= max(1e-4, 4e-6 * np.sqrt(top)) * np.abs(refi).max()
    assert np.abs(Y @ Z - np.eye(C)).max() <= max(1e-4, 4e-7 * np.sqrt(top) * 10)


@pytest.mark.parametrize("C,top", [(64, 3.0), (256, 40.0), (256, 4e3), (181, 1e5)])
def test_spd_sqrt_iteration_count_decided_on_device(dev, C, top):
    """The Newton-Schulz iterations that run are what the worst matrix of the launch needs (ns_init_kernel; the later
    launches switch themselves off): same square roots as the full count to fp32 round-off, for a batch that mixes a
    well and a badly conditioned matrix.  `optex::ns_adaptive` is an internal switch of the library, not ABI."""
    import ctypes
    from optimaltextures_amd import _lib, ops
    rng = np.random.default_rng(C)
    q, _ = np.linalg.qr(rng.standard_normal((C, C)))
    mats = []
    for t in (top, 2.0):
        w = np.concatenate([[1.0, 1.5], np.geomspace(2.0, t, C - 2)])
        mats.append(((q * w) @ q.T).astype(np.float32))
    A = cu(np.stack(mats), dev)
    flag = ctypes.c_bool.in_dll(_lib.lib(), "_ZN5optex11ns_adaptiveE")
    assert flag.value
    Ya, Za = [x.clone() for x in ops.spd_sqrt(A, lambda_min=1.0)]
    try:
        flag.value = False
        Yf, Zf = [x.clone() for x in ops.spd_sqrt(A, lambda_min=1.0)]
    finally:
        flag.value = True
    tol = max(2e-6, 2e-7 * np.sqrt(top))
    assert (Ya - Yf).abs().max().item() <= tol * Yf.abs().max().item()
    assert (Za - Zf).abs().max().item() <= tol * Zf.abs().max().item()
    for b, t in enumerate((top, 2.0)):
        w = np.concatenate([[1.0, 1.5], np.geomspace(2.0, t, C - 2)])
        ref = (q * np.sqrt(w)) @ q.T
        assert np.abs(Ya[b].cpu().numpy().astype(np.float64) - ref).max() <= 1e-5 * max(1.0, np.sqrt(top) / 30) * np.abs(ref).max()


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
@pytest.mark.parametrize("C,S,Ss", [(8, 1, 1), (23, 3, 1), (64, 2, 2), (181, 2, 1), (256, 2, 1)])
def test_transfer_operator_vs_oracle(dev, mode, C, S, Ss):
    """optex_transfer_operator (histmatch.py:24-42) against the oracle's fp64 operator built from the same features"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.ops import Seg
    rng = np.random.default_rng(C + S + Ss)
    n, ns = 3 * C + 200, 2 * C + 333
    t = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    s = relu_feat(rng, Ss, C, ns, scale=1.5, shift=0.5)
    _, cov_t = ops.linear_stats(Seg.of(cu(t, dev)), pool=False)
    _, cov_s = ops.linear_stats(Seg.of(cu(s, dev)), pool=False)
    for eps in (0.0, 1.0):  # spectrum bound unknown (the public default) / the eps * I that linear_stats added
        Tt = ops.transfer_operator_t(cov_t, cov_s, mode, eps).cpu().numpy()
        for i in range(S):
            _, T = orc.linear_match(t[i], 1, s[i if Ss > 1 else 0], 1, mode, return_T=True)
            assert np.abs(Tt[i].T - T).max() <= 2e-5 * max(np.abs(T).max(), 1.0), f"segment {i} eps {eps}"


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
@pytest.mark.parametrize("S,Ss,C,n,ns,blend", [(2, 1, 32, 1024, 768, False), (1, 1, 16, 576, 560, True),
                                                (3, 3, 23, 400, 300, True), (2, 1, 181, 2048, 1500, False)])
def test_ot_loop_linear_modes_vs_oracle_chain(dev, mode, S, Ss, C, n, ns, blend):
    """optex_ot_loop modes 2-4 over 4 iterations with explicit rotations: the literal sequence (rotate, statistics,
    operator, apply, rotate back, blend) against the same chain of oracle calls; the style side enters only through its
    statistics rotated as matrices (cov(S R) = R^T cov(S) R), which must not be visible at the tolerance.  The
    single-affine fast path (fuse_rotations) agrees with the literal one to fp32 round-off."""
    from optimaltextures_amd import ops
    rng = np.random.default_rng(S + C + n)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, Ss, C, ns, scale=1.5, shift=0.5)
    content = relu_feat(rng, S, C, n, scale=2.0) if blend else None
    lr = orc.LegacyRNG(78)
    iters = 4
    R = np.stack([orc.random_rotation(C, lr) for _ in range(iters)]).astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    want = np.empty_like(x)
    for s in range(S):
        w = x[s]
        for it in range(iters):
            rp, rs = orc.rotate_cm(w, R[it]), orc.rotate_cm(sty[s if Ss > 1 else 0], R[it])
            w = orc.unrotate_cm(orc.linear_match(rp, 1, rs, 1, mode), R[it])
            if blend:
                w = orc.content_blend(w, content[s], 0.05)
        want[s] = w
    outs = {}
    # default (apply + rotation back as one GEMM), single-affine fast path, literal three GEMMs, collapsed chain (SURVEY
    # 7.4-3: statistics propagated analytically, one feature-map GEMM per call; with a content blend 3 runs as 1)
    for fused in (0, 1, 2, 3):
        xd = cu(x, dev)
        ops.ot_loop(mode, xd, cu(sty, dev), cu(R, dev), cu(Rt, dev), content=cu(content, dev) if blend else None,
                    strength=0.05 if blend else 0.0, fuse_rotations=fused)
        outs[fused] = xd.cpu().numpy()
        err = np.abs(outs[fused] - want).max() / np.abs(want).max()
        print(f"{mode} fused={fused} C={C}: rel err {err:.2e}")
        assert err <= 3 * LIN_TOL, f"fused={fused}"
    assert np.abs(outs[1] - outs[0]).max() <= 5e-5 * np.abs(want).max()
    assert np.abs(outs[2] - outs[0]).max() <= 2e-5 * np.abs(want).max()
    assert np.abs(outs[3] - outs[0]).max() <= 1e-4 * np.abs(want).max()


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_collapsed_chain_at_bench_shape(dev, mode):
    """fuse_rotations = 3 at the bench's own shape (8 textures of [256, 16384] against a [256, 12288] style, 13 iterations —
    the longest chain of the default schedule): the collapsed chain stays within 2e-4 of the literal loop's result"""
    from optimaltextures_amd import ops, rotation
    rng = np.random.default_rng(77)
    S, C, n, ns, iters = 8, 256, 16384, 12288, 13
    mix = (rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)   # correlated channels, like VGG features
    x = np.maximum(np.einsum("ck,skn->scn", mix, rng.standard_normal((S, C, n)).astype(np.float32)) * 3 + 0.4, 0).astype(np.float32)
    sty = np.maximum(np.einsum("ck,skn->scn", mix, rng.standard_normal((1, C, ns)).astype(np.float32)) * 5 + 0.2, 0).astype(np.float32)
    R, Rt = rotation.rotations(C, iters, dev, rng=np.random.RandomState(3))
    outs = {}
    for fused in (0, 3):
        xd = cu(x, dev)
        ops.ot_loop(mode, xd, cu(sty, dev), R, Rt, fuse_rotations=fused)
        outs[fused] = xd
    scale = outs[0].abs().max().item()
    err = (outs[3] - outs[0]).abs().max().item() / scale
    print(f"{mode}: collapsed vs literal, 13 iterations at [8, 256, 16384]: {err:.2e} of max |x| = {scale:.2f}")
    assert err <= 2e-4
    # and the chain did something: the result carries the style's statistics
    mu = outs[3].mean(dim=2)
    assert (mu - cu(sty, dev)[0].mean(dim=1)[None]).abs().max().item() <= 1e-3 * scale


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
def test_ot_loop_linear_chain13_reference_golden(dev, golden, mode):
    """the reference's own 13-step chain (tests/golden/optimal_transport.npz, captured rotations) through optex_ot_loop"""
    from optimaltextures_amd import ops
    g = golden("optimal_transport.npz")
    past, sty = g["pastiche"], g["style"]          # NHWC [1, 24, 24, 16], [1, 20, 28, 16]
    C = past.shape[-1]
    R = g[f"chain13_R_{mode}"].astype(np.float32)
    Rt = np.ascontiguousarray(R.transpose(0, 2, 1))
    x = np.ascontiguousarray(past.reshape(-1, C).T)[None]
    s = np.ascontiguousarray(sty.reshape(-1, C).T)[None]
    for fused in (0, 1, 2, 3):
        xd = cu(x, dev)
        ops.ot_loop(mode, xd, cu(s, dev), cu(R, dev), cu(Rt, dev), fuse_rotations=fused)
        got = xd.cpu().numpy()[0].T.reshape(past.shape)
        want = g[f"chain13_out_{mode}"]
        assert np.abs(got - want).max() <= CHAIN_TOL * np.abs(want).max(), f"fused={fused}"


def test_driver_linear_modes_use_the_fused_loop(dev):
    """driver.ot_iterations routes chol / pca / sym through optex_ot_loop (no host-side factorization): profile classes
    of the device linalg kernels show up, and the result equals the oracle chain"""
    from optimaltextures_amd import ops
    from optimaltextures_amd.driver import ot_iterations
    rng = np.random.default_rng(5)
    S, C, n, ns, iters = 2, 48, 900, 640, 3
    x = relu_feat(rng, S, C, n, scale=2.0)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.3)
    ops.profile_collect()
    ops.profile_enable(True)
    got = ot_iterations(cu(x, dev), cu(sty, dev), "chol", iters, rng=np.random.RandomState(12)).cpu().numpy()
    ops.profile_enable(False)
    prof = ops.profile_collect()
    assert prof["chol_inv"]["launches"] == iters + 1 and prof["linalg_gemm"]["launches"] >= iters
    lr = orc.LegacyRNG(12)
    Rs = [orc.random_rotation(C, lr).astype(np.float32) for _ in range(iters)]  # one sequence shared by the segments
    for s in range(S):
        w = x[s]
        for R in Rs:
            w = orc.unrotate_cm(orc.linear_match(orc.rotate_cm(w, R), 1, orc.rotate_cm(sty[0], R), 1, "chol"), R)
        assert np.abs(got[s] - w).max() <= 3 * LIN_TOL * np.abs(w).max()


@pytest.mark.parametrize("mode", ["chol", "pca", "sym"])
@pytest.mark.parametrize("S,C,n,ns,blend,fused", [(3, 32, 1024, 768, False, 0), (2, 23, 400, 300, True, 2), (4, 181, 4096, 3072, False, 0),
                                                   (2, 64, 2048, 1500, True, 1), (3, 96, 2048, 1500, False, 3)])
def test_linear_modes_per_texture_rotation_streams(dev, mode, S, C, n, ns, blend, fused):
    """un-shared rotations in the linear modes (optex.py:168 run once per image): texture i of a batch driven by one numpy
    stream per texture equals the B = 1 run with stream i (to round-off: the split-K partition of the covariance depends on
    the batch), and the oracle chain with the same rotations"""
    from optimaltextures_amd import ops, rotation
    rng = np.random.default_rng(S * C + n)
    x = relu_feat(rng, S, C, n, scale=2.0, shift=0.3)
    sty = relu_feat(rng, 1, C, ns, scale=1.5, shift=0.5)
    content = relu_feat(rng, S, C, n, scale=2.0) if blend else None
    iters, seeds = 3, [500 + 11 * i for i in range(S)]
    R, Rt = rotation.rotations_per_segment(C, iters, dev, [np.random.RandomState(sd) for sd in seeds])
    xd = cu(x, dev)
    ops.ot_loop(mode, xd, cu(sty, dev), R, Rt, content=cu(content, dev) if blend else None, strength=0.05 if blend else 0.0,
                fuse_rotations=fused)
    batch = xd.cpu().numpy()
    Rh = R.cpu().numpy()
    for i in range(S):
        xi = cu(x[i:i + 1], dev)
        ops.ot_loop(mode, xi, cu(sty, dev), R[i].contiguous(), Rt[i].contiguous(), content=cu(content[i:i + 1], dev) if blend else None,
                    strength=0.05 if blend else 0.0, fuse_rotations=fused)
        one = xi.cpu().numpy()[0]
        assert np.abs(batch[i] - one).max() <= 2e-5 * np.abs(one).max(), f"texture {i}"
        w = x[i]
        for it in range(iters):
            rp, rs = orc.rotate_cm(w, Rh[i, it]), orc.rotate_cm(sty[0], Rh[i, it])
            w = orc.unrotate_cm(orc.linear_match(rp, 1, rs, 1, mode), Rh[i, it])
            if blend:
                w = orc.content_blend(w, content[i], 0.05)
        assert np.abs(batch[i] - w).max() <= 3 * LIN_TOL * np.abs(w).max(), f"texture {i} vs oracle"
    assert np.abs(batch[0] - batch[1]).max() > 1e-3 * np.abs(batch).max()
