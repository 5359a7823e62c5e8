#!/usr/bin/env python3
"""Generate the golden input/output vectors in tests/golden/*.npz.

Runs ONLY in the build container: it imports the reference implementation read-only from
/root/reference (with empty stub modules for the two packages the reference imports but the hot
path never touches: kornia and torchvision) and records inputs + outputs of the reference's own
functions.  Nothing of the reference's source is copied: the fixtures hold data only.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz

Reference functions exercised (file:line in /root/reference):
    util.get_iters_and_sizes  util.py:68-86        util.get_size / round32  util.py:33-42,93-94
    optex.random_rotation     optex.py:142-149     optex.optimal_transport  optex.py:167-177
    histmatch.hist_match      histmatch.py:5-46    histmatch.cdf_match      histmatch.py:49-69
    histmatch.interp          histmatch.py:72-92   optex.fit_pca            optex.py:180-190
    optex.mix_style_features  optex.py:193-206
plus the two ATen ops whose exact CPU semantics the cdf path depends on (torch.histc, torch.linspace).
Environment the vectors were captured with: torch 2.10.0+rocm7.0 (CPU), numpy 2.2.6, scipy 1.15.3.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in [
        "kornia", "kornia.color", "kornia.color.hls",
        "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
    ]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["kornia.color.hls"].hls_to_rgb = None
    sys.modules["kornia.color.hls"].rgb_to_hls = None
    sys.path.insert(0, REF)
    import histmatch  # noqa
    import optex  # noqa
    import util  # noqa
    return optex, histmatch, util


def relu_features(rng, b, h, w, c, scale=1.0, shift=0.0):
    """ReLU-like features: ~half zeros (ties), positive tail."""
    x = rng.standard_normal((b, h, w, c)).astype(np.float32) * scale + shift
    return np.maximum(x, 0).astype(np.float32)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def main():
    optex, histmatch, util = import_reference()
    torch.set_num_threads(1)  # reductions/GEMMs in a fixed order
    meta = dict(torch=torch.__version__, numpy=np.__version__)
    import scipy
    meta["scipy"] = scipy.__version__

    # ------------------------------------------------------------------ G-A0 schedule + sizes
    sched = {}
    for size in (256, 512, 1024, 2048):
        iters, sizes = util.get_iters_and_sizes(size, 500, 5, True)
        sched[f"iters_{size}"] = np.array(iters, dtype=np.int64)
        sched[f"sizes_{size}"] = np.array(sizes, dtype=np.int64)
    for (iters_total, passes) in ((300, 3), (1000, 7)):
        iters, sizes = util.get_iters_and_sizes(768, iters_total, passes, True)
        sched[f"iters_768_{iters_total}_{passes}"] = np.array(iters, dtype=np.int64)
        sched[f"sizes_768_{iters_total}_{passes}"] = np.array(sizes, dtype=np.int64)
    gs_in, gs_out = [], []
    for args in [(512, 1.0, 1141, 1600, True), (256, 1.0, 1141, 1600, True), (512, 1.0, 402, 402, True),
                 (512, 0.5, 736, 512, False), (320, 1.0, 736, 512, False), (1024, 1.0, 512, 512, True),
                 (448, 2.0, 416, 416, False), (512, 1.0, 512, 512, False)]:
        gs_in.append([args[0], args[1], args[2], args[3], int(args[4])])
        gs_out.append(list(util.get_size(*args)))
    sched["get_size_in"] = np.array(gs_in, dtype=np.float64)
    sched["get_size_out"] = np.array(gs_out, dtype=np.int64)
    r32 = np.array([0, 1, 31, 32, 33, 255, 256, 257, 500], dtype=np.int64)
    sched["round32_in"] = r32
    sched["round32_out"] = np.array([util.round32(int(v)) for v in r32], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **sched)

    # ------------------------------------------------------------------ G-A1 rotations
    rot = {}
    for n, seed in ((2, 1), (3, 2), (4, 0), (23, 3), (64, 4)):
        np.random.seed(seed)
        R = optex.random_rotation(n).numpy()
        assert R.dtype == np.float64
        rot[f"R_{n}_seed{seed}"] = R
    big = {}
    for n, seed in ((170, 5), (256, 6), (512, 7)):
        np.random.seed(seed)
        R = optex.random_rotation(n).numpy()
        big[n] = R
        rot[f"sha256_f64_{n}_seed{seed}"] = np.frombuffer(hashlib.sha256(R.tobytes()).digest(), dtype=np.uint8)
        rot[f"R32_{n}_seed{seed}"] = R.astype(np.float32)  # what optex.py:168 `.to(feature)` produces
        rot[f"det_{n}"] = np.array(np.linalg.det(R))
        rot[f"ortherr_{n}"] = np.array(np.abs(R @ R.T - np.eye(n)).max())
    # two consecutive draws share the legacy gaussian cache: pin the stream continuity
    np.random.seed(11)
    rot["R_5_seed11_first"] = optex.random_rotation(5).numpy()
    rot["R_5_seed11_second"] = optex.random_rotation(5).numpy()
    np.savez_compressed(os.path.join(OUT, "rotation.npz"), **rot)

    # ------------------------------------------------------------------ G-A8 interp
    it = {}
    it["ka1_x"] = np.array([-0.5, 0, 0.5, 1, 1.5, 2.5, 3], dtype=np.float32)
    it["ka1_xp"] = np.array([0, 1, 2, 3], dtype=np.float32)
    it["ka1_fp"] = np.array([0, 10, 40, 90], dtype=np.float32)
    it["ka1_out"] = histmatch.interp(t(it["ka1_x"]), t(it["ka1_xp"]), t(it["ka1_fp"])).numpy()
    it["ka2_x"] = np.array([0.5, 1, 1.5, 2], dtype=np.float32)
    it["ka2_xp"] = np.array([0, 1, 1, 1, 2], dtype=np.float32)
    it["ka2_fp"] = np.array([0, 1, 2, 3, 4], dtype=np.float32)
    it["ka2_out"] = histmatch.interp(t(it["ka2_x"]), t(it["ka2_xp"]), t(it["ka2_fp"])).numpy()
    rng = np.random.default_rng(100)
    for case in range(6):
        n_xp = 256
        inc = rng.random(n_xp).astype(np.float32)
        if case >= 2:  # flat runs / duplicates in xp (empty histogram bins)
            inc[rng.random(n_xp) < 0.4] = 0.0
        if case >= 4:  # flat runs in fp as well
            pass
        xp = np.cumsum(inc, dtype=np.float32)
        xp = (xp / xp[-1]).astype(np.float32)
        fp = np.cumsum(rng.random(n_xp).astype(np.float32) * (rng.random(n_xp) > (0.3 if case >= 4 else 0.0)),
                       dtype=np.float32)
        x = rng.random(2000).astype(np.float32)
        x[:50] = xp[rng.integers(0, n_xp, 50)]  # exact hits on knots
        x[50] = 0.0
        x[51] = 1.0
        x = np.minimum(x, xp[-1])
        it[f"rand{case}_x"], it[f"rand{case}_xp"], it[f"rand{case}_fp"] = x, xp, fp
        it[f"rand{case}_out"] = histmatch.interp(t(x), t(xp), t(fp)).numpy()
    np.savez_compressed(os.path.join(OUT, "interp.npz"), **it)

    # ------------------------------------------------------------------ histc / linspace raw semantics
    hl = {}
    rng = np.random.default_rng(200)
    xs, los, his, hs, es = [], [], [], [], []
    for case in range(12):
        n = 5000
        if case % 3 == 0:
            x = rng.standard_normal(n).astype(np.float32) * np.float32(10 ** rng.uniform(-3, 3))
        elif case % 3 == 1:
            x = np.maximum(rng.standard_normal(n), 0).astype(np.float32) * np.float32(rng.uniform(0.1, 8))
        else:
            x = (rng.integers(-40, 40, n) / np.float32(7.0)).astype(np.float32)  # lattice values on bin borders
        lo = np.float32(x.min() - (rng.random() if case % 2 else 0.0))
        hi = np.float32(x.max() + (rng.random() if case % 4 == 1 else 0.0))
        h = torch.histc(t(x), 256, float(lo), float(hi)).numpy()
        e = torch.linspace(torch.tensor(lo), torch.tensor(hi), 257).numpy()
        xs.append(x), los.append(lo), his.append(hi), hs.append(h), es.append(e)
    hl["x"], hl["lo"], hl["hi"] = np.stack(xs), np.array(los, np.float32), np.array(his, np.float32)
    hl["hist"], hl["edges257"] = np.stack(hs), np.stack(es)
    # many (lo,hi) pairs for linspace alone
    lo = (rng.standard_normal(400) * 10 ** rng.uniform(-2, 2, 400)).astype(np.float32)
    hi = (lo + np.abs(rng.standard_normal(400)).astype(np.float32) * np.float32(5) + np.float32(1e-3)).astype(np.float32)
    hl["ls_lo"], hl["ls_hi"] = lo, hi
    hl["ls_edges257"] = np.stack([torch.linspace(torch.tensor(a), torch.tensor(b), 257).numpy() for a, b in zip(lo, hi)])
    # degenerate lo == hi: histc widens to [lo-1, hi+1]
    xc = np.full(100, 3.0, np.float32)
    hl["const_hist"] = torch.histc(t(xc), 256, 3.0, 3.0).numpy()
    np.savez_compressed(os.path.join(OUT, "histc_linspace.npz"), **hl)

    # ------------------------------------------------------------------ G-A6 cdf_match with intermediates
    cm = {}
    rng = np.random.default_rng(300)
    C = 8
    tgt = relu_features(rng, 1, 16, 16, C, 2.0, 0.3)[0].reshape(-1, C).T.copy()  # [C, 256]
    src = relu_features(rng, 1, 20, 12, C, 1.5, -0.2)[0].reshape(-1, C).T.copy()  # [C, 240]
    tgt[5] = 2.0  # constant target channel
    src[6] = 0.5  # constant source channel
    tgt[7] = 3.0
    src[7] = 3.0  # both constant and equal: lo == hi
    # channels 0-1 dense (rotated-like, no ties)
    tgt[0] = rng.standard_normal(256).astype(np.float32)
    src[0] = (rng.standard_normal(240) * 2 + 1).astype(np.float32)
    tgt[1] = (rng.standard_normal(256) * 0.01 + 5).astype(np.float32)
    src[1] = (rng.standard_normal(240) * 3 - 5).astype(np.float32)
    cm["target"], cm["source"] = tgt, src
    cm["out"] = histmatch.cdf_match(t(tgt), t(src)).numpy()
    lo_l, hi_l, ht_l, hs_l, be_l, rm_l = [], [], [], [], [], []
    for tc, sc in zip(t(tgt), t(src)):
        lo = torch.min(tc.min(), sc.min())
        hi = torch.max(tc.max(), sc.max())
        ht = torch.histc(tc, 256, lo, hi)
        hs = torch.histc(sc, 256, lo, hi)
        be = torch.linspace(lo, hi, 257)[1:]
        tcdf = ht.cumsum(0)
        tcdf = tcdf / tcdf[-1]
        scdf = hs.cumsum(0)
        scdf = scdf / scdf[-1]
        rm = histmatch.interp(tcdf, scdf, be)
        lo_l.append(lo.numpy()), hi_l.append(hi.numpy()), ht_l.append(ht.numpy()), hs_l.append(hs.numpy())
        be_l.append(be.numpy()), rm_l.append(rm.numpy())
    cm["lo"], cm["hi"] = np.array(lo_l), np.array(hi_l)
    cm["hist_t"], cm["hist_s"], cm["bin_edges"], cm["remapped"] = map(np.stack, (ht_l, hs_l, be_l, rm_l))
    # a larger dense case: rotated-like gaussian columns
    tg2 = (rng.standard_normal((16, 4096)) * rng.uniform(0.5, 4, (16, 1)) + rng.uniform(-2, 2, (16, 1))).astype(np.float32)
    sr2 = (rng.standard_normal((16, 3000)) * rng.uniform(0.5, 4, (16, 1)) + rng.uniform(-2, 2, (16, 1))).astype(np.float32)
    cm["target2"], cm["source2"] = tg2, sr2
    cm["out2"] = histmatch.cdf_match(t(tg2), t(sr2)).numpy()
    # tie-heavy un-rotated relu features (mix_style_features calls hist_match un-rotated)
    tg3 = relu_features(rng, 1, 32, 32, 6, 1.0)[0].reshape(-1, 6).T.copy()
    sr3 = relu_features(rng, 1, 32, 24, 6, 2.0)[0].reshape(-1, 6).T.copy()
    cm["target3"], cm["source3"] = tg3, sr3
    cm["out3"] = histmatch.cdf_match(t(tg3), t(sr3)).numpy()
    # appendix-B style degenerate channels
    cm["deg_const_t_out"] = histmatch.cdf_match(torch.full((1, 64), 2.0), torch.linspace(0, 4, 80)[None]).numpy()
    cm["deg_const_s_out"] = histmatch.cdf_match(torch.linspace(0, 1, 64)[None], torch.full((1, 80), 0.5)).numpy()
    cm["deg_both_out"] = histmatch.cdf_match(torch.full((1, 64), 3.0), torch.full((1, 80), 3.0)).numpy()
    np.savez_compressed(os.path.join(OUT, "cdf_match.npz"), **cm)

    # ------------------------------------------------------------------ G-A5 hist_match (all modes)
    hm = {}
    rng = np.random.default_rng(400)
    tg = relu_features(rng, 1, 16, 16, 8, 2.0, 0.5)
    sr = relu_features(rng, 1, 20, 12, 8, 1.0, 0.2)
    tgb = relu_features(rng, 2, 12, 12, 8, 2.0, 0.5)  # pooled B=2 target
    tgb[1] += 1.5
    srb2 = relu_features(rng, 2, 10, 14, 8, 1.0, 0.2)  # B_s == B_t
    hm["target"], hm["source"], hm["target_b2"], hm["source_b2"] = tg, sr, tgb, srb2
    for mode in ("chol", "pca", "sym", "cdf"):
        hm[f"out_{mode}"] = histmatch.hist_match(t(tg), t(sr), mode).contiguous().numpy()
        hm[f"out_b2_{mode}"] = histmatch.hist_match(t(tgb), t(sr), mode).contiguous().numpy()
    for mode in ("chol", "pca", "sym"):
        hm[f"out_b2s2_{mode}"] = histmatch.hist_match(t(tgb), t(srb2), mode).contiguous().numpy()
    tgc = tg.copy()
    tgc[..., 3] = 1.25  # constant target channel stays finite thanks to eps = 1
    hm["target_const"] = tgc
    hm["out_const_chol"] = histmatch.hist_match(t(tgc), t(sr), "chol").contiguous().numpy()
    # a wider one: C = 32
    tgw = relu_features(rng, 1, 24, 24, 32, 3.0, 0.1)
    srw = relu_features(rng, 1, 20, 28, 32, 2.0, 0.4)
    hm["target_w"], hm["source_w"] = tgw, srw
    for mode in ("chol", "pca", "sym"):
        hm[f"out_w_{mode}"] = histmatch.hist_match(t(tgw), t(srw), mode).contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, "hist_match.npz"), **hm)

    # ------------------------------------------------------------------ G-A2/A10 optimal_transport
    ot = {}
    rng = np.random.default_rng(500)
    C = 16
    past = relu_features(rng, 1, 24, 24, C, 2.0, 0.3)
    sty = relu_features(rng, 1, 20, 28, C, 1.5, 0.5)
    ot["pastiche"], ot["style"] = past, sty
    captured = []
    orig_rr = optex.random_rotation

    def recording_rr(N, device="cpu", impl="scipy"):
        R = orig_rr(N, device, impl)
        captured.append(R.numpy().copy())
        return R

    optex.random_rotation = recording_rr
    for mode in ("chol", "pca", "sym", "cdf"):
        captured.clear()
        np.random.seed(42)
        out = optex.optimal_transport(t(past), t(sty), mode)
        R = captured[0]
        ot[f"R_{mode}"] = R  # fp64, as returned by random_rotation
        R32 = torch.from_numpy(R).to(torch.float32)
        rp, rs = t(past) @ R32, t(sty) @ R32
        ot[f"rotated_pastiche_{mode}"] = rp.numpy()
        ot[f"rotated_style_{mode}"] = rs.numpy()
        ot[f"matched_{mode}"] = histmatch.hist_match(rp, rs, mode).contiguous().numpy()
        ot[f"out_{mode}"] = out.contiguous().numpy()
    for mode in ("chol", "pca", "sym"):
        captured.clear()
        np.random.seed(43)
        x = t(past)
        for _ in range(13):
            x = optex.optimal_transport(x, t(sty), mode)
        ot[f"chain13_R_{mode}"] = np.stack(captured)
        ot[f"chain13_out_{mode}"] = x.contiguous().numpy()
    # content blend epilogue of the caller (optex.py:115-117), l = 2 (relu3_1): strength = cs / 2**(4-2)
    captured.clear()
    np.random.seed(44)
    content = relu_features(rng, 1, 24, 24, C, 2.0, 0.0)
    x = t(past).clone()
    for _ in range(3):
        x = optex.optimal_transport(x, t(sty), "chol")
        x += (0.2 / 2 ** (4 - 2)) * (t(content) - x)
    ot["blend_content"], ot["blend_R"], ot["blend_out"] = content, np.stack(captured), x.contiguous().numpy()
    # batch semantics: B=2 pooled
    pastb = relu_features(rng, 2, 12, 12, C, 2.0, 0.3)
    ot["pastiche_b2"] = pastb
    for mode in ("chol", "cdf"):
        captured.clear()
        np.random.seed(45)
        ot[f"out_b2_{mode}"] = optex.optimal_transport(t(pastb), t(sty), mode).contiguous().numpy()
        ot[f"R_b2_{mode}"] = captured[0]
    # colour-transfer shaped call: C = 3, mode cdf (optex.py:131-134)
    captured.clear()
    np.random.seed(46)
    p3 = rng.random((1, 32, 32, 3)).astype(np.float32)
    s3 = rng.random((1, 32, 32, 3)).astype(np.float32) ** 2
    ot["pastiche_c3"], ot["style_c3"] = p3, s3
    ot["out_c3_cdf"] = optex.optimal_transport(t(p3), t(s3), "cdf").contiguous().numpy()
    ot["R_c3_cdf"] = captured[0]
    optex.random_rotation = orig_rr
    np.savez_compressed(os.path.join(OUT, "optimal_transport.npz"), **ot)

    # ------------------------------------------------------------------ "next" rows: fit_pca, mix_style_features
    nx = {}
    rng = np.random.default_rng(600)
    base = rng.standard_normal((1, 20, 24, 6)).astype(np.float32)
    mixm = rng.standard_normal((6, 24)).astype(np.float32)
    feat = np.maximum(base @ mixm, 0).astype(np.float32)  # low-rank-ish relu features, C = 24
    nx["pca_in"] = feat
    f, e = optex.fit_pca(t(feat))
    nx["pca_features"], nx["pca_eigvecs"] = f.numpy(), e.numpy()
    nx["pca_k"] = np.array(e.shape[1])
    sfA = relu_features(rng, 1, 12, 12, 8, 2.0, 0.1)
    sfB = relu_features(rng, 1, 12, 12, 8, 1.0, 0.6)
    sf = np.concatenate([sfA, sfB])
    torch.manual_seed(7)
    mask = torch.ceil(torch.rand(6, 6) - 0.5)[None, None]
    nx["mix_style"], nx["mix_mask"] = sf, mask.numpy()
    for mode in ("chol", "cdf"):
        nx[f"mix_out_{mode}"] = optex.mix_style_features([t(sf)], mask, 0.5, mode)[0].contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, "next_rows.npz"), **nx)

    with open(os.path.join(OUT, "VERSIONS.txt"), "w") as fh:
        for k, v in meta.items():
            fh.write(f"{k} {v}\n")
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn:28s} {os.path.getsize(os.path.join(OUT, fn)):9d} B")


if __name__ == "__main__":
    main()
