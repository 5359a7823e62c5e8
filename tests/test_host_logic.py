"""CPU checks of the host-side plumbing against the reference's own outputs (tests/golden/schedule.npz, captured by
tests/golden/gen_golden.py from util.py:33-42,68-86,93-94): the iteration / size schedule that DEFINES the metric's
workload ("default iters"), the size helpers, the CLI surface."""
import numpy as np
import pytest

from optimaltextures_amd.util import get_iters_and_sizes, get_size, layer_iters, round32


@pytest.mark.parametrize("key,args", [("256", (256, 500, 5)), ("512", (512, 500, 5)), ("1024", (1024, 500, 5)),
                                      ("2048", (2048, 500, 5)), ("768_300_3", (768, 300, 3)), ("768_1000_7", (768, 1000, 7))])
def test_schedule_matches_reference(golden, key, args):
    g = golden("schedule.npz")
    iters, sizes = get_iters_and_sizes(args[0], args[1], args[2], True)
    assert np.array_equal(np.array(iters), g[f"iters_{key}"])
    assert np.array_equal(np.array(sizes), g[f"sizes_{key}"])


def test_default_workload_constants(golden):
    """relu3_1 at --size 512 runs 13/12/10/9/8 = 52 iterations on 64^2..128^2 pixels (466 176 pixel-iterations);
    all five layers 493; the table is read with [l - 1] (optex.py:112), so relu5_1 takes the LAST column."""
    table, sizes = get_iters_and_sizes(512, 500, 5, True)
    assert sizes == [256, 320, 384, 448, 512]
    relu3 = [layer_iters(table, p, 2) for p in range(5)]   # encoder list index 2 == relu3_1
    assert relu3 == [13, 12, 10, 9, 8] and sum(relu3) == 52
    assert sum(it * (s // 4) ** 2 for it, s in zip(relu3, sizes)) == 466176
    assert sum(layer_iters(table, p, l) for p in range(5) for l in range(5)) == 493
    assert [layer_iters(table, 0, l) for l in range(5)] == [40, 8, 13, 22, 40]  # relu5_1, 4_1, 3_1, 2_1, 1_1
    assert sum(layer_iters(table, p, 4) for p in range(5)) == 160                # relu1_1 only


def test_get_size_and_round32_match_reference(golden):
    g = golden("schedule.npz")
    for (size, scale, h, w, over), want in zip(g["get_size_in"], g["get_size_out"]):
        assert list(get_size(int(size), float(scale), int(h), int(w), bool(over))) == list(want)
    assert [round32(int(v)) for v in g["round32_in"]] == list(g["round32_out"])


def test_no_multires_returns_full_size_schedule():
    """the reference crashes here (util.py:80,86 `.tolist()` on a list); ours returns the evident intent"""
    table, sizes = get_iters_and_sizes(512, 500, 5, False)
    assert sizes == [512] * 5 and len(table) == 5 and all(len(r) == 5 for r in table)


def test_cli_keeps_every_reference_flag():
    """optex.py:222-244: flag names and defaults"""
    import optex as cli
    a = cli.build_parser().parse_args([])
    want = dict(style=["style/graffiti.jpg"], content=None, batch=1, size=512, passes=5, iters=500, hist_mode="chol",
                color_transfer=None, content_strength=0.01, style_scale=1.0, mixing_alpha=0.5, no_pca=False,
                no_multires=False, seed=None, no_tf32=False, cudnn_benchmark=False, compile=False, script=False,
                device=None, memory_format="contiguous", output_dir="output/")
    for k, v in want.items():
        assert getattr(a, k) == v, k
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["--hist_mode", "nope"])
    b = cli.build_parser().parse_args(["-s", "a.jpg", "b.jpg", "-c", "c.jpg", "--hist_mode", "cdf", "--no_pca"])
    assert b.style == ["a.jpg", "b.jpg"] and b.content == "c.jpg" and b.hist_mode == "cdf" and b.no_pca


def test_cli_extension_flags_and_constructor_arguments():
    """--pca_fit / --codec_layout (the literal SVD route stays selectable for parity runs; VERDICT r3 item 9: the
    codec layout is a constructor argument, not an environment variable)"""
    import optex as cli
    from optimaltextures_amd import vgg
    from optimaltextures_amd.driver import OptimalTexture
    a = cli.build_parser().parse_args([])
    assert a.pca_fit == "gram" and a.codec_layout == "mixed"
    b = cli.build_parser().parse_args(["--pca_fit", "svd", "--codec_layout", "nchw"])
    assert b.pca_fit == "svd" and b.codec_layout == "nchw"
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["--pca_fit", "qr"])
    t = OptimalTexture(size=256, layers=(1,), codec_layout="nchw", pca_fit="svd")
    assert t.pca_fit == "svd" and t.encoders[0].codec_layout == "nchw" and t.decoders[0].codec_layout == "nchw"
    with pytest.raises(ValueError):
        OptimalTexture(size=256, layers=(1,), pca_fit="qr")
    with pytest.raises(ValueError):
        OptimalTexture(size=256, layers=(1,), codec_layout="nhwc")
    import inspect
    assert "os.environ" not in inspect.getsource(vgg) and "getenv" not in inspect.getsource(vgg)


def test_kept_rank_rule_and_margin():
    """optex.py:184-185: k = index of the first cumulative singular-value share ABOVE 0.9 (SURVEY 8f N1's known answer:
    shares [0.54, 0.808, 0.912, ...] keep k = 2), and the distance of that crossing from 0.9, which decides whether the Gram
    route's k can be trusted (driver.PCA_RANK_MARGIN)"""
    import torch
    from optimaltextures_amd import driver
    sing = torch.tensor([0.54, 0.268, 0.104, 0.05, 0.038])
    k, margin = driver._kept_rank(sing)
    assert int(k) == 2 and abs(float(margin) - 0.012) < 1e-6
    # a crossing inside round-off of 0.9: the margin says so
    sing = torch.tensor([0.5, 0.4 + 5e-7, 0.1 - 5e-7])
    k, margin = driver._kept_rank(sing)
    assert float(margin) < driver.PCA_RANK_MARGIN
    # batched, and the CPU route of fit_pca_cm is the literal SVD with the same rule
    k, margin = driver._kept_rank(torch.tensor([[0.54, 0.268, 0.104, 0.088], [0.95, 0.03, 0.01, 0.01]]))
    assert k.tolist() == [2, 0] and float(margin[1]) > 0.04
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 6, 500, generator=g) * torch.tensor([5.0, 3.0, 1.0, 0.5, 0.2, 0.1]).view(1, 6, 1)
    a = x.permute(0, 2, 1).reshape(-1, 6) - x.mean()
    sv = torch.linalg.svdvals(a)
    want_k = int((torch.cumsum(sv / sv.sum(), 0) > 0.9).to(torch.int32).argmax())
    assert int(driver._kept_rank(sv)[0]) == want_k


def test_hls_conversion_matches_colorsys_and_round_trips():
    """driver.rgb_to_hls / hls_to_rgb stand in for kornia.color.hls (optex.py:126-131; kornia is not installed):
    h in radians [0, 2pi), channel order (h, l, s) like kornia."""
    import colorsys

    import torch

    from optimaltextures_amd.driver import hls_to_rgb, rgb_to_hls
    g = torch.Generator().manual_seed(0)
    img = torch.rand(2, 3, 8, 8, generator=g)
    img[0, :, 0, 0] = 0.5          # grey pixel: hue / saturation 0
    img[0, :, 0, 1] = torch.tensor([1.0, 0.0, 0.0])
    hls = rgb_to_hls(img)
    for b in range(2):
        for y in range(8):
            for x in range(8):
                r, gg, bb = (float(v) for v in img[b, :, y, x])
                h, l, s = colorsys.rgb_to_hls(r, gg, bb)
                got = [float(v) for v in hls[b, :, y, x]]
                assert abs(got[0] - h * 2 * np.pi) < 1e-4 and abs(got[1] - l) < 1e-6 and abs(got[2] - s) < 1e-4
    assert torch.allclose(hls_to_rgb(hls), img, atol=1e-5)


# ------------------------------------------------------------------------------------------------ VGG interface / assets
REF = "/root/reference"


def test_missing_weight_file_raises_and_synthetic_is_opt_in():
    """vgg.py:147,166: the reference raises when a weight file is absent; so do we unless synthetic weights are asked for"""
    import os

    from conftest import ROOT
    from optimaltextures_amd.vgg import Decoder, Encoder
    models = os.path.join(ROOT, "assets", "models")
    with pytest.raises(FileNotFoundError, match="conv4_1"):
        Encoder(4, models)
    with pytest.raises(FileNotFoundError, match="conv5_1"):
        Decoder(5, models)
    assert Encoder(4, models, allow_synthetic=True).weights.startswith("synthetic")
    assert Encoder(3, models).weights.startswith("pretrained") and Decoder(1, models).weights.startswith("pretrained")
    assert Encoder(2, None).weights.startswith("synthetic")  # models_dir=None: all-synthetic codec, explicit


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_pretrained_codecs_load_and_keep_the_reference_key_order(depth):
    """assets/models/*.pth (the reference's state_dicts, keys "0.weight", "2.weight", ...) load into our nn.Sequential
    unchanged: same keys in the same order, same shapes"""
    import os

    import torch

    from conftest import ROOT
    from optimaltextures_amd.vgg import Decoder, Encoder
    models = os.path.join(ROOT, "assets", "models")
    for cls, fn in ((Encoder, f"vgg_normalised_conv{depth}_1.pth"), (Decoder, f"feature_invertor_conv{depth}_1.pth")):
        sd = torch.load(os.path.join(models, fn), map_location="cpu", weights_only=True)
        m = cls(depth, models)
        ours = m.model.state_dict()
        assert list(ours.keys()) == list(sd.keys())
        for k in sd:
            assert torch.equal(ours[k], sd[k])


@pytest.mark.skipif(not __import__("os").path.isdir(REF), reason="build-container only: needs /root/reference")
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_codecs_equal_the_reference_modules_with_real_weights(depth):
    """The reference's own Encoder / Decoder (vgg.py:138-171, imported read-only) against ours with the real
    conv{1,2,3}_1 weights on the CPU: outputs bit-identical, strides included (the encoder returns an NHWC VIEW of NCHW
    memory, vgg.py:153), and the shipped asset files are byte-identical to the reference's."""
    import hashlib
    import os
    import sys

    import torch

    from conftest import ROOT
    from optimaltextures_amd.vgg import Decoder, Encoder
    import types
    stubs = ["torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils"]
    added = [n for n in stubs if n not in sys.modules]
    for n in added:  # the reference's util.py imports torchvision (absent here); its vgg.py only needs to_nchw / to_nhwc
        sys.modules[n] = types.ModuleType(n)
    sys.path.insert(0, REF)
    try:
        import vgg as ref_vgg
    finally:
        sys.path.remove(REF)
        for n in added + ["util", "vgg"]:
            sys.modules.pop(n, None)
    models = os.path.join(ROOT, "assets", "models")
    for fn in (f"vgg_normalised_conv{depth}_1.pth", f"feature_invertor_conv{depth}_1.pth"):
        a = hashlib.sha256(open(os.path.join(models, fn), "rb").read()).hexdigest()
        b = hashlib.sha256(open(os.path.join(REF, "models", fn), "rb").read()).hexdigest()
        assert a == b, fn
    x = torch.rand(2, 3, 40, 56, generator=torch.Generator().manual_seed(depth))
    with torch.inference_mode():
        re_, rd = ref_vgg.Encoder(depth).eval(), ref_vgg.Decoder(depth).eval()
        oe, od = Encoder(depth, models).eval(), Decoder(depth, models).eval()
        fr, fo = re_(x), oe(x)
        assert fr.shape == fo.shape and fr.stride() == fo.stride() and torch.equal(fr, fo)
        assert torch.equal(oe.features(x), fr.permute(0, 3, 1, 2))
        dr, do = rd(fr), od(fo)
        assert dr.shape == do.shape == x.shape and torch.equal(dr, do)
        assert torch.equal(od.decode(fo.permute(0, 3, 1, 2)), dr)


def test_image_io_round_trip_and_names(tmp_path):
    """util.py:27-30,45-65: PIL-only load (LANCZOS == the removed ANTIALIAS, to_tensor == /255) and save; the style's long
    side flips role between load (PIL gives (width, height)) and the per-pass resize — sizes pinned from the reference
    (graffiti loads as 736x512 (HxW) at --size 512; lava-small 402^2 -> 416^2)"""
    import os
    from argparse import Namespace

    import torch
    from PIL import Image

    from conftest import ROOT
    from optimaltextures_amd.util import load_image, load_styles, maybe_load_content, output_name, save_image
    g = load_styles([os.path.join(ROOT, "assets/style/graffiti.jpg")], size=512, scale=1)[0]
    assert tuple(g.shape) == (1, 3, 736, 512) and g.dtype == torch.float32 and 0 <= float(g.min()) and float(g.max()) <= 1
    lava = load_styles([os.path.join(ROOT, "assets/style/lava-small.jpg")], size=512, scale=1)[0]
    assert tuple(lava.shape) == (1, 3, 416, 416)  # never upsampled at load (the inverted oversize flag, util.py:16)
    c = maybe_load_content(os.path.join(ROOT, "assets/content/rocket.jpg"), size=256)
    assert tuple(c.shape) == (1, 3, 256, 256) and maybe_load_content(None, 256) is None
    args = Namespace(style=["style/a.jpg", "x/b.png"], mixing_alpha=0.25, content="content/c.jpg", content_strength=0.05,
                     hist_mode="pca", no_pca=True, no_multires=False, style_scale=0.5, color_transfer="lum", size=1024,
                     output_dir=str(tmp_path))
    assert output_name(args) == "a_b_blend0.25_c_strength0.05_pcahist_no_pca_scale0.5_lum_1024"
    img = torch.rand(2, 3, 16, 24, generator=torch.Generator().manual_seed(0))
    paths = save_image(img, args)
    assert [os.path.basename(p) for p in paths] == [output_name(args) + "_1.png", output_name(args) + "_2.png"]
    back = load_image(paths[0], 24, oversize=True)  # PIL size (24, 16) -> get_size(24, 1, 24, 16, True) = (32, 32)
    assert tuple(back.shape) == (1, 3, 32, 32)
    raw = torch.from_numpy(__import__("numpy").array(Image.open(paths[1]).convert("RGB"))).permute(2, 0, 1) / 255.0
    assert float((raw - img[1].clamp(0, 1)).abs().max()) <= 0.5 / 255 + 1e-6


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5])
def test_encoder_out_shape_matches_the_convolutions(depth):
    """Encoder.out_shape (what a rank that only receives style features uses to know their shapes, dist.StyleSync) against
    the real layer stack, odd sizes included (ceil_mode pooling, vgg.py:26)"""
    import torch
    from optimaltextures_amd.vgg import Encoder
    enc = Encoder(depth).eval()
    for h, w in [(32, 32), (37, 50), (96, 65), (33, 129)]:
        with torch.inference_mode():
            f = enc.features(torch.rand(1, 3, h, w))
        assert tuple(f.shape[1:]) == enc.out_shape(h, w), (depth, h, w)


def test_polar_method_is_parallel_over_four_word_attempts():
    """The structure optex_legacy_normals (csrc/rotation.hip) relies on, checked against numpy itself on the host: an attempt of
    numpy's legacy polar method consumes exactly four MT19937 words whether it is accepted or not, so the normals are a pure
    function of the word stream — attempt j from words 4j .. 4j + 3, outputs = the accepted attempts in order, second value of
    a pair first.  Evaluated that way (vectorised accept / reject, the host libm's scalar log like legacy_gauss uses) the
    values equal RandomState.normal bit for bit, and the state after the draw is the state after exactly the words of the
    last accepted attempt."""
    import math
    for seed, pairs in ((42, 4000), (7, 313)):
        words_rng, ref = np.random.RandomState(seed), np.random.RandomState(seed)
        w = words_rng.randint(0, 2 ** 32, size=4 * 2 * pairs, dtype=np.uint32).astype(np.uint64)   # raw tempered words
        unit = lambda a, b: ((a >> 5).astype(np.float64) * 67108864.0 + (b >> 6).astype(np.float64)) / 9007199254740992.0
        x1, x2 = 2.0 * unit(w[0::4], w[1::4]) - 1.0, 2.0 * unit(w[2::4], w[3::4]) - 1.0
        r2 = x1 * x1 + x2 * x2
        acc = ~((r2 >= 1.0) | (r2 == 0.0))
        assert acc.sum() >= pairs
        idx = np.flatnonzero(acc)[:pairs]
        f = np.sqrt(-2.0 * np.array([math.log(v) for v in r2[idx]]) / r2[idx])
        got = np.empty(2 * pairs)
        got[0::2], got[1::2] = f * x2[idx], f * x1[idx]
        assert np.array_equal(got, ref.normal(size=2 * pairs))
        # the reference stream has consumed the words up to and including the last accepted attempt's — no more, no less
        consumed = 4 * (int(idx[-1]) + 1)
        probe = np.random.RandomState(seed)
        probe.randint(0, 2 ** 32, size=consumed, dtype=np.uint32)
        assert np.array_equal(probe.get_state()[1], ref.get_state()[1]) and probe.get_state()[2] == ref.get_state()[2]


def test_rotation_schedule_of_a_forward_call():
    """OptimalTexture.rotation_schedule: (C, iterations) per (pass, layer) in the order the loops ask for their rotations — what
    bench.py prefetches a step ahead (rotation.DeviceNormals.prefetch) and forward() checks with DeviceNormals.covers().  The
    headline configuration: relu3_1 only, no_pca, 13 / 12 / 10 / 9 / 8 iterations at C = 256 (util.py:68-86 read through
    optex.py:112)."""
    from optimaltextures_amd.driver import OptimalTexture
    tex = OptimalTexture(size=512, iters=500, passes=5, hist_mode="cdf", no_pca=True, layers=(3,), independent=True)
    assert tex.rotation_schedule() == [(256, 13), (256, 12), (256, 10), (256, 9), (256, 8)]
    tex = OptimalTexture(size=256, iters=500, passes=2, hist_mode="cdf", no_pca=True, layers=(3, 1), color_transfer="opt")
    sched = tex.rotation_schedule()
    assert [c for c, _ in sched] == [256, 64, 256, 64, 3] and sched[-1] == (3, 3) and all(k > 0 for _, k in sched)
