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
