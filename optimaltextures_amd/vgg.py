"""Truncated normalised VGG-19 encoders and their learned inverters (reference vgg.py:14-171).  These stay on
PyTorch-ROCm (MIOpen) by the north-star's scoping; only the interface is kept: Encoder(depth) maps NCHW images to
features, Decoder(depth) maps features back.  Module order inside the nn.Sequential matches the reference so that its
state_dicts (`models/vgg_normalised_conv{d}_1.pth`, `models/feature_invertor_conv{d}_1.pth`, keys "2.weight" ...) load
unchanged from `models_dir` (the relu1_1..relu3_1 files ship in assets/models; the reference's relu4_1 / relu5_1 files
are absent from its own repository, `.MISSING_LARGE_BLOBS`).  A missing file RAISES like the reference does
(vgg.py:147,166); seeded synthetic weights (variance-preserving init: same shapes, FLOPs and therefore throughput) are
an explicit opt-in: `models_dir=None`, or `allow_synthetic=True` for the depths whose files do not exist."""
import os

import torch
import torch.nn as nn

from .util import to_nchw, to_nhwc

WIDTHS = (64, 128, 256, 512, 512)
# number of extra 3x3 convs at the block's own width before the pool that leads to the NEXT block
_SAME_WIDTH_CONVS = (1, 1, 3, 3)


def _conv(cin, cout):
    return [nn.ReflectionPad2d((1, 1, 1, 1)), nn.Conv2d(cin, cout, (3, 3)), nn.ReLU()]


def encoder_layers(depth: int):
    mods = [nn.Conv2d(3, 3, (1, 1))] + _conv(3, WIDTHS[0])
    for d in range(1, depth):
        w = WIDTHS[d - 1]
        for _ in range(_SAME_WIDTH_CONVS[d - 1]):
            mods += _conv(w, w)
        mods.append(nn.MaxPool2d((2, 2), (2, 2), (0, 0), ceil_mode=True))
        mods += _conv(w, WIDTHS[d])
    return mods


def decoder_layers(depth: int):
    mods = []
    for d in range(depth, 1, -1):
        w = WIDTHS[d - 2]
        mods += _conv(WIDTHS[d - 1], w)
        mods.append(nn.UpsamplingNearest2d(scale_factor=2))
        for _ in range(_SAME_WIDTH_CONVS[d - 2]):
            mods += _conv(w, w)
    mods += [nn.ReflectionPad2d((1, 1, 1, 1)), nn.Conv2d(WIDTHS[0], 3, (3, 3))]
    return mods


def _synthetic_init(model: nn.Sequential, seed: int):
    gen = torch.Generator().manual_seed(seed)
    for m in model:
        if isinstance(m, nn.Conv2d):
            fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * (2.0 / fan_in) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.05)


# Layout policy of the fused codec path: "mixed" runs every 3x3 convolution with >= 64 output channels
# channels-last (MIOpen's MFMA implicit-GEMM kernels: 8-20 % faster than the planar Winograd assembly at these shapes,
# scripts/conv_layout_probe.py) and keeps the 3-channel ends and the feature hand-off to the OT kernels planar; the
# glue pass between two convolutions changes the layout for free.  "nchw" keeps everything planar.  A constructor argument
# of Encoder / Decoder / OptimalTexture (`codec_layout`); this is only its default — nothing is read from the environment.
CODEC_LAYOUT = "mixed"
CODEC_LAYOUTS = ("mixed", "nchw")


def _conv_channels_last(m: nn.Conv2d, layout: str) -> bool:
    # 64 -> 3 (the last decoder convolution) is the one wide 3x3 shape MIOpen runs faster planar (2.5 vs 3.0 ms)
    return layout == "mixed" and m.kernel_size == (3, 3) and m.out_channels >= 64


def _weight(m: nn.Conv2d, channels_last: bool):
    """the convolution's weight in the layout of its input; the channels-last copy is cached per module and refreshed
    when the parameter is replaced, moved or (for tensors that track it) modified in place"""
    if not channels_last:
        return m.weight
    try:
        version = m.weight._version
    except RuntimeError:  # inference tensors do not track a version counter
        version = -1
    key = (m.weight.data_ptr(), str(m.weight.device), version)
    if getattr(m, "_optex_w_cl_key", None) != key:
        m._optex_w_cl = m.weight.detach().contiguous(memory_format=torch.channels_last)
        m._optex_w_cl_key = key
    return m._optex_w_cl


def run_fused(model: nn.Sequential, x, layout=None):
    """Run one of the Sequentials above with every convolution on MIOpen (bias-free) and everything BETWEEN two
    convolutions — bias, ReLU, max-pool / upsample, reflection pad, and the change of memory layout the next
    convolution wants — in one pass of optex_vgg_glue_layout (csrc/glue.hip).  With layout "nchw" the result is
    bit-identical to model(x) (only the kernel boundaries move); with "mixed" the convolutions run through other MIOpen
    kernels, i.e. a different fp32 summation order inside the convolution.  layout None = the module default CODEC_LAYOUT."""
    from . import ops
    layout = CODEC_LAYOUT if layout is None else layout
    cur, bias = x, None
    relu = pool = up = False
    pad = 0

    def flush(next_nhwc):
        nonlocal cur, bias, relu, pool, up, pad
        cur_nhwc = (not cur.is_contiguous()) and cur.permute(0, 2, 3, 1).is_contiguous()
        if bias is not None or relu or pool or up or pad or cur_nhwc != next_nhwc or not (cur_nhwc or cur.is_contiguous()):
            cur = ops.vgg_glue(cur, bias, relu=relu, pool=pool, up=up, pad=pad, out_nhwc=next_nhwc)
        bias, relu, pool, up, pad = None, False, False, False, 0

    for m in model:
        if isinstance(m, nn.Conv2d):
            cl = _conv_channels_last(m, layout)
            flush(cl)
            cur = torch.nn.functional.conv2d(cur, _weight(m, cl), None)
            bias = m.bias
        elif isinstance(m, nn.ReLU):
            assert not (pool or up or pad), "glue order is bias, relu, pool / upsample, pad"
            relu = True
        elif isinstance(m, nn.MaxPool2d):
            assert not (up or pad)
            pool = True
        elif isinstance(m, nn.UpsamplingNearest2d):
            assert not (pool or pad)
            up = True
        elif isinstance(m, nn.ReflectionPad2d):
            pad = 1
        else:
            raise TypeError(f"unexpected module {type(m).__name__} in the VGG codec")
    flush(False)
    return cur


class _Codec(nn.Module):
    FILE = ""

    def __init__(self, depth: int, layers, models_dir=None, seed=0, allow_synthetic=False, codec_layout=None):
        super().__init__()
        assert isinstance(depth, int) and 1 <= depth <= 5
        if codec_layout is not None and codec_layout not in CODEC_LAYOUTS:
            raise ValueError(f"codec_layout must be one of {CODEC_LAYOUTS}, got {codec_layout!r}")
        self.depth = depth
        self.codec_layout = codec_layout  # None = the module default (vgg.CODEC_LAYOUT) at call time
        self.model = nn.Sequential(*layers)
        path = os.path.join(models_dir, self.FILE.format(depth)) if models_dir else None
        if path and os.path.exists(path):
            self.model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
            self.weights = "pretrained:" + path
        elif path and not allow_synthetic:
            raise FileNotFoundError(
                f"{path} not found (the reference fails the same way, vgg.py:147,166).  Pass allow_synthetic=True "
                "(CLI: --synthetic_weights) to run this depth with seeded random weights, or models_dir=None for an "
                "all-synthetic codec")
        else:
            _synthetic_init(self.model, seed + depth)
            self.weights = "synthetic(seed=%d)" % (seed + depth)


class Encoder(_Codec):
    FILE = "vgg_normalised_conv{}_1.pth"

    def __init__(self, depth: int, models_dir=None, allow_synthetic=False, codec_layout=None):
        super().__init__(depth, encoder_layers(depth), models_dir, seed=100, allow_synthetic=allow_synthetic,
                         codec_layout=codec_layout)

    def out_shape(self, h: int, w: int):
        """(C, H', W') of features() for an H x W image, from the layer list alone (no convolution is run): what a rank
        that only RECEIVES style features needs in order to know their shapes in advance (dist.StyleSync)"""
        c = 3
        for m in self.model:
            if isinstance(m, nn.Conv2d):
                c = m.out_channels
                h, w = h - (m.kernel_size[0] - 1), w - (m.kernel_size[1] - 1)
            elif isinstance(m, nn.ReflectionPad2d):
                l, r, t, b = m.padding
                h, w = h + t + b, w + l + r
            elif isinstance(m, nn.MaxPool2d):
                h, w = (h + 1) // 2, (w + 1) // 2  # kernel 2, stride 2, ceil_mode=True (vgg.py:26)
        return c, h, w

    def features(self, x):
        """NCHW image -> NCHW feature (channel-major per image: the layout every OT kernel wants)"""
        if x.is_cuda and not torch.is_grad_enabled():
            return run_fused(self.model, x, self.codec_layout)
        return self.model(x)

    def forward(self, x):
        return to_nhwc(self.model(x))  # the reference's contract: an NHWC view of NCHW memory (vgg.py:153)


class Decoder(_Codec):
    FILE = "feature_invertor_conv{}_1.pth"

    def __init__(self, depth: int, models_dir=None, allow_synthetic=False, codec_layout=None):
        super().__init__(depth, decoder_layers(depth), models_dir, seed=200, allow_synthetic=allow_synthetic,
                         codec_layout=codec_layout)

    def decode(self, feat_nchw):
        if feat_nchw.is_cuda and not torch.is_grad_enabled():
            return run_fused(self.model, feat_nchw, self.codec_layout)
        return self.model(feat_nchw)

    def forward(self, x):
        return self.model(to_nchw(x))
