#!/usr/bin/env python3
"""Command line of the MI355X build: same flags and defaults as the reference's optex.py:222-244, plus
--layers / --models_dir / --synthetic_weights / --independent / --np_seed (extensions; the reference hard-codes all five
layers, loads weights from ./models and leaves numpy's RNG — which drives the rotations — unseeded).

Paths are taken as given; a relative path that does not exist is looked up under this repository's assets/ directory
(assets/style/graffiti.jpg, assets/models/...: the reference's data files, see assets/PROVENANCE.md), so the reference's
own command lines (`python optex.py -s style/graffiti.jpg`) work unchanged."""
import argparse
import os
from time import time

# --cudnn_benchmark (MIOpen's find mode) times every applicable solver of a convolution once per shape, its naive reference
# kernel included (0.3-1 s per launch at 512^2 batches); that solver never wins and is taken out of the search.  Without the flag
# nothing is searched and this does nothing.  (An environment variable of MIOpen: set before torch loads the library.)
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from optimaltextures_amd import dist as otdist  # noqa: E402
from optimaltextures_amd.driver import OptimalTexture  # noqa: E402
from optimaltextures_amd.util import load_styles, maybe_load_content, save_image  # noqa: E402


def required_length(nmin, nmax):
    class RequiredLength(argparse.Action):
        def __call__(self, parser, args, values, option_string=None):
            if not nmin <= len(values) <= nmax:
                raise argparse.ArgumentTypeError(f'argument "{self.dest}" requires between {nmin} and {nmax} arguments')
            setattr(args, self.dest, values)

    return RequiredLength


ROOT = os.path.dirname(os.path.abspath(__file__))


def resolve(path):
    """the path itself, or its counterpart under assets/ when only that exists"""
    if path is None or os.path.exists(path) or os.path.isabs(path):
        return path
    alt = os.path.join(ROOT, "assets", path)
    return alt if os.path.exists(alt) else path


def build_parser():
    p = argparse.ArgumentParser(description="Optimal-transport texture synthesis on MI355X")
    p.add_argument("-s", "--style", type=str, nargs="+", action=required_length(1, 2), default=["style/graffiti.jpg"])
    p.add_argument("-c", "--content", type=str, default=None)
    p.add_argument("--batch", type=int, default=1)
    p.add_argument("--size", type=int, default=512)
    p.add_argument("--passes", type=int, default=5)
    p.add_argument("--iters", type=int, default=500)
    p.add_argument("--hist_mode", type=str, choices=["sym", "pca", "chol", "cdf", "sort"], default="chol")
    p.add_argument("--color_transfer", type=str, default=None, choices=["lum", "opt"])
    p.add_argument("--content_strength", type=float, default=0.01)
    p.add_argument("--style_scale", type=float, default=1.0)
    p.add_argument("--mixing_alpha", type=float, default=0.5)
    p.add_argument("--no_pca", action="store_true")
    p.add_argument("--no_multires", action="store_true")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--no_tf32", action="store_true", help="accepted for compatibility: gfx950 has no TF32, fp32 is exact")
    p.add_argument("--cudnn_benchmark", action="store_true", help="MIOpen find mode")
    p.add_argument("--compile", action="store_true", help="accepted for compatibility, ignored (no tracing compiler)")
    p.add_argument("--script", action="store_true", help="accepted for compatibility, ignored")
    p.add_argument("--device", type=str, default=None, help="accepted for compatibility (the reference ignores it too)")
    p.add_argument("--memory_format", type=str, default="contiguous", choices=["contiguous", "channels_last"])
    p.add_argument("--output_dir", type=str, default="output/")
    # extensions
    p.add_argument("--layers", type=int, nargs="+", default=[5, 4, 3, 2, 1], help="VGG depths to run (reluN_1)")
    p.add_argument("--models_dir", type=str, default="models", help="directory with the pretrained .pth files")
    p.add_argument("--synthetic_weights", action="store_true",
                   help="depths whose .pth file is missing (relu4_1 / relu5_1 are absent from the reference repository) run "
                        "with seeded random weights instead of raising FileNotFoundError")
    p.add_argument("--independent", action="store_true", help="--batch images are independent textures (not pooled)")
    p.add_argument("--np_seed", type=int, default=None, help="seed numpy's global RNG (drives the rotations)")
    p.add_argument("--pca_fit", type=str, default="gram", choices=["gram", "svd"],
                   help="how fit_pca finds its basis: eigenvectors of the fp64 Gram matrix (default, fast on ROCm) or the "
                        "reference's literal torch.linalg.svd call (optex.py:183) for parity runs")
    p.add_argument("--codec_layout", type=str, default="mixed", choices=["mixed", "nchw"],
                   help="memory layout of the VGG convolutions inside the fused codec path (vgg.py)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.backends.cudnn.benchmark = args.cudnn_benchmark
    memory_format = torch.contiguous_format if args.memory_format == "contiguous" else torch.channels_last
    rank, world, device = otdist.init_distributed()
    if device.type != "cuda":
        raise SystemExit("optex.py needs an MI355X: the HIP path has no CPU fallback")
    # Seeding (optimaltextures_amd/dist.py).  --independent (an extension: the reference has no such flag): every rank seeds
    # torch and numpy identically and texture i's noise comes from its own generator, texture_seed(seed, i) — the job writes
    # the same images on 1, 2, 4 or 8 GPUs (and therefore NOT the reference's single `torch.rand(batch, ...)` draw, not even
    # on one GPU: INTEGRATION.md).  Without --independent (pooled --batch, content image) a multi-rank run is N replicas of
    # the job: rank r seeds with seed + r / np_seed + r, so the ranks write N different variants, not N copies of one
    # image; rank 0 — and any single-process run — is the reference's own draw (optex.py:253-254, 263-265).
    variant = rank if (world > 1 and not args.independent) else 0
    if args.seed is not None:
        torch.manual_seed(args.seed + variant)
    if args.np_seed is not None:
        np.random.seed(args.np_seed + variant)

    if world > 1 and args.independent and args.batch < world:
        raise SystemExit(f"--independent --batch {args.batch} cannot be sharded over {world} ranks: every rank needs at "
                         "least one texture (use --batch >= the number of GPUs)")
    style_files, content_file = [resolve(s) for s in args.style], resolve(args.content)

    with torch.inference_mode():
        styles = load_styles(style_files, size=args.size, scale=args.style_scale, device=device, memory_format=memory_format)
        if len(styles) > 1:
            assert styles[0].shape == styles[1].shape, "Style images must have the same shape"
        content = maybe_load_content(content_file, size=args.size, device=device, memory_format=memory_format)
        lo, hi = otdist.shard_range(args.batch, rank, world) if (world > 1 and args.independent) else (0, args.batch)
        if content is None and args.independent and args.seed is not None:
            pastiche = otdist.texture_noise(lo, hi - lo, (3, args.size, args.size), device, seed=args.seed, on_cpu=True)
            pastiche = pastiche.contiguous(memory_format=memory_format)
        else:  # the reference's draw (optex.py:263-265)
            shape = content.shape if content is not None else (hi - lo, 3, args.size, args.size)
            pastiche = torch.rand(shape).to(device=device, memory_format=memory_format)

        texturizer = OptimalTexture(
            size=args.size, iters=args.iters, passes=args.passes, hist_mode=args.hist_mode,
            color_transfer=args.color_transfer, content_strength=args.content_strength, style_scale=args.style_scale,
            mixing_alpha=args.mixing_alpha, no_pca=args.no_pca, no_multires=args.no_multires, layers=args.layers,
            models_dir=resolve(args.models_dir), independent=args.independent,
            allow_synthetic=args.synthetic_weights, codec_layout=args.codec_layout, pca_fit=args.pca_fit).to(device)
        if rank == 0:
            for enc, dec in zip(texturizer.encoders, texturizer.decoders):
                print(f"relu{enc.depth}_1 weights: encoder {enc.weights} | decoder {dec.weights}")
        if world > 1:
            texturizer.style_sync = otdist.StyleSync(device, spread=True)

        t = time()
        pastiche = texturizer.forward(pastiche, styles, content, verbose=rank == 0)
        torch.cuda.synchronize()
        if texturizer.style_sync is not None:
            texturizer.style_sync.verify(block=True)
        if rank == 0:
            print("Took:", time() - t)
    if world > 1 and args.independent:
        paths = save_image(pastiche, args, first=lo, total=args.batch)   # one directory, files numbered by global texture index
    else:
        if world > 1:  # pooled batch or content image: every rank holds the whole (replicated) job
            args.output_dir = f"{args.output_dir.rstrip('/')}/rank{rank}"
        paths = save_image(pastiche, args)
    print("\n".join(paths))
    return paths


if __name__ == "__main__":
    try:
        main()
    finally:
        otdist.shutdown()   # multi-rank runs: every rank leaves together (no-op for a single process)
