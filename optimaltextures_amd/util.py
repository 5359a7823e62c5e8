"""Host-side plumbing around the hot path: the iteration/size schedule that defines "default iters"
(reference util.py:68-86), size helpers (util.py:33-42,93-94) and PIL-only image IO (util.py:13-30,45-65; torchvision and
kornia are not available on the target image)."""
import os
from typing import Tuple

import numpy as np
import torch
from PIL import Image
from torch import Tensor
from torch.nn.functional import interpolate

LAYER_WIDTHS = (64, 128, 256, 512, 512)  # relu1_1 .. relu5_1


def round32(integer: int) -> int:
    """round UP to a multiple of 32 (util.py:93-94)"""
    return (int(integer) + 31) // 32 * 32


def get_size(size: int, scale: float, h: int, w: int, oversize: bool = False) -> Tuple[int, int]:
    """util.py:33-42.  NB the reference's load_image passes PIL's (width, height) as (h, w)."""
    target = size * scale
    other = int(float(w) * (target / float(h)))
    first = size
    if oversize:
        first = min(int(target), h)
        other = min(other, w)
    return round32(first), round32(other)


def get_iters_and_sizes(size: int, iters: int, passes: int, use_multires: bool):
    """util.py:68-86: per-pass budgets ~ arange(2P, P, -1), per-layer share ~ widths + 64, truncated to int32; sizes are
    32 * round(linspace(256, size, P) / 32) (numpy round-half-even).  Returned as nested lists [pass][column]; the
    driver reads column [l - 1] for encoder index l (optex.py:112), which rotates the table by one.
    use_multires=False crashes in the reference (`.tolist()` on a list, util.py:80,86); here it returns the evident
    intent: equal budgets at full size."""
    if use_multires:
        weights = np.arange(2 * passes, passes, -1)
        per_pass = weights / np.sum(weights) * iters
        sizes = (32 * np.round(np.linspace(256, size, passes) / 32)).astype(np.int32).tolist()
    else:
        per_pass = np.ones(passes) * int(iters / passes)
        sizes = [int(size)] * passes
    share = np.array(LAYER_WIDTHS) + 64
    share = share / np.sum(share)
    table = (per_pass[:, None] * share[None, :]).astype(np.int32)
    return table.tolist(), sizes


def layer_iters(table, p: int, enc_index: int) -> int:
    """iterations the reference runs in pass p for encoder list index l (0 = relu5_1 .. 4 = relu1_1): table[p][l - 1]"""
    return int(table[p][enc_index - 1])


def to_nchw(x: Tensor) -> Tensor:
    return x.permute(0, 3, 1, 2)


def to_nhwc(x: Tensor) -> Tensor:
    return x.permute(0, 2, 3, 1)


def resize(x: Tensor, size: Tuple[int, int]) -> Tensor:
    return interpolate(x, size=size, mode="bicubic", align_corners=False, antialias=True)


def name(filepath: str) -> str:
    return os.path.basename(filepath).split(".")[0]


def _to_tensor(img: Image.Image) -> Tensor:
    arr = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)


def load_image(path, size, scale=1, oversize=True, device="cpu", memory_format=torch.contiguous_format) -> Tensor:
    img = Image.open(path).convert(mode="RGB")
    img = img.resize(get_size(size, scale, img.size[0], img.size[1], oversize), Image.LANCZOS)
    return _to_tensor(img).unsqueeze(0).to(device, memory_format=memory_format)


def load_styles(style_files, size, scale, oversize=False, device="cpu", memory_format=torch.contiguous_format):
    # the reference inverts the flag here (util.py:16): styles are never upsampled at load
    return [load_image(f, size, scale, not oversize, device=device, memory_format=memory_format) for f in style_files]


def maybe_load_content(content_file, size, device="cpu", memory_format=torch.contiguous_format):
    if content_file is None:
        return None
    return load_image(content_file, size, oversize=False, device=device, memory_format=memory_format)


def output_name(args) -> str:
    """file stem built from the flags like util.py:46-61"""
    parts = [name(s) for s in args.style]
    if len(args.style) > 1:
        parts.append("blend" + str(args.mixing_alpha))
    if args.content is not None:
        parts += [name(args.content), "strength" + str(args.content_strength)]
    parts.append(args.hist_mode + "hist")
    if args.no_pca:
        parts.append("no_pca")
    if args.no_multires:
        parts.append("no_multires")
    if args.style_scale != 1:
        parts.append("scale" + str(args.style_scale))
    if args.color_transfer is not None:
        parts.append(args.color_transfer)
    parts.append(str(args.size))
    return "_".join(parts)


def save_image(output: Tensor, args, first: int = 0, total: int = None) -> list:
    """first / total (extension): `output` holds textures first .. first + len(output) - 1 of a job of `total` (a rank's
    shard): files are numbered by the texture's GLOBAL index, so a sharded job writes the same file names as one process"""
    os.makedirs(args.output_dir, exist_ok=True)
    stem = output_name(args)
    paths = []
    many = (total if total is not None else len(output)) > 1
    for o, out in enumerate(output):
        arr = out.detach().clamp(0, 1).mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
        path = os.path.join(args.output_dir, stem + (f"_{first + o + 1}" if many else "") + ".png")
        Image.fromarray(arr).save(path)
        paths.append(path)
    return paths
