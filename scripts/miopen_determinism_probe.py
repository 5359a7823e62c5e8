"""MIOpen run-to-run reproducibility of the VGG codec at the bench shapes (why tests/test_gpu_dist.py compares a hooked and an
un-hooked forward call stage by stage instead of end to end): the same encoder call differs from itself by 1e-6 .. 7e-6."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from optimaltextures_amd import dist as otdist
from optimaltextures_amd.driver import OptimalTexture
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

def shard(mode, layers=(3,), size=512, B=8):
    tex = OptimalTexture(size=size, iters=500, passes=5, hist_mode=mode, no_pca=True, layers=layers, independent=True).to(dev).eval()
    tex.rng = otdist.rotation_rng(0, 3)
    style = torch.rand(1, 3, 736, 512, generator=torch.Generator().manual_seed(5)).to(dev)
    past = otdist.texture_noise(24, B, (3, size, size), dev, seed=0)
    rec = []
    with torch.inference_mode():
        out = tex.forward(past, [style], None, on_layer=lambda p, l, img: rec.append(img.clone()) or None)
    return out, rec, tex

for mode in ("cdf", "chol"):
    a, ra, tex = shard(mode)
    b, rb, _ = shard(mode)
    print(mode, "run-to-run max diff", float((a - b).abs().max()), [float((x - y).abs().max()) for x, y in zip(ra, rb)])
enc, dec = tex.encoders[0], tex.decoders[0]
with torch.inference_mode():
    for B in (1, 8):
        for s in (256, 512):
            x = torch.rand(B, 3, s, s, device=dev)
            f1, f2 = enc.features(x), enc.features(x)
            d1, d2 = dec.decode(f1), dec.decode(f1)
            print("codec B", B, "size", s, "enc diff", float((f1 - f2).abs().max()), "dec diff", float((d1 - d2).abs().max()))
    x = torch.rand(8, 3, 512, 512, device=dev)
    r1 = torch.nn.functional.interpolate(x, size=(256, 256), mode="bilinear", align_corners=False)
    r2 = torch.nn.functional.interpolate(x, size=(256, 256), mode="bilinear", align_corners=False)
    print("resize diff", float((r1 - r2).abs().max()))
