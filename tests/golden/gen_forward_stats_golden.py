#!/usr/bin/env python3
"""End-to-end STATISTICS of the reference's `OptimalTexture.forward` in the headline mode (SURVEY 8c "G-E2E",
hist_mode="cdf") -> tests/golden/forward_stats.npz.

`cdf` chains cannot be compared element-wise: the reference's transfer function is a discontinuous sawtooth
(histmatch.py:77-81), a 1e-6 perturbation decorrelates a chain within ~5 iterations.  What IS
comparable is what the algorithm promises: after the run the relu3_1 feature distribution of the image matches the style's,
channel by channel.  So this fixture runs the reference itself (imported read-only from /root/reference, CPU path, its own
pretrained relu3_1 encoder / decoder, the same shims as gen_forward_golden.py) at 256^2, relu3_1 only, `no_pca`, the default
5 passes / 500 iterations (160 OT iterations at [1, 64, 64, 256]: a one-encoder list reads column [l - 1] = [-1] of the
schedule, optex.py:112), TWICE — with two different numpy seeds, i.e. two different rotation sequences from the same noise —
and records, per run:

  * of the relu3_1 features of the OUTPUT image (re-encoded): per-channel mean, variance and the 1 / 25 / 50 / 75 / 99 %
    quantiles;
  * of the image: mean, std, min, max and per-colour-channel mean / std;

plus the same feature statistics of the style (the target) and of the input noise.  The difference between the two
reference runs is the spread of these statistics under a change of rotations in the REFERENCE ITSELF; the GPU test holds
the HIP driver (run on the first seed) to a small multiple of that spread (tests/test_gpu_configs.py).
"""
import os
import sys
import time

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)
sys.path.insert(0, OUT)

from gen_forward_golden import REF, import_reference, ref_codec, sha  # noqa: E402

QUANTILES = (0.01, 0.25, 0.5, 0.75, 0.99)
CASES = {
    "cdf_nopca_3_lava_256": dict(size=256, iters=500, passes=5, hist_mode="cdf", no_pca=True, layers=(3,),
                                 styles=["style/lava-small.jpg"], seed=8, np_seeds=(108, 208)),
}


def feature_stats(encoder, img):
    """per-channel statistics of the encoder's features of img [1, 3, H, W]: mean [C], var [C], quantiles [5, C]"""
    f = encoder(img)                       # reference Encoder: NHWC
    f = f.reshape(-1, f.shape[-1]).double()
    q = torch.quantile(f, torch.tensor(QUANTILES, dtype=torch.float64), dim=0)
    return f.mean(0).float().numpy(), f.var(0, unbiased=False).float().numpy(), q.float().numpy()


def image_stats(img):
    return np.array([float(img.mean()), float(img.std()), float(img.min()), float(img.max())] +
                    [float(v) for v in img.mean((0, 2, 3))] + [float(v) for v in img.std((0, 2, 3))], dtype=np.float32)


def main():
    optex, util, vgg = import_reference()
    from optimaltextures_amd.util import load_styles
    torch.set_num_threads(8)

    class RefTexture(optex.OptimalTexture):
        def __init__(self, cfg):  # the attribute block of optex.py:31-41 without the weight loading of :42-43
            torch.nn.Module.__init__(self)
            self.hist_mode, self.color_transfer = cfg["hist_mode"], None
            self.content_strength, self.style_scale = 0.1, 1
            self.mixing_alpha, self.use_pca = 0.5, not cfg["no_pca"]
            self.passes = cfg["passes"]
            self.iters_per_pass_and_layer, self.sizes = util.get_iters_and_sizes(cfg["size"], cfg["iters"], cfg["passes"], True)
            self.encoders = torch.nn.ModuleList([ref_codec(vgg, vgg.Encoder, l) for l in cfg["layers"]])
            self.decoders = torch.nn.ModuleList([ref_codec(vgg, vgg.Decoder, l) for l in cfg["layers"]])

    out = {}
    for name, cfg in CASES.items():
        styles = load_styles([os.path.join(ROOT, "assets", s) for s in cfg["styles"]], size=cfg["size"], scale=1)
        ref_styles = util.load_styles([os.path.join(REF, s) for s in cfg["styles"]], size=cfg["size"], scale=1)
        assert all(torch.equal(a, b) for a, b in zip(styles, ref_styles)), "load_styles differs from the reference"
        torch.manual_seed(cfg["seed"])
        pastiche = torch.rand(1, 3, cfg["size"], cfg["size"])
        tex = RefTexture(cfg).eval()
        enc = tex.encoders[0]
        out[f"{name}__sha_pastiche"] = sha(pastiche)
        out[f"{name}__sha_style0"] = sha(styles[0])
        out[f"{name}__sizes"] = np.array(tex.sizes)
        out[f"{name}__iters"] = np.array([tex.iters_per_pass_and_layer[p][-1] for p in range(cfg["passes"])])
        with torch.inference_mode():
            for tag, img in (("style", styles[0]), ("noise", pastiche)):
                m, v, q = feature_stats(enc, img)
                out[f"{name}__{tag}_feat_mean"], out[f"{name}__{tag}_feat_var"], out[f"{name}__{tag}_feat_q"] = m, v, q
            for r, np_seed in enumerate(cfg["np_seeds"]):
                t0 = time.time()
                np.random.seed(np_seed)  # the rotations (optex.py:149) — the only randomness inside forward() here
                result = tex.forward(pastiche.clone(), [s.clone() for s in styles], None)
                m, v, q = feature_stats(enc, result)
                out[f"{name}__run{r}_feat_mean"], out[f"{name}__run{r}_feat_var"], out[f"{name}__run{r}_feat_q"] = m, v, q
                out[f"{name}__run{r}_image"] = image_stats(result)
                out[f"{name}__run{r}_np_seed"] = np.array(np_seed)
                print(f"{name} run {r} (np seed {np_seed}): {time.time() - t0:.1f} s, image mean {float(result.mean()):.4f} "
                      f"std {float(result.std()):.4f}; mean |feature mean - style's| "
                      f"{np.abs(m - out[f'{name}__style_feat_mean']).mean():.4f} (noise: "
                      f"{np.abs(out[f'{name}__noise_feat_mean'] - out[f'{name}__style_feat_mean']).mean():.4f})")
        a, b = out[f"{name}__run0_feat_mean"], out[f"{name}__run1_feat_mean"]
        print(f"{name}: spread between the two reference runs, per-channel feature mean: mean {np.abs(a - b).mean():.4e} "
              f"max {np.abs(a - b).max():.4e}")
    path = os.path.join(OUT, "forward_stats.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
