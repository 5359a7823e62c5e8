#!/usr/bin/env python3
"""How well-posed is a forward() fixture?  Runs the REFERENCE (build container only, /root/reference) on one case of
tests/golden/gen_forward_golden.py three ways — fp32 as recorded, fp64 end to end, and fp32 with the start image perturbed
by 1e-6 — and prints how far the reference moves away from its own fp32 output.  A HIP-vs-reference tolerance below that
self-sensitivity would test LAPACK round-off, not the implementation.
    python scripts/ref_forward_sensitivity.py sym_pca_54321"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import gen_forward_golden as G  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sym_pca_54321"
    cfg = G.CASES[name]
    optex, util, vgg = G.import_reference()
    from optimaltextures_amd.util import load_styles, maybe_load_content
    torch.set_num_threads(8)

    class RefTexture(optex.OptimalTexture):
        def __init__(self, cfg):
            torch.nn.Module.__init__(self)
            self.hist_mode, self.color_transfer = cfg["hist_mode"], None
            self.content_strength, self.style_scale = cfg["content_strength"], 1
            self.mixing_alpha, self.use_pca = cfg.get("mixing_alpha", 0.5), not cfg["no_pca"]
            self.passes = cfg["passes"]
            self.iters_per_pass_and_layer, self.sizes = util.get_iters_and_sizes(cfg["size"], cfg["iters"], cfg["passes"], True)
            self.encoders = torch.nn.ModuleList([G.ref_codec(vgg, vgg.Encoder, l) for l in cfg["layers"]])
            self.decoders = torch.nn.ModuleList([G.ref_codec(vgg, vgg.Decoder, l) for l in cfg["layers"]])

    styles = load_styles([os.path.join(ROOT, "assets", s) for s in cfg["styles"]], size=cfg["size"], scale=1)
    content = maybe_load_content(os.path.join(ROOT, "assets", cfg["content"]) if cfg["content"] else None, size=cfg["size"])

    def run(dtype, noise):
        torch.manual_seed(cfg["seed"])
        shape = content.shape if content is not None else (cfg.get("batch", 1), 3, cfg["size"], cfg["size"])
        pastiche = torch.rand(shape)
        if noise:
            pastiche = pastiche + noise * torch.randn(shape, generator=torch.Generator().manual_seed(99))
        np.random.seed(cfg["np_seed"])
        tex = RefTexture(cfg).eval().to(dtype)
        torch.manual_seed(cfg["seed"] + 1000)
        with torch.inference_mode():
            out = tex.forward(pastiche.to(dtype), [s.to(dtype) for s in styles], None if content is None else content.to(dtype))
        return out.double().numpy()

    golden = np.load(os.path.join(ROOT, "tests", "golden", "forward.npz"))[f"{name}__out"].astype(np.float64)
    base = run(torch.float32, 0.0)
    print(f"{name}: rerun fp32 vs golden max {np.abs(base - golden).max():.3e}")
    noises = [float(v) for v in sys.argv[2:]] or [1e-6]
    runs = [("fp64", lambda: run(torch.float64, 0.0))] if not sys.argv[2:] else []
    runs += [(f"fp32 + {v:g} input noise", (lambda v=v: run(torch.float32, v))) for v in noises]
    for label, fn in runs:
        out = fn()
        err = np.abs(out - golden)
        print(f"{name}: reference {label} vs its own fp32 output: max {err.max():.3e} (rel {err.max() / np.abs(golden).max():.3e}), "
              f"mean {err.mean():.3e}")


if __name__ == "__main__":
    main()
