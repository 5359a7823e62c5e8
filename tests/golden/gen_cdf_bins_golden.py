"""Golden vectors for histmatch.cdf_match(target, source, bins) with bins != 256 (histmatch.py:49-69), made by importing the
reference in the build container (CPU, torch's ATen histc / linspace / cumsum / searchsorted):

    python tests/golden/gen_cdf_bins_golden.py        ->  tests/golden/cdf_match_bins.npz

No caller inside the reference passes another bin count; these vectors pin the generalised oracle (orc_cdf_match_bins) and the
HIP entry optex_cdf_match_bins for a direct call.  Needs /root/reference, so it runs in the build container only.
"""
import os

import numpy as np
import torch

from gen_golden import import_reference, relu_features, t

OUT = os.path.dirname(os.path.abspath(__file__))
BINS = (1, 2, 3, 16, 100, 255, 257, 1000, 2048, 2049, 5000)


def main():
    _, histmatch, _ = import_reference()
    torch.set_num_threads(1)
    rng = np.random.default_rng(4900)
    C, nt, ns = 6, 700, 520
    tgt = relu_features(rng, 1, 28, 25, C, 2.0, 0.3)[0].reshape(-1, C).T.copy()  # ties at zero
    src = relu_features(rng, 1, 26, 20, C, 1.5, -0.2)[0].reshape(-1, C).T.copy()
    tgt[0] = rng.standard_normal(nt).astype(np.float32)  # dense, rotated-like
    src[0] = (rng.standard_normal(ns) * 2 + 1).astype(np.float32)
    tgt[1] = (rng.standard_normal(nt) * 0.01 + 5).astype(np.float32)  # disjoint ranges
    src[1] = (rng.standard_normal(ns) * 3 - 5).astype(np.float32)
    tgt[4] = 2.0  # constant target channel
    src[5] = 0.5  # constant source channel
    g = dict(target=tgt, source=src, bins=np.array(BINS))
    for b in BINS:
        g[f"out_{b}"] = histmatch.cdf_match(t(tgt), t(src), b).numpy()
    # both constant and equal (lo == hi: histc widens the range, linspace does not)
    g["deg_both_out_7"] = histmatch.cdf_match(torch.full((1, 64), 3.0), torch.full((1, 80), 3.0), 7).numpy()
    np.savez_compressed(os.path.join(OUT, "cdf_match_bins.npz"), **g)
    print("wrote cdf_match_bins.npz", {k: v.shape for k, v in g.items() if k.startswith("out_")})


if __name__ == "__main__":
    main()
