"""GPU diagnostic: why do the PCA forward fixtures differ?  Runs the two PCA cases with (a) the driver's GPU SVD and
(b) the SVD taken on the CPU exactly like the reference (torch.svd of the same features), printing k per (pass, layer)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_configs import FORWARD_CASES, MODELS, _asset
from optimaltextures_amd import driver
from optimaltextures_amd.driver import OptimalTexture, project_cm
from optimaltextures_amd.util import load_styles, maybe_load_content

g = np.load(os.path.join(ROOT, "tests/golden/forward.npz"))
dev = torch.device("cuda:0")
orig = driver.fit_pca_cm
log = []

def cpu_fit(style_cm):
    b, c, n = style_cm.shape
    t = style_cm.permute(0, 2, 1).reshape(-1, c).cpu()
    A = t - t.mean()
    _, ev, V = torch.svd(A)
    share = torch.cumsum(ev / torch.sum(ev), dim=0)
    k = int((share > 0.9).max(0).indices)
    e = V[:, :k].contiguous().to(style_cm.device)
    log.append(("cpu", c, k, share[k - 1:k + 1].tolist()))
    return project_cm(style_cm, e), e

def gpu_fit(style_cm):
    out = orig(style_cm)
    b, c, n = style_cm.shape
    a = style_cm.permute(0, 2, 1).reshape(-1, c) - style_cm.mean()
    sing = torch.linalg.svdvals(a)
    share = torch.cumsum(sing / sing.sum(), 0)
    k = out[1].shape[1]
    # subspace vs the CPU basis
    t = style_cm.permute(0, 2, 1).reshape(-1, c).cpu(); A = t - t.mean(); _, ev, V = torch.svd(A)
    kc = int((torch.cumsum(ev / ev.sum(), 0) > 0.9).max(0).indices)
    E, Ec = out[1].cpu().double(), V[:, :kc].double()
    ang = float((E @ E.T - Ec @ Ec.T).abs().max()) if k == kc else float("nan")
    log.append(("gpu", c, k, share[k - 1:k + 1].tolist(), "cpu k", kc, "projector diff", ang,
                "sv rel diff", float(((sing.cpu() - ev) / ev).abs().max())))
    return out

for name in ("pca_pca_321_content", "sym_pca_54321"):
    cfg = FORWARD_CASES[name]
    styles = load_styles([_asset(s) for s in cfg["styles"]], size=cfg["size"], scale=1)
    content = maybe_load_content(_asset(cfg["content"]) if cfg["content"] else None, size=cfg["size"])
    for label, fit in (("gpu-svd", gpu_fit), ("cpu-svd", cpu_fit)):
        driver.fit_pca_cm = fit
        log.clear()
        torch.manual_seed(cfg["seed"])
        pastiche = torch.rand(content.shape if content is not None else (1, 3, cfg["size"], cfg["size"]))
        tex = OptimalTexture(size=cfg["size"], iters=cfg["iters"], passes=cfg["passes"], hist_mode=cfg["hist_mode"],
                             content_strength=cfg["content_strength"], no_pca=False, layers=cfg["layers"], models_dir=MODELS,
                             allow_synthetic=True, index_by_position=True).to(dev).eval()
        np.random.seed(cfg["np_seed"])
        with torch.inference_mode():
            out = tex.forward(pastiche.to(dev), [s.to(dev) for s in styles], None if content is None else content.to(dev))
        err = np.abs(out.cpu().numpy() - g[f"{name}__out"])
        print(f"{name} [{label}]: max err {err.max():.3e} mean {err.mean():.3e}")
        for l in log:
            print("   ", l)
