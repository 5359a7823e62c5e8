"""The RCCL call sequence of the multi-GPU path on ONE GPU (pytest -m gpu): a one-rank "nccl" process group in a child
process runs what bench.py --gpus N runs per rank — the packed style broadcast (int64 header + asynchronous fp32
payload on the communicator's stream), the barrier, the max-over-ranks and per-rank gathers of the timing — and a whole
OptimalTexture.forward with the broadcast hook against one without.  The world-size-2 semantics are covered on CPU over
gloo (tests/test_dist.py); 8-GPU runs are the driver's."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from optimaltextures_amd import dist as otdist
from optimaltextures_amd.driver import OptimalTexture
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sync = otdist.StyleSync(dev, always=True)
g = torch.Generator(device=dev).manual_seed(1)
payload = [torch.rand(1, 11, 40, device=dev, generator=g), torch.rand(32, 11, device=dev, generator=g), torch.empty(0, 0, device=dev)]
got, ints = sync.broadcast_packed(payload, [8, 12, 5], counts=(3, 3))
assert ints == [8, 12, 5] and len(got) == 3
for a, b in zip(got, payload):
    assert a.shape == b.shape and bool((a == b).all())
assert sync.messages == 2 and sync.bytes_moved > 0        # counts known on every rank: header + payload
got, ints = sync.broadcast_packed(payload, [8, 12, 5])
assert ints == [8, 12, 5] and len(got) == 3 and sync.messages == 5   # + the announcing message
# more tensors than any fixed header would hold (ADVICE r2: --passes >= 17 with five layers), still two messages
many = [torch.rand(3, 7, device=dev, generator=g) for _ in range(2 * 20 * 5)]
got, ints = sync.broadcast_packed(many, list(range(1 + 20 * 11)), counts=(200, 221))
assert len(got) == 200 and ints[-1] == 220 and all(bool((a == b).all()) for a, b in zip(got, many))
try:
    sync.broadcast_packed(payload, [1], counts=(2, 1))        # wrong announcement: EVERY rank raises, after the exchange
    raise SystemExit("expected ValueError")
except ValueError:
    pass
otdist.barrier()
assert otdist.all_reduce_max(3.5, dev) == 3.5
assert otdist.all_gather_floats(1.25, dev) == [1.25]
# a forward call with the hook (style side prefetched and broadcast once) equals one without, bit for bit
def run(hook):
    torch.manual_seed(3)
    import numpy as np
    np.random.seed(3)
    tex = OptimalTexture(size=128, iters=20, passes=2, hist_mode="cdf", no_pca=False, layers=(2, 1), models_dir=None,
                         allow_synthetic=True).to(dev)
    if hook:
        tex.style_sync = otdist.StyleSync(dev, always=True)
    style = torch.rand(1, 3, 96, 128, generator=torch.Generator().manual_seed(5)).to(dev)
    past = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(6)).to(dev)
    return tex.forward(past, [style], None)
a, b = run(False), run(True)
assert bool((a == b).all()), float((a - b).abs().max())
# BASELINE config 4's per-GPU shard: 8 independent 512^2 textures, relu3_1, C = 256 (no_pca), the full 5-pass schedule
# (52 OT iterations), the style side arriving through the RCCL broadcast hook (VERDICT r2 item 1b).
# MIOpen's fp32 convolutions are not reproducible run to run at these shapes (scripts/miopen_determinism_probe.py: the same encoder call differs
# by 1e-6 .. 7e-6 from itself, and `cdf` amplifies one ulp to a whole bin within five iterations, SURVEY 0), so two whole
# forward calls cannot be compared bit for bit.  What the hook must not change is checked exactly where it is exact:
#   (1) the style side the hook delivers == the style side computed locally, bit for bit, for all five passes, when both
#       come from the SAME encoder outputs (the hook only packs, broadcasts and re-slices them);
#   (2) the OT loop of every pass, fed the recorded inputs of the hooked run, reproduces the hooked run's loop output bit
#       for bit (the kernels are deterministic; only the convolutions around them are not);
#   (3) the un-hooked forward agrees with the hooked one as closely as two un-hooked runs agree with each other (chol:
#       round-off; cdf: image statistics).
def make(mode):
    tex = OptimalTexture(size=512, iters=500, passes=5, hist_mode=mode, no_pca=True, layers=(3,), independent=True).to(dev).eval()
    tex.rng = otdist.rotation_rng(0, 3)
    return tex
style = torch.rand(1, 3, 736, 512, generator=torch.Generator().manual_seed(5)).to(dev)
def noise():
    return otdist.texture_noise(24, 8, (3, 512, 512), dev, seed=0)
with torch.inference_mode():
    # (1) pack -> broadcast -> slice is the identity on the style sides
    tex = make("cdf")
    local = [(False,) + tex._compute_style_side(tex._style_tensors([style], sz, True)) for sz in tex.sizes]
    tex.style_sync = otdist.StyleSync(dev, always=True)
    got = tex._sync_style_sides(local, tex.passes)
    assert tex.style_sync.messages == 2                      # one packed exchange for all five passes
    for (r0, f0, e0, h0), (r1, f1, e1, h1) in zip(local, got):
        assert r0 == r1 and h0 == h1 and all(bool((x == y).all()) for x, y in zip(f0, f1))
        assert all(x.shape == y.shape for x, y in zip(e0, e1))
    # (2) + (3)
    from optimaltextures_amd import driver as drv
    for mode in ("cdf", "chol"):
        calls = []
        real = drv.ot_iterations
        def spy(x, style_f, hist_mode, iters, **kw):
            state = kw["rng"].get_state()
            xin = x.clone()
            out = real(x, style_f, hist_mode, iters, **kw)
            calls.append((xin, style_f.clone(), iters, state, out.clone()))
            return out
        drv.ot_iterations = spy
        tex = make(mode)
        tex.style_sync = otdist.StyleSync(dev, always=True)
        hooked = tex.forward(noise(), [style], None)
        drv.ot_iterations = real
        assert tex.style_sync.messages == 5 and len(calls) == 5   # no PCA: shapes known everywhere, one payload per pass (its source = pass mod world)
        assert hooked.shape == (8, 3, 512, 512) and bool(torch.isfinite(hooked).all())
        assert not bool((hooked[0] == hooked[1]).all())      # independent textures
        import numpy as np
        for xin, sf, iters, state, want in calls:            # the loop replayed on the recorded inputs
            rng = np.random.RandomState()
            rng.set_state(state)
            again = real(xin.clone(), sf, mode, iters, pooled=False, rng=rng)
            assert bool((again == want).all()), (mode, iters)
        plain = make(mode).forward(noise(), [style], None)
        plain2 = make(mode).forward(noise(), [style], None)
        if mode == "chol":
            noise_floor = float((plain - plain2).abs().max())
            assert float((hooked - plain).abs().max()) <= max(10 * noise_floor, 1e-3), (float((hooked - plain).abs().max()), noise_floor)
        else:
            for im in (plain, plain2):
                assert abs(float(hooked.mean() - im.mean())) < 5e-3 and abs(float(hooked.std() - im.std())) < 5e-3
torch.cuda.synchronize()
dist.destroy_process_group()
print("ok")
"""


def test_rccl_call_sequence_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
