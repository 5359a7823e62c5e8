"""Multi-process checks of the N > 1 path on CPU (gloo, world_size 2): the texture sharding and the one collective the
hot path has — the per-(pass, layer) broadcast of style-side data (optimaltextures_amd/dist.py).  On the GPU
box the same code runs over RCCL ("nccl" backend); nothing here needs a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from optimaltextures_amd import dist as otdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    r, w, device = otdist.init_distributed("gloo")
    assert (r, w) == (rank, world) and device.type == "cpu"
    try:
        ret[rank] = fn(rank, world, device)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def run_world(fn, world=2):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
        return dict(ret)


# ------------------------------------------------------------------------------------------------ sharding (no processes)
@pytest.mark.parametrize("total,world", [(64, 8), (64, 1), (7, 2), (3, 8), (0, 4), (65, 8)])
def test_shard_range_partitions_textures(total, world):
    ranges = [otdist.shard_range(total, r, world) for r in range(world)]
    covered = [i for lo, hi in ranges for i in range(lo, hi)]
    assert covered == list(range(total))                       # disjoint, ordered, complete
    sizes = [hi - lo for lo, hi in ranges]
    assert max(sizes) - min(sizes) <= 1                        # balanced
    if total == 64 and world == 8:
        assert sizes == [8] * 8                                # BASELINE config 4: 8 textures per GPU


# ------------------------------------------------------------------------------------------------ broadcast of style data
def _style_sync_job(rank, world, device):
    sync = otdist.StyleSync(device)
    g = torch.Generator().manual_seed(123)
    out = []
    # two (pass, layer) rounds with different, data-dependent shapes (the PCA rank k changes per pass)
    for (c, k, ns) in [(16, 5, 96), (32, 11, 40)]:
        if sync.is_source:
            payload = [torch.rand(1, k, ns, generator=g), torch.rand(c, k, generator=g), torch.tensor([8.0, 12.0])]
        else:
            payload = None
        got = sync(payload)
        out.append([t.numpy().copy() for t in got])
    # no-PCA round: an empty eigvec tensor must survive the trip
    payload = [torch.rand(1, 8, 24, generator=g), torch.empty(0, 0), torch.tensor([4.0, 6.0])] if sync.is_source else None
    got = sync(payload)
    out.append([t.numpy().copy() for t in got])
    return out, sync.bytes_moved, otdist.all_reduce_max(float(rank + 1), device)


def _style_sync_counts_job(rank, world, device):
    """the header has no fixed capacity (2 * passes * layers tensors overflowed a 160-tensor header on the source
    rank only and left the others waiting): 400 tensors in one exchange, counts known on every rank -> 2 messages;
    a source whose lists do not match the announced counts makes EVERY rank raise, after the exchange"""
    sync = otdist.StyleSync(device)
    g = torch.Generator().manual_seed(7)
    many = [torch.rand(2, 3, generator=g) for _ in range(400)] if sync.is_source else None
    ints = list(range(901)) if sync.is_source else None
    got, gi = sync.broadcast_packed(many, ints, counts=(400, 901))
    msgs = sync.messages
    raised = False
    try:
        sync.broadcast_packed([torch.zeros(1)] if sync.is_source else None, [1] if sync.is_source else None, counts=(2, 1))
    except ValueError:
        raised = True
    # still in step: a third exchange works on both ranks
    again, _ = sync.broadcast_packed([torch.full((4,), 2.5)] if sync.is_source else None, [] if sync.is_source else None)
    return float(sum(t.sum() for t in got)), len(got), gi[-1], msgs, raised, again[0].tolist()


def test_style_sync_unbounded_header_and_symmetric_errors_gloo_world2():
    res = run_world(_style_sync_counts_job, 2)
    assert res[0] == res[1]
    total, n, last, msgs, raised, again = res[0]
    assert n == 400 and last == 900 and msgs == 2 and raised and again == [2.5] * 4


def test_style_sync_broadcast_gloo_world2():
    res = run_world(_style_sync_job, 2)
    (a, bytes_a, max_a), (b, bytes_b, max_b) = res[0], res[1]
    assert max_a == max_b == 2.0                               # bench.py's max-over-ranks timing reduction
    assert bytes_a == bytes_b > 0
    for ra, rb in zip(a, b):
        assert len(ra) == len(rb) == 3
        for ta, tb in zip(ra, rb):
            assert ta.shape == tb.shape and np.array_equal(ta, tb)
    assert a[2][1].shape == (0, 0)
    assert a[0][0].shape == (1, 5, 96) and a[1][1].shape == (32, 11)


# ------------------------------------------------------------------------------------------------ sharded job == single job
def _sharded_textures_job(rank, world, device):
    """Each rank synthesises its shard of `total` independent textures with the CPU oracle standing in for the HIP
    kernels (this is a test of the SHARDING logic: seeds, ranges, broadcast), and returns them."""
    from oracle import oracle as orc
    total, C, n, ns, iters = 5, 8, 64, 48, 2
    sync = otdist.StyleSync(device)
    style = None
    if sync.is_source:
        style = [torch.from_numpy(np.maximum(np.random.default_rng(9).standard_normal((1, C, ns)), 0).astype(np.float32))]
    style = sync(style)[0].numpy()[0]
    lo, hi = otdist.shard_range(total, rank, world)
    outs = {}
    for i in range(lo, hi):
        x = np.maximum(np.random.default_rng(100 + i).standard_normal((C, n)), 0).astype(np.float32)
        rng = orc.LegacyRNG(1000 + i)                            # per-texture rotation stream: rank-independent
        for _ in range(iters):
            R = orc.random_rotation(C, rng).astype(np.float32)
            x = orc.unrotate_cm(orc.cdf_match(orc.rotate_cm(x, R), orc.rotate_cm(style, R)), R)
        outs[i] = x
    return outs


def test_sharded_textures_equal_single_process():
    two = run_world(_sharded_textures_job, 2)
    one = run_world(_sharded_textures_job, 1)
    merged = {**two[0], **two[1]}
    assert sorted(merged) == sorted(one[0]) == list(range(5))
    assert sorted(two[0]) == [0, 1, 2] and sorted(two[1]) == [3, 4]
    for i in range(5):
        assert np.array_equal(merged[i], one[0][i])              # sharding changes nothing, bit for bit


# ------------------------------------------------------------------------------------------------ style prefetch of the driver
def _prefetch_job(rank, world, device):
    """driver.OptimalTexture.prefetch_style_sides over gloo: rank 0 encodes the style for every pass (torch-CPU VGG,
    no_pca so that no HIP kernel is involved), rank 1 only receives."""
    from optimaltextures_amd.driver import OptimalTexture
    torch.manual_seed(0)
    tex = OptimalTexture(size=288, iters=10, passes=2, hist_mode="cdf", no_pca=True, layers=(1,)).eval()
    tex.style_sync = otdist.StyleSync(device)
    g = torch.Generator().manual_seed(5)
    style = torch.rand(1, 3, 96, 128, generator=g) if tex.style_sync.is_source else torch.zeros(1, 3, 96, 128)
    with torch.inference_mode():
        sides = tex.prefetch_style_sides((288, 288), [style], None)
    return [(bool(r), [f.numpy().copy() for f in sf], [tuple(e.shape) for e in eig], list(hw)) for r, sf, eig, hw in sides]


def test_driver_style_prefetch_gloo_world2():
    res = run_world(_prefetch_job, 2)
    a, b = res[0], res[1]
    assert len(a) == len(b) == 2                                   # one entry per pass
    assert [x[0] for x in a] == [x[0] for x in b] == [True, True]   # 288 -> 256 for pass 0, 256 -> 288 for pass 1
    for (ra, fa, ea, ha), (rb, fb, eb, hb) in zip(a, b):
        assert ha == hb and ea == eb == [(0, 0)]
        for x, y in zip(fa, fb):
            assert x.shape == y.shape and x.shape[1] == 64 and np.array_equal(x, y)   # rank 1 holds rank 0's features
    assert a[0][1][0].shape[2] == a[0][3][0][0] * a[0][3][0][1]      # [1, C, Hs * Ws] and its (Hs, Ws)


# ------------------------------------------------------------------------------------------------ bench.py's step() over gloo
def _oracle_backed_ops():
    """Stand-ins for the two GPU entry points the driver's forward() needs, built from the CPU oracle, so that the
    ORCHESTRATION (style prefetch + packed broadcast + per-rank textures) can run under gloo without a GPU.  Test
    infrastructure only: the product path never routes through the oracle."""
    from oracle import oracle as orc
    from optimaltextures_amd import ops, rotation

    def rotations(N, count, device, rng=None, want64=False):
        normals = rotation.draw_normals(N, count, rng)
        R = np.stack([orc.random_rotation_from_normals(normals[i], N) for i in range(count)]).astype(np.float32)
        return torch.from_numpy(R), torch.from_numpy(np.ascontiguousarray(R.transpose(0, 2, 1)))

    def ot_loop(mode, x, style, R32, Rt32, content=None, strength=0.0, fuse_rotations=False):
        xs, st = x.numpy(), style.numpy()
        for s in range(xs.shape[0]):
            w = xs[s]
            for R in R32.numpy():
                w = orc.unrotate_cm(orc.hist_match_cm(orc.rotate_cm(w, R), 1, orc.rotate_cm(st[s if st.shape[0] > 1 else 0], R), 1,
                                                      mode), R)
            xs[s] = w
        return x

    rotation.rotations = rotations
    ops.ot_loop = ot_loop


def _bench_step_job(rank, world, device, spread=False):
    """what bench.py's step() does on every rank: B independent textures, relu-layer encode -> OT iterations -> decode over
    the multi-resolution passes, the style side arriving through StyleSync — from rank 0 alone (spread=False: ranks != 0 hold a
    blank style) or pass p from rank p mod world (spread=True, the default of bench.py and the CLI)"""
    from optimaltextures_amd.driver import OptimalTexture
    _oracle_backed_ops()
    tex = OptimalTexture(size=288, iters=20, passes=2, hist_mode="cdf", no_pca=True, layers=(1,), independent=True).eval()
    if world > 1:
        tex.style_sync = otdist.StyleSync(device, spread=spread)
    g = torch.Generator().manual_seed(77)
    style = torch.rand(1, 3, 64, 96, generator=g)
    if world > 1 and rank != 0 and not spread:
        style = torch.zeros_like(style)          # spread=False: only the source rank's style may matter
    # bench.py's seeding rule: this rank's step 0 is rotation group `rank` = textures 2 * rank, 2 * rank + 1
    tex.rng = otdist.rotation_rng(0, rank)
    pastiche = otdist.texture_noise(2 * rank, 2, (3, 288, 288), device, seed=0)
    with torch.inference_mode():
        out = tex.forward(pastiche, [style])
    sync = tex.style_sync
    return out.numpy().copy(), (sync.messages, sync.bytes_moved) if sync is not None else (0, 0)


def test_bench_step_with_style_sync_gloo_world2():
    """Both ranks run the whole forward() with the packed style broadcast; rank 1 (blank local style) must produce exactly
    what a single process computes for rank 1's seeds with the real style: the broadcast delivered rank 0's style side for
    every pass, and nothing else crossed ranks.  Without PCA every shape is known in advance: one header-free payload per
    pass, no host synchronisation — all from rank 0, or (spread, the default) pass p from rank p mod 2."""
    ref = run_world(_single_rank1_job, 1)[0]
    for job in (_bench_step_job, _bench_step_spread_job):
        res = run_world(job, 2)
        (out0, (msgs0, bytes0)), (out1, (msgs1, bytes1)) = res[0], res[1]
        assert msgs0 == msgs1 == 2 and bytes0 == bytes1 > 0    # one payload per pass (two passes), nothing else
        assert out0.shape == out1.shape == (2, 3, 288, 288) and np.isfinite(out0).all() and np.isfinite(out1).all()
        assert not np.array_equal(out0, out1)                      # different seeds per rank: different textures
        assert np.array_equal(out1, ref)


def _bench_step_spread_job(rank, world, device):
    return _bench_step_job(rank, world, device, spread=True)


def _single_rank1_job(rank, world, device):
    """rank 1's work in a world of one: real style, seeds of rank 1"""
    from optimaltextures_amd.driver import OptimalTexture
    _oracle_backed_ops()
    tex = OptimalTexture(size=288, iters=20, passes=2, hist_mode="cdf", no_pca=True, layers=(1,), independent=True).eval()
    tex.rng = otdist.rotation_rng(0, 1)
    style = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(77))
    pastiche = otdist.texture_noise(2, 2, (3, 288, 288), device, seed=0)
    with torch.inference_mode():
        return tex.forward(pastiche, [style]).numpy().copy()


# ------------------------------------------------------------------------------------------------ one seeding rule (dist.py)
def _seeded_job(rank, world, device):
    """a job of 4 rotation groups x 2 textures, sharded over `world` ranks by bench.py's rule (group q of step k on rank r:
    q = k * world + r): returns {global texture index: image}"""
    from optimaltextures_amd.driver import OptimalTexture
    _oracle_backed_ops()
    tex = OptimalTexture(size=288, iters=12, passes=2, hist_mode="cdf", no_pca=True, layers=(1,), independent=True).eval()
    if world > 1:
        tex.style_sync = otdist.StyleSync(device)
    style = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(77))
    B, groups, out = 2, 4, {}
    with torch.inference_mode():
        for k in range(groups // world):
            q = k * world + rank
            tex.rng = otdist.rotation_rng(5, q)
            img = tex.forward(otdist.texture_noise(q * B, B, (3, 288, 288), device, seed=5), [style]).numpy()
            for j in range(B):
                out[q * B + j] = img[j].copy()
    return out


def test_texture_i_is_the_same_image_whatever_the_world_size():
    one = run_world(_seeded_job, 1)[0]
    two = run_world(_seeded_job, 2)
    merged = {**two[0], **two[1]}
    assert sorted(one) == sorted(merged) == list(range(8))
    assert sorted(two[0]) == [0, 1, 4, 5] and sorted(two[1]) == [2, 3, 6, 7]
    for i in range(8):
        assert np.array_equal(one[i], merged[i]), i
    assert not np.array_equal(one[0], one[1]) and not np.array_equal(one[0], one[2])


def test_seed_helpers_are_pure_functions_of_seed_and_index():
    a = otdist.texture_noise(3, 2, (3, 8, 8), "cpu", seed=1)
    b = otdist.texture_noise(4, 1, (3, 8, 8), "cpu", seed=1)
    assert torch.equal(a[1], b[0]) and not torch.equal(a[0], a[1])
    assert not torch.equal(otdist.texture_noise(4, 1, (3, 8, 8), "cpu", seed=2)[0], b[0])
    assert otdist.rotation_seed(0, 3) != otdist.rotation_seed(0, 4) and 0 <= otdist.rotation_seed(10 ** 6, 10 ** 7) < 2 ** 32
    r1, r2 = otdist.rotation_rng(0, 3), otdist.rotation_rng(0, 3)
    assert r1.standard_normal() == r2.standard_normal()


# ------------------------------------------------------------------------------------------------ bench.py --gpus N launches itself
def test_bench_gpus2_spawns_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) must start two ranks itself and report
    n_gpus = 2 (it used to benchmark ONE GPU with a warning).  --dry_run: the launch path only, gloo on CPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "8", "--dry_run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                              # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["steps"] == 3 and d["warmup"] == 1
    assert d["textures_total"] == 8 * 2 * 3
    assert d["first_timed_texture_by_rank"] == [16, 24]           # step 1 (after one warm-up step): groups 2 and 3


def test_bench_total_is_strong_scaling_dry_run():
    """`--total 64 --gpus 2` = BASELINE config 4's partitioning (a fixed job split evenly over the ranks): 32 textures per
    rank per step, "scaling": "strong", the same global texture numbering; an uneven split is refused"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--total", "64", "--dry_run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["config"]["textures_per_gpu_per_step"] == 32
    assert d["textures_total"] == 64 * 2 and d["first_timed_texture_by_rank"] == [64, 96]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total", "9", "--dry_run"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "does not split evenly" in (r.stdout + r.stderr)


def test_bench_refuses_a_world_that_differs_from_gpus():
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry_run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def _known_shapes_job(rank, world, device):
    """StyleSync.broadcast_known: shapes agreed in advance, one message; a source whose tensors do not fit marks the payload
    and EVERY rank raises — after the collective, so nobody is left waiting in it (the receiver used to carry
    on with a NaN payload)"""
    sync = otdist.StyleSync(device)
    g = torch.Generator().manual_seed(3)
    shapes = [(1, 8, 24), (2, 5), (0, 0)]
    payload = [torch.rand(sh, generator=g) for sh in shapes] if sync.is_source else None
    got = sync.broadcast_known(payload, shapes)
    raised = False
    try:
        sync.broadcast_known([torch.zeros(3)] if sync.is_source else None, [(4,)])
    except ValueError:
        raised = True
    again = sync.broadcast_known([torch.full((4,), 1.5)] if sync.is_source else None, [(4,)])
    return [t.numpy().copy() for t in got], sync.messages, raised, again[0].tolist()


def _known_shapes_round_robin_job(rank, world, device):
    """broadcast_known with a source per exchange (the driver spreads the style sides of a call's passes over the ranks):
    exchange p comes from rank p mod world; every rank ends up with every payload, whoever computed it"""
    sync = otdist.StyleSync(device)
    got = []
    for p in range(5):
        src = p % world
        g = torch.Generator().manual_seed(100 + p)
        payload = [torch.rand(1, 4, 6 + p, generator=g)] if rank == src else None   # only the source holds the data
        got.append(sync.broadcast_known(payload, [(1, 4, 6 + p)], src=src)[0].numpy().copy())
    return got, sync.messages


def _known_shapes_deferred_error_job(rank, world, device):
    """in spread mode a source whose tensors are bad must still JOIN the call's later exchanges (their sources are
    other ranks, and the receivers only poll the mark): with defer=True it completes all five and raises in raise_deferred();
    the other rank raises from the mark.  Nobody hangs."""
    sync = otdist.StyleSync(device, spread=True)
    issued, raised = 0, False
    try:
        for p in range(5):
            src = p % world
            payload = None
            if rank == src:
                payload = [torch.zeros(3)] if (p == 1) else [torch.full((1, 4, 6), float(p))]   # pass 1's source (rank 1) is bad
            sync.broadcast_known(payload, [(1, 4, 6)], src=src, defer=True)
            issued += 1
        sync.raise_deferred()
        sync.verify(block=True)
    except ValueError:
        raised = True
    return issued, raised


def test_style_sync_spread_bad_source_joins_every_exchange_before_raising_gloo_world2():
    res = run_world(_known_shapes_deferred_error_job, 2)
    assert res[1] == (5, True)            # the bad source issued all five exchanges, then raised
    assert res[0][1] is True              # the receiver raised from the mark (gloo: at the exchange it arrived with)


def test_style_sync_defaults_to_one_source():
    """the hook's default is spread=False — placeholders on ranks != src are a supported contract"""
    assert otdist.StyleSync(torch.device("cpu")).spread is False


def test_style_sync_known_shapes_one_source_per_pass_gloo_world2():
    res = run_world(_known_shapes_round_robin_job, 2)
    (a, ma), (b, mb) = res[0], res[1]
    assert ma == mb == 5
    for p, (x, y) in enumerate(zip(a, b)):
        want = torch.rand(1, 4, 6 + p, generator=torch.Generator().manual_seed(100 + p)).numpy()
        assert np.array_equal(x, want) and np.array_equal(y, want)


def test_style_sync_known_shapes_single_message_gloo_world2():
    res = run_world(_known_shapes_job, 2)
    (a, ma, ra, aa), (b, mb, rb, ab) = res[0], res[1]
    assert ma == mb == 3 and ra and rb and aa == ab == [1.5] * 4
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x, y)
    assert a[2].shape == (0, 0)
