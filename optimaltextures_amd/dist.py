"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) on xGMI.

The hot path shards by independent textures: texture i is synthesised entirely on one rank, nothing is
exchanged during iteration.  The only collective is a broadcast, once per forward call, of the style side of every
(pass, layer) — style features + PCA basis, packed into one buffer (<= 12 MB per layer at 512^2) — so that one rank
encodes / SVD-fits the style and the others receive it over xGMI: latency-bound, far from the per-link bandwidth.  On CPU (tests) the same code runs over gloo."""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

MAX_DIMS = 4


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(backend: Optional[str] = None):
    """Join the job described by RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).  Returns (rank, world, device)."""
    rank, world, local = env_world()
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if use_cuda else {}
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world, **kw)
    return rank, world, device


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous balanced partition of `total` independent textures: ranks < total % world get one extra"""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _padded(numel: int) -> int:
    return (numel + 63) // 64 * 64


class StyleSync:
    """Hook for OptimalTexture.style_sync: rank `src` owns the style side of a forward call, everyone gets it back.

    One exchange = TWO messages whatever the number of tensors: an int64 header (tensor count, shapes, and any integers
    that travel along, e.g. feature-map sizes) and ONE flat fp32 payload holding every tensor back to back.  The header is
    the only host synchronisation of the receiving ranks (the PCA rank k is data dependent, so shapes cannot be known in
    advance); the payload broadcast is issued asynchronously — over RCCL it runs on the communicator's own stream and the
    consumers' stream merely waits for it — and the received tensors are views into the flat buffer.

    The header has no fixed capacity: it is sized from `counts = (n_tensors, n_ints)`, which the caller passes when every
    rank can compute them (OptimalTexture.forward: 2 * passes * layers tensors); without `counts` a two-integer message
    announces them first.  Nothing is checked on one rank only between collectives: a source whose lists do not match the
    announced counts marks the header, the exchange completes, and EVERY rank raises.  The header-free exchange
    (broadcast_known) carries its mark in the first word of the payload: the source raises at once, a receiver as soon as it
    has looked at the word — immediately on a host backend (gloo), and on a GPU without ever blocking the host: the word is
    copied to pinned memory behind the broadcast and examined by `verify()`, which every later exchange calls first and
    which a caller can make blocking where it synchronises anyway (bench.py: after the timed region; optex.py: before
    saving).  No rank is left computing on a marked payload past its next exchange."""

    STATUS_WORDS = 64  # the mark travels in front of the payload; 64 floats keep every tensor on its 256-byte boundary

    def __init__(self, device, src: int = 0, group=None, always: bool = False, spread: bool = False):
        """always=True: issue the broadcasts even in a world of one (tests: the RCCL call sequence on one GPU).
        spread=False (default): everything comes from `src` (a rank of `group`), the other ranks' style images are never
        looked at — placeholders of the right shape are enough there.  spread=True (bench.py and the CLI opt in: they load
        the real style images on every rank): when every shape is known in advance (no PCA) the driver lets group rank
        (src + p) mod world encode and send the style side of pass p — EVERY rank must then hold the real style images;
        a rank that holds placeholders would send garbage for its passes, and nothing can tell."""
        self.device, self.src, self.group, self.always = torch.device(device), src, group, always
        self.spread = bool(spread)
        self._deferred = None   # a bad-source error of a spread exchange, raised once the call's exchanges are all issued
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bytes_moved = 0    # payload + header bytes this rank sent or received
        self.messages = 0       # broadcast calls issued
        self._pending = []      # (pinned host word, event) of header-free exchanges whose mark has not been looked at yet

    def verify(self, block: bool = False):
        """Raise ValueError if a header-free exchange delivered a payload its source had marked as not matching the announced
        shapes.  block=False looks only at marks whose copy has completed (never waits); block=True waits for all of them."""
        keep, bad = [], False
        for host, ev in self._pending:
            if ev is not None and not block and not ev.query():
                keep.append((host, ev))
                continue
            if ev is not None:
                ev.synchronize()
            bad = bad or float(host[0]) != 0.0
        self._pending = keep
        if bad:
            raise ValueError("StyleSync.broadcast_known: the source rank's tensors did not have the announced shapes "
                             "(the payload of that exchange was marked invalid)")

    @property
    def is_source(self) -> bool:
        return self.rank == self.src

    @staticmethod
    def header_len(n_tensors: int, n_ints: int) -> int:
        return 3 + n_tensors * (1 + MAX_DIMS) + n_ints

    def raise_deferred(self):
        """Behind the last `defer=True` exchange of a call: raise the bad-source error kept back on the source, and look at the
        marks received so far (all of them on a host backend; on a GPU those whose copy has completed — never blocking)."""
        err, self._deferred = self._deferred, None
        if err is not None:
            raise err
        self.verify()

    def broadcast_known(self, tensors: Optional[List[torch.Tensor]], shapes: List[Tuple[int, ...]], src: Optional[int] = None,
                        defer: bool = False):
        """The exchange when every rank can compute every SHAPE in advance (no PCA: the rank k is the only data-dependent
        size): no header, no host synchronisation at all — ONE asynchronous payload broadcast; the receiving ranks keep
        enqueueing kernels behind it.  source: list of fp32 tensors of exactly these shapes -> everyone: the tensors.
        src: the rank that holds the tensors of THIS exchange (default: the hook's source rank) — the driver spreads the style
        sides of a call's passes over the ranks, one source each (OptimalTexture.prefetch_style_sides).
        defer=True: nothing raises here — neither a source whose tensors are bad nor a receiver that has seen a mark — but in
        raise_deferred(): for a sequence of exchanges with different sources every rank has to issue ALL of them, or the
        ranks that go on hang in the next collective."""
        if self.world == 1 and not self.always:
            return list(tensors)
        src = self.src if src is None else int(src)  # a rank of `group` (like self.rank), mapped to the global rank below
        is_source = self.rank == src
        if not defer:
            self.verify()
        numels = [int(torch.Size(sh).numel()) for sh in shapes]
        head = self.STATUS_WORDS
        total = head + sum(_padded(k) for k in numels)
        flat = torch.empty(total, dtype=torch.float32, device=self.device)
        bad = False
        if is_source:
            bad = tensors is None or len(tensors) != len(shapes) or any(tuple(t.shape) != tuple(sh) or t.dtype != torch.float32
                                                                       for t, sh in zip(tensors or [], shapes))
            if bad:
                # the payload still goes out (nobody is left waiting in a collective), marked: word 0 != 0, the rest NaN
                flat.fill_(float("nan"))
                flat[:head] = 1.0
            else:
                flat[:head] = 0.0
                off = head
                for t, k in zip(tensors, numels):
                    flat[off:off + k].copy_(t.reshape(-1))
                    off += _padded(k)
        gsrc = src if self.group is None else dist.get_global_rank(self.group, src)
        work = dist.broadcast(flat, gsrc, group=self.group, async_op=True)
        self.messages += 1
        self.bytes_moved += total * 4
        out, off = [], head
        for sh, k in zip(shapes, numels):
            out.append(flat[off:off + k].view(tuple(sh)))
            off += _padded(k)
        work.wait()  # RCCL: the current stream waits for the communicator's stream, the host does not block
        if is_source and bad:
            err = ValueError("StyleSync.broadcast_known: the source rank's tensors do not have the announced shapes")
            if defer:
                # a call with several exchanges from rotating sources (spread mode): this rank must still JOIN the later ones —
                # the receivers only poll the mark and go on to the next broadcast — so the error is kept until
                # raise_deferred(), which the driver calls behind the last exchange of the call
                self._deferred = self._deferred or err
                return out
            raise err
        if flat.is_cuda:
            host = torch.empty(1, dtype=torch.float32, pin_memory=True)
            host.copy_(flat[:1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((host, ev))
            if not defer:
                self.verify()
        else:
            self._pending.append((flat[:1], None))
            if not defer:
                self.verify(block=True)
        return out

    def broadcast_packed(self, tensors: Optional[List[torch.Tensor]], ints: Optional[List[int]] = None,
                         counts: Optional[Tuple[int, int]] = None):
        """source: (list of fp32 tensors, list of ints) -> everyone: (list of tensors, list of ints).
        counts = (n_tensors, n_ints) if known on every rank (saves the announcing message)."""
        if self.world == 1 and not self.always:
            return list(tensors), list(ints or [])
        self.verify()
        if self.is_source:
            ints = [int(v) for v in (ints or [])]
            tensors = list(tensors or [])
        if counts is None:
            meta = torch.zeros(2, dtype=torch.int64)
            if self.is_source:
                meta[0], meta[1] = len(tensors), len(ints)
            meta = meta.to(self.device)
            dist.broadcast(meta, self.src, group=self.group)
            n_t, n_i = (int(v) for v in meta.tolist())
            self.messages += 1
            self.bytes_moved += 16
        else:
            n_t, n_i = int(counts[0]), int(counts[1])
        hlen = self.header_len(n_t, n_i)
        header = torch.zeros(hlen, dtype=torch.int64)
        flat = None
        if self.is_source:
            ok = (len(tensors) == n_t and len(ints) == n_i and
                  all(t.dim() <= MAX_DIMS and t.dtype == torch.float32 for t in tensors))
            if not ok:
                header[0] = -1  # every rank raises after the exchange: no rank is left waiting in a collective
            else:
                header[0], header[1] = n_t, n_i
                for i, t in enumerate(tensors):
                    base = 3 + i * (1 + MAX_DIMS)
                    header[base] = t.dim()
                    for d, sz in enumerate(t.shape):
                        header[base + 1 + d] = sz
                header[2] = sum(_padded(t.numel()) for t in tensors)
                if ints:
                    header[3 + n_t * (1 + MAX_DIMS):] = torch.tensor(ints, dtype=torch.int64)
                flat = torch.empty(int(header[2]), dtype=torch.float32, device=self.device)
                off = 0
                for t in tensors:  # every tensor starts on a 256-byte boundary: the kernels' 16-byte vector paths stay usable
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                    off += _padded(t.numel())
        header = header.to(self.device)
        dist.broadcast(header, self.src, group=self.group)
        h = header.tolist()  # the one host synchronisation of a receiving rank
        self.messages += 1
        self.bytes_moved += hlen * 8
        if h[0] < 0:
            raise ValueError(f"StyleSync.broadcast_packed: the source rank's lists do not match the announced counts "
                             f"({n_t} tensors, {n_i} ints) or hold a non-fp32 / >{MAX_DIMS}-d tensor")
        total = h[2]
        if flat is None:
            flat = torch.empty(total, dtype=torch.float32, device=self.device)
        work = dist.broadcast(flat, self.src, group=self.group, async_op=True) if total else None
        self.messages += 1 if total else 0
        self.bytes_moved += total * 4
        out, off = [], 0
        for i in range(n_t):
            base = 3 + i * (1 + MAX_DIMS)
            shape = h[base + 1:base + 1 + h[base]]
            numel = 1
            for sz in shape:
                numel *= sz
            out.append(flat[off:off + numel].view(shape))
            off += _padded(numel)
        if work is not None:
            work.wait()  # RCCL: the current stream waits for the communicator's stream, the host does not block
        ibase = 3 + n_t * (1 + MAX_DIMS)
        return out, h[ibase:ibase + n_i]

    def __call__(self, payload: Optional[List[torch.Tensor]]) -> List[torch.Tensor]:
        return self.broadcast_packed(payload)[0]


# ------------------------------------------------------------------------------------------------ seeding rule of sharded jobs
# Independent textures are numbered globally (texture i of a job, whatever rank synthesises it).  Everything random about
# texture i is a function of (seed, i) alone, so that a job gives the same images on 1, 2, 4 or 8 GPUs:
#   * its noise initialisation comes from a generator seeded texture_seed(seed, i);
#   * rotations: textures that share one rotation sequence (the reference shares R across its batch, optex.py:168-170)
#     form a ROTATION GROUP of `group_size` consecutive textures, group q draws from RandomState(rotation_seed(seed, q));
#     a rank always processes whole groups.  group_size = 1 gives every texture its own sequence (the reference run as
#     separate B = 1 jobs).
NOISE_STRIDE = 1_000_003


def texture_seed(seed: int, index: int) -> int:
    return (int(seed) * NOISE_STRIDE + int(index)) % (1 << 63)


def rotation_seed(seed: int, group: int) -> int:
    return (1000 + int(seed) * 7919 + int(group)) % (1 << 32)  # numpy's legacy RandomState takes 32-bit seeds


def texture_noise(first: int, count: int, chw, device, seed: int = 0, on_cpu: bool = False) -> torch.Tensor:
    """uniform [0, 1) noise images of textures first .. first + count - 1: [count, *chw] on `device`.  on_cpu=True draws
    from a CPU generator and moves the result (what the reference CLI does, optex.py:263-265); otherwise the device's."""
    gdev = torch.device("cpu") if on_cpu else torch.device(device)
    g = torch.Generator(device=gdev)
    out = torch.empty((count, *chw), dtype=torch.float32, device=device)
    for j in range(count):
        g.manual_seed(texture_seed(seed, first + j))
        out[j].copy_(torch.rand(tuple(chw), generator=g, device=gdev), non_blocking=True)
    return out


def rotation_rng(seed: int, group: int):
    import numpy as np
    return np.random.RandomState(rotation_seed(seed, group))


def rotation_stream(seed: int, groups, device):
    """the same numpy stream(s) as rotation_rng, advanced on the GPU (rotation.DeviceNormals): `groups` is one rotation group
    (one sequence shared by its textures) or a list of them (one sequence per texture, group size 1)"""
    from .rotation import DeviceNormals
    many = isinstance(groups, (list, tuple, range))
    return DeviceNormals([rotation_seed(seed, g) for g in (groups if many else [groups])], device)   # seeded on the device


def barrier():
    if dist.is_initialized():
        dist.barrier()


def shutdown():
    """Leave the job in step: a barrier, then the process group is destroyed explicitly.  A rank that simply exits while a peer
    is still tearing down its backend can make that peer abort (seen with gloo: SIGABRT in rank 0 of a two-rank dry run, one
    run in ten) — and the launcher then reports a failed job although the result line was printed."""
    if dist.is_available() and dist.is_initialized():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()


def all_gather_floats(value: float, device) -> List[float]:
    """every rank's value, in rank order (bench.py reports per-rank step times so a scaling loss can be located)"""
    if not dist.is_initialized():
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def all_reduce_max(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
