"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) on xGMI.

The hot path shards by independent textures: texture i is synthesised entirely on one rank, nothing is
exchanged during iteration.  The only collective is a broadcast, once per (pass, layer), of the style-side data
(PCA basis + style features) so that one rank encodes / SVD-fits the style and the others receive <= 12 MB over xGMI
— latency-bound, far from the per-link bandwidth.  On CPU (tests) the same code runs over gloo."""
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


class StyleSync:
    """Callable hook for OptimalTexture.style_sync: rank `src` passes its list of tensors, everyone gets them back."""

    def __init__(self, device, src: int = 0, group=None):
        self.device, self.src, self.group = torch.device(device), src, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bytes_moved = 0

    @property
    def is_source(self) -> bool:
        return self.rank == self.src

    def __call__(self, payload: Optional[List[torch.Tensor]]) -> List[torch.Tensor]:
        if self.world == 1:
            return payload
        # 1. shapes (count and dims differ per pass: the PCA rank k is data dependent)
        meta = torch.zeros(1 + 8 * (1 + MAX_DIMS), dtype=torch.int64, device=self.device)
        if self.is_source:
            assert payload is not None and len(payload) <= 8
            meta[0] = len(payload)
            for i, t in enumerate(payload):
                assert t.dim() <= MAX_DIMS and t.dtype == torch.float32
                meta[1 + i * (1 + MAX_DIMS)] = t.dim()
                for d, sz in enumerate(t.shape):
                    meta[2 + i * (1 + MAX_DIMS) + d] = sz
        dist.broadcast(meta, self.src, group=self.group)
        m = meta.tolist()
        out = []
        for i in range(m[0]):
            nd = m[1 + i * (1 + MAX_DIMS)]
            shape = m[2 + i * (1 + MAX_DIMS):2 + i * (1 + MAX_DIMS) + nd]
            if self.is_source:
                t = payload[i].to(self.device).contiguous()
            else:
                t = torch.empty(shape, dtype=torch.float32, device=self.device)
            if t.numel():
                dist.broadcast(t, self.src, group=self.group)
                self.bytes_moved += t.numel() * 4
            out.append(t)
        return out


def barrier():
    if dist.is_initialized():
        dist.barrier()


def all_reduce_max(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
