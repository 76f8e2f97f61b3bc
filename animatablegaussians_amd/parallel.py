"""View-sharded data parallelism: one process per GPU, every rank renders its share of the step's
camera views of the SAME Gaussians, and the only exchange is the sum over views of the per-Gaussian attribute
gradients (and, once the networks are in the step, of the StyleUNet gradients) -- one all-reduce per step.

Backend-agnostic on purpose: ``nccl`` (= RCCL over xGMI) on the GPU node, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def views_of_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of the step's views; every view is rendered exactly once."""
    return list(range(rank, n_views, world))


class GradSync:
    """Packs a fixed list of gradient tensors into one flat buffer and all-reduces it (sum, optionally mean).

    The flat buffer is allocated once; ``start`` packs and launches the collective (async), ``finish`` waits and
    scatters the reduced values back into the ``.grad`` fields."""

    def __init__(self, params: Sequence[torch.Tensor], average: bool = False):
        self.params = list(params)
        self.average = average
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.work = None

    def start(self):
        off = 0
        for p in self.params:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            self.flat[off:off + p.numel()].copy_(g.reshape(-1))
            off += p.numel()
        if dist.is_initialized() and dist.get_world_size() > 1:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.average and dist.is_initialized():
            self.flat /= dist.get_world_size()
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
