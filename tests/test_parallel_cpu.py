"""World-size-2 gloo test (CPU) of the view sharding + gradient exchange used by bench.py / training at N > 1."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    from animatablegaussians_amd.parallel import GradSync, views_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, n_views = 1000, 8
    g = torch.Generator().manual_seed(0)
    params = [torch.zeros(P, c, requires_grad=True) for c in (3, 3, 4, 1, 3)]
    # every view v contributes a deterministic gradient field f(v); rank r only renders its own views
    def grad_of_view(v, p):
        return torch.full_like(p, float(v + 1)) * torch.arange(p.numel()).reshape(p.shape) / p.numel()
    mine = views_of_rank(n_views, rank, world)
    for p in params:
        p.grad = sum(grad_of_view(v, p) for v in mine)
    sync = GradSync(params)
    sync.start()
    sync.finish()
    want = [sum(grad_of_view(v, p) for v in range(n_views)) for p in params]
    ok = all(torch.allclose(p.grad, w, rtol=1e-6) for p, w in zip(params, want))
    covered = sorted(sum((views_of_rank(n_views, r, world) for r in range(world)), []))
    out[rank] = bool(ok and covered == list(range(n_views)))
    dist.destroy_process_group()


def test_view_sharding_and_grad_allreduce_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(out) == {0: True, 1: True}


def test_views_of_rank_partition():
    from animatablegaussians_amd.parallel import views_of_rank
    for world in (1, 2, 4, 8):
        got = sorted(sum((views_of_rank(16, r, world) for r in range(world)), []))
        assert got == list(range(16))
        assert max(len(views_of_rank(16, r, world)) for r in range(world)) == 16 // world
