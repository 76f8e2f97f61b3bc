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


def _bucket_worker(rank, world, port, out):
    """Two replicas of a small MLP, different data per rank: after BucketedGradSync the gradients equal the mean of the
    two single-rank gradients, buckets fire during backward, and a parameter without gradient does not stall finish()."""
    from animatablegaussians_amd.parallel import BucketedGradSync
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 8))
    unused = torch.nn.Parameter(torch.ones(5))
    params = list(net.parameters()) + [unused]
    sync = BucketedGradSync(params, bucket_bytes=1024)          # several buckets
    assert len(sync.buckets) >= 3

    def data(r):
        g = torch.Generator().manual_seed(100 + r)
        return torch.randn(32, 16, generator=g), torch.randn(32, 8, generator=g)

    ok = True
    for step in range(2):                                       # two steps: zero() re-arms the buckets
        sync.zero()
        x, y = data(rank + 10 * step)
        ((net(x) - y) ** 2).mean().backward()
        sync.finish()
        got = [p.grad.clone() for p in net.parameters()]
        # reference: plain autograd on both ranks' data, averaged
        want = None
        for r in range(world):
            ref = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
            xr, yr = data(r + 10 * step)
            h = torch.relu(torch.nn.functional.linear(xr, ref[0], ref[1]))
            h = torch.relu(torch.nn.functional.linear(h, ref[2], ref[3]))
            ((torch.nn.functional.linear(h, ref[4], ref[5]) - yr) ** 2).mean().backward()
            gs = [p.grad for p in ref]
            want = gs if want is None else [a + b for a, b in zip(want, gs)]
        want = [w / world for w in want]
        ok = ok and all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(got, want))
        ok = ok and bool((unused.grad == 0).all())
        ok = ok and all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in params)   # still views of the flat buffer
    # a second backward in the same step must not silently write into buckets that are already reducing
    sync.zero()
    x, y = data(rank)
    ((net(x) - y) ** 2).mean().backward()
    try:
        ((net(x) - y) ** 2).mean().backward()
        ok = False
    except RuntimeError as e:
        ok = ok and "second gradient" in str(e)
    sync.finish()
    sync.close()
    # defer_to_finish: two backward calls per step (one loss per view), everything reduced in finish()
    for p in params:
        p.grad = None
    sync2 = BucketedGradSync(params, bucket_bytes=1024, defer_to_finish=True)
    sync2.zero()
    for v in range(2):
        x, y = data(rank + 10 * v)
        (((net(x) - y) ** 2).mean() / 2).backward()
    sync2.finish()
    got = [p.grad.clone() for p in net.parameters()]
    want = None
    for r in range(world):
        for v in range(2):
            ref = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
            xr, yr = data(r + 10 * v)
            h = torch.relu(torch.nn.functional.linear(xr, ref[0], ref[1]))
            h = torch.relu(torch.nn.functional.linear(h, ref[2], ref[3]))
            (((torch.nn.functional.linear(h, ref[4], ref[5]) - yr) ** 2).mean() / 2).backward()
            gs = [p.grad for p in ref]
            want = gs if want is None else [a + b for a, b in zip(want, gs)]
    want = [w / world for w in want]
    ok = ok and all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(got, want))
    out[rank] = bool(ok)
    sync2.close()
    dist.destroy_process_group()


def test_bucketed_grad_sync_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(out) == {0: True, 1: True}


def _order_worker(rank, world, port, out):
    """The two ranks receive their gradients in DIFFERENT orders (rank 0: last parameter first, as a plain backward; rank 1: first
    parameter first, then a shuffle) -- what a backward spread over several HIP streams can produce.  Collectives must still be issued in
    the same (bucket-index) order on both ranks, and the sums must be right."""
    from animatablegaussians_amd.parallel import BucketedGradSync
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = [torch.nn.Parameter(torch.zeros(300)) for _ in range(12)]
    sync = BucketedGradSync(params, bucket_bytes=2 * 300 * 4, average=False)       # 2 parameters per bucket -> 6 buckets
    assert len(sync.buckets) == 6
    ok = True
    orders = {0: [list(range(11, -1, -1)), [5, 4, 11, 10, 1, 0, 7, 6, 3, 2, 9, 8]],
              1: [list(range(12)), [0, 7, 3, 10, 1, 6, 11, 4, 9, 2, 5, 8]],
              2: [[6, 7, 8, 9, 10, 11, 0, 1, 2, 3, 4, 5], [11, 0, 10, 1, 9, 2, 8, 3, 7, 4, 6, 5]],          # (world size 4, round 6)
              3: [[1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10], [2, 9, 4, 7, 0, 11, 6, 1, 8, 3, 10, 5]]}
    launch_orders = []
    for step in range(2):
        sync.zero()
        for i in orders[rank][step]:                       # drive the accumulate hooks one parameter at a time, in this rank's order
            (params[i] * float((rank + 1) * (i + 1))).sum().backward()
        mid = list(sync.launch_order)                      # what went out during the "backward"
        sync.finish()
        launch_orders.append(list(sync.launch_order))
        ok = ok and sync.launch_order == sorted(sync.launch_order) == list(range(6)) and mid == sorted(mid)
        for i, p in enumerate(params):
            ok = ok and bool(torch.allclose(p.grad, torch.full((300,), float(sum((r + 1) * (i + 1) for r in range(world))))))
    out[rank] = (bool(ok), launch_orders)
    sync.close()
    dist.destroy_process_group()


def test_bucket_collectives_are_issued_in_the_same_order_on_every_rank():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(out)
    assert res[0][0] and res[1][0], res
    assert res[0][1] == res[1][1] == [list(range(6))] * 2


def test_world_size_one_pass_through_surface():
    """at world size 1 the exchange object is a pass-through by default.  `.flat` then raises a clear error instead of being
    None, `zero()` / `finish()` / an optimizer step work through the default constructor, and `flat_when_single=True` keeps the buffer."""
    import pytest
    from animatablegaussians_amd.parallel import BucketedGradSync
    net = torch.nn.Linear(4, 3)
    sync = BucketedGradSync(list(net.parameters()))
    with pytest.raises(AttributeError, match="flat_when_single"):
        sync.flat
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    before = net.weight.detach().clone()
    for _ in range(2):
        sync.zero()
        assert all(p.grad is None for p in net.parameters())
        net(torch.ones(2, 4)).sum().backward()
        sync.finish()
        opt.step()
    assert not torch.equal(before, net.weight)
    sync.close()
    kept = BucketedGradSync(list(net.parameters()), flat_when_single=True)
    kept.zero()
    net(torch.ones(2, 4)).sum().backward()
    kept.finish()
    assert kept.flat.numel() == 15 and all(p.grad.data_ptr() >= kept.flat.data_ptr() for p in net.parameters())
    assert float(kept.flat.abs().sum()) > 0
    kept.close()


def test_pose_per_rank_dealing_of_the_training_step():
    """bench_avatar.TrainingStep's two sharding modes, host logic only: (a) the views of ONE pose are dealt round-robin over the
    ranks -- every rank has the same joint transforms and the ranks' cameras of a step are disjoint; (b) ``pose_per_rank``: every rank has its own pose
    (its own joint transforms) and needs no disjoint cameras."""
    import numpy as np
    import torch
    import bench_avatar as ba
    J = 55
    same = [ba.joint_transforms(J, "cpu", seed=7 + 0) for _ in range(2)]
    assert torch.equal(same[0], same[1])
    a, b = ba.joint_transforms(J, "cpu", seed=7), ba.joint_transforms(J, "cpu", seed=8)
    assert not torch.equal(a, b)
    for A in (a, b):                                                  # rigid: rotation blocks orthonormal
        R = A[:, :3, :3]
        assert float((R @ R.transpose(1, 2) - torch.eye(3)).abs().max()) < 1e-5

    class _Step:                                                      # the camera dealing without a GPU: the method only reads these attributes
        cameras = ba.TrainingStep.cameras
    world, V, n_cam = 4, 2, 8
    for mode in (False, True):
        per_rank = []
        for rank in range(world):
            s = _Step()
            s.world, s.rank, s.pose_per_rank, s.views = world, rank, mode, list(range(n_cam))
            per_rank.append([s.cameras(i, V) for i in range(3)])
        for i in range(3):
            dealt = [c for r in range(world) for c in per_rank[r][i]]
            if not mode:
                assert sorted(dealt) == sorted(set(dealt)) and len(dealt) == world * V          # one pose: no camera rendered twice in a step
            assert all(len(per_rank[r][i]) == V for r in range(world))


def test_split_pairs_node_equals_the_row_slices_it_replaces():
    """grouped._SplitPairs (one autograd node for the (network, view) image pairs of a stacked decoder output): same values and the same gradient
    as the per-view slices ``img[2 v : 2 v + 2].reshape(1, 2 C, H, W)`` it replaces, including views that receive no gradient (host logic, CPU)."""
    import torch
    from animatablegaussians_amd.grouped import _SplitPairs
    g = torch.Generator().manual_seed(0)
    img = torch.randn(6, 3, 5, 4, generator=g)
    a = img.clone().requires_grad_(True)
    b = img.clone().requires_grad_(True)
    pairs = _SplitPairs.apply(a)
    want = [b[2 * v:2 * v + 2].reshape(1, 6, 5, 4) for v in range(3)]
    assert len(pairs) == 3 and all(torch.equal(p, w) for p, w in zip(pairs, want))
    ups = [torch.randn(1, 6, 5, 4, generator=g) for _ in range(3)]
    (pairs[0] * ups[0]).sum().backward(retain_graph=True)               # only view 0: the others' rows must come back zero
    (want[0] * ups[0]).sum().backward(retain_graph=True)
    assert torch.equal(a.grad, b.grad) and not a.grad[2:].any()
    a.grad = b.grad = None
    sum((p * u).sum() for p, u in zip(pairs, ups)).backward()
    sum((w * u).sum() for w, u in zip(want, ups)).backward()
    assert torch.equal(a.grad, b.grad)


def _exchange_worker(rank, world, port, out):
    """bench.py's N > 1 exchange (parallel.ViewGradExchange) on two gloo ranks: every rank submits the gradient arrays of its own view per
    step through 3 slots; after every iteration of `every` steps the accumulator must hold the sum over BOTH ranks and the iteration's
    steps, the held-back mode must leave the rank's own sum, and the checksums must add up."""
    from animatablegaussians_amd.parallel import ViewGradExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, cols, every, n_slots = 257, (3, 3, 4, 1, 3), 4, 3

    def grads_of(step, r):
        g = torch.Generator().manual_seed(1000 * r + step)
        return [torch.randn(P, c, generator=g) for c in cols]

    x = ViewGradExchange(P, sum(cols), "cpu", n_slots, every)
    ok = True
    for it in range(3):
        for j in range(every):
            step = it * every + j
            x.submit(step % n_slots, grads_of(step, rank))
        x.join()
        want = sum(torch.cat(grads_of(it * every + j, r), dim=1) for j in range(every) for r in range(world))
        ok = ok and torch.allclose(x.acc, want, rtol=1e-5, atol=1e-5)
    ok = ok and x.reduced == 3 and x.count == 3 * every
    # held back (bench.py's untimed check): the rank's own sum, checksums of what was packed
    x.reset()
    x.hold_back, x.checksums = True, torch.zeros(2, dtype=torch.float64)
    for j in range(every):
        x.submit(j % n_slots, grads_of(100 + j, rank))
    mine = sum(torch.cat(grads_of(100 + j, rank), dim=1) for j in range(every))
    ok = ok and torch.allclose(x.acc, mine, rtol=1e-5, atol=1e-5) and x.reduced == 3
    ok = ok and abs(float(x.checksums[0]) - float(mine.double().sum())) <= 1e-6 * float(x.checksums[1])
    # an iteration that is cut short leaves no collective behind: the next reset() starts a fresh one
    x.hold_back, x.checksums = False, None
    x.reset()
    x.submit(0, grads_of(200, rank))
    x.reset()
    for j in range(every):
        x.submit(j % n_slots, grads_of(300 + j, rank))
    want = sum(torch.cat(grads_of(300 + j, r), dim=1) for j in range(every) for r in range(world))
    ok = ok and torch.allclose(x.acc, want, rtol=1e-5, atol=1e-5) and x.reduced == 4
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_view_grad_exchange_accumulates_and_reduces_once_per_iteration_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(out) == {0: True, 1: True}


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: world size 4 with UNEVEN view counts (10 views over 4 ranks: 3, 3, 2, 2) -- the ranks then take different numbers of steps per
# iteration but must issue the same collectives in the same order
# ---------------------------------------------------------------------------------------------------------------------------------
def _uneven_worker(rank, world, port, out):
    from animatablegaussians_amd.parallel import GradSync, ViewGradExchange, views_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_views, P, cols, n_slots = 10, 131, (3, 3, 4, 1, 3), 3
    mine = views_of_rank(n_views, rank, world)
    ok = len(mine) == (3 if rank < 2 else 2)

    # (1) GradSync: per-parameter gradients summed over a rank's own views, one all-reduce
    params = [torch.zeros(P, c, requires_grad=True) for c in cols]

    def grad_of_view(v, p):
        return torch.full_like(p, float(v + 1)) * torch.arange(p.numel()).reshape(p.shape) / p.numel()

    for p in params:
        p.grad = sum(grad_of_view(v, p) for v in mine)
    sync = GradSync(params)
    sync.start()
    sync.finish()
    ok = ok and all(torch.allclose(p.grad, sum(grad_of_view(v, p) for v in range(n_views)), rtol=1e-6) for p in params)

    # (2) ViewGradExchange: `every` = the rank's OWN number of views per iteration; one all-reduce per iteration on every rank
    def grads_of(it, v):
        g = torch.Generator().manual_seed(100 * it + v)
        return [torch.randn(P, c, generator=g) for c in cols]

    x = ViewGradExchange(P, sum(cols), "cpu", n_slots, every=len(mine))
    for it in range(3):
        for j, v in enumerate(mine):
            x.submit(j % n_slots, grads_of(it, v))
        x.join()
        want = sum(torch.cat(grads_of(it, v), dim=1) for v in range(n_views))
        ok = ok and torch.allclose(x.acc, want, rtol=1e-5, atol=1e-5)
        x.release()
    ok = ok and x.reduced == 3

    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_world4_uneven_views_gradsync_and_exchange_gloo():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(out) == {0: True, 1: True, 2: True, 3: True}


def _run_world(worker, world, timeout=240):
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    res = dict(out)
    assert set(res) == set(range(world)) and all((v[0] if isinstance(v, tuple) else v) is True for v in res.values()), res
    return res


def test_world4_bucketed_grad_sync_order_and_exchange_gloo():
    """The three world-2 workers above at world size 4 (they are written for any world size): bucketed all-reduce = the mean over FOUR ranks'
    gradients with buckets firing during backward, four different gradient-arrival orders issuing one collective order, and bench.py's
    ViewGradExchange over four ranks (sums per iteration, held-back mode, cut-short iterations)."""
    _run_world(_bucket_worker, 4)
    res = _run_world(_order_worker, 4)
    assert all(res[r][1] == res[0][1] for r in range(4))                     # the same collective order on every rank, both steps
    _run_world(_exchange_worker, 4)
