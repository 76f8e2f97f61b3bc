"""FusedAdam (include/ag_optim.h) against torch.optim.Adam: the same parameters after several steps, the same state_dict layout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd,maximize", [(0.0, False), (0.01, False), (0.0, True)])
def test_fused_adam_equals_torch_adam(wd, maximize):
    """Tensors of awkward lengths (1, 3, a prime, > one chunk, a view offset that breaks 16-byte alignment), 5 steps with fresh gradients:
    parameters and both moments agree with torch.optim.Adam to fp32 rounding (different association of the same expression)."""
    import torch
    from animatablegaussians_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(1,), (3,), (4099,), (64, 64, 3, 3), (512, 512), (7, 13)] + [(17,)] * 60        # > 48 tensors: two native calls
    base = [torch.randn(*s, generator=g) for s in shapes]
    pool = torch.randn(1003, generator=g).cuda()
    mis = torch.nn.Parameter(pool[1:1001])                      # data pointer 4 bytes off a 16-byte boundary
    ours = [torch.nn.Parameter(t.clone().cuda()) for t in base] + [mis]
    ref = [torch.nn.Parameter(t.clone().cuda()) for t in base] + [torch.nn.Parameter(pool[1:1001].clone())]
    kw = dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, maximize=maximize)
    oa, ob = FusedAdam(ours, **kw), torch.optim.Adam(ref, **kw)
    for step in range(5):
        for a, b in zip(ours, ref):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (step - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    for a, b in zip(ours, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), float((a - b).abs().max())
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == 5.0
        for k in ("exp_avg", "exp_avg_sq"):      # sums of terms of mixed sign: rounding relative to the tensor's scale, not to the element
            assert float((sa[k] - sb[k]).abs().max()) <= 2e-6 * float(sb[k].abs().max()) + 1e-12, k


def test_fused_adam_state_dict_loads_into_torch_adam_and_back():
    import copy
    import torch
    from animatablegaussians_amd.optim import FusedAdam
    ps = [torch.nn.Parameter(torch.randn(33, 5).cuda()), torch.nn.Parameter(torch.randn(100).cuda())]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a, b = FusedAdam(ps, lr=1e-2), torch.optim.Adam(qs, lr=1e-2, fused=True)
    for p in ps + qs:
        p.grad = torch.ones_like(p)
    a.step()
    b.load_state_dict(copy.deepcopy(a.state_dict()))      # ours -> torch (a copy: load_state_dict keeps tensors that already have the right device and type)
    for p, q in zip(ps, qs):
        q.data.copy_(p.data)
    for p in ps + qs:
        p.grad = torch.full_like(p, 0.5)
    a.step()
    b.step()
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)
    a2 = FusedAdam(ps, lr=1e-2)
    a2.load_state_dict(copy.deepcopy(b.state_dict()))     # torch -> ours
    assert float(a2.state[ps[0]]["step"]) == 2.0 and a2.state[ps[0]]["step"].device.type == "cpu"
    h = torch.nn.Parameter(torch.randn(4).cuda().half())
    h.grad = torch.ones_like(h)
    with pytest.raises(RuntimeError):           # no fallback: anything but dense fp32 on one GPU is refused
        FusedAdam([h]).step()


def test_fused_adam_counts_steps_per_parameter_like_torch():
    """The reference pretrains position / other networks first (main_avatar.py:126-160: the colour network gets no gradient), then trains all three:
    parameters of one optimizer have then taken different numbers of steps, and Adam's bias corrections are per parameter.  Two steps in which only
    half the tensors have gradients, then two with all: identical to torch.optim.Adam."""
    import torch
    from animatablegaussians_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(4)
    base = [torch.randn(257, generator=g) for _ in range(6)]
    ours = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    ref = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    oa, ob = FusedAdam(ours, lr=1e-2), torch.optim.Adam(ref, lr=1e-2)
    for step in range(4):
        for i, (a, b) in enumerate(zip(ours, ref)):
            if step < 2 and i % 2:
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for i, (a, b) in enumerate(zip(ours, ref)):
        assert float(oa.state[a]["step"]) == float(ob.state[b]["step"]) == (2.0 if i % 2 else 4.0)
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (i, float((a - b).abs().max()))


def test_fused_adam_follows_replaced_storage_and_replaced_state():
    """Round-5 advice: the filled argument structures were cached on id(p), so a parameter whose STORAGE was replaced (p.data = ..., module.to())
    or an optimizer state replaced without load_state_dict was updated through stale addresses.  Every pointer is now refilled per step: after
    either replacement the step equals torch.optim.Adam's on the same history."""
    import torch
    from animatablegaussians_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(9)
    base = [torch.randn(1031, generator=g) for _ in range(3)]
    ours = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    ref = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    oa, ob = FusedAdam(ours, lr=1e-2), torch.optim.Adam(ref, lr=1e-2)

    def one_step():
        for a, b in zip(ours, ref):
            gr = torch.randn(a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()

    one_step()
    old_storage = ours[1].data
    ours[1].data = ours[1].data.clone()                   # same Parameter object, new storage
    keep_old = old_storage.clone()
    one_step()
    assert torch.equal(old_storage, keep_old)             # the old storage is no longer written
    for a, b in zip(ours, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    # the moment buffers replaced behind the optimizer's back (same values, new tensors)
    st = oa.state[ours[0]]
    old_m = st["exp_avg"]
    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
    keep_m = old_m.clone()
    one_step()
    assert torch.equal(old_m, keep_m)
    for a, b in zip(ours, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    # two parameter groups keep their own structures (the second used to evict the first)
    p1, p2 = torch.nn.Parameter(torch.ones(5).cuda()), torch.nn.Parameter(torch.ones(7).cuda())
    o2 = FusedAdam([{"params": [p1]}, {"params": [p2], "lr": 1e-1}], lr=1e-2)
    for _ in range(2):
        p1.grad, p2.grad = torch.ones_like(p1), torch.ones_like(p2)
        o2.step()
    assert len(o2._calls) == 2
    assert abs(float(p1[0]) - (1 - 2e-2)) < 1e-6 and abs(float(p2[0]) - (1 - 2e-1)) < 1e-6
