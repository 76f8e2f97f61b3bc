"""Guards that are statistical or depend on the state of the box (round 4: collected LAST -- `test_zz_*` -- so that `pytest -x` can never
hide a parity test behind one of them; the bench-contract tests that spawn whole benchmark processes sit in test_zz_bench_contract_gpu.py
for the same reason)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    import torch
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(31359)
    return AvatarNet.synthetic({'with_viewdirs': True})


def _items(net):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_avatar_net_gpu import _items as make
    return make(net)


def _step_fn(net, items, target):
    import torch
    params = [(n, p) for n, p in net.named_parameters()]

    def step():
        torch.manual_seed(77)                                              # the training-mode view-direction jitter
        net.zero_grad(set_to_none=True)
        out = net.render(items, bg_color=(0., 0., 0.))
        loss = (out['rgb_map'] - target).abs().mean() + 0.005 * torch.linalg.norm(out['offset'], dim=-1).mean()
        loss.backward()
        torch.cuda.synchronize()
        return {k: out[k].detach().clone() for k in ('rgb_map', 'mask_map', 'offset', 'pos_map')}, [p.grad.clone() for _, p in params]
    return params, step


N_NOISE = 8          # serialised repeats behind every tensor's noise estimate
SLACK = 6.0          # allowed multiple of the largest deviation seen among them
FLOOR = 2e-4         # + this fraction of the tensor's largest gradient


def test_training_step_is_reproducible_and_streams_do_not_disturb_it(net, monkeypatch):
    """The whole training iteration (three StyleUNets, assembly, LBS, rasterizer, loss, backward) run repeatedly from the same state.

    * Forward products are deterministic: every map of every repeat must be BIT-equal -- in the default grouped single-chain mode, in the
      one-network-at-a-time mode with the networks on their six HIP streams, and with that mode serialised on one stream (AG_SINGLE_STREAM=1).
      This is the check that caught the cross-wave packed-fp32 disturbance in round 2 (profiles/r03_packed_fp32_hazard.md; the library has
      no packed-fp32 instruction since: tests/test_no_packed_fp32_cpu.py).
    * Gradients: the bias / noise-strength / style reductions are deterministic since round 4 (two-stage, fixed order), but their INPUTS are
      not bit-equal between runs: the rasterizer's backward and the split-K weight gradients accumulate with float atomics in arrival
      order.  Each parameter tensor is therefore held to its OWN run-to-run noise, measured over N_NOISE serialised repeats (round 3 used
      two, and a 12 x excursion of one scalar made the record red): deviation of every concurrent run from the reference run
      <= SLACK x the largest deviation among the N_NOISE repeats + FLOOR x the tensor's largest gradient.  With 8 repeats the chance that
      an honest 9th..11th draw exceeds 6 x the maximum of the first 8 is negligible for any light-tailed noise, and the floor covers
      tensors whose repeats happen to agree to the bit."""
    import torch
    items = _items(net)
    net.get_pose_map(items)
    net.train()
    target = torch.rand(1024, 1024, 3, generator=torch.Generator().manual_seed(5)).cuda()
    params, step = _step_fn(net, items, target)
    rel_of = lambda a, b: float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))   # noqa: E731

    def check(ref_maps, maps, what):
        for k in ref_maps:
            assert torch.equal(ref_maps[k], maps[k]), (what, k, float((ref_maps[k] - maps[k]).abs().max()))

    # ---- default mode: the grouped chain --------------------------------------------------------------------------------------
    ref_maps_g, ref_grads_g = step()
    worst_g = 0.0
    for rep in range(3):
        maps, grads = step()
        check(ref_maps_g, maps, f"grouped repeat {rep}")
        worst_g = max(worst_g, max(rel_of(a, b) for a, b in zip(ref_grads_g, grads)))
        del maps, grads
    del ref_grads_g

    # ---- one network at a time: serialised noise, then the six-stream runs ------------------------------------------------------
    prev = net.set_grouped(False)
    try:
        monkeypatch.setenv("AG_SINGLE_STREAM", "1")
        ref_maps, ref_grads = step()
        noise_t = [0.0] * len(params)
        for rep in range(N_NOISE):
            maps2, grads2 = step()
            check(ref_maps, maps2, f"serialised repeat {rep}")
            noise_t = [max(n, rel_of(a, b)) for n, a, b in zip(noise_t, ref_grads, grads2)]
            del maps2, grads2
        monkeypatch.delenv("AG_SINGLE_STREAM")
        rows = []
        for rep in range(3):
            maps, grads = step()
            check(ref_maps, maps, f"concurrent run {rep}")
            for (name, _), a, b, nt in zip(params, ref_grads, grads, noise_t):
                rows.append((rel_of(a, b) / (SLACK * nt + FLOOR), rel_of(a, b), nt, name, rep))
            del maps, grads
    finally:
        net.set_grouped(prev)
    # the grouped chain and the one-by-one path compute the same maps up to split-K summation order
    for k in ref_maps:
        d = float((ref_maps[k] - ref_maps_g[k]).abs().max() / ref_maps[k].abs().max().clamp_min(1e-30))
        assert d <= (1e-2 if k in ('rgb_map', 'mask_map') else 5e-5), (k, d)       # images pass the rasterizer's discrete decisions
    rows.sort(reverse=True)
    print(f"grouped chain: 3 repeats bit-equal forward, worst gradient deviation {worst_g:.1e}; concurrent vs serialised: "
          f"{sum(p.numel() for _, p in params) / 1e6:.1f} M gradients x 3 runs, worst ratios to {SLACK:g} x the largest of {N_NOISE} serialised "
          f"deviations + {FLOOR:g}: " + "; ".join(f"{r:.2f} ({n}: {d:.1e} vs noise {t:.1e})" for r, d, t, n, _ in rows[:3]))
    assert rows[0][0] <= 1.0, rows[:5]
    net.zero_grad(set_to_none=True)


def test_training_passes_do_not_pile_up_memory_without_the_garbage_collector():
    """The layer-level autograd nodes hand every saved tensor (their own OUTPUT among them) to autograd's save_for_backward and drop
    their private references when the backward returns: with Python's cycle collector switched off the allocated memory is the same
    after every pass (a node -> output -> node cycle would add ~5 GB per pass until the collector runs)."""
    import gc
    import torch
    from animatablegaussians_amd import synth
    from animatablegaussians_amd.styleunet import DualStyleUNet

    dev = torch.device("cuda:0")
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
    pose = synth.pose_map(512).to(dev)
    style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
    G = torch.randn(1, 6, 1024, 1024, device=dev)
    was = gc.isenabled()
    gc.disable()
    try:
        seen = []
        for _ in range(4):
            for p in net.parameters():
                p.grad = None
            images, _ = net([style], pose, randomize_noise=False)
            (images * G).sum().backward()
            del images
            torch.cuda.synchronize()
            seen.append(torch.cuda.memory_allocated(dev))
    finally:
        if was:
            gc.enable()
    assert max(seen[1:]) - min(seen[1:]) < (64 << 20), seen


