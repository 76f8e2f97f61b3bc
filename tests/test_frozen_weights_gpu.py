"""Frozen weights (include/ag_layers.h AgGroupedLayerArgs.packed_weights / weights_cached; grouped.GroupedStyleUNets._frozen_cache_for): in eval mode
under no_grad the StyleUNets keep their weights' modulation, maxima and packed images between frames.  The cached path must give the SAME BITS as
the uncached one, on the frame that fills the cache, on the frames that use it, for another pose, and after any parameter update."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    import torch
    from animatablegaussians_amd.avatar import AvatarNet
    torch.manual_seed(31359)
    n = AvatarNet.synthetic({'with_viewdirs': True})
    n.eval()
    return n


def _render(net, seed, frozen):
    import torch
    from test_avatar_net_gpu import _items
    os.environ["AG_FROZEN_WEIGHTS"] = "1" if frozen else "0"
    try:
        items = _items(net, seed=seed)
        with torch.no_grad():
            net.get_pose_map(items)
            out = net.render(items)
        return out['rgb_map'].clone(), out['mask_map'].clone()
    finally:
        os.environ.pop("AG_FROZEN_WEIGHTS", None)


def _cache_of(net):
    g = net._grouped_nets()
    return getattr(g, "_frozen_cache", {}) if g is not None else {}


def test_cached_frames_equal_uncached_frames_bit_for_bit(net):
    import torch
    if net._grouped_nets() is None:
        pytest.skip("the grouped chain is off")
    net._grouped_nets().invalidate_frozen()
    want3, want5 = _render(net, 3, False), _render(net, 5, False)
    assert not _cache_of(net), "the switch must keep the cache out of the way"
    first = _render(net, 3, True)                  # fills the cache
    n_entries = len(_cache_of(net))
    assert n_entries > 50, f"only {n_entries} cache entries: the layer calls did not register"
    again = _render(net, 3, True)                  # runs from it
    other = _render(net, 5, True)                  # another pose, same weights
    assert len(_cache_of(net)) == n_entries
    for got, want in ((first, want3), (again, want3), (other, want5)):
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert not torch.equal(want3[0], want5[0])


def test_a_parameter_update_empties_the_cache(net):
    import torch
    from animatablegaussians_amd.optim import FusedAdam
    if net._grouped_nets() is None:
        pytest.skip("the grouped chain is off")
    before = _render(net, 3, True)
    p = net.color_net._p("convs1.0.conv.weight")
    with torch.no_grad():
        p.mul_(1.5)                                # an in-place update: the version counter moves
    after, want = _render(net, 3, True), _render(net, 3, False)
    assert torch.equal(after[0], want[0]) and not torch.equal(after[0], before[0])
    # the fused optimizer writes the parameters from a native kernel: it must move the version counters itself
    q = net.color_net._p("convs1.1.conv.weight")
    opt = FusedAdam([q], lr=1e-2)
    q.grad = torch.ones_like(q)
    v0 = q._version
    opt.step()
    assert q._version > v0
    after2, want2 = _render(net, 3, True), _render(net, 3, False)
    assert torch.equal(after2[0], want2[0]) and not torch.equal(after2[0], after[0])
    q.grad = None


def test_training_mode_and_autograd_never_use_the_cache(net):
    import torch
    from test_avatar_net_gpu import _items
    if net._grouped_nets() is None:
        pytest.skip("the grouped chain is off")
    net._grouped_nets().invalidate_frozen()
    items = _items(net, seed=3)
    net.get_pose_map(items)
    net.eval()
    out = net.render(items)                        # autograd on: no cache
    assert out['rgb_map'].requires_grad and not _cache_of(net)
    net.train()
    try:
        with torch.no_grad():
            net.render(items)                      # training mode (random styles are per-call temporaries): no cache
        assert not _cache_of(net)
    finally:
        net.eval()
