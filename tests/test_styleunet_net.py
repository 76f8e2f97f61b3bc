"""DualStyleUNet on the MI355X kernels vs the reference's own module.

Fixture: tests/golden/dual_styleunet_512_1024.npz, written by tests/golden/make_golden_dual_styleunet.py, which ran the
reference's DualStyleUNet (CPU, fixed noise) at the product size with parameters filled by synth.named_fill -- a pure
function of the state_dict key, so the GPU box rebuilds the identical 74M-parameter network without the reference.

Tolerance.  The network is ~45 fp32 convolutions deep (K up to 9216 products per output) and its backward carries
leaky-ReLU slope selections of activations that sit within rounding of zero, so two correct fp32 evaluations do not
agree to 1e-4 on every gradient: the reference's OWN fp32 run deviates from its float64 run by up to 6e-2 of a
tensor's max |grad| (noise-strength scalars: sums of ~1e7 signed terms), 4e-3 on the pose-map gradient, median 9e-5.
The fixture therefore stores the float64 values plus, per tensor, err32 = that reference-fp32 deviation, and this
test requires of our fp32 path (dev = |ours - ref64| / max|ref64| per tensor):
  * forward images: dev <= 1e-4 (the north-star tolerance; measured 3e-6, the reference's fp32 1e-6);
  * gradients, as a distribution over the 219 tensors: our 50/75/90/95/99th percentiles of dev are within 3x the same
    percentiles of err32 (measured 1.5-2.7x: sequential-K MFMA accumulation is a little noisier than oneDNN's blocked
    sums, and more forward noise selects a few more leaky-ReLU slopes differently);
  * every single tensor: dev <= 1e-2 (5e-2 for the twelve scalar noise strengths, 3e-2 for the pose-map gradient) -- a mis-wired
    layer is O(1).
The same bars hold for the single-network path in both arithmetic modes AND for the product's grouped launch chain (grouped.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dual_styleunet_512_1024.npz")


# full-tensor statistics (round 6): our percentiles within FULL_MULT x the reference-fp32's own; per-tensor caps on the block sums (of a block's sum of
# magnitudes), the tensor sum (of its sum of magnitudes) and the sum of squares (relative).  A wrong output-channel block of one layer moves its
# block sum by O(1) of the block's magnitude; values measured on the three paths are in profiles/r06_styleunet_fullstats/.
# Measured (profiles/r06_styleunet_fullstats/, four paths): ours / reference-fp32 = 2.0-3.0 at the median and 1.2-1.9 at the 90th percentile of the three
# statistics; largest block-sum deviation of a TENSOR 5.0e-3 of the block's magnitude (an activate.bias), of a one-number noise strength 3.3e-2 of its
# value (convs2.5, where the reference's own fp32 run is 6.1e-2 off).
FULL_MULT = float(os.environ.get("AG_TEST_FULL_MULT", "5"))
FULL_CAP_SUM = float(os.environ.get("AG_TEST_FULL_CAP_SUM", "1e-2"))
FULL_CAP_BLK = float(os.environ.get("AG_TEST_FULL_CAP_BLK", "2e-2"))
FULL_CAP_SQ = float(os.environ.get("AG_TEST_FULL_CAP_SQ", "2e-2"))
FULL_CAP_SCALAR = float(os.environ.get("AG_TEST_FULL_CAP_SCALAR", "8e-2"))      # one-element tensors (the twelve noise strengths): relative error of the number


def _sub(t, n=256):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy()


def test_reference_state_dict_names_and_shapes():
    """Host logic on CPU: our parameter table is the reference's state_dict (names + shapes), and the loader is strict."""
    import torch
    from animatablegaussians_amd.styleunet import DualStyleUNet

    gold = np.load(GOLD)
    want = {k[len("shape:"):]: tuple(int(v) for v in gold[k]) for k in gold.files if k.startswith("shape:")}
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)
    have = {k: tuple(v.shape) for k, v in net.reference_state_dict().items()}
    assert have == want
    assert net.n_latent == 14 and net.num_layers == 12
    sd = net.reference_state_dict()
    assert "conv_in.0.kernel" not in sd             # constants: part of state_dict(), optional for load_reference_state_dict
    net.load_reference_state_dict(sd)
    net.load_reference_state_dict(net.state_dict())
    with pytest.raises(RuntimeError):
        net.load_reference_state_dict({**sd, "bogus.weight": torch.zeros(1)})
    sd.pop("style.1.bias")
    with pytest.raises(RuntimeError):
        net.load_reference_state_dict(sd)


def test_state_dict_is_the_reference_modules_layout():
    """``state_dict()`` keys / order / shapes, ``named_parameters()`` order (what Adam indexes its state by) and the constant FIR /
    Haar buffers equal the reference module's -- fixture produced by the reference's own DualStyleUNet
    (tests/golden/make_golden_state_layout.py) for both configurations network/avatar.py:34-36 builds.  This is what lets
    main_avatar.py:777-813 (strict ``load_state_dict``, ``optm.load_state_dict``) exchange files between the two."""
    import json
    import torch
    from animatablegaussians_amd.styleunet import DualStyleUNet
    layout = json.load(open(os.path.join(os.path.dirname(GOLD), "state_layout.json")))
    for out_ch in (3, 8):
        g = layout[f"out_ch_{out_ch}"]
        net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=out_ch, out_size=1024, style_dim=512, n_mlp=2)
        sd = net.state_dict()
        assert [[k, list(v.shape)] for k, v in sd.items()] == [[k, s] for k, s, _ in g["keys"]]
        params = dict(net.named_parameters())
        assert [k for k, _, kind in g["keys"] if kind == "param"] == [k for k in sd if k in params]
        assert list(params) == g["param_order"] == net._learnable
        assert sum(p.numel() for p in params.values()) == g["n_params"]
        for k, v in g["constants"].items():
            assert torch.equal(sd[k], torch.tensor(v)), k
        # the reference's strict loader semantics on the full dict
        net.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        with pytest.raises(RuntimeError):
            net.load_state_dict({k: v for k, v in sd.items() if k != "iwt.ll"}, strict=True)


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["split_bf16", "fp32", "split_f16"])
def test_dual_styleunet_forward_backward_vs_reference_golden(math):
    """Both arithmetic paths of the convolutions (include/ag_conv.h: the default six-product bf16 split and the fp32 MFMA) end to end
    against the fixture the reference module produced."""
    import torch
    from animatablegaussians_amd import conv as agc
    from animatablegaussians_amd import synth
    from animatablegaussians_amd.styleunet import DualStyleUNet

    prev = agc.set_math(math)
    try:
        _golden_body(math)
    finally:
        agc.set_math(prev)


@pytest.mark.gpu
def test_grouped_chain_forward_backward_vs_reference_golden():
    """The PRODUCT path -- the grouped launch chain of grouped.py (here over one network: one encoder instance, its two decoders as a
    group of two, the comb convolutions without the concatenation) -- against the same fixture of the reference module, same bars."""
    from animatablegaussians_amd import conv as agc
    _golden_body(agc.get_math(), grouped=True)


def _golden_body(math, grouped=False):
    import torch
    from animatablegaussians_amd import synth
    from animatablegaussians_amd.styleunet import DualStyleUNet

    gold = np.load(GOLD)
    dev = torch.device("cuda:0")
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)
    net.load_reference_state_dict(synth.named_fill(net.reference_state_dict()))
    net = net.to(dev)
    pose = synth.pose_map(512).to(dev).requires_grad_(True)
    style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
    if grouped:
        from animatablegaussians_amd.grouped import GroupedStyleUNets
        images = GroupedStyleUNets([net]).forward([style], pose)[0]
    else:
        images, _ = net([style], pose, randomize_noise=False)
    assert images.shape == (1, 6, 1024, 1024)

    scale = float(gold["images_max"])
    fwd_rows = []
    for key, got in (("images_sub16", images[0, :, ::16, ::16]), ("images_crop_a", images[0, :, 500:532, 500:532]),
                     ("images_crop_b", images[0, :, 100:132, 700:732])):
        d = np.abs(got.detach().cpu().numpy() - gold[key]).max()
        assert d <= 1e-4 * scale, f"{key}: max diff {d:.3e} vs scale {scale:.3f}"
        fwd_rows.append((d / scale, float(gold["err32:" + key]), key))
    assert abs(float(images.detach().abs().double().mean()) - float(gold["images_absmean"])) <= 1e-5 * scale

    G = torch.randn(images.shape, generator=torch.Generator().manual_seed(4242)).to(dev)
    (images * G).sum().backward()
    rows = []   # (ours, reference fp32, name)
    pose_dev = np.abs(pose.grad[0, :, ::8, ::8].cpu().numpy() - gold["pose_grad_sub8"]) / float(gold["pose_grad_max"])
    d = pose_dev.max()
    rows.append((float(d), float(gold["err32:pose_grad_sub8"]), "pose"))
    for ref_name in net._learnable:
        g = net._p(ref_name).grad
        assert g is not None, ref_name
        gmax = float(gold["gmax:" + ref_name])
        d = np.abs(_sub(g) - gold["grad:" + ref_name]).max() / max(gmax, 1e-30)
        rows.append((float(d), float(gold["err32:grad:" + ref_name]), ref_name))
    ours, ref = np.array([o for o, _, _ in rows]), np.array([r for _, r, _ in rows])
    # Round 6: statistics over EVERY gradient element (the probe above reads 256 samples per tensor, 0.08 % of 74 M elements: a fault confined to one
    # output-channel block or one edge tile can sit between them).  Per tensor, in float64 on the device: the sum and the sums of NBLK = 16 contiguous
    # blocks of the flattened tensor (dimension 0 = output channels is the slowest) against the reference module's float64 values, normalised by the
    # (block's) sum of magnitudes, and the sum of squares relative to the reference's.  Yardstick as above: the reference's OWN float32 run (e32*).
    full = []     # (name, dev_sum, ref32_sum, dev_blk, ref32_blk, dev_sq, ref32_sq)
    for ref_name in list(net._learnable) + ["@pose"]:
        g = (pose.grad if ref_name == "@pose" else net._p(ref_name).grad).detach().double().flatten()
        n = g.numel()
        edges = [(n * b) // 16 for b in range(17)]
        blk = np.array([float(g[edges[b]:edges[b + 1]].sum()) for b in range(16)])
        d_sum = abs(float(g.sum()) - float(gold["fsum:" + ref_name])) / max(float(gold["fabs:" + ref_name]), 1e-300)
        d_blk = float(np.max(np.abs(blk - gold["fblk:" + ref_name]) / np.maximum(gold["fblkabs:" + ref_name], 1e-300)))
        d_sq = abs(float((g * g).sum()) - float(gold["fsq:" + ref_name])) / max(float(gold["fsq:" + ref_name]), 1e-300)
        full.append((ref_name, d_sum, float(gold["e32sum:" + ref_name]), d_blk, float(gold["e32blk:" + ref_name]), d_sq, float(gold["e32sq:" + ref_name]), n))
    fo = {k: np.array([r[i] for r in full]) for k, i in (("sum", 1), ("rsum", 2), ("blk", 3), ("rblk", 4), ("sq", 5), ("rsq", 6))}
    print(f"\n[parity] full-tensor gradient statistics over {len(full)} tensors ({math}{', grouped' if grouped else ''}), ours / reference fp32 at p50 p90 p99 max: "
          + "; ".join(f"{k}: " + " ".join(f"{np.percentile(fo[k], q):.1e}/{np.percentile(fo['r' + k], q):.1e}" for q in (50, 90, 99, 100)) for k in ("sum", "blk", "sq")))
    out_dir = os.environ.get("AG_TEST_REPORT_DIR")
    if out_dir:
        with open(os.path.join(out_dir, f"styleunet_grad_fullstats_{math}{'_grouped' if grouped else ''}.txt"), "w") as f:
            for r in sorted(full, key=lambda r: -r[3]):
                f.write(f"blk ours {r[3]:.3e} ref32 {r[4]:.3e} | sum ours {r[1]:.3e} ref32 {r[2]:.3e} | sq ours {r[5]:.3e} ref32 {r[6]:.3e} | {r[0]}\n")
    for k, caps in (("sum", FULL_CAP_SUM), ("blk", FULL_CAP_BLK), ("sq", FULL_CAP_SQ)):
        for q in (50, 90):
            assert np.percentile(fo[k], q) <= FULL_MULT * max(np.percentile(fo["r" + k], q), 1e-7), (k, q, np.percentile(fo[k], q), np.percentile(fo["r" + k], q))
        col = {"sum": 1, "blk": 3, "sq": 5}[k]
        for r in full:
            cap = (2 * FULL_CAP_SCALAR if k == "sq" else FULL_CAP_SCALAR) if r[7] == 1 else caps
            assert r[col] <= cap, (k, r)
    if out_dir:
        with open(os.path.join(out_dir, f"styleunet_grad_report_{math}{'_grouped' if grouped else ''}.txt"), "w") as f:
            for o, r, n in fwd_rows:
                f.write(f"forward {n}: ours {o:.3e} ref32 {r:.3e}\n")
            # the pose-map gradient's row is a MAXIMUM over 12 288 samples: its distribution tells isolated slope flips from a broad error
            f.write("pose-map gradient deviation / max|grad|: " + " ".join(f"p{q}={np.percentile(pose_dev, q):.2e}" for q in (50, 90, 99, 99.9, 100)) +
                    f" rms={np.sqrt((pose_dev ** 2).mean()):.2e} samples>1e-3: {int((pose_dev > 1e-3).sum())} of {pose_dev.size}\n")
            for q in (50, 75, 90, 95, 99, 100):
                f.write(f"p{q}: ours {np.percentile(ours, q):.3e} ref32 {np.percentile(ref, q):.3e}\n")
            for o, r, n in sorted(rows, reverse=True):
                f.write(f"ours {o:.3e} ref32 {r:.3e} {n}\n")
    # Round 4 (bias / noise-strength / style reductions deterministic, two-stage): measured ours / reference-fp32 = 2.7, 1.7-2.2, 1.6-2.1,
    # 1.7-2.3, 1.5-1.6 at the 50/75/90/95/99th percentiles (profiles/r04_styleunet_grad_report_*.txt, all three paths): bar 3x (was 4x)
    for q in (50, 75, 90, 95):
        assert np.percentile(ours, q) <= 3 * np.percentile(ref, q), (q, np.percentile(ours, q), np.percentile(ref, q))
    # 99th percentile.  Over the parameter TENSORS on both sides: 3x (measured 1.5-2.3x in every mode and on the grouped chain).  Over ALL rows
    # (round 5: asserted again) it is the third largest of ~230 rows, and the largest rows are the thirteen one-number quantities -- twelve noise
    # strengths and the pose-map gradient's maximum --, which are heavy-tailed: re-associating the comb convolutions moves single noise strengths by
    # 4-15x in the REFERENCE'S OWN fp32 arithmetic (tests/test_styleunet_oracle_cpu.py::test_the_one_number_gradients_move_by_factors...), and the
    # bisect of round 5 (profiles/r05_grad_bisect/, r05_comb_split_flip_diag.txt) shows that the product chain's comb convolutions as two halves are
    # exactly such a re-association: forward 1.4e-6 apart, pose-gradient deviations identical at the 50 / 90 / 99th percentile of the SAMPLES
    # (1.2e-5 / 1.0e-4 / 6.3e-4 with the split and without, in fp16 and in exact-product fp32), differences above 5e-3 confined to 219 pixels in 6
    # spots.  Measured all-rows p99 / reference: 1.13 (comb off) .. 3.35 (product chain), 2.35 in exact-product fp32 with the split: bar 4x; the
    # results are deterministic since round 5 (fixed-order weight-gradient sums), so these are the same numbers on every run and box.
    tens = [(o, r) for o, r, n in rows if not n.endswith("noise.weight") and n != "pose"]
    o99, r99 = np.percentile([o for o, _ in tens], 99), np.percentile([r for _, r in tens], 99)
    assert o99 <= 3 * r99, (o99, r99)
    assert np.percentile(ours, 99) <= 4 * np.percentile(ref, 99), (np.percentile(ours, 99), np.percentile(ref, 99))
    # the pose-map gradient by DISTRIBUTION over its 12 288 stored samples (deviation / max|grad|): the bulk is what an arithmetic error would move,
    # the maximum is what a single slope flip moves.  Measured 6.3e-4 / 2.0-2.9e-3 at the 99th / 99.9th percentile on every path and mode.
    assert np.percentile(pose_dev, 99) <= 1.5e-3 and np.percentile(pose_dev, 99.9) <= 6e-3, (np.percentile(pose_dev, 99), np.percentile(pose_dev, 99.9))
    # per-tensor caps (fraction of the tensor's largest gradient; were 3e-2 / 0.12 with the atomically summed scalars): 1e-2 for
    # parameter tensors (measured worst 5.4e-3); 5e-2 for the twelve noise-strength scalars -- one number each, a sum over a whole
    # feature map of products with mixed signs, where the reference's OWN fp32 run is off by 6.1e-2 (convs2.5; ours 3.0e-2 there, every
    # other one <= 5.4e-3); 3e-2 for the pose-map gradient (an activation gradient, max over 98 k samples: isolated leaky-ReLU slope
    # flips; reference fp32 4.3e-3, ours 1.7e-2 on the grouped chain)
    for o, _, n in rows:
        assert o <= (5e-2 if n.endswith("noise.weight") else 3e-2 if n == "pose" else 1e-2), (n, o)


@pytest.mark.gpu
def test_f16_mode_is_as_close_to_float64_as_the_reference_under_its_own_cudnn_tf32():
    """The opt-in AG_CONV_MATH_F16 arithmetic (one fp16 part per operand, include/ag_conv.h) on the product's grouped chain, against the
    REFERENCE MODULE's float64 golden, held to the deviation the reference module itself shows when its convolutions round their operands to TF32
    -- the arithmetic cuDNN gives it by default on its own hardware (tests/golden/make_golden_dual_styleunet_tf32.py emulates exactly that in the
    reference's code): forward within 2x of the reference-under-TF32's deviation, gradient rows within 2x at the 50 / 75 / 90 / 95 / 99th percentile,
    every row within 3x of the largest reference-under-TF32 row.  No fp32 parity claim is made in this mode; this is its own contract."""
    import torch
    from animatablegaussians_amd import conv as agc, synth
    from animatablegaussians_amd.grouped import GroupedStyleUNets
    from animatablegaussians_amd.styleunet import DualStyleUNet

    gold = np.load(GOLD)
    tf32 = np.load(os.path.join(os.path.dirname(GOLD), "dual_styleunet_512_1024_tf32.npz"))
    dev = torch.device("cuda:0")
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)
    net.load_reference_state_dict(synth.named_fill(net.reference_state_dict()))
    net = net.to(dev)
    pose = synth.pose_map(512).to(dev).requires_grad_(True)
    style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
    prev = agc.set_math("f16")
    try:
        images = GroupedStyleUNets([net]).forward([style], pose)[0]
        G = torch.randn(images.shape, generator=torch.Generator().manual_seed(4242)).to(dev)
        (images * G).sum().backward()
        torch.cuda.synchronize()
        agc.check_status()
    finally:
        agc.set_math(prev)
    scale = float(gold["images_max"])
    for key, got in (("images_sub16", images[0, :, ::16, ::16]), ("images_crop_a", images[0, :, 500:532, 500:532]),
                     ("images_crop_b", images[0, :, 100:132, 700:732])):
        d = np.abs(got.detach().cpu().numpy() - gold[key]).max() / scale
        assert d <= 2 * float(tf32["errtf32:" + key]), (key, d, float(tf32["errtf32:" + key]))
    rows = [(float(np.abs(pose.grad[0, :, ::8, ::8].cpu().numpy() - gold["pose_grad_sub8"]).max() / float(gold["pose_grad_max"])),
             float(tf32["errtf32:pose_grad_sub8"]), "pose")]
    for ref_name in net._learnable:
        g = net._p(ref_name).grad
        d = np.abs(_sub(g) - gold["grad:" + ref_name]).max() / max(float(gold["gmax:" + ref_name]), 1e-30)
        rows.append((float(d), float(tf32["errtf32:grad:" + ref_name]), ref_name))
    ours, ref = np.array([o for o, _, _ in rows]), np.array([r for _, r, _ in rows])
    out_dir = os.environ.get("AG_TEST_REPORT_DIR")
    if out_dir:
        with open(os.path.join(out_dir, "styleunet_grad_report_f16_grouped_vs_reference_tf32.txt"), "w") as f:
            for q in (50, 75, 90, 95, 99, 100):
                f.write(f"p{q}: ours {np.percentile(ours, q):.3e} reference under TF32 {np.percentile(ref, q):.3e}\n")
            for o, r, n in sorted(rows, reverse=True):
                f.write(f"ours {o:.3e} reftf32 {r:.3e} {n}\n")
    print("f16 mode vs the reference under TF32, gradient rows: " + ", ".join(f"p{q} {np.percentile(ours, q):.2e} / {np.percentile(ref, q):.2e}" for q in (50, 90, 99, 100)))
    for q in (50, 75, 90, 95, 99):
        assert np.percentile(ours, q) <= 2 * np.percentile(ref, q), (q, np.percentile(ours, q), np.percentile(ref, q))
    assert ours.max() <= 3 * ref.max(), (ours.max(), ref.max())


@pytest.mark.gpu
@pytest.mark.parametrize("grouped", [True, False])
def test_styleunet_gradients_are_bit_reproducible(grouped):
    """Round 5: the weight gradient's pixel slices are added in a fixed order (csrc/ag_conv.hip wgrad_reduce_kernel; float atomics in arrival
    order before), which was the last non-deterministic sum of the StyleUNet path: two forward + backward passes from the same state give
    BIT-equal images, pose-map gradient and all parameter gradients -- on the product's grouped chain and one network at a time."""
    import torch
    from animatablegaussians_amd import synth
    from animatablegaussians_amd.styleunet import DualStyleUNet

    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
    pose0 = synth.pose_map(512).to(dev)
    style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
    G = torch.randn(1, 6, 1024, 1024, generator=torch.Generator().manual_seed(4242)).to(dev)
    runs = []
    for _ in range(3):
        for p in net.parameters():
            p.grad = None
        pose = pose0.clone().requires_grad_(True)
        if grouped:
            from animatablegaussians_amd.grouped import GroupedStyleUNets
            images = GroupedStyleUNets([net]).forward([style], pose)[0]
        else:
            images, _ = net([style], pose, randomize_noise=False)
        (images * G).sum().backward()
        torch.cuda.synchronize()
        runs.append((images.detach().clone(), pose.grad.clone(), {n: net._p(n).grad.clone() for n in net._learnable}))
    for im, pg, gr in runs[1:]:
        assert torch.equal(im, runs[0][0]) and torch.equal(pg, runs[0][1])
        for n, g in gr.items():
            assert torch.equal(g, runs[0][2][n]), (n, float((g - runs[0][2][n]).abs().max()))


@pytest.mark.gpu
def test_layer_level_autograd_nodes_equal_the_per_kernel_chain():
    """fused_layers.py: ConvLayer / StyledConv / ToRGB as one autograd node each run the same kernels in the same order as the chain of
    per-kernel Functions -- images bit-equal; gradients equal up to the order in which float atomics sum (weight gradients of the
    pixel-sliced convolutions, bias / noise-strength / style reductions: two runs of the SAME path differ by up to ~2e-5 of a tensor's
    scale, measured): within 1e-4 of the tensor's scale."""
    import torch
    from animatablegaussians_amd import styleunet, synth
    from animatablegaussians_amd.styleunet import DualStyleUNet

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).to(dev)
    pose = synth.pose_map(512).to(dev)
    style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
    view = torch.randn(1, 128, 128, 128, generator=torch.Generator().manual_seed(2)).to(dev) * 0.1
    G = torch.randn(1, 6, 1024, 1024, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    prev = styleunet.set_fused_layers(True)
    try:
        for fused in (True, False):
            styleunet.set_fused_layers(fused)
            for p in net.parameters():
                p.grad = None
            x = pose.clone().requires_grad_(True)
            images, _ = net([style], x, randomize_noise=False, view_feature1=view, view_feature2=view)
            (images * G).sum().backward()
            res[fused] = (images.detach().clone(), x.grad.clone(), {k: net._p(k).grad.clone() for k in net._learnable})
    finally:
        styleunet.set_fused_layers(prev)
    assert torch.equal(res[True][0], res[False][0])
    worst = 0.0
    for a, b in [(res[True][1], res[False][1])] + [(res[True][2][k], res[False][2][k]) for k in res[True][2]]:
        worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
    assert worst <= 1e-4, worst
    assert set(res[True][2]) == set(res[False][2])


@pytest.mark.gpu
def test_stacked_modulation_gemm_equals_the_per_layer_equal_linear():
    """DualStyleUNet._stage_styles: the EqualLinear modulation (dual_styleunet.py:152-155, lr_mul 1: F.linear(w, W * (1 / sqrt(512)), b)) of
    every StyledConv / ToRGB of a stage group as ONE native launch (include/ag_linear.h; rounds 3-5: one addmm on the stacked weights) -- same
    values as the per-layer formula (fp64 as the yardstick: both fp32 forms are within 2e-6 of it relative to the style scale), for both branches
    and both stage groups."""
    import math
    import torch
    import torch.nn.functional as F
    from animatablegaussians_amd.styleunet import DualStyleUNet

    torch.manual_seed(3)
    net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2).cuda()
    w_latent = torch.randn(1, 512).cuda()
    n_stages = len(net.dec)
    for branch in (1, 2):
        for stages in (range(0, min(net.VIEW_STAGE + 1, n_stages)), range(net.VIEW_STAGE + 1, n_stages)):
            stages = list(stages)
            if not stages:
                continue
            got = net._stage_styles(branch, stages, w_latent)
            assert len(got) == 3 * len(stages)
            for prefix, style in got.items():
                mw, mb = net._p(f"{prefix}.modulation.weight"), net._p(f"{prefix}.modulation.bias")
                ref32 = F.linear(w_latent, mw * (1 / math.sqrt(mw.shape[1])), bias=mb * 1.0)
                ref64 = F.linear(w_latent.double(), mw.double() * (1 / math.sqrt(mw.shape[1])), bias=mb.double())
                assert style.shape == ref32.shape == (1, mw.shape[0])
                scale = float(ref64.abs().max())
                assert float((style.double() - ref64).abs().max()) <= 2e-6 * scale
                assert float((ref32.double() - ref64).abs().max()) <= 2e-6 * scale
