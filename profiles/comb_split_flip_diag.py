"""Where do the grouped chain's gradients differ between the comb convolution with and without the concatenation?   (GPU)
    python profiles/comb_split_flip_diag.py
Same network, same inputs, same upstream gradient; only grouped.set_comb_split differs.  Prints how the two pose-map gradients (3 x 512 x 512,
every sample) and the forward images differ: a broad difference means one path is less accurate, a difference confined to a few spots of the
size of a receptive field means leaky-ReLU slope selections that flipped on pre-activations within rounding of zero."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animatablegaussians_amd import grouped as gr, synth  # noqa: E402
from animatablegaussians_amd.styleunet import DualStyleUNet  # noqa: E402

dev = torch.device("cuda:0")
net = DualStyleUNet(inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2)
net.load_reference_state_dict(synth.named_fill(net.reference_state_dict()))
net = net.to(dev)
style = (torch.ones(1, 512) / np.sqrt(512)).to(dev)
G = torch.randn(1, 6, 1024, 1024, generator=torch.Generator().manual_seed(4242)).to(dev)
res = {}
for on in (True, False):
    prev = gr.set_comb_split(on)
    for p in net.parameters():
        p.grad = None
    pose = synth.pose_map(512).to(dev).requires_grad_(True)
    images = gr.GroupedStyleUNets([net]).forward([style], pose)[0]
    (images * G).sum().backward()
    torch.cuda.synchronize()
    res[on] = (images.detach().double(), pose.grad.detach().double()[0], {n: net._p(n).grad.detach().double().clone() for n in net._learnable})
    gr.set_comb_split(prev)
im1, pg1, gr1 = res[True]
im0, pg0, gr0 = res[False]
print(f"forward images: max |difference| / max|image| = {float((im1 - im0).abs().max() / im0.abs().max()):.2e}")
d = ((pg1 - pg0).abs() / pg0.abs().max()).cpu().numpy()
print("pose-map gradient, comb split on vs off, |difference| / max|grad| over all 786 432 samples: " +
      " ".join(f"p{q}={np.percentile(d, q):.2e}" for q in (50, 90, 99, 99.9, 99.99, 100)))
big = d.max(0) > 1e-3
print(f"pixels with a difference > 1e-3: {int(big.sum())} of {big.size}; 16 x 16 blocks containing one: {int(big.reshape(32, 16, 32, 16).any(axis=(1, 3)).sum())} of 1024")
for thr in (1e-4, 1e-3, 5e-3):
    b = d.max(0) > thr
    ys, xs = np.nonzero(b)
    print(f"   > {thr:.0e}: {int(b.sum()):6d} pixels" + (f", bounding boxes of their 64 x 64 cells: {len(set(zip(ys // 64, xs // 64)))} cells" if len(ys) else ""))
rows = []
for n in gr1:
    rows.append((float((gr1[n] - gr0[n]).abs().max() / gr0[n].abs().max().clamp_min(1e-30)), n))
rows.sort(reverse=True)
print("parameter gradients, largest |difference| / max|grad| between the two paths:")
for v, n in rows[:8]:
    print(f"   {v:.2e}  {n}")
print("   median over the tensors: %.2e" % np.median([v for v, _ in rows]))
