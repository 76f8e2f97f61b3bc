"""CPU oracle (plain torch, any float dtype, autograd for the backward) of the whole DualStyleUNet.  TEST INFRASTRUCTURE ONLY:
imported by tests/ and by bench.py's `cpu_baseline_styleunet` leg, never by the product.

A functional restatement of the reference module `network/styleunet/dual_styleunet.py:DualStyleUNet.forward` (:792-911) at batch 1
with one style vector, fixed noise (`randomize_noise=False`) and no conditioning -- the only way `network/avatar.py:94,107,120`
calls it -- over a state_dict in the REFERENCE'S keys (the product's `DualStyleUNet.reference_state_dict()` or a reference
checkpoint).  Layer semantics, each from the cited lines:
  EqualConv2d        conv2d(x, weight * 1/sqrt(fan_in))                                                     :93-128
  ConvLayer          [Blur pad (2,2)] + EqualConv2d (stride 2 pad 0 | stride 1 pad k//2) + FusedLeakyReLU     :326-371
  ModulatedConv2d    weight = scale * W * style; demodulate by rsqrt(sum w^2 + 1e-8); upsample: conv_transpose2d stride 2 + Blur
                     pad (1,1) with the x4 kernel; else conv2d pad k//2                                      :225-300
  StyledConv         ModulatedConv2d + NoiseInjection (out + weight * noise) + FusedLeakyReLU                :570-604
  ToRGB              ModulatedConv2d 1x1 without demodulation + bias; skip: iwt -> Upsample -> dwt, added     :607-633
  FromRGB(use_wt=False)  Downsample + ConvLayer 1x1, + skip                                                   :442-470
  Haar / InverseHaar four upfirdn2d calls each                                                                :374-425
  mapping network    PixelNorm + n_mlp x EqualLinear(lr_mul 0.01, fused leaky ReLU)                          :13-18,131-165
`upfirdn2d` and `fused_leaky_relu` are the CPU branches of the reference's own ops as restated in styleunet_oracle.py.

Pinned: tests/test_styleunet_oracle_cpu.py holds this file to tests/golden/dual_styleunet_512_1024.npz, the fixture the reference
module itself produced (make_golden_dual_styleunet.py): forward and every parameter gradient, float64 and float32.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import styleunet_oracle as so


def _fir(taps=(1, 3, 3, 1), gain=1.0, dtype=torch.float32):
    k = torch.tensor(taps, dtype=dtype)
    k = k[None, :] * k[:, None]
    return k / k.sum() * gain


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):            # upfirdn2d.py:167-183 on NCHW
    n, c, h, w = x.shape
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    y = so.upfirdn2d(x.reshape(n * c, h, w), kernel.to(x.dtype), up, up, down, down, pad[0], pad[1], pad[2], pad[3])
    return y.reshape(n, c, y.shape[-2], y.shape[-1])


def fused_leaky_relu(x, bias):                                   # fused_act.py:118-129 (CPU branch)
    shape = [1, -1] + [1] * (x.dim() - 2)
    return F.leaky_relu(x + bias.view(*shape), 0.2) * (2 ** 0.5)


def _haar(dtype):
    a = 1 / (2 ** 0.5)
    lo, hi = torch.tensor([[a, a]], dtype=dtype), torch.tensor([[-a, a]], dtype=dtype)
    return lo.T * lo, hi.T * lo, lo.T * hi, hi.T * hi          # ll, lh, hl, hh


def haar_split(x):
    return torch.cat([upfirdn2d(x, k, down=2) for k in _haar(x.dtype)], 1)


def haar_merge(x):
    ll, lh, hl, hh = _haar(x.dtype)
    parts = x.chunk(4, 1)
    return sum(upfirdn2d(p, k, up=2, pad=(1, 0, 1, 0)) for p, k in zip(parts, (ll, -lh, -hl, hh)))


class DualStyleUNetOracle:
    def __init__(self, sd, inp_size=512, out_size=1024, middle_size=8, n_mlp=2, lr_mlp=0.01, comb_as_two_halves=False):
        self.sd = sd
        # conditioning probe (tests/test_styleunet_oracle_cpu.py): the decoders' comb convolutions summed as conv(out, W[:, :C1]) + conv(level, W[:, C1:])
        # -- the same expression re-associated, as ag_grouped_comb_* evaluates it -- instead of one convolution of the concatenation
        self.comb_as_two_halves = comb_as_two_halves
        self.n_mlp, self.lr_mlp = n_mlp, lr_mlp
        self.log_in, self.log_mid, self.log_out = int(math.log2(inp_size)), int(math.log2(middle_size)), int(math.log2(out_size)) - 1
        self.n_enc = self.log_in - 2 - self.log_mid + 1
        self.n_dec = self.log_out - self.log_mid

    def p(self, k):
        return self.sd[k]

    def conv_layer(self, x, prefix, downsample=False):
        base = 1 if downsample else 0
        w = self.p(f"{prefix}.{base}.weight")
        k = w.shape[-1]
        scale = 1 / math.sqrt(w.shape[1] * k * k)
        if downsample:
            x = upfirdn2d(x, _fir(dtype=x.dtype), pad=(2, 2))
            x = F.conv2d(x, w * scale, stride=2, padding=0)
        else:
            x = F.conv2d(x, w * scale, stride=1, padding=k // 2)
        return fused_leaky_relu(x, self.p(f"{prefix}.{base + 1}.bias"))

    def style_of(self, prefix, w_latent):                       # EqualLinear, lr_mul 1, bias_init 1 (:152-155)
        w = self.p(f"{prefix}.modulation.weight")
        return F.linear(w_latent, w * (1 / math.sqrt(w.shape[1])), self.p(f"{prefix}.modulation.bias"))

    def modulated_conv(self, x, prefix, w_latent, demodulate, upsample):
        w = self.p(f"{prefix}.weight")                           # [1, Cout, Cin, k, k]
        _, cout, cin, k, _ = w.shape
        scale = 1 / math.sqrt(cin * k * k)
        style = self.style_of(prefix, w_latent).view(1, 1, cin, 1, 1)
        weight = scale * w * style
        if demodulate:
            weight = weight * torch.rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8).view(1, cout, 1, 1, 1)
        weight = weight[0]
        if upsample:
            x = F.conv_transpose2d(x, weight.transpose(0, 1), stride=2, padding=0)
            return upfirdn2d(x, _fir(gain=4.0, dtype=x.dtype), pad=(1, 1))
        return F.conv2d(x, weight, padding=k // 2)

    def styled_conv(self, x, prefix, w_latent, noise, upsample):
        x = self.modulated_conv(x, f"{prefix}.conv", w_latent, True, upsample)
        x = x + self.p(f"{prefix}.noise.weight") * noise
        return fused_leaky_relu(x, self.p(f"{prefix}.activate.bias"))

    def to_rgb(self, x, prefix, w_latent, skip):
        out = self.modulated_conv(x, f"{prefix}.conv", w_latent, False, False) + self.p(f"{prefix}.bias")
        if skip is not None:
            s = haar_merge(skip)
            s = upfirdn2d(s, _fir(gain=4.0, dtype=x.dtype), up=2, pad=(2, 1))
            out = out + haar_split(s)
        return out

    def latent(self, z):
        x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
        for i in range(self.n_mlp):
            w = self.p(f"style.{i + 1}.weight")
            x = F.linear(x, w * ((1 / math.sqrt(w.shape[1])) * self.lr_mlp))
            x = fused_leaky_relu(x, self.p(f"style.{i + 1}.bias") * self.lr_mlp)
        return x

    def forward(self, style, pose, view_feature1=None, view_feature2=None):
        w_latent = self.latent(style)
        img = pose
        out = self.conv_layer(img, "conv_in", downsample=True)
        levels = [out]
        for n in range(self.n_enc):
            img = upfirdn2d(img, _fir(dtype=img.dtype), down=2, pad=(1, 1))
            out = self.conv_layer(img, f"from_rgbs.{n}.conv") + out
            out = self.conv_layer(out, f"cond_convs.{n}.conv1")
            out = self.conv_layer(out, f"cond_convs.{n}.conv2", downsample=True)
            levels.append(out)
        n_comb = self.n_enc + 1
        images = []
        for b, vf in ((1, view_feature1), (2, view_feature2)):
            out = skip = None
            for n in range(self.n_dec):
                if n == 0:
                    out = self.conv_layer(levels[-1], f"comb_convs.{n_comb - 1}")
                elif n < n_comb and self.comb_as_two_halves:
                    w = self.p(f"comb_convs.{n_comb - 1 - n}.0.weight")
                    c1, sc = out.shape[1], 1 / math.sqrt(w.shape[1] * 9)
                    y = F.conv2d(out, w[:, :c1] * sc, padding=1) + F.conv2d(levels[-1 - n], w[:, c1:] * sc, padding=1)
                    out = fused_leaky_relu(y, self.p(f"comb_convs.{n_comb - 1 - n}.1.bias"))
                elif n < n_comb:
                    out = self.conv_layer(torch.cat([out, levels[-1 - n]], 1), f"comb_convs.{n_comb - 1 - n}")
                out = self.styled_conv(out, f"convs{b}.{2 * n}", w_latent, self.p(f"noises.noise_{2 * n}"), True)
                out = self.styled_conv(out, f"convs{b}.{2 * n + 1}", w_latent, self.p(f"noises.noise_{2 * n + 1}"), False)
                skip = self.to_rgb(out, f"to_rgbs{b}.{n}", w_latent, skip)
                if n == 4 and vf is not None:               # i == 8 at :881-883: after the stage's ToRGB, before the next comb conv
                    out = out + F.interpolate(vf, out.shape[-2:], mode="bilinear")
            images.append(haar_merge(skip))
        return torch.cat(images, 1)
