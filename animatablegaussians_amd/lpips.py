"""LPIPS(net='vgg', version 0.1) on the MI355X kernels -- SURVEY.md §8(f)-1, the perceptual term of the training loss
(reference ``network/lpips/lpips.py:21-127``, used by ``main_avatar.py:117-124,227-238`` on a 512^2 crop every iteration).

Same module surface and state_dict keys as the reference's ``LPIPS`` restricted to what the trainer instantiates
(``LPIPS(net='vgg')``: linear calibration on, spatial off, eval mode, frozen parameters):

  * trunk   = torchvision ``vgg16().features[0:30]`` as sliced by ``pretrained_networks.py:97-134`` -- thirteen 3x3
              convolutions (MFMA kernels, ``conv.py``) each followed by bias + ReLU in one pass (``noise_bias_act`` with slope 0),
              four 2x2 max-pools (``ag_maxpool2x2_*``); taps after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3;
  * per tap = unit-normalise both feature stacks over channels, squared difference, channel weights ``lin{k}.model.1.weight``,
              mean over pixels -- one fused kernel each way (``ag_lpips_level_*``);
  * the ground-truth image's branch runs without autograd; the trunk is frozen, so the prediction's backward is input gradients only.

torchvision (and its ImageNet weights) are third-party and absent here: the trunk's structure is restated, its weights come from
``load_reference_state_dict`` (the reference module's ``state_dict()``, which contains them after ``LPIPS(net='vgg')`` downloaded
them) -- parity of the TRUNK WEIGHTS is therefore unpinned; the arithmetic is pinned against the reference's own class
(tests/golden/make_golden_lpips.py)."""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _lib
from . import conv as agc
from .styleunet_ops import noise_bias_act

# torchvision vgg16 `features` indices of the convolutions, grouped by the reference's slices; a pool opens slices 2..5
_SLICES = (((0, 3, 64), (2, 64, 64)),
           ((5, 64, 128), (7, 128, 128)),
           ((10, 128, 256), (12, 256, 256), (14, 256, 256)),
           ((17, 256, 512), (19, 512, 512), (21, 512, 512)),
           ((24, 512, 512), (26, 512, 512), (28, 512, 512)))
CHNS = (64, 128, 256, 512, 512)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if x.dim() != 4 or x.shape[0] != 1 or not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("maxpool2x2: float32 GPU tensor [1, C, H, W]")
        x = x.contiguous()
        _, C, H, W = x.shape
        y = torch.empty((1, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        arg = torch.empty((C, H // 2, W // 2), dtype=torch.uint8, device=x.device) if x.requires_grad else None
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ag_maxpool2x2_forward(_p(y), _p(arg), _p(x), C, H, W, _stream(x.device)), "ag_maxpool2x2_forward")
        ctx.shape = (C, H, W)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, gy):
        (arg,) = ctx.saved_tensors
        C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty((1, C, H, W), dtype=torch.float32, device=gy.device)
        with _lib.on_device(gy.device):
            _lib.check(_lib.lib().ag_maxpool2x2_backward(_p(gx), _p(gy), _p(arg), C, H, W, _stream(gy.device)), "ag_maxpool2x2_backward")
        return gx


def maxpool2x2(x):
    """``nn.MaxPool2d(2, 2)`` at batch 1."""
    return _MaxPool2x2.apply(x)


class _LpipsLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f0, f1, lin):
        f0, f1, lin = f0.contiguous(), f1.contiguous(), lin.contiguous()
        C, HW = int(f0.shape[1]), int(f0.shape[2] * f0.shape[3])
        if f1.shape != f0.shape or lin.numel() != C:
            raise RuntimeError("lpips level: feature stacks of equal shape and one weight per channel")
        out = torch.zeros(1, dtype=torch.float32, device=f0.device)
        with _lib.on_device(f0.device):
            _lib.check(_lib.lib().ag_lpips_level_forward(_p(out), _p(f0), _p(f1), _p(lin), C, HW, _stream(f0.device)),
                       "ag_lpips_level_forward")
        ctx.save_for_backward(f0, f1, lin)
        return out

    @staticmethod
    def backward(ctx, g):
        f0, f1, lin = ctx.saved_tensors
        C, HW = int(f0.shape[1]), int(f0.shape[2] * f0.shape[3])
        gf0 = torch.empty_like(f0)
        g = g.contiguous()
        with _lib.on_device(f0.device):
            _lib.check(_lib.lib().ag_lpips_level_backward(_p(gf0), _p(g), _p(f0), _p(f1), _p(lin), C, HW, _stream(f0.device)),
                       "ag_lpips_level_backward")
        return gf0, None, None        # the ground-truth features and the frozen channel weights receive no gradient


class LPIPS(nn.Module):
    def __init__(self, pretrained=True, net='vgg', version='0.1', lpips=True, spatial=False, pnet_rand=False, pnet_tune=False,
                 use_dropout=True, model_path=None, eval_mode=True, verbose=False):
        super().__init__()
        if net not in ('vgg', 'vgg16') or version != '0.1' or not lpips or spatial or pnet_tune:
            raise RuntimeError("LPIPS (MI355X path): net='vgg', version '0.1', lpips=True, spatial=False, frozen trunk")
        self.chns, self.L = list(CHNS), len(CHNS)
        self._names = {}
        for si, convs in enumerate(_SLICES):
            for idx, cin, cout in convs:
                # torchvision's initialiser for VGG convolutions: kaiming_normal(fan_out, relu), zero bias
                w = torch.randn(cout, cin, 3, 3) * (2.0 / (cout * 9)) ** 0.5
                self._add(f"net.slice{si + 1}.{idx}.weight", w)
                self._add(f"net.slice{si + 1}.{idx}.bias", torch.zeros(cout))
        for k, c in enumerate(CHNS):
            # nn.Conv2d(c, 1, 1, bias=False) default init; the trained weights are non-negative
            self._add(f"lin{k}.model.1.weight", (torch.rand(1, c, 1, 1) * 2 - 1) / c ** 0.5)
        self.register_buffer("scaling_layer__shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scaling_layer__scale", torch.tensor([.458, .448, .450])[None, :, None, None])
        for p in self.parameters():
            p.requires_grad = False                      # main_avatar.py:343-344 freezes everything
        if eval_mode:
            self.eval()

    def _add(self, ref_name, value):
        attr = ref_name.replace(".", "__")
        self.register_parameter(attr, nn.Parameter(value))
        self._names[ref_name] = attr

    def _w(self, ref_name):
        return getattr(self, self._names[ref_name])

    @torch.no_grad()
    def load_reference_state_dict(self, sd, strict=True):
        """``sd`` = ``state_dict()`` of the reference's ``LPIPS(net='vgg')`` (trunk + lin layers + scaling buffers)."""
        seen = set()
        for key, value in sd.items():
            if key in self._names:
                self._w(key).copy_(value)
                seen.add(key)
            elif key in ("scaling_layer.shift", "scaling_layer.scale"):
                getattr(self, key.replace(".", "__")).copy_(value)
            elif key.startswith("lins."):
                continue                                 # the reference's ModuleList alias of lin0..lin4 (same tensors)
            elif strict:
                raise RuntimeError(f"unexpected key in reference state_dict: {key}")
        missing = [k for k in self._names if k not in seen]
        if strict and missing:
            raise RuntimeError(f"missing keys in reference state_dict: {missing[:5]}")

    def reference_state_dict(self):
        sd = {ref: getattr(self, attr).detach() for ref, attr in self._names.items()}
        sd["scaling_layer.shift"], sd["scaling_layer.scale"] = self.scaling_layer__shift, self.scaling_layer__scale
        return sd

    def features(self, x):
        """The five ReLU taps of the VGG16 trunk for one image [1, 3, H, W] (pretrained_networks.py:121-134)."""
        taps = []
        h = x
        for si, convs in enumerate(_SLICES):
            if si:
                h = maxpool2x2(h)
            for idx, _, _ in convs:
                h = agc.conv2d(h, self._w(f"net.slice{si + 1}.{idx}.weight"), stride=1, padding=1)
                h = noise_bias_act(h, None, None, self._w(f"net.slice{si + 1}.{idx}.bias"), 0.0, 1.0)     # bias + ReLU
            taps.append(h)
        return taps

    def forward(self, in0, in1, retPerLayer=False, normalize=False):
        """``in0`` (the prediction, differentiable) and ``in1`` (the reference image) [1, 3, H, W]; returns [1, 1, 1, 1]."""
        if in0.shape[0] != 1 or in1.shape != in0.shape:
            raise RuntimeError("LPIPS (MI355X path): batch 1, equal shapes")
        if normalize:                                    # inputs in [0, 1] -> [-1, 1]  (lpips.py:85-87)
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        s0 = (in0 - self.scaling_layer__shift) / self.scaling_layer__scale
        s1 = (in1 - self.scaling_layer__shift) / self.scaling_layer__scale
        outs0 = self.features(s0.contiguous())
        with torch.no_grad():
            outs1 = self.features(s1.contiguous())
        res = [_LpipsLevel.apply(outs0[k], outs1[k], self._w(f"lin{k}.model.1.weight").reshape(-1)).view(1, 1, 1, 1)
               for k in range(self.L)]
        val = res[0]
        for r in res[1:]:
            val = val + r
        if retPerLayer:
            # the reference accumulates with `val += res[l]` on `val = res[0]` (lpips.py:105-107): its per-layer list comes back
            # with the TOTAL in slot 0; kept for drop-in fidelity
            return val, [val] + res[1:]
        return val


def lpips_named_fill(state, seed: int = 31359):
    """Deterministic synthetic LPIPS parameters as a pure function of (state_dict key, shape): He-scaled normal trunk weights,
    small biases, non-negative channel weights -- what the parity fixture and the GPU test both rebuild (no weights are shipped)."""
    import zlib
    out = {}
    for name, t in state.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        shape = tuple(t.shape)
        if name.startswith("lin"):
            v = torch.rand(shape, generator=g) / shape[1] ** 0.5
        elif name.endswith(".weight"):
            v = torch.randn(shape, generator=g) * (2.0 / (shape[1] * 9)) ** 0.5
        else:
            v = 0.05 * torch.randn(shape, generator=g)
        out[name] = v
    return out
