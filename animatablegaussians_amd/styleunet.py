"""DualStyleUNet on the MI355X kernels: the pose-conditioned generator the avatar uses three
times per step (``network/avatar.py:34-36``), rebuilt as a flat, table-driven network over

  * ``conv.conv2d`` / ``conv.conv_transpose2d``  -- MFMA fp32 implicit-GEMM kernels (include/ag_conv.h)
  * ``styleunet_ops.upfirdn2d_nchw``             -- FIR resampling (include/ag_styleunet.h)
  * ``styleunet_ops.fused_leaky_relu``           -- bias + leaky-ReLU * sqrt(2)

It is numerically the reference's ``network/styleunet/dual_styleunet.py:DualStyleUNet`` at batch 1 with a single style
vector (the only way ``network/avatar.py:94,107,120`` calls it) and takes the reference's checkpoints:
``load_reference_state_dict`` accepts the ``state_dict`` of the reference module key for key (FIR / Haar kernels are
constants here, every learnable tensor and the fixed noise maps are loaded).  The arithmetic order of the modulated
convolution follows the reference's fused branch (``dual_styleunet.py:254-298``): the modulated, demodulated weight is
formed first and then convolved, so results agree to fp32 summation-order noise.

Small glue (the 512-wide mapping / modulation GEMVs, weight modulation, noise and skip additions, channel concat) is
plain torch tensor algebra on the GPU; all convolutions, FIR filters and activations are this package's HIP kernels.
There is no CPU path.
"""
from __future__ import annotations

import math
import os

import torch

from . import conv as agc
from . import fused_layers
from . import linear_ops
from .styleunet_ops import haar_merge, haar_split, modulate_weight, noise_bias_act, skip_chain, upfirdn2d_nchw

_SQRT2 = 2 ** 0.5
# ConvLayer / StyledConv / ToRGB as one autograd node each (fused_layers.py: same kernels, same order, bit-identical results, a third of
# the autograd nodes) -- with one exception: ToRGB's skip path iwt -> Upsample -> dwt runs as ONE composed kernel by default
# (AG_SKIP_CHAIN, fused_layers.set_skip_chain), which equals the three-kernel chain to 1e-6 of the largest value, not to the bit
# (tests/test_styleunet_ops.py::test_skip_chain_*).  AG_UNFUSED_LAYERS=1 or set_fused_layers(False) runs the per-kernel chain (A/B and
# the equality test).
_FUSED_LAYERS = os.environ.get("AG_UNFUSED_LAYERS") != "1"


def set_fused_layers(on: bool) -> bool:
    global _FUSED_LAYERS
    prev, _FUSED_LAYERS = _FUSED_LAYERS, bool(on)
    return prev


def _fir(taps, gain=1.0):
    k = torch.tensor(taps, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    k /= k.sum()
    return k * gain


def _haar():
    a = 1 / (2 ** 0.5)
    lo, hi = torch.tensor([[a, a]]), torch.tensor([[-a, a]])
    # dual_styleunet.py:374-384
    return {"ll": lo.T * lo, "lh": hi.T * lo, "hl": lo.T * hi, "hh": hi.T * hi}


class _Holder(torch.nn.Module):
    """Bare container: gives a tensor the reference's dotted state_dict path.  Never called."""


class DualStyleUNet(torch.nn.Module):
    """Encoder (pose map -> 6 feature levels) + two style-modulated decoders (front / back maps)."""

    def __init__(self, inp_size=512, inp_ch=3, out_ch=3, out_size=1024, style_dim=512, n_mlp=2, middle_size=8,
                 channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01):
        super().__init__()
        if tuple(blur_kernel) != (1, 3, 3, 1):
            raise ValueError("only the [1,3,3,1] blur kernel of the product is supported")
        cm = channel_multiplier
        ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm, 1024: 16 * cm}
        self.inp_size, self.inp_ch, self.out_ch, self.out_size = inp_size, inp_ch, out_ch, out_size
        self.style_dim, self.n_mlp, self.lr_mlp = style_dim, n_mlp, lr_mlp
        log_in, log_mid, log_out = int(math.log2(inp_size)), int(math.log2(middle_size)), int(math.log2(out_size)) - 1
        self._slots = {}                    # reference state_dict key -> (holder module, leaf name)
        self._learnable = []                # keys of the parameters, in named_parameters() order

        # ---- every tensor of the reference module's state_dict, under the reference's names AND in its order ------------
        # The tensors live in a tree of bare holder modules shaped like the reference's module tree (dual_styleunet.py:
        # 636-760), declared in the reference's registration order, so that ``state_dict()`` / ``load_state_dict()`` /
        # ``parameters()`` are interchangeable with the reference's: main_avatar.py:777-813 saves and strictly loads
        # ``avatar_net.state_dict()``, and Adam's state is indexed by parameter order.  The holders never run; the forward
        # below is table-driven.  Constant FIR / Haar kernels are stored (the reference stores them) but not read.
        for top in ("style", "comb_convs", "from_rgbs", "cond_convs", "conv_in", "convs1", "convs2", "to_rgbs1", "to_rgbs2",
                    "noises", "iwt"):
            self.add_module(top, _Holder())
        blur, blur_up = _fir(blur_kernel), _fir(blur_kernel, 4.0)
        for i in range(n_mlp):
            self._param(f"style.{i + 1}.weight", torch.randn(style_dim, style_dim) / lr_mlp)
            self._param(f"style.{i + 1}.bias", torch.zeros(style_dim))

        c0 = ch[inp_size // 2]
        self._conv_layer_params("comb_convs.0", 2 * c0, c0, 3)
        self.enc = []                       # (level index, in channels, out channels)
        cin = c0
        for n, i in enumerate(range(log_in - 2, log_mid - 1, -1)):
            cout = ch[2 ** i]
            self._const(f"from_rgbs.{n}.downsample.kernel", blur)
            self._conv_layer_params(f"from_rgbs.{n}.conv", inp_ch, cin, 1)
            self._conv_layer_params(f"cond_convs.{n}.conv1", cin, cin, 3)
            self._conv_layer_params(f"cond_convs.{n}.conv2", cin, cout, 3, downsample=True)
            self._conv_layer_params(f"comb_convs.{n + 1}", cout * (2 if i > log_mid else 1), cout, 3)
            self.enc.append((n, cin, cout))
            cin = cout
        self._conv_layer_params("conv_in", inp_ch, c0, 3, downsample=True)
        self.n_comb = len(self.enc) + 1

        self.num_layers = 2 * (log_out - log_mid)
        for layer in range(self.num_layers):
            res = 2 ** ((layer + 2 * (log_mid + 1)) // 2)
            self._const(f"noises.noise_{layer}", torch.randn(1, 1, res, res))
        self.dec = []                       # (stage, in channels, out channels)
        cin = ch[middle_size]
        for n, i in enumerate(range(log_mid + 1, log_out + 1)):
            cout = ch[2 ** i]
            for b in (1, 2):
                self._styled_conv_params(f"convs{b}.{2 * n}", cin, cout, blur_up)
                self._styled_conv_params(f"convs{b}.{2 * n + 1}", cout, cout, None)
                self._param(f"to_rgbs{b}.{n}.bias", torch.zeros(1, out_ch * 4, 1, 1))
                self._const(f"to_rgbs{b}.{n}.upsample.kernel", blur_up)
                self._haar_consts(f"to_rgbs{b}.{n}.iwt", inverse=True)
                self._haar_consts(f"to_rgbs{b}.{n}.dwt", inverse=False)
                self._param(f"to_rgbs{b}.{n}.conv.weight", torch.randn(1, out_ch * 4, cout, 1, 1))
                self._param(f"to_rgbs{b}.{n}.conv.modulation.weight", torch.randn(cout, style_dim))
                self._param(f"to_rgbs{b}.{n}.conv.modulation.bias", torch.ones(cout))
            self.dec.append((n, cin, cout))
            cin = cout
        self._haar_consts("iwt", inverse=True)
        self.n_latent = log_out * 2 - (log_mid * 2 - 1) + 1
        self._learnable = [k for k, _ in self.named_parameters()]

        # ---- the filters the forward reads (non-persistent: not part of the checkpoint) ---------------------------
        self.register_buffer("_k_blur", blur.clone(), persistent=False)            # Blur before stride-2 conv, Downsample
        self.register_buffer("_k_blur_up", blur_up.clone(), persistent=False)      # Blur after conv_transpose, Upsample

    # ---- parameter bookkeeping ------------------------------------------------------------------------------------
    def _declare(self, ref_name, value, learnable):
        *path, leaf = ref_name.split(".")
        m = self
        for part in path:
            if part not in m._modules:
                m.add_module(part, _Holder())
            m = m._modules[part]
        if learnable:
            m.register_parameter(leaf, torch.nn.Parameter(value))
        else:
            m.register_buffer(leaf, value.clone())
        self._slots[ref_name] = (m, leaf)

    def _param(self, ref_name, value):
        self._declare(ref_name, value, True)

    def _const(self, ref_name, value):
        self._declare(ref_name, value, False)

    def _haar_consts(self, prefix, inverse):                                  # dual_styleunet.py:387-416
        for name, k in _haar().items():
            self._const(f"{prefix}.{name}", -k if inverse and name in ("lh", "hl") else k)

    def _p(self, ref_name):
        m, leaf = self._slots[ref_name]
        return getattr(m, leaf)

    def _conv_layer_params(self, prefix, cin, cout, k, downsample=False):
        # ConvLayer = [Blur] + EqualConv2d(bias=False) + FusedLeakyReLU(bias)   (dual_styleunet.py:326-371)
        base = 1 if downsample else 0
        if downsample:
            self._const(f"{prefix}.0.kernel", _fir((1, 3, 3, 1)))
        self._param(f"{prefix}.{base}.weight", torch.randn(cout, cin, k, k))
        self._param(f"{prefix}.{base + 1}.bias", torch.zeros(cout))

    def _styled_conv_params(self, prefix, cin, cout, blur_up):
        self._param(f"{prefix}.conv.weight", torch.randn(1, cout, cin, 3, 3))
        if blur_up is not None:                                               # upsampling ModulatedConv2d owns a Blur (:186-193)
            self._const(f"{prefix}.conv.blur.kernel", blur_up)
        self._param(f"{prefix}.conv.modulation.weight", torch.randn(cin, self.style_dim))
        self._param(f"{prefix}.conv.modulation.bias", torch.ones(cin))
        self._param(f"{prefix}.noise.weight", torch.zeros(1))
        self._param(f"{prefix}.activate.bias", torch.zeros(cout))

    _CONST_SUFFIX = (".kernel", ".ll", ".lh", ".hl", ".hh")

    def reference_state_dict(self):
        """Learnable tensors + noise maps under the reference module's state_dict keys (``state_dict()`` minus the constant
        FIR / Haar kernels)."""
        return {k: v.detach() for k, v in self.state_dict().items() if not k.endswith(self._CONST_SUFFIX)}

    @torch.no_grad()
    def load_reference_state_dict(self, sd, strict=True):
        """Load the ``state_dict`` of the reference's DualStyleUNet, with or without its constant buffers (``*.kernel``,
        ``*.ll`` ...: they are the same constants here).  Anything else unknown or missing raises when ``strict``.
        ``load_state_dict`` itself is the reference's (every key required when strict)."""
        own = self.state_dict()
        full = {k: v for k, v in own.items() if k.endswith(self._CONST_SUFFIX)}
        for key, value in sd.items():
            if key in own:
                if tuple(own[key].shape) != tuple(value.shape):
                    raise RuntimeError(f"{key}: checkpoint shape {tuple(value.shape)} != {tuple(own[key].shape)}")
                full[key] = value
            elif strict:
                raise RuntimeError(f"unexpected key in reference state_dict: {key}")
        missing = [k for k in own if k not in full]
        if strict and missing:
            raise RuntimeError(f"missing keys in reference state_dict: {missing[:5]}{' ...' if len(missing) > 5 else ''}")
        self.load_state_dict(full, strict=False)

    # ---- building blocks ------------------------------------------------------------------------------------------
    def _conv_layer(self, x, prefix, downsample=False):
        base = 1 if downsample else 0
        w = self._p(f"{prefix}.{base}.weight")
        k = w.shape[-1]
        scale = 1 / math.sqrt(w.shape[1] * k * k)                        # EqualConv2d: conv(x, weight * scale) (:100-117); the
        bias = self._p(f"{prefix}.{base + 1}.bias")                      # product is formed inside the weight re-pack
        if _FUSED_LAYERS:                                                # one autograd node for [blur +] conv + bias / activation
            return fused_layers.conv_layer(x, w, bias, self._k_blur if downsample else None, scale, downsample)
        if downsample:
            x = upfirdn2d_nchw(x, self._k_blur, pad=(2, 2))              # p = (4 - 2) + (k - 1), k = 3 (:339-345)
            x = agc.conv2d(x, w, stride=2, padding=0, weight_scale=scale)
        else:
            x = agc.conv2d(x, w, stride=1, padding=k // 2, weight_scale=scale)
        return noise_bias_act(x, None, None, bias)

    def _stage_styles(self, branch, stages, w_latent):
        """The modulation vectors of every modulated convolution of ``stages`` (two StyledConvs and one ToRGB each) from ONE native launch:
        EqualLinear(w_latent) with lr_mul 1 (:152-155) is ``w_latent @ (W * c)^T + b`` per convolution; the 3 * len(stages) layers are jobs of one
        ``ag_equal_linear_forward`` call that reads the parameter tensors where they lie (``c`` is the call's alpha: c * (w_latent @ W^T) + b, the same
        value up to the rounding of the scaling) and, backward, writes every parameter's gradient where autograd wants it.  Rounds 3-5 stacked the
        weights (a 12.6-MB concatenation per decoder branch and step) for one hipBLASLt GEMV forward and two backward: ~0.6 ms of a 33-ms
        training step (profiles/r05b_glue_v1.txt).  Returns {convolution prefix: [B, C] style}."""
        prefixes = [p for n in stages for p in (f"convs{branch}.{2 * n}.conv", f"convs{branch}.{2 * n + 1}.conv", f"to_rgbs{branch}.{n}.conv")]
        ws = [self._p(f"{p}.modulation.weight") for p in prefixes]
        bs = [self._p(f"{p}.modulation.bias") for p in prefixes]
        out = {}
        for c0 in range(0, len(prefixes), linear_ops.MAX_JOBS):
            c1 = min(len(prefixes), c0 + linear_ops.MAX_JOBS)
            out.update(zip(prefixes[c0:c1], linear_ops.equal_linear_group(w_latent, ws[c0:c1], bs[c0:c1])))
        return out

    def _modulated_weight(self, prefix, styles, demodulate, transposed=False):
        w = self._p(f"{prefix}.weight")                                  # [1, Cout, Cin, k, k]
        k = w.shape[-1]
        # (scale * W) * style, demodulated (:254-259), in one kernel; [Cout, Cin, k, k] or transposed for conv_transpose2d
        return modulate_weight(w, styles[prefix], 1 / math.sqrt(w.shape[2] * k * k), demodulate, transposed)

    def _styled_conv(self, x, prefix, w_latent, noise, upsample):
        if _FUSED_LAYERS:
            w = self._p(f"{prefix}.conv.weight")
            k = w.shape[-1]
            if noise is None:
                hw = (2 * x.shape[2], 2 * x.shape[3]) if upsample else (x.shape[2], x.shape[3])
                noise = torch.randn(1, 1, hw[0], hw[1], device=x.device, dtype=x.dtype)
            return fused_layers.styled_conv(x, w, w_latent[f"{prefix}.conv"], noise, self._p(f"{prefix}.noise.weight"),
                                            self._p(f"{prefix}.activate.bias"), self._k_blur_up if upsample else None,
                                            1 / math.sqrt(w.shape[2] * k * k), upsample)
        weight = self._modulated_weight(f"{prefix}.conv", w_latent, True, transposed=upsample)
        if upsample:
            x = agc.conv_transpose2d(x, weight, stride=2, padding=0)
            x = upfirdn2d_nchw(x, self._k_blur_up, pad=(1, 1))           # p = (4 - 2) - (3 - 1) = 0 -> pad (1, 1) (:188-193)
        else:
            x = agc.conv2d(x, weight, stride=1, padding=1)
        if noise is None:
            noise = torch.randn(1, 1, x.shape[2], x.shape[3], device=x.device, dtype=x.dtype)
        # NoiseInjection (:301-311) + FusedLeakyReLU (:596) in one pass
        return noise_bias_act(x, noise, self._p(f"{prefix}.noise.weight"), self._p(f"{prefix}.activate.bias"))

    def _haar_split(self, x):                                             # HaarTransform (:387-403), one kernel
        return haar_split(x)

    def _haar_merge(self, x):                                             # InverseHaarTransform (:406-425), one kernel
        return haar_merge(x)

    def _to_rgb(self, x, prefix, w_latent, skip):
        if _FUSED_LAYERS:
            w = self._p(f"{prefix}.conv.weight")
            return fused_layers.to_rgb(x, w, w_latent[f"{prefix}.conv"], self._p(f"{prefix}.bias").reshape(-1), skip, self._k_blur_up,
                                       1 / math.sqrt(w.shape[2] * w.shape[-1] * w.shape[-1]))
        weight = self._modulated_weight(f"{prefix}.conv", w_latent, False)
        out = agc.conv2d(x, weight, bias=self._p(f"{prefix}.bias").reshape(-1), stride=1, padding=0)   # bias in the conv epilogue
        if skip is not None:
            if fused_layers._SKIP_CHAIN:
                out = out + skip_chain(skip, self._k_blur_up)            # iwt -> Upsample -> dwt as one kernel
            else:
                s = self._haar_merge(skip)
                s = upfirdn2d_nchw(s, self._k_blur_up, up=2, pad=(2, 1))     # Upsample (:32-50)
                out = out + self._haar_split(s)
        return out

    def get_latent(self, z):
        """Mapping network: PixelNorm + n_mlp x EqualLinear(lr_mul, fused leaky ReLU)  (:594-610)."""
        return latents_of([self], [z])[0]

    def encode(self, condition_img):
        """Pose map [1, inp_ch, S, S] -> feature levels, finest first  (:854-864)."""
        img = condition_img
        out = self._conv_layer(img, "conv_in", downsample=True)
        levels = [out]
        for n, _, _ in self.enc:
            img = upfirdn2d_nchw(img, self._k_blur, down=2, pad=(1, 1))                  # Downsample (:53-71)
            out = self._conv_layer(img, f"from_rgbs.{n}.conv") + out                      # FromRGB, use_wt=False (:455-468)
            out = self._conv_layer(out, f"cond_convs.{n}.conv1")
            out = self._conv_layer(out, f"cond_convs.{n}.conv2", downsample=True)
            levels.append(out)
        return levels

    VIEW_STAGE = 4      # the view-direction feature is added to the stage-4 activations (i == 8 at :881-883)

    def _decode_stages(self, branch, levels, w_latent, noise, out, skip, stages):
        stages = list(stages)
        if not stages:
            return out, skip
        w_latent = self._stage_styles(branch, stages, w_latent)          # from here on: {prefix: style}
        for n in stages:
            if n == 0:
                out = self._conv_layer(levels[-1], f"comb_convs.{self.n_comb - 1}")
            elif n < self.n_comb:
                out = self._conv_layer(torch.cat([out, levels[-1 - n]], 1), f"comb_convs.{self.n_comb - 1 - n}")
            out = self._styled_conv(out, f"convs{branch}.{2 * n}", w_latent, noise[2 * n], True)
            out = self._styled_conv(out, f"convs{branch}.{2 * n + 1}", w_latent, noise[2 * n + 1], False)
            skip = self._to_rgb(out, f"to_rgbs{branch}.{n}", w_latent, skip)
        return out, skip

    def decode_shared(self, branch, levels, w_latent, noise):
        """View-independent part of one decoder: stages 0 .. VIEW_STAGE  (77 % of the decoder's FLOPs)."""
        return self._decode_stages(branch, levels, w_latent, noise, None, None, range(0, min(self.VIEW_STAGE + 1, len(self.dec))))

    def decode_view(self, branch, levels, w_latent, noise, out, skip, view_feature=None):
        """View-dependent tail of one decoder: add the view feature, run the remaining stages, inverse wavelet."""
        if view_feature is not None and len(self.dec) > self.VIEW_STAGE:
            if view_feature.shape[-2:] != out.shape[-2:]:
                view_feature = linear_ops.bilinear_resize(view_feature, out.shape[-2:])       # F.interpolate(..., mode="bilinear") (:881-883)
            out = out + view_feature
        out, skip = self._decode_stages(branch, levels, w_latent, noise, out, skip, range(self.VIEW_STAGE + 1, len(self.dec)))
        return self._haar_merge(skip)

    def decode(self, branch, levels, w_latent, noise, view_feature=None):
        """One decoder (branch 1 = front map, 2 = back map)  (:869-905)."""
        out, skip = self.decode_shared(branch, levels, w_latent, noise)
        return self.decode_view(branch, levels, w_latent, noise, out, skip, view_feature)

    # The two decoders are independent given the encoder levels: branch 2 runs on a side HIP stream so that its small
    # layers (a few workgroups each at 16^2 .. 64^2) fill the CUs branch 1 leaves idle, forward and -- because autograd replays
    # every node on the stream it was recorded on -- backward.  AG_SINGLE_STREAM=1 disables it (A/B measurements, captures).
    def _two_branches(self, fn, shared=()):
        """``shared``: tensors allocated on the current stream that branch 2 reads on the side stream (encoder levels, latent,
        shared decoder state): they are registered with the side stream so the caching allocator does not hand their memory
        out again while side-stream kernels -- forward now, backward later -- may still be reading it."""
        import os
        cur = torch.cuda.current_stream()
        capturing = torch.cuda.is_current_stream_capturing()
        if os.environ.get("AG_SINGLE_STREAM") == "1" or (capturing and os.environ.get("AG_CAPTURE_BRANCHES") != "1"):
            return [fn(1), fn(2)]
        if getattr(self, "_side_stream", None) is None or self._side_stream.device != cur.device:
            self._side_stream = torch.cuda.Stream(cur.device)
            # gradients of branch-2 parameters are produced on the side stream and accumulated on the parameters' own
            # (default) stream; autograd synchronises the two, the mismatch is intended
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
        side = self._side_stream
        side.wait_stream(cur)
        for t in shared:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(side)
        with torch.cuda.stream(side):
            r2 = fn(2)
        r1 = fn(1)
        cur.wait_stream(side)
        for t in (r2 if isinstance(r2, (tuple, list)) else (r2,)):
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)          # allocated on the side stream, consumed (and later freed) on the main one
        return [r1, r2]

    def _latent_and_noise(self, styles, input_is_latent, noise, randomize_noise):
        w_latent = styles[0] if input_is_latent else self.get_latent(styles[0])
        if w_latent.dim() == 3:
            w_latent = w_latent[:, 0]
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else \
                [self._p(f"noises.noise_{i}") for i in range(self.num_layers)]
        return w_latent, noise

    def forward_views(self, styles, condition_img, view_features, input_is_latent=False, noise=None, randomize_noise=True):
        """Several evaluations that differ only in ``(view_feature1, view_feature2)`` -- the colour network seen from V
        cameras of the same pose.  Encoder and decoder stages 0..4 are computed once (and, under autograd, back-propagated
        once with the summed gradient); only stage 5 runs per view.  Returns a list of V image tensors, each equal to
        ``forward(..., view_feature1=f1, view_feature2=f2)[0]``."""
        if randomize_noise and noise is None:
            raise RuntimeError("forward_views shares activations between views: use fixed noise (randomize_noise=False)")
        w_latent, noise = self._latent_and_noise(styles, input_is_latent, noise, randomize_noise)
        levels = self.encode(condition_img)
        shared = self._two_branches(lambda b: self.decode_shared(b, levels, w_latent, noise), shared=[*levels, w_latent])
        images = []
        for f1, f2 in view_features:
            parts = self._two_branches(lambda b: self.decode_view(b, levels, w_latent, noise, shared[b - 1][0], shared[b - 1][1],
                                                                  f1 if b == 1 else f2), shared=[f2])
            images.append(torch.cat(parts, 1))
        return images

    def forward(self, styles, condition_img, cond=None, return_latents=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True, view_feature1=None,
                view_feature2=None):
        """Same call as the reference (:616-911); returns ``(images [1, 2*out_ch, out_size, out_size], latent|None)``."""
        if cond is not None or truncation < 1 or len(styles) != 1 or inject_index is not None:
            raise RuntimeError("DualStyleUNet (MI355X path): one style vector, no conditioning / truncation / style mixing")
        if condition_img.shape[0] != 1 or styles[0].shape[0] != 1:
            raise RuntimeError("DualStyleUNet (MI355X path): batch 1")
        if not condition_img.is_cuda:
            raise RuntimeError("DualStyleUNet (MI355X path) runs on the GPU only")
        w_latent, noise = self._latent_and_noise(styles, input_is_latent, noise, randomize_noise)
        levels = self.encode(condition_img)
        image1, image2 = self._two_branches(lambda b: self.decode(b, levels, w_latent, noise, view_feature1 if b == 1 else view_feature2),
                                            shared=[*levels, w_latent, view_feature2])
        images = torch.cat([image1, image2], 1)
        if return_latents:
            return images, w_latent.unsqueeze(1).repeat(1, self.n_latent, 1)
        return images, None


def latents_of(nets, zs):
    """The mapping networks (PixelNorm + n_mlp x EqualLinear(lr_mul, fused leaky ReLU), dual_styleunet.py:594-610) of several networks: layer i of
    ALL of them is one native launch (include/ag_linear.h; the PixelNorm rides in the first).  Returns one latent [B, style_dim] per network."""
    n0 = nets[0]
    if any((n.n_mlp, n.style_dim, n.lr_mlp) != (n0.n_mlp, n0.style_dim, n0.lr_mlp) for n in nets):
        return [latents_of([n], [z])[0] for n, z in zip(nets, zs)]
    xs = list(zs)
    norm_inside = not any(z.requires_grad for z in zs)          # (the native PixelNorm option has no input gradient: a learnable z takes the torch form)
    if not norm_inside:
        xs = [z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8) for z in zs]
    for i in range(n0.n_mlp):
        xs = list(linear_ops.equal_linear_group(xs, [n._p(f"style.{i + 1}.weight") for n in nets], [n._p(f"style.{i + 1}.bias") for n in nets],
                                                lr_mul=n0.lr_mlp, activation=True, normalize_input=norm_inside and i == 0))
    if n0.n_mlp == 0 and norm_inside:
        xs = [z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8) for z in zs]
    return xs
