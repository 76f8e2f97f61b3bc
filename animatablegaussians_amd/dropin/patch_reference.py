"""Route the reference's convolution call sites to the MFMA kernels without editing the reference tree.

``network/styleunet/conv2d_gradfix.py`` is imported relatively (``from . import conv2d_gradfix``,
``dual_styleunet.py:10``), so it cannot be shadowed through ``PYTHONPATH`` like the three extension modules; its two
public functions are replaced in place instead.  Call once before the networks run, e.g. at the top of
``main_avatar.py``::

    from patch_reference import apply; apply()

At batch 1 the reference's grouped calls (``groups = batch``, ``dual_styleunet.py:266-298``) are ordinary convolutions;
``conv.conv2d`` / ``conv.conv_transpose2d`` take the same arguments and raise on anything outside the product's
configurations (no fallback)."""


def apply():
    import importlib

    from animatablegaussians_amd import conv as agc
    g = importlib.import_module("network.styleunet.conv2d_gradfix")

    def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1 and groups == input.shape[0] == 1:
            groups = 1
        return agc.conv2d(input, weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)

    def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
        return agc.conv_transpose2d(input, weight, bias=bias, stride=stride, padding=padding,
                                    output_padding=output_padding, groups=groups, dilation=dilation)

    g.conv2d, g.conv_transpose2d = conv2d, conv_transpose2d
    return g
