"""Top-level module ``fused`` expected by the reference's ``network/styleunet/fused_act.py:30`` (``import fused``)."""
from animatablegaussians_amd.styleunet_ops import fused_bias_act  # noqa: F401
