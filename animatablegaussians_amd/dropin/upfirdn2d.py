"""Top-level module ``upfirdn2d`` expected by the reference's ``network/styleunet/upfirdn2d.py:8``
(``import upfirdn2d as upfirdn2d_op``)."""
from animatablegaussians_amd.styleunet_ops import upfirdn2d  # noqa: F401
