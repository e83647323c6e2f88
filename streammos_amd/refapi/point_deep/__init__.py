"""``point_deep``: the two extension modules of the reference's deep_point package
(deep_point/setup.py:4-19), re-implemented over the C ABI."""
__smos_refapi__ = True
from . import cpu_kernel, cuda_kernel  # noqa: E402,F401
