"""Import-path shim: the reference exposes MSDeformAttn from deformattn/modules/ms_deform_attn.py."""
from .._msda import MSDeformAttn, MSDeformAttnFunction  # noqa: F401
