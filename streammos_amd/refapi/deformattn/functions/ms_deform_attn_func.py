"""Import-path shim: the reference exposes these names from deformattn/functions/ms_deform_attn_func.py."""
from .._msda import MSDeformAttnFunction, ms_deform_attn_core_pytorch  # noqa: F401
