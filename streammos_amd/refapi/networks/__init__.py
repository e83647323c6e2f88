__smos_refapi__ = True
from . import backbone, multi_view_encoder  # noqa: E402,F401
