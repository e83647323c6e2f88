__smos_refapi__ = True
from . import StreamMOS  # noqa: E402,F401
