__smos_refapi__ = True
from . import StreamMOS, StreamMOS_seg  # noqa: E402,F401
