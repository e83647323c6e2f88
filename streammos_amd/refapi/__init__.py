"""Host-side mirror of the reference's operator / module interface for the inference path.

The sub-packages carry the reference's own import names (``deep_point``, ``point_deep``,
``MultiScaleDeformableAttention``, ``deformattn``, ``networks``, ``models``, ``config``) because those
names *are* the interface.  ``install()`` publishes them under those top-level
names so that the reference's entry scripts (``val_StreamMOS.py`` does ``from models import *`` and
``eval("StreamMOS.AttNet")``) pick up this implementation unchanged -- see INTEGRATION.md.
"""
import importlib
import sys

_TOP_LEVEL = ("point_deep", "deep_point", "MultiScaleDeformableAttention", "deformattn", "networks", "models", "config")


def install(force=False):
    """Register the mirror packages in ``sys.modules`` under the reference's top-level names."""
    for name in _TOP_LEVEL:
        if name in sys.modules and not force:
            mod = sys.modules[name]
            if getattr(mod, "__smos_refapi__", False):
                continue
            raise RuntimeError("refapi.install(): a different module named %r is already imported (%s); "
                               "install() must run before the reference's own packages are imported"
                               % (name, getattr(mod, "__file__", "?")))
        mod = importlib.import_module(__name__ + "." + name)
        sys.modules[name] = mod
        # publish already-imported submodules too (e.g. point_deep.cuda_kernel, models.StreamMOS)
        prefix = __name__ + "." + name + "."
        for full, sub in list(sys.modules.items()):
            if full.startswith(prefix) and sub is not None:
                sys.modules[name + "." + full[len(prefix):]] = sub
    return [sys.modules[n] for n in _TOP_LEVEL]
