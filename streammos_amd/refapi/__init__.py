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
    """Register the mirror packages -- and every sub-module of them -- in ``sys.modules`` under the
    reference's top-level names.  Sub-modules are imported under their real (streammos_amd.refapi.*) names
    first and then aliased, so their relative imports keep working whichever name they are reached by."""
    import pkgutil
    for name in _TOP_LEVEL:
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__smos_refapi__", False) and not force:
            raise RuntimeError("refapi.install(): a different module named %r is already imported (%s); "
                               "install() must run before the reference's own packages are imported"
                               % (name, getattr(mod, "__file__", "?")))
    here = sys.modules[__name__]
    for info in pkgutil.walk_packages(here.__path__, prefix=__name__ + "."):
        importlib.import_module(info.name)
    prefix = __name__ + "."
    for full, sub in list(sys.modules.items()):
        if sub is not None and full.startswith(prefix) and full[len(prefix):].split(".")[0] in _TOP_LEVEL:
            sys.modules[full[len(prefix):]] = sub
    return [sys.modules[n] for n in _TOP_LEVEL]
