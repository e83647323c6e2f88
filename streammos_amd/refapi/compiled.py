"""Loader for the COMPILED pybind11 twins of ``point_deep.cuda_kernel`` and ``MultiScaleDeformableAttention``
(csrc/shim/pybind_shims.cpp, built by ``streammos_amd.build.build_pybind_shims``): what a reference maintainer who
keeps the C++ binding layer (deep_point/src/point_deep_cuda.cpp:59-62, deformattn/src/vision.cpp:13-16) would link
against libsmos_hip.so.  ``refapi.install()`` publishes the ctypes-backed Python modules; ``load(name)`` returns the
compiled one for the same name, with the same functions and argument lists."""
import importlib.machinery
import importlib.util
import os

from .. import _lib, build

NAMES = tuple(name for name, _ in build.SHIMS)


def load(name):
    if name not in NAMES:
        raise ValueError("no compiled shim %r (have %s)" % (name, ", ".join(NAMES)))
    path = build.shim_path(name)
    if not os.path.isfile(path):
        raise RuntimeError("%s is not built: run python -m streammos_amd.build --shims" % path)
    _lib.load()     # torch first, then libsmos_hip.so on torch's HIP runtime; the shim's DT_NEEDED entries resolve to both
    spec = importlib.util.spec_from_loader(name, importlib.machinery.ExtensionFileLoader(name, path))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module
