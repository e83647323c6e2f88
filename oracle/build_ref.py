"""Build recipe for ``oracle/_ref`` -- TEST INFRASTRUCTURE, not product code.

Compiles the reference's *own* CPU implementation of the point->grid max-pool
(``/root/reference/deep_point/src/point_deep.cpp``, the ``point_deep.cpu_kernel``
pybind module of SURVEY.md section 8b) from where it lies, straight into
``oracle/_ref/``.  No reference source is copied into this repository; only the
resulting shared object lands in ``oracle/_ref`` (git-ignored, travels to the GPU
box).  Nothing here runs the reference's own build system (``deep_point/setup.py``).

The deformable-attention CUDA extension of the reference has no CPU build
(``deformattn/src/cpu/ms_deform_attn_cpu.cpp`` only raises) and its CUDA sources
cannot be compiled by this image, so it is "unbuildable here"; its CPU form is the
reference's pure-PyTorch ``ms_deform_attn_core_pytorch`` (imported, not built).
"""
import os
import sys

REF_ROOT = os.environ.get("SMOS_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
MOD_NAME = "smos_ref_point_deep_cpu"


def ref_available():
    return os.path.isfile(os.path.join(REF_ROOT, "deep_point", "src", "point_deep.cpp"))


def built_path():
    p = os.path.join(OUT_DIR, MOD_NAME + ".so")
    return p if os.path.isfile(p) else None


def build(verbose=False):
    """Compile the reference's point_deep.cpp -> oracle/_ref/<MOD_NAME>.so. Returns the path."""
    if not ref_available():
        return built_path()
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    src = os.path.join(REF_ROOT, "deep_point", "src", "point_deep.cpp")
    so = os.path.join(OUT_DIR, MOD_NAME + ".so")
    if os.path.isfile(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    cpp_extension.load(name=MOD_NAME, sources=[src], build_directory=OUT_DIR,
                       extra_cflags=["-O2", "-DVERSION_GE_1_3"], verbose=verbose,
                       is_python_module=True)
    return so


def load():
    """Import the built module (needs torch imported first). Returns None if it was never built."""
    so = built_path()
    if so is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location(MOD_NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
