"""Import harness for the Python reference -- TEST INFRASTRUCTURE (this container only).

Used by ``tests/golden/make_golden.py`` and the ``needs_reference`` tests to import
``/root/reference`` on CPU so that golden vectors can be generated and the oracle's
restatement can be validated against the real thing.  The reference never travels to
the GPU box; every caller must cope with ``reference_available() == False``.

Recipe:
  * ``point_deep.cpu_kernel``  -> the reference's own point_deep.cpp, built by build_ref.py
  * ``point_deep.cuda_kernel`` -> empty module (imported by deep_point/__init__.py:5, never
    called for CPU tensors)
  * ``MultiScaleDeformableAttention`` -> module whose ms_deform_attn_forward is the
    reference's own ms_deform_attn_core_pytorch (deformattn/functions/ms_deform_attn_func.py:41-61;
    the equivalence of the two is what deformattn/test.py:31-60 asserts) and whose
    ms_deform_attn_backward is its autograd derivative (deformattn/test.py:63-78, gradcheck)
  * ``cv2`` -> empty module (imported at utils/boundary_loss.py:4, unused on this path)
"""
import ast
import os
import sys
import types

from . import build_ref

REF_ROOT = build_ref.REF_ROOT
_state = {}


def reference_available():
    return build_ref.ref_available()


def import_reference():
    """Returns a namespace with the reference's modules (models, networks, deep_point, ...)."""
    if "ns" in _state:
        return _state["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import torch  # noqa: F401
    build_ref.build()
    cpu_kernel = build_ref.load()

    pkg = types.ModuleType("point_deep")
    pkg.__path__ = []
    cuda_stub = types.ModuleType("point_deep.cuda_kernel")
    sys.modules["point_deep"] = pkg
    sys.modules["point_deep.cpu_kernel"] = cpu_kernel
    sys.modules["point_deep.cuda_kernel"] = cuda_stub
    pkg.cpu_kernel = cpu_kernel
    pkg.cuda_kernel = cuda_stub
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")

    msda = types.ModuleType("MultiScaleDeformableAttention")
    sys.modules["MultiScaleDeformableAttention"] = msda

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # our own package mirrors some of these names; make sure the reference's win here
    for name in ("deep_point", "deformattn", "networks", "models", "config", "utils", "datasets"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(REF_ROOT):
            del sys.modules[name]

    from deformattn.functions import ms_deform_attn_func as _f

    def _fwd(value, shapes, lsi, loc, w, step):
        n, lq = loc.shape[0], loc.shape[1]
        return _f.ms_deform_attn_core_pytorch(value, shapes, loc, w).reshape(n, lq, -1)

    msda.ms_deform_attn_forward = _fwd

    def _bwd(value, shapes, lsi, loc, w, grad_output, step):
        # the CUDA backward (deformattn/src/cuda/ms_deform_im2col_cuda.cuh:301-920) as the autograd derivative of the
        # reference's own ms_deform_attn_core_pytorch -- the pair deformattn/test.py:63-78 holds together with gradcheck
        import torch
        n, lq = loc.shape[0], loc.shape[1]
        with torch.enable_grad():
            v, l, a = (t.detach().requires_grad_(True) for t in (value, loc, w))
            y = _f.ms_deform_attn_core_pytorch(v, shapes, l, a).reshape(n, lq, -1)
            return torch.autograd.grad(y, (v, l, a), grad_output)

    msda.ms_deform_attn_backward = _bwd

    import deep_point
    import networks.backbone
    import networks.multi_view_encoder
    import models.StreamMOS
    import config.StreamMOS
    import utils.transforms
    import deformattn.modules

    ns = types.SimpleNamespace(
        deep_point=deep_point, backbone=networks.backbone, mve=networks.multi_view_encoder,
        StreamMOS=models.StreamMOS, config=config.StreamMOS, transforms=utils.transforms,
        msda_func=_f, msda_modules=deformattn.modules, cpu_kernel=cpu_kernel,
        voting=_extract_voting_functions())
    _state["ns"] = ns
    return ns


def _extract_voting_functions():
    """voxel_voting.py runs its whole pipeline at import (argparse + dataset walk at module
    level, voxel_voting.py:128-252), so only its pure functions (:13-91) are pulled out of the
    source text and executed with torch/numpy in scope."""
    import numpy as np
    import torch
    path = os.path.join(REF_ROOT, "voxel_voting.py")
    tree = ast.parse(open(path).read(), path)
    wanted = {"map", "Quantize", "determine_voxel_labels", "get_point_labels_from_voxel_labels"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    mod = types.ModuleType("smos_ref_voxel_voting_functions")
    mod.__dict__.update(np=np, torch=torch)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), mod.__dict__)
    return mod


def extract_instance_voting(namespace_overrides):
    """voxel_instance_voting.py also runs at import (argparse, yaml.load, a dataset walk and a process pool at module
    level, :272-352), so its functions (:17-270: map, min_bounding_box_3d, in_hull, get_point_labels_from_voxel_labels,
    determine_voxel_labels, Quantize, get_data, cluster, post_processing) are pulled out of the source text and executed in
    a module whose globals the caller provides (what the script's module level would have defined: files, poses_list,
    data_path, pred_path, pred_bf_path, save_path, task_cfg, crop_to_fov, frames_num_max, utils).  Only the degenerate-hull
    `except` branch of in_hull touches names that no longer exist (np.bool, scipy.spatial.qhull); nothing else is altered."""
    import copy
    import numpy as np
    import scipy
    import torch
    from scipy.spatial import ConvexHull, Delaunay
    from sklearn.cluster import DBSCAN
    path = os.path.join(REF_ROOT, "voxel_instance_voting.py")
    tree = ast.parse(open(path).read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    mod = types.ModuleType("smos_ref_voxel_instance_voting_functions")
    mod.__dict__.update(np=np, torch=torch, scipy=scipy, copy=copy, os=os, DBSCAN=DBSCAN, ConvexHull=ConvexHull, Delaunay=Delaunay)
    mod.__dict__.update(namespace_overrides)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), mod.__dict__)
    return mod
