"""Tensor-level entry points of the HIP kernels (torch tensors in, C ABI underneath).

torch is used for device memory and streams only: every function launches on
``torch.cuda.current_stream`` of the tensors' device and returns without synchronising.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, profiling

_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.float64: 2}
_flag_ws = {}
_ws_namespace = 0


def set_workspace_namespace(tag):
    """Per-stream scratch (``_stream_workspace``, ``_flag``) is keyed by (device, HIP stream, namespace).  Work that is
    captured on ONE stream but replayed concurrently on several (the TTA groups of StreamRunner(graph=True, split=k) all use
    torch's capture stream) selects a namespace per group while it is being captured, so that the groups' graphs do not
    share scratch.  Returns the previous namespace."""
    global _ws_namespace
    prev, _ws_namespace = _ws_namespace, tag
    return prev


# The two private torch._C entry points behind torch.cuda.current_stream(...).cuda_stream / torch.cuda.current_device(),
# resolved once; a torch build without them gets the public (slower: ~4 us + ~3 us per launch) calls instead.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
if _raw_stream is None:
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream
if _raw_device is None:
    _raw_device = torch.cuda.current_device


def _dev_index(device):
    return device.index if device.index is not None else _raw_device()


def _stream(t):
    """raw handle of torch's current HIP stream on t's device (the C call behind torch.cuda.current_stream(...).cuda_stream:
    the wrapper objects cost ~4 us per launch on the host)."""
    return _raw_stream(_dev_index(t.device))


class _NoGuard:
    """`with` target for a launch on the device that is already current: nothing to switch, nothing to restore."""
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """Device guard for a launch: torch.cuda.device(...) only where the tensor's device is not the current one (the usual
    one-process-per-GPU case never switches; the guard object and its two device queries cost ~3 us per launch)."""
    return _NO_GUARD if device.index is None or _raw_device() == device.index else torch.cuda.device(device)


def _require_cuda(name, *tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("%s: expected a tensor in GPU memory, got device=%s (no CPU path here)" % (name, t.device))


def _flag(device):
    """4-byte "saw a negative feature" scratch of the scatter; one per (device, stream) so that launches on different
    HIP streams never share it."""
    key = (device, _raw_stream(_dev_index(device)), _ws_namespace)
    ws = _flag_ws.get(key)
    if ws is None:
        ws = torch.zeros(4, dtype=torch.int32, device=device)
        _flag_ws[key] = ws
    return ws


def _dtype_code(name, t):
    code = _DTYPE_CODE.get(t.dtype)
    if code is None:
        raise RuntimeError("%s: unsupported dtype %s" % (name, t.dtype))
    return code


def voxel_maxpool_fwd(feat, ind, out, out_size, scale, voxel_max_idx=None):
    """feat [BS,C,N(,1)] any strides, ind [BS,N,D(,1)] contiguous, out [BS,C,*out_size] zero-filled."""
    _require_cuda("voxel_maxpool_fwd", feat, ind, out, voxel_max_idx)
    if feat.dtype != ind.dtype or feat.dtype != out.dtype:
        raise RuntimeError("voxel_maxpool_fwd: feat/ind/out dtypes differ (%s, %s, %s)" % (feat.dtype, ind.dtype, out.dtype))
    if not ind.is_contiguous():
        raise RuntimeError("voxel_maxpool_fwd: pcds_ind must be contiguous")
    bs, c, n = feat.shape[0], feat.shape[1], feat.shape[2]
    d = ind.shape[2]
    if len(out_size) != d or len(scale) != d or ind.shape[0] != bs or ind.shape[1] != n:
        raise RuntimeError("voxel_maxpool_fwd: shape mismatch feat %s ind %s out_size %s" % (tuple(feat.shape), tuple(ind.shape), tuple(out_size)))
    if tuple(out.shape) != (bs, c) + tuple(int(s) for s in out_size):
        raise RuntimeError("voxel_maxpool_fwd: out has shape %s" % (tuple(out.shape),))
    if voxel_max_idx is not None and (voxel_max_idx.dtype != torch.int64 or not voxel_max_idx.is_contiguous()):
        raise RuntimeError("voxel_maxpool_fwd: voxel_max_idx must be contiguous int64")
    lib = _lib.load()
    label = "voxel_maxpool_fwd[%dx%dx%d->%s]" % (bs, c, n, "x".join(str(int(s)) for s in out_size))
    with _on(feat.device), profiling.span(label):
        rc = lib.smos_voxel_maxpool_fwd(
            feat.data_ptr(), _lib.i64_array(feat.stride()[:3]), ind.data_ptr(), out.data_ptr(),
            _lib.i64_array(out.stride()), voxel_max_idx.data_ptr() if voxel_max_idx is not None else None,
            bs, c, n, d, _lib.i64_array(out_size), _lib.f32_array(scale), _dtype_code("voxel_maxpool_fwd", feat),
            _flag(feat.device).data_ptr(), _stream(feat))
    _lib.check(rc, "smos_voxel_maxpool_fwd")
    return out


def voxel_maxpool_bwd(feat, ind, out, grad_out, grad_feat, out_size, scale):
    _require_cuda("voxel_maxpool_bwd", feat, ind, out, grad_out, grad_feat)
    # grad_out is read with out's strides and grad_feat written with feat's: same logical layout required
    # (strides of size-1 dims are arbitrary, so compare through contiguity / explicit equality on real dims)
    def _same_layout(a, b):
        return all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)
    if not _same_layout(grad_out, out):
        grad_out = grad_out.contiguous()
        if not _same_layout(grad_out, out):
            raise RuntimeError("voxel_maxpool_bwd: grad_out layout must match out")
    if not _same_layout(grad_feat, feat):
        raise RuntimeError("voxel_maxpool_bwd: grad_feat layout must match feat")
    bs, c, n = feat.shape[0], feat.shape[1], feat.shape[2]
    lib = _lib.load()
    with _on(feat.device):
        rc = lib.smos_voxel_maxpool_bwd(
            feat.data_ptr(), _lib.i64_array(feat.stride()[:3]), ind.data_ptr(), out.data_ptr(), grad_out.data_ptr(),
            _lib.i64_array(out.stride()), grad_feat.data_ptr(), bs, c, n, ind.shape[2], _lib.i64_array(out_size),
            _lib.f32_array(scale), _dtype_code("voxel_maxpool_bwd", feat), _stream(feat))
    _lib.check(rc, "smos_voxel_maxpool_bwd")
    return grad_feat


def bilinear_gather(grid, coord, scale, out=None, point_major=False):
    """grid [B,C,H,W] (any strides), coord [B,N,K(,1)] -> out [B,C,N] (a view of a [B,N,C] buffer when
    point_major)."""
    _require_cuda("bilinear_gather", grid, coord, out)
    if grid.dtype != torch.float32 or coord.dtype != torch.float32:
        raise RuntimeError("bilinear_gather: float32 only (got %s, %s)" % (grid.dtype, coord.dtype))
    if coord.dim() == 4:
        coord = coord[..., 0]
    if not coord.is_contiguous():
        coord = coord.contiguous()
    b, c, h, w = grid.shape
    n, k = coord.shape[1], coord.shape[2]
    if out is None:
        if point_major:
            out = torch.empty((b, n, c), dtype=torch.float32, device=grid.device).permute(0, 2, 1)
        else:
            out = torch.empty((b, c, n), dtype=torch.float32, device=grid.device)
    lib = _lib.load()
    with _on(grid.device), profiling.span_f("bilinear_gather[%dx%dx%dx%d->%d]", (b, c, h, w, n)):
        rc = lib.smos_bilinear_gather_fwd(grid.data_ptr(), _lib.i64_array(grid.stride()), coord.data_ptr(), k,
                                          out.data_ptr(), _lib.i64_array(out.stride()[:3]), b, c, h, w, n,
                                          _lib.f32_array(scale), _stream(grid))
    _lib.check(rc, "smos_bilinear_gather_fwd")
    return out


def msda_fwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """value [N,S,M,D], shapes [L,2] int64 (device), level_start [L] int64 (device), loc [N,Lq,M,L,P,2],
    attn [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    _require_cuda("ms_deform_attn_forward", value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
    n, s, m, d = value.shape
    lq, l, p = sampling_loc.shape[1], sampling_loc.shape[3], sampling_loc.shape[4]
    out = torch.empty((n, lq, m * d), dtype=value.dtype, device=value.device)
    lib = _lib.load()
    with _on(value.device), profiling.span_f("msda_fwd[%dx%dx%dx%d]", (n, lq, m, d)):
        rc = lib.smos_msda_fwd(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                               sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(), n, s, m, d, l, lq, p,
                               _dtype_code("ms_deform_attn_forward", value), _stream(value))
    _lib.check(rc, "smos_msda_fwd")
    return out


def msda_bwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output):
    """Returns (grad_value, grad_sampling_loc, grad_attn_weight)."""
    _require_cuda("ms_deform_attn_backward", value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    for name, t in (("value", value), ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)):
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
    n, s, m, d = value.shape
    lq, l, p = sampling_loc.shape[1], sampling_loc.shape[3], sampling_loc.shape[4]
    g_value = torch.zeros_like(value)
    g_loc = torch.empty_like(sampling_loc)
    g_attn = torch.empty_like(attn_weight)
    lib = _lib.load()
    with _on(value.device), profiling.span_f("msda_bwd[%dx%dx%dx%d]", (n, lq, m, d)):
        rc = lib.smos_msda_bwd(grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                               sampling_loc.data_ptr(), attn_weight.data_ptr(), g_value.data_ptr(), g_loc.data_ptr(),
                               g_attn.data_ptr(), n, s, m, d, l, lq, p, _dtype_code("ms_deform_attn_backward", value),
                               _stream(value))
    _lib.check(rc, "smos_msda_bwd")
    return g_value, g_loc, g_attn


def tta_argmax(pred, want_prob=False):
    """pred [B,K,N(,1)] float32 -> labels [N] uint8 (and prob [N,K])."""
    _require_cuda("tta_argmax", pred)
    if pred.dim() == 4:
        pred = pred[..., 0]
    pred = pred.contiguous().float()
    b, k, n = pred.shape
    labels = torch.empty(n, dtype=torch.uint8, device=pred.device)
    prob = torch.empty((n, k), dtype=torch.float32, device=pred.device) if want_prob else None
    lib = _lib.load()
    with _on(pred.device):
        rc = lib.smos_tta_argmax(pred.data_ptr(), b, k, n, labels.data_ptr(), prob.data_ptr() if want_prob else None,
                                 _stream(pred))
    _lib.check(rc, "smos_tta_argmax")
    return (labels, prob) if want_prob else labels


def vote_clear(table):
    _require_cuda("vote_clear", table)
    lib = _lib.load()
    with _on(table.device):
        rc = lib.smos_vote_clear(table.data_ptr(), _stream(table))
    _lib.check(rc, "smos_vote_clear")


def vote_accumulate(points, labels, table, pose_diff=None, recip_quantize=False):
    """points [n, >=3] float32 rows, labels [n] uint8, pose_diff 4x4 (numpy float64) or None."""
    _require_cuda("vote_accumulate", points, labels, table)
    if points.dtype != torch.float32 or labels.dtype != torch.uint8 or points.stride(1) != 1:
        raise RuntimeError("vote_accumulate: points must be float32 rows and labels uint8")
    pose = None
    if pose_diff is not None:
        pose = _lib.f64_array([float(v) for v in pose_diff.reshape(-1)[:16]])
    lib = _lib.load()
    with _on(points.device):
        rc = lib.smos_vote_accumulate(points.data_ptr(), points.shape[0], points.stride(0), labels.data_ptr(), pose,
                                      1 if recip_quantize else 0, table.data_ptr(), _stream(points))
    _lib.check(rc, "smos_vote_accumulate")


def vote_accumulate_frames(frames, table, recip_quantize=False):
    """The whole voting window in one launch.  frames: list of (points [n, >=3] float32 rows, labels [n] uint8, pose_diff
    4x4 numpy float64 or None for the current frame)."""
    if not frames:
        return
    count = len(frames)
    pts, lab = (ctypes.c_void_p * count)(), (ctypes.c_void_p * count)()
    n, stride = (ctypes.c_int64 * count)(), (ctypes.c_int64 * count)()
    pose = (_lib.c_f64p * count)()
    keep = []
    for f, (points, labels, pose_diff) in enumerate(frames):
        _require_cuda("vote_accumulate_frames", points, labels, table)
        if points.dtype != torch.float32 or labels.dtype != torch.uint8 or points.stride(1) != 1 or points.device != table.device:
            raise RuntimeError("vote_accumulate_frames: points must be float32 rows and labels uint8, on the table's device")
        if labels.shape[0] != points.shape[0] or not labels.is_contiguous():
            raise RuntimeError("vote_accumulate_frames: one contiguous label per point")
        pts[f], lab[f], n[f], stride[f] = points.data_ptr(), labels.data_ptr(), points.shape[0], points.stride(0)
        if pose_diff is not None:
            arr = np.ascontiguousarray(pose_diff, dtype=np.float64).reshape(-1)
            if arr.size < 16:
                raise RuntimeError("vote_accumulate_frames: pose_diff must be 4x4")
            keep.append(arr)                          # the numpy buffer is read during the call
            pose[f] = arr.ctypes.data_as(_lib.c_f64p)
    lib = _lib.load()
    with _on(table.device), profiling.span_f("vote_accumulate[%dx%d]", (count, max(n))):
        rc = lib.smos_vote_accumulate_frames(count, pts, n, stride, lab, pose, 1 if recip_quantize else 0, table.data_ptr(),
                                             _stream(table))
    _lib.check(rc, "smos_vote_accumulate_frames")


def vote_resolve(points, labels, table, lut=None, recip_quantize=False):
    _require_cuda("vote_resolve", points, labels, table, lut)
    out = torch.empty(points.shape[0], dtype=torch.int32, device=points.device)
    lib = _lib.load()
    with _on(points.device):
        rc = lib.smos_vote_resolve(points.data_ptr(), points.shape[0], points.stride(0), labels.data_ptr(),
                                   1 if recip_quantize else 0, table.data_ptr(),
                                   lut.data_ptr() if lut is not None else None, out.data_ptr(), _stream(points))
    _lib.check(rc, "smos_vote_resolve")
    return out


def dbscan(points, eps, min_samples, max_sweeps=4096):
    """sklearn.cluster.DBSCAN(eps, min_samples).fit_predict(points[:, :3]) on the device (voxel_instance_voting.py:150-153).
    points [n, >=3] float32 rows.  Returns int32 [n]: the index of the cluster's lowest-index core point, -1 for noise
    (rank the distinct values to get scikit-learn's 0,1,2.. numbering).  Synchronises the stream."""
    _require_cuda("dbscan", points)
    if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] < 3 or points.stride(1) != 1:
        raise RuntimeError("dbscan: points must be float32 rows [n, >=3]")
    n = points.shape[0]
    labels = torch.empty(n, dtype=torch.int32, device=points.device)
    if n == 0:
        return labels
    lib = _lib.load()
    need = int(lib.smos_dbscan_work_bytes(n))
    if need <= 0:
        raise RuntimeError("dbscan: workspace query failed for n=%d" % n)
    work = torch.empty(need + 256, dtype=torch.uint8, device=points.device)
    base = (work.data_ptr() + 255) // 256 * 256
    with _on(points.device):
        rc = lib.smos_dbscan(points.data_ptr(), n, points.stride(0), float(eps), int(min_samples), labels.data_ptr(),
                             base, need, int(max_sweeps), _stream(points))
    _lib.check(rc, "smos_dbscan")
    return labels


def box_vote(points, labels, boxes, counts, pose_diff=None):
    """counts [K,3] uint32/int32 (zero-filled by the caller) += per-box class counts of one frame's kept points
    (voxel_instance_voting.py:170-179).  boxes [K,6] float32 (lo xyz, hi xyz), closed intervals."""
    _require_cuda("box_vote", points, labels, boxes, counts)
    if points.dtype != torch.float32 or labels.dtype != torch.uint8 or points.stride(1) != 1:
        raise RuntimeError("box_vote: points must be float32 rows and labels uint8")
    if boxes.dtype != torch.float32 or not boxes.is_contiguous() or boxes.dim() != 2 or boxes.shape[1] != 6:
        raise RuntimeError("box_vote: boxes must be a contiguous float32 [K, 6] tensor")
    if counts.dtype not in (torch.int32, torch.uint32) or not counts.is_contiguous() or counts.numel() != boxes.shape[0] * 3:
        raise RuntimeError("box_vote: counts must be a contiguous 32-bit [K, 3] tensor")
    pose = None
    if pose_diff is not None:
        pose = _lib.f64_array([float(v) for v in pose_diff.reshape(-1)[:16]])
    lib = _lib.load()
    with _on(points.device):
        rc = lib.smos_box_vote(points.data_ptr(), points.shape[0], points.stride(0), labels.data_ptr(), pose,
                               boxes.data_ptr(), boxes.shape[0], counts.data_ptr(), _stream(points))
    _lib.check(rc, "smos_box_vote")
    return counts


# ---------------------------------------------------------------------------------------------
# fused encoder epilogues (inference engine)
# ---------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _planes(name, t):
    """(batch stride, channel stride, H*W) of a 4-D tensor whose (H, W) planes are contiguous."""
    if t.dim() != 4 or t.stride(3) != 1 or t.stride(2) != t.shape[3]:
        raise RuntimeError("%s: expected contiguous (H, W) planes, got shape %s strides %s" % (name, tuple(t.shape), t.stride()))
    return t.stride(0), t.stride(1), t.shape[2] * t.shape[3]


def bias_act(x, bias, act, out=None, residual=None):
    """out = act(x + bias[c] (+ residual)); x, residual, out: [B,C,H,W] with contiguous planes (out may be a
    channel slice of a larger buffer, or x itself)."""
    _require_cuda("bias_act", x, bias, out, residual)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    xb, xc, hw = _planes("bias_act", x)
    ob, oc, _ = _planes("bias_act", out)
    rb = rc = 0
    if residual is not None:
        rb, rc, _ = _planes("bias_act", residual)
    lib = _lib.load()
    with _on(x.device):
        rc_ = lib.smos_bias_act(x.data_ptr(), xb, xc, bias.data_ptr() if bias is not None else None,
                                residual.data_ptr() if residual is not None else None, rb, rc, out.data_ptr(), ob, oc,
                                x.shape[0], x.shape[1], hw, act, _stream(x))
    _lib.check(rc_, "smos_bias_act")
    return out


def downsample_epilogue(a, p, bias, stride, out=None):
    """relu(a + bias[c] + maxpool3x3(p, stride, pad 1)); a [B,C,Ho,Wo], p [B,C,H,W] any strides."""
    _require_cuda("downsample_epilogue", a, p, bias, out)
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    ob, oc, _ = _planes("downsample_epilogue", out)
    lib = _lib.load()
    with _on(a.device):
        rc = lib.smos_downsample_epilogue(a.data_ptr(), _lib.i64_array(a.stride()), p.data_ptr(), _lib.i64_array(p.stride()),
                                          bias.data_ptr(), out.data_ptr(), ob, oc, p.shape[0], p.shape[1], p.shape[2],
                                          p.shape[3], stride, _stream(a))
    _lib.check(rc, "smos_downsample_epilogue")
    return out


def channel_gate_residual(y, bias, w1, b1, w2, b2, xres, sums_ws, out=None):
    _require_cuda("channel_gate_residual", y, bias, w1, b1, w2, b2, xres, sums_ws, out)
    if out is None:
        out = torch.empty(y.shape, dtype=y.dtype, device=y.device)
    yb, yc, hw = _planes("channel_gate_residual", y)
    rb, rc, _ = _planes("channel_gate_residual", xres)
    ob, oc, _ = _planes("channel_gate_residual", out)
    lib = _lib.load()
    with _on(y.device):
        rc_ = lib.smos_channel_gate_residual(y.data_ptr(), yb, yc, bias.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                             w2.data_ptr(), b2.data_ptr(), xres.data_ptr(), rb, rc, out.data_ptr(), ob, oc,
                                             sums_ws.data_ptr(), y.shape[0], y.shape[1], w1.shape[0], hw, _stream(y))
    _lib.check(rc_, "smos_channel_gate_residual")
    return out


def upsample_concat(sources, size, out=None):
    """Bilinear (align_corners=True) resize of up to three [B,C_i,H_i,W_i] maps to `size`, concatenated along C."""
    import ctypes
    _require_cuda("upsample_concat", *sources)
    b = sources[0].shape[0]
    ctot = sum(s.shape[1] for s in sources)
    if out is None:
        out = torch.empty((b, ctot, size[0], size[1]), dtype=torch.float32, device=sources[0].device)
    strides = [_planes("upsample_concat", s) for s in sources]
    ptrs = (ctypes.c_void_p * len(sources))(*[s.data_ptr() for s in sources])
    lib = _lib.load()
    with _on(out.device):
        rc = lib.smos_upsample_concat(ptrs, _lib.i64_array([s.shape[1] for s in sources]),
                                      _lib.i64_array([s.shape[2] for s in sources]),
                                      _lib.i64_array([s.shape[3] for s in sources]),
                                      _lib.i64_array([st[0] for st in strides]), _lib.i64_array([st[1] for st in strides]),
                                      len(sources), out.data_ptr(), b, size[0], size[1], _stream(out))
    _lib.check(rc, "smos_upsample_concat")
    return out


# ---------------------------------------------------------------------------------------------
# fused point-side kernels (inference engine)
# ---------------------------------------------------------------------------------------------
def _rows(name, t, c):
    """(batch pitch, row pitch) of a point-row view [B, N, c] whose last dim is contiguous."""
    if t.dim() != 3 or t.shape[2] != c or t.stride(2) != 1:
        raise RuntimeError("%s: expected point rows [B, N, %d] with contiguous channels, got %s / %s" % (name, c, tuple(t.shape), t.stride()))
    return t.stride(0), t.stride(1)


def pointnet_scatter(xyzi, coord, w1, b1, w2, b2, bev, pts_out=None, zero_fill=False):
    """xyzi [B,T,7,N(,1)], coord [B,T,N,K(,1)], bev [B,H,W,T*64] zero-filled channels-last (zero_fill=True clears it
    here, inside the profiled span), pts_out [B,N,64] rows."""
    _require_cuda("pointnet_scatter", xyzi, coord, w1, b1, w2, b2, bev, pts_out)
    b, t, cin, n = xyzi.shape[:4]
    k = coord.shape[3]
    if not (xyzi.is_contiguous() and coord.is_contiguous() and bev.is_contiguous()):
        raise RuntimeError("pointnet_scatter: xyzi, coord and bev must be contiguous")
    h, w = bev.shape[1], bev.shape[2]
    cout = w2.shape[0]
    if bev.shape[3] != t * cout:
        raise RuntimeError("pointnet_scatter: bev has %d channels, expected %d" % (bev.shape[3], t * cout))
    po_b = po_n = 0
    if pts_out is not None:
        po_b, po_n = _rows("pointnet_scatter", pts_out, cout)
    lib = _lib.load()
    with _on(xyzi.device), profiling.span_f("pointnet_scatter[%dx%dx%d->%dx%d]", (b, t, n, h, w)):
        if zero_fill:
            bev.zero_()
        rc = lib.smos_pointnet_scatter(xyzi.data_ptr(), coord.data_ptr(), k, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                       b2.data_ptr(), bev.data_ptr(), pts_out.data_ptr() if pts_out is not None else None,
                                       po_b, po_n, b, t, n, h, w, cin, w1.shape[0], cout, _stream(xyzi))
    _lib.check(rc, "smos_pointnet_scatter")
    return bev


def point_head_prepare(l1, l2, l3):
    """(W1 [96,192(,1,1)], b1), (W2 [64,96], b2), (W3 [M3,64], b3) -> the flat weight block smos_point_head expects."""
    w1, b1 = l1[0].reshape(l1[0].shape[0], -1).float(), l1[1].float()
    w2, b2 = l2[0].reshape(l2[0].shape[0], -1).float(), l2[1].float()
    w3, b3 = l3[0].reshape(l3[0].shape[0], -1).float(), l3[1].float()
    if tuple(w1.shape) != (96, 192) or tuple(w2.shape) != (64, 96) or w3.shape[1] != 64 or w3.shape[0] > 32:
        raise RuntimeError("point_head_prepare: expected 192 -> 96 -> 64 -> (<=32), got %s %s %s" % (tuple(w1.shape), tuple(w2.shape), tuple(w3.shape)))
    dev = w1.device
    lane = torch.arange(64, device=dev)
    m, h = lane & 31, lane >> 5
    a1 = w1.view(3, 32, 2, 96).permute(0, 3, 2, 1).reshape(3, 96, 64)                     # (mt, s, h*32 + m)

    def acc_order(n_steps):                                                               # ch(s, h) for s < n_steps
        s = torch.arange(n_steps, device=dev)[:, None]
        return 32 * (s >> 4) + 8 * ((s & 15) >> 2) + 4 * h[None, :] + (s & 3)            # [n_steps, 64]
    ch2 = acc_order(48)
    a2 = torch.stack([w2[mt * 32 + m[None, :].expand(48, 64), ch2] for mt in range(2)])  # (mt, s, lane)
    w3p = torch.zeros((32, 64), device=dev)
    w3p[:w3.shape[0]] = w3
    a3 = w3p[m[None, :].expand(32, 64), acc_order(32)]
    b3p = torch.zeros(32, device=dev)
    b3p[:b3.shape[0]] = b3
    flat = torch.cat([a1.reshape(-1), a2.reshape(-1), a3.reshape(-1), b1, b2, b3p]).contiguous()
    if flat.numel() != int(_lib.load().smos_point_head_weight_floats()):
        raise RuntimeError("point_head_prepare: weight block has %d floats" % flat.numel())
    return flat, int(w3.shape[0])


def point_head(rows, wprep, m3, out=None, n_live=None):
    """rows [B, N, >=192] float32 (row stride a multiple of 4 floats) -> logits [B, m3, N].
    n_live: optional device int32 tensor (its first element is read): the number of REAL points at the front of every sample;
    the logits of the scan's padding tail [n_live, N) come back as zeros without being computed (the streaming runner's form:
    val_StreamMOS.py:113 cuts them off)."""
    _require_cuda("point_head", rows, wprep, out, n_live)
    if rows.dtype != torch.float32 or rows.dim() != 3 or rows.stride(2) != 1 or rows.stride(0) != rows.shape[1] * rows.stride(1):
        raise RuntimeError("point_head: rows must be a float32 [B, N, C] tensor with dense point rows")
    if n_live is not None and (n_live.dtype != torch.int32 or n_live.numel() < 1):
        raise RuntimeError("point_head: n_live must be a device int32 tensor")
    b, n = rows.shape[0], rows.shape[1]
    if out is None:
        out = torch.empty((b, m3, n), dtype=torch.float32, device=rows.device)
    lib = _lib.load()
    with _on(rows.device), profiling.span_f("point_head[%dx%d]", (b, n)):
        rc = lib.smos_point_head_live(rows.data_ptr(), rows.stride(1), wprep.data_ptr(), out.data_ptr(), b, n, 192, 96, 64, m3,
                                      n_live.data_ptr() if n_live is not None else None, _stream(rows))
    _lib.check(rc, "smos_point_head_live")
    return out


def conv_prepare(w, mt, order="taps"):
    """w [Cout, Cin, KH, KW] (BatchNorm folded) -> the weight block of smos_conv_cl in MFMA operand order for `mt`
    32-channel output blocks per wave (include/smos.h): [cout tile][stage = (ky, kx, cin chunk)][k-step / 4][mt][lane][k-step % 4]
    with lane = h * 32 + m holding w[ct*32*mt + mt_i*32 + m][chunk*32 + 8*i4 + 4*h + c][ky][kx].
    order="rows" (smos_conv_rows_cl, mt <= 2): stage = (ky, cin chunk, kx) -- the kx taps of a staged input row together."""
    cout, cin, kh, kw = w.shape
    if cin % 32 or cout % (32 * mt) or mt not in (1, 2, 4) or order not in ("taps", "rows") or (order == "rows" and mt > 2):
        raise RuntimeError("conv_prepare: Cin %% 32 == 0 and Cout %% (32 * mt) == 0 required (got %s, mt=%d, order=%s)" % (tuple(w.shape), mt, order))
    #           ct                mt_i  m   chunk      i4  h  c   ky  kx
    v = w.float().reshape(cout // (32 * mt), mt, 32, cin // 32, 4, 2, 4, kh, kw)
    if order == "rows":
        v = v.permute(0, 7, 3, 8, 4, 1, 5, 2, 6)     # -> [ct, ky, chunk, kx, i4, mt_i, h, m, c]
    else:
        v = v.permute(0, 7, 8, 3, 4, 1, 5, 2, 6)     # -> [ct, ky, kx, chunk, i4, mt_i, h, m, c]
    return v.reshape(-1).contiguous()


def conv_cl_supported(x, cout, kernel, stride=1, residual=None, out=None):
    """The hard limits of smos_conv_cl / smos_conv_rows_cl (the SMOS_REQUIREs of csrc/conv_igemm.hip:486-500), as a
    predicate the engine asks BEFORE it routes a layer to the own kernels: Cin / Cout multiples of 32, Cout <= 2048,
    kernel <= 7x7, stride 1 or 2, and every operand below 2 GiB (the kernels address through 32-bit buffer offsets; a
    16-stream batch of 128-channel 256x256 maps is 2.1 GB).  Shapes only: works on meta tensors."""
    kh, kw = kernel
    b, cin, h, w = x.shape
    if cin % 32 or cout % 32 or cout > 2048 or not (1 <= kh <= 7 and 1 <= kw <= 7) or stride not in (1, 2):
        return False
    ho, wo = (h + 2 * (kh // 2) - kh) // stride + 1, (w + 2 * (kw // 2) - kw) // stride + 1
    if ho <= 0 or wo <= 0:
        return False

    def pitch(t, c):
        return t.stride(3) if t is not None and t.dim() == 4 and t.stride(1) == 1 else c
    limit = 1 << 31
    if b * h * w * pitch(x, cin) * 4 >= limit or b * ho * wo * pitch(out, cout) * 4 >= limit:
        return False
    if residual is not None and b * ho * wo * pitch(residual, cout) * 4 >= limit:
        return False
    return True


def conv_rows_ok(kernel, stride, cin, cout):
    """Shapes smos_conv_rows_cl covers (stride 1, "same" padding, KW in {3, 5, 7})."""
    kh, kw = kernel
    return stride == 1 and kw in (3, 5, 7) and kh in (1, 3, 5, 7) and cin % 32 == 0 and cout % 32 == 0


def conv_rows_cl(x, wprep, bias, act, cout, kernel, mt=1, residual=None, out=None, chan_sums=None):
    """conv_cl for stride 1 / "same" padding / KW in {3, 5, 7} at 32 * mt (mt in {1, 2}) output channels per block, with the
    input rows staged through LDS once per kernel row instead of one global request per tap (csrc/conv_rows.hip).
    wprep = conv_prepare(w, mt, order="rows")."""
    _require_cuda("conv_rows_cl", x, wprep, bias, residual, out, chan_sums)
    b, cin, h, w = x.shape
    kh, kw = kernel
    if not conv_rows_ok(kernel, 1, cin, cout) or wprep.numel() != cout * cin * kh * kw or mt not in (1, 2) or cout % (32 * mt):
        raise RuntimeError("conv_rows_cl: unsupported shape %s k%dx%d -> %d" % (tuple(x.shape), kh, kw, cout))
    if out is None:
        out = empty_cl(b, cout, h, w, x.device)
    elif tuple(out.shape) != (b, cout, h, w):
        raise RuntimeError("conv_rows_cl: out has shape %s" % (tuple(out.shape),))
    if residual is not None and tuple(residual.shape) != (b, cout, h, w):
        raise RuntimeError("conv_rows_cl: residual has shape %s" % (tuple(residual.shape),))
    if chan_sums is not None and (residual is not None or not chan_sums.is_contiguous() or
                                  tuple(chan_sums.shape) != (b, conv_sum_chunks(h, w), cout)):
        raise RuntimeError("conv_rows_cl: chan_sums must be contiguous [B, conv_sum_chunks(H, W), Cout], without a residual")
    lib = _lib.load()
    label = "conv_cl[%dx%dx%dx%d->%dx%dx%dk%dx%d%s]" % (b, cin, h, w, cout, h, w, kh, kw, "+res" if residual is not None else "")
    args = (x.data_ptr(), _cl("conv_rows_cl", x), wprep.data_ptr(), bias.data_ptr() if bias is not None else None,
            residual.data_ptr() if residual is not None else None, _cl("conv_rows_cl", residual) if residual is not None else 0,
            out.data_ptr(), _cl("conv_rows_cl", out), b, h, w, cin, cout, kh, kw, int(mt), int(act),
            chan_sums.data_ptr() if chan_sums is not None else None)
    with _on(x.device), profiling.span(label, "conv_rows"):
        rc = lib.smos_conv_rows_cl(*args, _stream(x))
    _lib.check(rc, "smos_conv_rows_cl")
    if profiling._replay_label == label:
        keep = (x, wprep, bias, residual, out, chan_sums)

        def again(keep=keep):
            with _on(keep[0].device), profiling.span(label, "conv_rows"):
                _lib.check(lib.smos_conv_rows_cl(*args, _stream(keep[0])), "smos_conv_rows_cl")
        profiling.offer_replay(label, again)
    return out


_CONV_MIN_WAVE_TILES = int(os.environ.get("SMOS_CONV_MIN_WAVE_TILES", "2048"))   # tuning knob (tools/ubench_conv.py)


def conv_mt(cout, n_pixels, residual=False):
    """Output blocks per wave for smos_conv_cl: wider waves re-use each activation load more often, narrower ones give
    more wave tiles; aim at >= 2 waves for each of the 1024 SIMDs.  With a residual input the kernel holds the residual
    tile in registers, which limits it to mt <= 2."""
    tiles32 = (n_pixels + 31) // 32
    for mt in ((2, 1) if residual else (4, 2, 1)):
        if cout % (32 * mt) == 0 and tiles32 * (cout // (32 * mt)) >= _CONV_MIN_WAVE_TILES:
            return mt
    return 1


def conv_sum_chunks(ho, wo):
    """Row segments per sample in the channel-sum table of conv_cl(..., chan_sums=...)."""
    return ((ho + 3) // 4) * ((wo + 31) // 32) * 4


def conv_cl(x, wprep, bias, act, cout, kernel, stride=1, padding=None, mt=1, residual=None, out=None, chan_sums=None):
    """act(conv(x) + bias [+ residual]) on channels-last [B,C,H,W] views in one launch (csrc/conv_igemm.hip).
    wprep = conv_prepare(w, mt); kernel = (KH, KW); padding defaults to "same" for odd kernels.
    chan_sums: optional float32 [B, conv_sum_chunks(Ho, Wo), Cout] that receives the per-row-segment channel sums of the
    output (the average-pool input of a ChannelAtt block); not together with a residual."""
    _require_cuda("conv_cl", x, wprep, bias, residual, out, chan_sums)
    b, cin, h, w = x.shape
    kh, kw = kernel
    ph, pw = padding if padding is not None else (kh // 2, kw // 2)
    ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    if wprep.numel() != cout * cin * kh * kw:
        raise RuntimeError("conv_cl: weight block has %d floats, expected %d" % (wprep.numel(), cout * cin * kh * kw))
    if out is None:
        out = empty_cl(b, cout, ho, wo, x.device)
    elif tuple(out.shape) != (b, cout, ho, wo):
        raise RuntimeError("conv_cl: out has shape %s, expected %s" % (tuple(out.shape), (b, cout, ho, wo)))
    if residual is not None and tuple(residual.shape) != (b, cout, ho, wo):
        raise RuntimeError("conv_cl: residual has shape %s" % (tuple(residual.shape),))
    if chan_sums is not None and (residual is not None or not chan_sums.is_contiguous() or chan_sums.dtype != torch.float32 or
                                  tuple(chan_sums.shape) != (b, conv_sum_chunks(ho, wo), cout)):
        raise RuntimeError("conv_cl: chan_sums must be contiguous float32 [B, conv_sum_chunks(Ho, Wo), Cout], without a residual")
    lib = _lib.load()
    label = ("conv_cl[%dx%dx%dx%d->%dx%dx%dk%dx%d%s]" % (b, cin, h, w, cout, ho, wo, kh, kw, "+res" if residual is not None else "")
             if profiling.enabled() else None)
    args = (x.data_ptr(), _cl("conv_cl", x), wprep.data_ptr(), bias.data_ptr() if bias is not None else None,
            residual.data_ptr() if residual is not None else None, _cl("conv_cl", residual) if residual is not None else 0,
            out.data_ptr(), _cl("conv_cl", out), b, h, w, cin, cout, kh, kw, stride, ph, pw, mt, int(act),
            chan_sums.data_ptr() if chan_sums is not None else None)
    with _on(x.device), profiling.span(label, "conv_igemm"):
        rc = lib.smos_conv_cl(*args, _stream(x))
    _lib.check(rc, "smos_conv_cl")
    if label is not None and profiling._replay_label == label:
        keep = (x, wprep, bias, residual, out, chan_sums)          # the closure keeps the operands alive

        def again(keep=keep):
            with _on(keep[0].device), profiling.span(label, "conv_igemm"):
                _lib.check(lib.smos_conv_cl(*args, _stream(keep[0])), "smos_conv_cl")
        profiling.offer_replay(label, again)
    return out


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def conv_wino_prepare(w, mb):
    """w [Cout, Cin, 3, 3] (BatchNorm folded) -> the weight block of smos_conv_wino_cl: U = G w G^T per (cout, cin) in
    float64, rounded once to float32, in MFMA operand order [cout tile][cin chunk of 16][k-step i][mb][xi][lane][nu] with
    lane = q * 16 + m holding U[xi][nu] of w[ct*16*mb + mb_i*16 + m][chunk*16 + 4*q + i] (include/smos.h)."""
    cout, cin, kh, kw = w.shape
    if (kh, kw) != (3, 3) or cin % 16 or mb not in (1, 2) or cout % (16 * mb):
        raise RuntimeError("conv_wino_prepare: 3x3 kernel, Cin %% 16 == 0 and Cout %% (16 * mb) == 0 required (got %s, mb=%d)"
                           % (tuple(w.shape), mb))
    g = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    u = (g @ w.double() @ g.t()).float()                       # [Cout, Cin, xi, nu]
    #          ct               mb_i m   chunk     q  i  xi nu
    v = u.reshape(cout // (16 * mb), mb, 16, cin // 16, 4, 4, 4, 4)
    v = v.permute(0, 3, 5, 1, 6, 4, 2, 7)                      # -> [ct, chunk, i, mb_i, xi, q, m, nu]
    return v.reshape(-1).contiguous()


def conv_wino_ok(kernel, stride, cin, cout):
    """Shapes smos_conv_wino_cl covers: 3x3, stride 1, "same" padding, Cin and Cout multiples of 16."""
    return tuple(kernel) == (3, 3) and stride == 1 and cin % 16 == 0 and cout % 16 == 0


def conv_wino_mb(cout, n_items16=None):
    """16-channel output blocks per wave: 2 wherever Cout allows it -- half the operand traffic per multiply-add, and faster
    on every layer of the network, the ones with fewer items than resident blocks included (tools/ubench_wino.py)."""
    return 2 if cout % 32 == 0 else 1


def conv_wino_sum_chunks(h, w):
    """Chunks per sample in the channel-sum table of conv_wino_cl(..., chan_sums=...)."""
    return ((h + 7) // 8) * ((w + 31) // 32) * 4


def conv_wino_cl(x, wprep, bias, act, cout, mb=2, residual=None, out=None, chan_sums=None):
    """act(conv3x3(x) + bias [+ residual]) (stride 1, "same" padding) on channels-last [B,C,H,W] views in one launch of the
    Winograd F(2x2, 3x3) kernel (csrc/conv_wino.hip).  wprep = conv_wino_prepare(w, mb).
    chan_sums: optional float32 [B, conv_wino_sum_chunks(H, W), Cout]; not together with a residual."""
    _require_cuda("conv_wino_cl", x, wprep, bias, residual, out, chan_sums)
    b, cin, h, w = x.shape
    if not conv_wino_ok((3, 3), 1, cin, cout) or mb not in (1, 2) or cout % (16 * mb) or wprep.numel() != 16 * cout * cin:
        raise RuntimeError("conv_wino_cl: unsupported shape %s -> %d (mb=%d)" % (tuple(x.shape), cout, mb))
    if out is None:
        out = empty_cl(b, cout, h, w, x.device)
    elif tuple(out.shape) != (b, cout, h, w):
        raise RuntimeError("conv_wino_cl: out has shape %s" % (tuple(out.shape),))
    if residual is not None and tuple(residual.shape) != (b, cout, h, w):
        raise RuntimeError("conv_wino_cl: residual has shape %s" % (tuple(residual.shape),))
    if chan_sums is not None and (residual is not None or not chan_sums.is_contiguous() or chan_sums.dtype != torch.float32 or
                                  tuple(chan_sums.shape) != (b, conv_wino_sum_chunks(h, w), cout)):
        raise RuntimeError("conv_wino_cl: chan_sums must be contiguous float32 [B, conv_wino_sum_chunks(H, W), Cout], without a residual")
    lib = _lib.load()
    args = (x.data_ptr(), _cl("conv_wino_cl", x), wprep.data_ptr(), bias.data_ptr() if bias is not None else None,
            residual.data_ptr() if residual is not None else None, _cl("conv_wino_cl", residual) if residual is not None else 0,
            out.data_ptr(), _cl("conv_wino_cl", out), b, h, w, cin, cout, int(mb), int(act),
            chan_sums.data_ptr() if chan_sums is not None else None)
    fn = lib.smos_conv_wino_cl
    if not profiling.enabled():                      # the hot path: no label, no span
        dev = x.device
        with _on(dev):
            rc = fn(*args, _raw_stream(_dev_index(dev)))
        if rc:
            _lib.check(rc, "smos_conv_wino_cl")
        return out
    label = "conv_cl[%dx%dx%dx%d->%dx%dx%dk3x3%s]" % (b, cin, h, w, cout, h, w, "+res" if residual is not None else "")
    with _on(x.device), profiling.span(label, "conv_wino"):
        rc = fn(*args, _stream(x))
    _lib.check(rc, "smos_conv_wino_cl")
    if profiling._replay_label == label:
        keep = (x, wprep, bias, residual, out, chan_sums)

        def again(keep=keep):
            with _on(keep[0].device), profiling.span(label, "conv_wino"):
                _lib.check(fn(*args, _stream(keep[0])), "smos_conv_wino_cl")
        profiling.offer_replay(label, again)
    return out


class WinoChainWorkspace:
    """Completion counters of smos_conv_wino_chain_cl for one (layer count, map shape): zeroed once, then monotonic -- launch k
    waits for k * nct -- so nothing is cleared between launches.  One per call site AND stream (two launches that share it
    must not overlap)."""

    def __init__(self, n_layers, b, h, w, device):
        lib = _lib.load()
        self.key = (n_layers, b, h, w)
        self.ws = torch.zeros(lib.smos_conv_wino_chain_ws_ints(n_layers, b, h, w), dtype=torch.int32, device=device)
        self.launch_no = 0

    def gave_up(self):
        """True if a wait of any launch so far ran out of polls (the results of that launch are invalid).  Synchronises."""
        return bool(self.ws[-1].item())


def conv_wino_chain_cl(x, layers, chain_ws, chan_sums=None, mb=2):
    """EXPERIMENTAL.  A run of stride-1 3x3 convolutions C -> C of one map size in ONE launch (csrc/conv_wino_chain.hip): layer L
    reads x (L = 0) or layer L - 1's output.  layers: list of (wprep = conv_wino_prepare(w, mb), bias or None, res_from, out, act)
    with res_from -1 (none), 0 (= x) or j > 0 (= the output of layer j - 1); every `out` its own channels-last map of x's shape.
    chan_sums: the channel-sum table of the LAST layer (as conv_wino_cl's).  Bit-identical to the conv_wino_cl launches."""
    n = len(layers)
    b, c, h, w = x.shape
    if chain_ws.key != (n, b, h, w) or chain_ws.ws.device != x.device:
        raise RuntimeError("conv_wino_chain_cl: workspace made for %s, called with %s" % (chain_ws.key, (n, b, h, w)))
    _require_cuda("conv_wino_chain_cl", x, chan_sums, *[t for l in layers for t in (l[0], l[1], l[3])])
    wp, bs, outs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    rf, acts, pit = (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)(), (ctypes.c_int64 * n)()
    for i, (wprep, bias, res_from, out, act) in enumerate(layers):
        if wprep.numel() != 16 * c * c or tuple(out.shape) != (b, c, h, w):
            raise RuntimeError("conv_wino_chain_cl: layer %d: weight block or output of the wrong size" % i)
        wp[i], bs[i], outs[i] = wprep.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr()
        rf[i], acts[i], pit[i] = int(res_from), int(act), _cl("conv_wino_chain_cl", out)
    if chan_sums is not None and (not chan_sums.is_contiguous() or chan_sums.dtype != torch.float32 or
                                  tuple(chan_sums.shape) != (b, conv_wino_sum_chunks(h, w), c)):
        raise RuntimeError("conv_wino_chain_cl: chan_sums must be contiguous float32 [B, conv_wino_sum_chunks(H, W), C]")
    chain_ws.launch_no += 1
    dev = x.device
    with _on(dev):
        rc = _lib.load().smos_conv_wino_chain_cl(n, x.data_ptr(), _cl("conv_wino_chain_cl", x), wp, bs, rf, outs, pit, acts,
                                                 chan_sums.data_ptr() if chan_sums is not None else None, chain_ws.ws.data_ptr(),
                                                 chain_ws.launch_no, b, h, w, c, int(mb), _raw_stream(_dev_index(dev)))
    if rc:
        chain_ws.launch_no -= 1
        _lib.check(rc, "smos_conv_wino_chain_cl")
    return layers[-1][3]


class BasicBlockPlan:
    """Operands of one BasicBlock (networks/backbone.py:136-159) for smos_basic_block_cl: the two Winograd weight blocks, the
    biases and, for a ChannelAtt block, the gate MLP -- device addresses taken once (the tensors are kept alive here)."""
    __slots__ = ("c", "mb", "gated", "cr", "ptrs", "_keep")

    def __init__(self, w1, b1, w2, b2, gate=None):
        c = w1.shape[0]
        if tuple(w1.shape) != (c, c, 3, 3) or tuple(w2.shape) != (c, c, 3, 3) or c % 16:
            raise RuntimeError("BasicBlockPlan: two [C, C, 3, 3] weights with C %% 16 == 0 expected, got %s / %s"
                               % (tuple(w1.shape), tuple(w2.shape)))
        _require_cuda("BasicBlockPlan", w1, b1, w2, b2, *(gate or ()))
        self.c, self.mb = c, conv_wino_mb(c)
        u1, u2 = conv_wino_prepare(w1, self.mb), conv_wino_prepare(w2, self.mb)
        b1, b2 = b1.float().contiguous(), b2.float().contiguous()
        self.gated = gate is not None
        g = [t.float().contiguous() for t in gate] if self.gated else []
        if self.gated and (tuple(g[0].shape) != (g[0].shape[0], c) or tuple(g[2].shape) != (c, g[0].shape[0])):
            raise RuntimeError("BasicBlockPlan: gate MLP must be [Cr, C], [Cr], [C, Cr], [C]")
        self.cr = g[0].shape[0] if self.gated else 0
        self._keep = (u1, b1, u2, b2, g)
        self.ptrs = (u1.data_ptr(), b1.data_ptr(), u2.data_ptr(), b2.data_ptr()) + \
            (tuple(t.data_ptr() for t in g) if self.gated else (None, None, None, None)) + (self.cr,)


class UnbalanceBlockPlan:
    """Operands of one Unbalance_BasicBlock (networks/multi_view_encoder.py:478-497) for smos_unbalance_block_cl."""
    __slots__ = ("c", "mb", "ptrs_a", "ptrs_b", "ptrs_c", "_keep")

    def __init__(self, wa, ba, wb, bb, wc, bc):
        c = wa.shape[0]
        ka, kb = tuple(wa.shape[2:]), tuple(wb.shape[2:])
        if (tuple(wa.shape[:2]) != (c, c) or tuple(wb.shape[:2]) != (c, c) or tuple(wc.shape) != (c, 2 * c, 3, 3) or
                not conv_wino1d_ok(ka, 1, c, c) or not conv_wino1d_ok(kb, 1, c, c)):
            raise RuntimeError("UnbalanceBlockPlan: unexpected weights %s %s %s" % (tuple(wa.shape), tuple(wb.shape), tuple(wc.shape)))
        _require_cuda("UnbalanceBlockPlan", wa, ba, wb, bb, wc, bc)
        self.c, self.mb = c, conv_wino_mb(c)
        ua, ub = conv_wino1d_prepare(wa, self.mb), conv_wino1d_prepare(wb, self.mb)
        uc = conv_wino_prepare(wc, self.mb)
        ba, bb, bc = (t.float().contiguous() for t in (ba, bb, bc))
        self._keep = (ua, ub, uc, ba, bb, bc)
        self.ptrs_a = (ua.data_ptr(), ba.data_ptr(), ka[0], ka[1])
        self.ptrs_b = (ub.data_ptr(), bb.data_ptr(), kb[0], kb[1])
        self.ptrs_c = (uc.data_ptr(), bc.data_ptr())


def unbalance_block_cl(x, plan, both=None, out=None):
    """Unbalance_BasicBlock.forward on a channels-last [B,C,H,W] view in one foreign call (csrc/blocks.hip): conv_wino1d_cl x 2
    into the halves of `both` [B,2C,H,W], conv_wino_cl over it + x.  Bit-identical to the separate calls."""
    _require_cuda("unbalance_block_cl", x, both, out)
    b, c, h, w = x.shape
    if c != plan.c:
        raise RuntimeError("unbalance_block_cl: %d channels, the block has %d" % (c, plan.c))
    dev = x.device
    if both is None:
        both = empty_cl(b, 2 * c, h, w, dev)
    if out is None:
        out = empty_cl(b, c, h, w, dev)
    if both.shape != (b, 2 * c, h, w) or out.shape != x.shape:
        raise RuntimeError("unbalance_block_cl: both must be [B,2C,H,W] and out have x's shape %s" % (tuple(x.shape),))
    with _on(dev):
        rc = _lib.load().smos_unbalance_block_cl(x.data_ptr(), _cl("unbalance_block_cl", x), *plan.ptrs_a, *plan.ptrs_b, *plan.ptrs_c,
                                                 both.data_ptr(), _cl("unbalance_block_cl", both), out.data_ptr(),
                                                 _cl("unbalance_block_cl", out), b, h, w, c, plan.mb, _raw_stream(_dev_index(dev)))
    if rc:
        _lib.check(rc, "smos_unbalance_block_cl")
    return out


def basic_block_ok(c, gated):
    """Channel counts smos_basic_block_cl takes in the engine (those of the gate kernel where the block has one)."""
    return c % 16 == 0 and (not gated or (c % 32 == 0 and c <= 256 and 1024 % c == 0))


def basic_block_cl(x, plan, y=None, out=None, ws=None):
    """BasicBlock.forward on a channels-last [B,C,H,W] view in one foreign call (csrc/blocks.hip): the launches of
    conv_wino_cl, conv_wino_cl [, channel_gate_apply_cl] with the same arguments, bit-identical results.  y: scratch map for the
    first conv (allocated if None); ws: >= smos_basic_block_ws_floats floats for a gated block."""
    _require_cuda("basic_block_cl", x, y, out, ws)
    b, c, h, w = x.shape
    if c != plan.c:
        raise RuntimeError("basic_block_cl: %d channels, the block has %d" % (c, plan.c))
    dev = x.device
    if y is None:
        y = empty_cl(b, c, h, w, dev)
    if out is None:
        out = empty_cl(b, c, h, w, dev)
    if y.shape != x.shape or out.shape != x.shape:
        raise RuntimeError("basic_block_cl: y / out must have x's shape %s" % (tuple(x.shape),))
    lib = _lib.load()
    wsp = None
    if plan.gated:
        if ws is None or ws.dtype != torch.float32 or not ws.is_contiguous() or ws.numel() < lib.smos_basic_block_ws_floats(b, h, w, c):
            raise RuntimeError("basic_block_cl: a gated block needs smos_basic_block_ws_floats floats of contiguous scratch")
        wsp = ws.data_ptr()
    with _on(dev):
        rc = lib.smos_basic_block_cl(x.data_ptr(), _cl("basic_block_cl", x), *plan.ptrs, y.data_ptr(), _cl("basic_block_cl", y),
                                     out.data_ptr(), _cl("basic_block_cl", out), wsp, b, h, w, c, plan.mb,
                                     _raw_stream(_dev_index(dev)))
    if rc:
        _lib.check(rc, "smos_basic_block_cl")
    return out


def msda_fwd_qp(value, qp, h, w, points):
    """value [N, H*W, M, 32] contiguous, qp [N, H*W, M*P*3] contiguous (offsets | logits) -> [N, H*W, M*32]."""
    _require_cuda("msda_fwd_qp", value, qp)
    n, s, m, d = value.shape
    if not (value.is_contiguous() and qp.is_contiguous()) or s != h * w or qp.shape[-1] != m * points * 3 or value.dtype != torch.float32:
        raise RuntimeError("msda_fwd_qp: expected contiguous float32 value [N,H*W,M,D] and qp [N,H*W,M*P*3]")
    out = torch.empty((n, s, m * d), dtype=torch.float32, device=value.device)
    lib = _lib.load()
    with _on(value.device), profiling.span_f("msda_fwd[%dx%dx%dx%d]", (n, s, m, d)):
        rc = lib.smos_msda_fwd_qp(value.data_ptr(), qp.data_ptr(), out.data_ptr(), n, h, w, m, d, points, _stream(value))
    _lib.check(rc, "smos_msda_fwd_qp")
    return out


def add_layer_norm(x, res, gamma, beta, eps=1e-5):
    """LayerNorm(x + res) over the last dimension; x / res contiguous float32 [..., C]."""
    _require_cuda("add_layer_norm", x, res, gamma, beta)
    if not x.is_contiguous() or (res is not None and (not res.is_contiguous() or res.shape != x.shape)):
        raise RuntimeError("add_layer_norm: x and res must be contiguous and of equal shape")
    c = x.shape[-1]
    out = torch.empty_like(x)
    lib = _lib.load()
    with _on(x.device):
        rc = lib.smos_add_layer_norm(x.data_ptr(), res.data_ptr() if res is not None else None, gamma.data_ptr(), beta.data_ptr(),
                                     out.data_ptr(), x.numel() // c, c, float(eps), _stream(x))
    _lib.check(rc, "smos_add_layer_norm")
    return out


# ---------------------------------------------------------------------------------------------
# temporal fusion (csrc/tfusion.hip): the token-wise Linear / LayerNorm / FFN chain of a DeformAttnLayer on the matrix cores
# ---------------------------------------------------------------------------------------------
def _tf_pairs(w):
    """w [O, K] (O, K multiples of 16) -> [O/16, K/16, 64, 4]: pair (o, t) = 64 lanes x float4, lane q * 16 + m holding
    w[16 o + m][16 t + 4 q + 0..3] -- the A operands of the four v_mfma_f32_16x16x4_f32 that multiply k-tile t into output tile o."""
    o, k = w.shape
    v = w.float().reshape(o // 16, 16, k // 16, 4, 4)          # [o, m, t, q, r]
    return v.permute(0, 2, 3, 1, 4).reshape(o // 16, k // 16, 64, 4)


def tfusion_pack_linear(w):
    """w [cout, 128] -> the weight stream of one smos_tfusion_project job: cout zero-padded to a multiple of 64, slot o = the
    eight pairs (o, t = 0..7)."""
    cout, k = w.shape
    if k != 128 or cout > 2048:
        raise RuntimeError("tfusion_pack_linear: expected [<=2048, 128], got %s" % (tuple(w.shape),))
    pad = (cout + 63) // 64 * 64
    wp = torch.zeros((pad, k), dtype=torch.float32, device=w.device)
    wp[:cout] = w
    return _tf_pairs(wp).reshape(-1).contiguous()


class TfusionLayer:
    """Weights of one DeformAttnLayer behind its sampler in the order smos_tfusion_layer streams them, plus (optionally) the
    NEXT layer's [sampling_offsets | attention_weights] projection, which the kernel applies to its fresh output."""

    def __init__(self, out, norm1, lin1, lin2, norm2, next_qproj=None):
        wo, bo = out
        w1, b1 = lin1
        w2, b2 = lin2
        ffn = w1.shape[0]
        if tuple(wo.shape) != (128, 128) or w1.shape[1] != 128 or tuple(w2.shape) != (128, ffn) or ffn % 32:
            raise RuntimeError("TfusionLayer: built for d_model 128 and an FFN width that is a multiple of 32")
        dev = wo.device
        po, p1, p2 = _tf_pairs(wo), _tf_pairs(w1), _tf_pairs(w2)          # [8,8,..], [F/16,8,..], [8,F/16,..]
        slots = [po.reshape(8, 8, 64, 4)]
        # the kernel runs one hidden tile ahead with linear1: linear1(0), then [linear1(j + 1), linear2(j)] ..., linear2(last)
        l2 = p2.permute(1, 0, 2, 3)                                         # [F/16, 8 output tiles, 64, 4]
        n_t = ffn // 16
        slots.append(p1[0:1])
        if n_t > 1:
            slots.append(torch.stack((p1[1:], l2[:-1]), 1).reshape(-1, 8, 64, 4))
        slots.append(l2[-1:])
        self.nq = 0
        bq = torch.zeros(64, dtype=torch.float32, device=dev)
        if next_qproj is not None:
            wq, bqv = next_qproj
            self.nq = int(wq.shape[0])
            if self.nq > 64 or self.nq % 4 or wq.shape[1] != 128:
                raise RuntimeError("TfusionLayer: the next projection must have 4..64 output channels (multiple of 4)")
            wqp = torch.zeros((64, 128), dtype=torch.float32, device=dev)
            wqp[:self.nq] = wq
            bq[:self.nq] = bqv
            slots.append(_tf_pairs(wqp).reshape(4, 8, 64, 4))
        self.stream = torch.cat(slots).reshape(-1).contiguous()
        self.params = torch.cat([bo.float(), norm1[0].float(), norm1[1].float(), b1.float(), b2.float(), norm2[0].float(),
                                 norm2[1].float(), bq]).contiguous()
        self.ffn = int(ffn)
        self.eps1 = float(norm1[2]) if len(norm1) > 2 else 1e-5
        self.eps2 = float(norm2[2]) if len(norm2) > 2 else 1e-5
        lib = _lib.load()
        if (self.stream.numel() != int(lib.smos_tfusion_layer_stream_floats(self.ffn, 1 if self.nq else 0)) or
                self.params.numel() != int(lib.smos_tfusion_layer_param_floats(self.ffn))):
            raise RuntimeError("TfusionLayer: packed sizes disagree with the kernel's")


def _token_rows(name, t):
    """(tokens, pitch) of a token matrix [..., C] with contiguous channels and evenly pitched rows."""
    c = t.shape[-1]
    if t.dtype != torch.float32 or t.stride(-1) != 1:
        raise RuntimeError("%s: float32 rows with contiguous channels expected" % name)
    rows = t.reshape(-1, c) if t.is_contiguous() else t
    if rows.dim() != 2:
        raise RuntimeError("%s: a dense [..., C] tensor or a pitched [tokens, C] view expected, got %s / %s" % (name, tuple(t.shape), t.stride()))
    return rows.shape[0], rows.stride(0)


def tfusion_project(jobs):
    """jobs: up to eight (x [..., 128] token rows, wstream = tfusion_pack_linear(W), bias [cout] -- or an int cout for a Linear
    without bias[, out: a [tokens, cout] view with contiguous channels, e.g. a column range of a wider matrix]) -> list of
    [tokens, cout] outputs, ONE launch (csrc/tfusion.hip): the fusion's projections that depend on no previous layer; the
    decoder's tap products.  The jobs may have different token counts."""
    n = len(jobs)
    if not 1 <= n <= 8:
        raise RuntimeError("tfusion_project: 1..8 jobs")
    xs, pit, ws, bs, outs, ops_, couts, toks = ((ctypes.c_void_p * n)(), (ctypes.c_int64 * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(),
                                                (ctypes.c_void_p * n)(), (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)())
    res = []
    for j, job in enumerate(jobs):
        x, w, b = job[0], job[1], job[2]
        bias = None if isinstance(b, int) else b
        _require_cuda("tfusion_project", x, w, bias)
        tk, pitch = _token_rows("tfusion_project", x)
        if x.shape[-1] != 128:
            raise RuntimeError("tfusion_project: 128-channel token rows expected")
        cout = int(b) if bias is None else int(bias.shape[0])
        if w.numel() != (cout + 63) // 64 * 64 * 128:
            raise RuntimeError("tfusion_project: weight stream of %d floats for %d outputs" % (w.numel(), cout))
        if len(job) > 3 and job[3] is not None:
            out = job[3]
            _require_cuda("tfusion_project", out)
            if out.dim() != 2 or tuple(out.shape) != (tk, cout) or out.stride(1) != 1 or out.dtype != torch.float32:
                raise RuntimeError("tfusion_project: out must be a float32 [tokens, cout] view with contiguous channels")
        else:
            out = torch.empty((tk, cout), dtype=torch.float32, device=x.device)
        res.append(out)
        xs[j], pit[j], ws[j], outs[j], ops_[j], couts[j], toks[j] = x.data_ptr(), pitch, w.data_ptr(), out.data_ptr(), out.stride(0), cout, tk
        bs[j] = bias.data_ptr() if bias is not None else None
    lib = _lib.load()
    x0 = jobs[0][0]
    label = ("tfusion_project[%s]" % ",".join("%dx128->%d" % (int(toks[j]), int(couts[j])) for j in range(n))
             if profiling.enabled() else None)
    with _on(x0.device), profiling.span(label):
        rc = lib.smos_tfusion_project(n, xs, pit, ws, bs, outs, ops_, couts, toks, _stream(x0))
    _lib.check(rc, "smos_tfusion_project")
    return res


def tfusion_layer(sampled, query, prep, out=None):
    """One DeformAttnLayer behind its sampler (multi_view_encoder.py:314-320) in one launch: norm2(q1 + FFN(q1)) with q1 =
    norm1(query + output_proj(sampled)); returns (out [tokens, 128], qp_next [tokens, nq] or None).  sampled: dense
    [..., 128]; query: token rows (may be pitched); prep: TfusionLayer."""
    _require_cuda("tfusion_layer", sampled, query, out)
    tokens, sp = _token_rows("tfusion_layer", sampled)
    tq, qp = _token_rows("tfusion_layer", query)
    if sp != 128 or tq != tokens or sampled.shape[-1] != 128 or query.shape[-1] != 128:
        raise RuntimeError("tfusion_layer: sampled must be dense [tokens, 128] and query hold as many 128-channel rows")
    if out is None:
        out = torch.empty((tokens, 128), dtype=torch.float32, device=sampled.device)
    to, op = _token_rows("tfusion_layer", out)
    if to != tokens or out.shape[-1] != 128:
        raise RuntimeError("tfusion_layer: out must hold [tokens, 128]")
    nxt = torch.empty((tokens, prep.nq), dtype=torch.float32, device=sampled.device) if prep.nq else None
    lib = _lib.load()
    with _on(sampled.device), profiling.span_f("tfusion_layer[%dx128x%d%s]", (tokens, prep.ffn, "+q%d" % prep.nq if prep.nq else "")):
        rc = lib.smos_tfusion_layer(sampled.data_ptr(), query.data_ptr(), qp, prep.stream.data_ptr(), prep.params.data_ptr(),
                                    out.data_ptr(), op, nxt.data_ptr() if nxt is not None else None, prep.nq, tokens, prep.ffn,
                                    prep.eps1, prep.eps2, _stream(sampled))
    _lib.check(rc, "smos_tfusion_layer")
    return out, nxt


class TapWeights:
    """The per-tap matrices of a 3x3 convolution's input channels [c0, c1) in the forms upconv3x3 uses.  Everything
    derived from the weights lives in this object (owned by the engine that folded them) -- never in a cache keyed by
    the weights' address: a freed model's address is handed to the next model."""

    def __init__(self, w, c0, c1):
        cout = w.shape[0]
        self.cout, self.cin = cout, c1 - c0
        self.nk = w[:, c0:c1].permute(2, 3, 0, 1).reshape(9 * cout, c1 - c0).float().contiguous()   # tap-major, t = 3 ky + kx
        self.kn = self.nk.t().contiguous()                      # [Cin, 9*Cout]: the layout the library GEMM is fastest in
        self.zero = torch.zeros(9 * cout, dtype=torch.float32, device=w.device)
        self._conv = None
        self._stream = None

    def stream(self, parts=1):
        """the tap matrices in the operand order of smos_tfusion_project (Cin = 128), cut into `parts` equal output ranges"""
        if self._stream is None:
            self._stream = {}
        if parts not in self._stream:
            n = self.nk.shape[0] // parts
            self._stream[parts] = [tfusion_pack_linear(self.nk[i * n:(i + 1) * n]) for i in range(parts)]
        return self._stream[parts]

    def conv_operand(self):
        if self._conv is None:
            self._conv = conv_prepare(self.nk.view(9 * self.cout, self.cin, 1, 1), 4)
        return self._conv


def upconv_tap_weights(w, c0, c1):
    """w [Cout, Cin, 3, 3] -> TapWeights of input channels [c0, c1)."""
    return TapWeights(w, c0, c1)


# The nine tap products: library GEMM by default; "conv" runs them on the own kernel as one 1x1 convolution with 9*C outputs
# (measured in the step: 236.6 vs 238.8 scans/s -- K = 128 is only four stages per tile, so the epilogue dominates).
# "tf" (default since round 4): the token-wise Linear kernel of the temporal fusion (csrc/tfusion.hip), both sources in one launch:
# 0.26 -> 0.2x ms per step against the library GEMMs ("mm")
_TAP_GEMM = os.environ.get("SMOS_TAP_GEMM", "tf")
_TAP_GEMM_OWN = _TAP_GEMM == "conv"
_TAP_PARTS = int(os.environ.get("SMOS_TAP_PARTS", "3"))      # column ranges per source in the tap-product launch (tuning knob)
# x pass and y pass in one launch (smos_upconv_xy) where the geometry allows; "0": always the two launches (A/B, same results)
_UPCONV_XY = os.environ.get("SMOS_UPCONV_XY", "1") != "0"


def upconv_tap_products(x, wt):
    """z [B*Hs*Ws, 9*C]: the nine tap products W_t x of a channels-last source map x [B,Cin,Hs,Ws] at its own resolution."""
    b, cin, hs, ws = x.shape
    if _TAP_GEMM_OWN and cin % 32 == 0 and (9 * wt.cout) % 128 == 0:
        return conv_cl(x, wt.conv_operand(), None, 0, 9 * wt.cout, (1, 1), mt=4).permute(0, 2, 3, 1).reshape(b * hs * ws, 9 * wt.cout)
    rows = x.permute(0, 2, 3, 1).reshape(b * hs * ws, cin)     # no copy for a dense channels-last map
    # addmm with a zero bias on the [Cin, 9*C] copy of the weights: the library picks a faster kernel for this form than
    # for mm(rows, nk.t()) (tools/ubench_tapgemm.py: 0.182 vs 0.207 ms at 65536 x 128 x 1152); adding 0 changes no value
    return torch.addmm(wt.zero, rows, wt.kn)


def upconv3x3(conv_a, bias, sources, act, out=None):
    """act(conv_a + bias + sum over sources of conv3x3(bilinear_up(x_src), W_src)) without upsampling (csrc/upconv.hip).
    conv_a: channels-last [B,C,Ho,Wo] view (direct conv of the non-upsampled channels); sources: list of (x_cl
    [B,Cin,Hs,Ws] channels-last view with dense rows, tap weights from upconv_tap_weights[, tap products from
    upconv_tap_products: computed here when absent]) -- one or two."""
    _require_cuda("upconv3x3", conv_a, bias, out)
    b, c, ho, wo = conv_a.shape
    if out is None:
        out = conv_a
    lib = _lib.load()
    st = _stream(conv_a)
    if not 1 <= len(sources) <= 2:
        raise RuntimeError("upconv3x3: one or two upsampled sources, got %d" % len(sources))
    fused = _UPCONV_XY and all(lib.smos_upconv_xy_ok(src[0].shape[2], ho) for src in sources)
    zs, ts = [], []
    pre = {}
    if _TAP_GEMM == "tf":
        # the tap products of every source that has none yet, as jobs of the token-wise Linear kernel: one launch for the usual
        # sizes; a product matrix of 2 GiB and more (8 and more concurrent streams) is cut into row ranges below 2 GiB (a job
        # addresses its operands with 32-bit buffer offsets) and takes as many launches of up to 8 jobs as that needs
        todo = [i for i, src in enumerate(sources) if (len(src) < 3 or src[2] is None) and src[0].shape[1] == 128 and 9 * src[1].cout <= 2048]
        if todo:
            # every source's outputs in `parts` column ranges = parts x len(todo) jobs of 64-token blocks:
            # column ranges shorten the last, partly empty round of resident blocks (tools/ubench_taps.py: 0.230 / 0.213 / 0.211 ms for 1 / 2 / 3 ranges; library GEMMs 0.278)
            parts = _TAP_PARTS if len(todo) * _TAP_PARTS <= 8 and (9 * sources[todo[0]][1].cout) % (64 * _TAP_PARTS) == 0 else 1
            jobs = []
            for i in todo:
                x, wt = sources[i][0], sources[i][1]
                rows = x.permute(0, 2, 3, 1).reshape(-1, 128)
                z = torch.empty((rows.shape[0], 9 * wt.cout), dtype=torch.float32, device=x.device)
                pre[i] = z
                n = 9 * wt.cout // parts
                limit = ((1 << 31) - 4096) // (9 * wt.cout * 4) // 64 * 64          # rows whose z (and x) slice stays below 2 GiB
                for r0 in range(0, rows.shape[0], limit):
                    r1 = min(r0 + limit, rows.shape[0])
                    for k, ws_k in enumerate(wt.stream(parts)):
                        jobs.append((rows[r0:r1], ws_k, n, z[r0:r1, k * n:(k + 1) * n]))
            for j0 in range(0, len(jobs), 8):
                tfusion_project(jobs[j0:j0 + 8])
    with _on(conv_a.device):
        for i_src, src in enumerate(sources):
            x, wt = src[0], src[1]
            hs, ws, cin = x.shape[2], x.shape[3], x.shape[1]
            if wt.cin != cin or wt.cout != c:
                raise RuntimeError("upconv3x3: tap weights are for %d -> %d channels, got %d -> %d" % (wt.cin, wt.cout, cin, c))
            z = src[2] if len(src) > 2 and src[2] is not None else (pre[i_src] if i_src in pre else upconv_tap_products(x, wt))
            if tuple(z.shape) != (b * hs * ws, 9 * c) or not z.is_contiguous():
                raise RuntimeError("upconv3x3: tap products must be a contiguous [B*Hs*Ws, 9*C] matrix, got %s" % (tuple(z.shape),))
            zs.append((z, hs, ws))
            if fused:
                continue
            t = torch.empty((b, 3, hs, wo, c), dtype=torch.float32, device=conv_a.device)
            with profiling.span_f("upconv_xpass[%dx%dx%dx%d->%d]", (b, hs, ws, c, wo)):
                _lib.check(lib.smos_upconv_xpass(z.data_ptr(), t.data_ptr(), b, hs, ws, c, wo, st), "smos_upconv_xpass")
            ts.append((t, hs))
        if fused:
            z1, h1, w1 = zs[0]
            z2, h2, w2 = zs[1] if len(zs) > 1 else (None, 0, 0)
            with profiling.span("upconv_xy[%dx%dx%dx%d<-%s]" % (b, ho, wo, c, "+".join("%dx%d" % (h, w) for _, h, w in zs))
                                if profiling.enabled() else None):
                _lib.check(lib.smos_upconv_xy(conv_a.data_ptr(), _cl("upconv3x3", conv_a), bias.data_ptr(), z1.data_ptr(), h1, w1,
                                              z2.data_ptr() if z2 is not None else None, h2, w2, out.data_ptr(), _cl("upconv3x3", out),
                                              b, ho, wo, c, int(act), st), "smos_upconv_xy")
            return out
        t1, h1 = ts[0]
        t2, h2 = ts[1] if len(ts) > 1 else (None, 0)
        with profiling.span_f("upconv_ypass[%dx%dx%dx%d]", (b, ho, wo, c)):
            _lib.check(lib.smos_upconv_ypass(conv_a.data_ptr(), _cl("upconv3x3", conv_a), bias.data_ptr(), t1.data_ptr(), h1,
                                             t2.data_ptr() if t2 is not None else None, h2, out.data_ptr(), _cl("upconv3x3", out),
                                             b, ho, wo, c, int(act), st), "smos_upconv_ypass")
    return out


STEM_TAPS = (1, 2, 2, 4)      # 3x3 taps that reach an output pixel under stride 2, per parity class (y&1)*2 + (x&1)


def stem_prepare_weights(wa, wp):
    """wa [Cout,Cin,3,3], wp [Cout,Cin,1,1] (BatchNorm folded) -> 4 tensors in the MFMA operand order smos_stem_gemm
    expects: [(taps+1), Cin/2, 64], entry (mt, s, lane) = W[mt*32 + (lane & 31)][(lane >> 5) * (Cin/2) + s]."""
    cout, cin = wa.shape[0], wa.shape[1]
    out = []
    for cls in range(4):
        ey, ex = cls >> 1, cls & 1
        kys, kxs = ((0, 2) if ey else (1,)), ((0, 2) if ex else (1,))
        w = torch.cat([wa[:, :, ky, kx] for ky in kys for kx in kxs] + [wp[:, :, 0, 0]], 0).float()      # [(taps+1)*Cout, Cin]
        km = w.shape[0] // 32
        out.append(w.view(km, 32, 2, cin // 2).permute(0, 3, 2, 1).reshape(km, cin // 2, 64).contiguous())
    return out


_stem_ws = {}


def _stream_workspace(tag, shape, dtype, device, zero=False):
    """Scratch of the sparse first stage, one flat buffer per (device, HIP stream, tag), grown to the largest request
    seen: consecutive frames on a stream reuse it in stream order instead of holding a fresh allocation per frame that
    the host has enqueued ahead of the GPU.  The returned view is valid until the next request with the same tag on the
    same stream (callers consume it within the frame).  zero=True: zero-filled when (re)allocated -- for buffers whose
    users leave them zero (the occupancy flags).  ``release_stream_workspaces`` frees them."""
    key = (str(device), _raw_stream(_dev_index(device)), tag, dtype, _ws_namespace)
    need = 1
    for d in shape:
        need *= int(d)
    buf = _stem_ws.get(key)
    if buf is None or buf.numel() < need:
        buf = _stem_ws[key] = (torch.zeros if zero else torch.empty)(need, dtype=dtype, device=device)
    return buf[:need].view(shape)


class BlockScratch(dict):
    """{(HIP stream, namespace): buffer} of one ChannelAtt block (engine._block_ws); a dict that can be weakly referenced,
    so the registry below does not keep a dead engine's scratch alive."""
    __slots__ = ("__weakref__", "device")
    __hash__ = object.__hash__            # identity: the registry is a WeakSet
    __eq__ = object.__eq__


import weakref as _weakref

_block_ws_tables = _weakref.WeakSet()
_owner_tokens = iter(range(1, 1 << 62))


def new_workspace_owner():
    """A token for scratch namespaces that is never handed out twice (id(obj) can be: a later runner allocated at a freed
    runner's address would alias its namespaces)."""
    return next(_owner_tokens)


def new_block_scratch(device=None):
    table = BlockScratch()
    table.device = None if device is None else str(device)
    _block_ws_tables.add(table)
    return table


def release_stream_workspaces(device=None, stream=None, owner=None):
    """Drops the per-stream scratch: all of it, one device's, one HIP stream's, or (owner = the token a graph runner put
    into its namespace, ``new_workspace_owner``) what that runner's captured graphs use -- the sparse-stage buffers, the
    scatter's flag words and the channel-attention blocks' plane sums alike.  StreamRunner.close() calls this."""
    def owned(ns):
        return isinstance(ns, tuple) and len(ns) > 1 and ns[1] == owner

    def graph_ns(ns):
        # scratch baked into some runner's captured graphs: only that runner (the owner path) may drop it -- torch hands
        # pooled streams to several runners, so a stream handle alone does not say whose scratch an entry is
        return isinstance(ns, tuple) and len(ns) > 1 and ns[0] == "graph"

    def selected(dev, st, ns):
        if owner is not None:
            return owned(ns)
        if (device is not None or stream is not None) and graph_ns(ns):
            return False
        return (device is None or dev is None or dev == str(device)) and (stream is None or st == stream)
    for key in list(_stem_ws):                       # (device, stream, tag, dtype, namespace)
        if selected(key[0], key[1], key[4]):
            del _stem_ws[key]
    for key in list(_flag_ws):                       # (device, stream, namespace)
        if selected(str(key[0]), key[1], key[2]):
            del _flag_ws[key]
    for table in list(_block_ws_tables):             # engine._block_ws: {(stream, namespace): buffer} per ChannelAtt block
        for key in list(table):
            if selected(table.device, key[0], key[1]):
                del table[key]


class StemPlan:
    """Occupancy of the input grid of one frame (csrc/stem.hip): which cells hold points, in which compact row.
    meta (device, 12 x int32): [4..7] first row of each parity class, [8..11] one past its last row; meta[11] = rows."""

    def __init__(self, b, h, w, row_cell, row_of, meta, rows):
        self.b, self.h, self.w = b, h, w
        self.row_cell, self.row_of, self.meta, self.rows = row_cell, row_of, meta, rows

    def class_rows(self):
        """device tensor [4]: occupied cells per parity class (y & 1) * 2 + (x & 1)"""
        return self.meta[8:12] - self.meta[4:8]


def stem_plan(coord, h, w, row_floats=0):
    """coord [B,T,N,K(,1)] float32 contiguous -> StemPlan: marks the cells the points fall into and compacts them (rows
    ordered by parity class, sample, position) in two launches (mark, single-pass scan).  row_floats > 0 also provides the
    compact row table ``plan.rows`` [min(B*H*W, B*T*N), row_floats] with its existing rows zero-filled (by the scan kernel).
    Everything stays on the device; every buffer is per-stream scratch, valid until the next plan on the stream."""
    _require_cuda("stem_plan", coord)
    if coord.dtype != torch.float32 or not coord.is_contiguous():
        raise RuntimeError("stem_plan: coord must be contiguous float32")
    b, t, n, k = coord.shape[:4]
    dev = coord.device
    cells = b * h * w
    lib = _lib.load()
    words = int(lib.smos_stem_scan_state_words(cells))
    if words < 0:
        raise RuntimeError("stem_plan: grid too large for the scan")
    flags = _stream_workspace("stem_flags", (cells,), torch.int32, dev, zero=True)      # left all-zero by the scan
    state = _stream_workspace("stem_scan_state", (words,), torch.int64, dev)
    row_cell = _stream_workspace("stem_row_cell", (cells,), torch.int32, dev)
    row_of = _stream_workspace("stem_row_of", (cells,), torch.int32, dev)
    meta = _stream_workspace("stem_meta", (12,), torch.int32, dev, zero=True)
    rows = None
    if row_floats:
        # capacity: an occupied cell holds at least one point, so min(cells, points) rows always suffice
        rows = _stream_workspace("stem_rows", (min(cells, b * t * n), row_floats), torch.float32, dev)
    st = _stream(coord)
    with _on(dev), profiling.span_f("stem_mark+scan[%dx%dx%d]", (b, h, w)):
        _lib.check(lib.smos_stem_mark(coord.data_ptr(), k, b, t, n, h, w, flags.data_ptr(), state.data_ptr(), st), "smos_stem_mark")
        _lib.check(lib.smos_stem_scan(flags.data_ptr(), b, h, w, state.data_ptr(), row_cell.data_ptr(), row_of.data_ptr(),
                                      meta.data_ptr(), rows.data_ptr() if rows is not None else None, row_floats, st), "smos_stem_scan")
    return StemPlan(b, h, w, row_cell, row_of, meta, rows)


def pointnet_scatter_rows(xyzi, coord, w1, b1, w2, b2, plan, pts_out=None, n_live=None):
    """pointnet_scatter into the COMPACT row table of `plan` (stem_plan(..., row_floats=T*64)): returns plan.rows, a view of
    this stream's scratch (valid until the next plan on the stream); only the first plan.meta[11] rows exist (zero-filled by
    the plan's scan) -- the dense grid is never materialised."""
    _require_cuda("pointnet_scatter_rows", xyzi, coord, w1, b1, w2, b2, pts_out, n_live)
    if n_live is not None and (n_live.dtype != torch.int32 or n_live.numel() < 1):
        raise RuntimeError("pointnet_scatter_rows: n_live must be a device int32 tensor")
    b, t, cin, n = xyzi.shape[:4]
    k = coord.shape[3]
    if not (xyzi.is_contiguous() and coord.is_contiguous()):
        raise RuntimeError("pointnet_scatter_rows: xyzi and coord must be contiguous")
    cout = w2.shape[0]
    rows = plan.rows
    if rows is None or rows.shape[1] != t * cout:
        raise RuntimeError("pointnet_scatter_rows: the plan carries no row table of %d floats per row" % (t * cout))
    po_b = po_n = 0
    if pts_out is not None:
        po_b, po_n = _rows("pointnet_scatter_rows", pts_out, cout)
    lib = _lib.load()
    st = _stream(xyzi)
    # the span holds exactly ONE kernel, so its HIP-event mean is comparable with rocprofv3's per-kernel mean
    with _on(xyzi.device), profiling.span_f("pointnet_scatter[%dx%dx%d->%dx%d]", (b, t, n, plan.h, plan.w)):
        rc = lib.smos_pointnet_scatter_rows_live(xyzi.data_ptr(), coord.data_ptr(), k, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                 b2.data_ptr(), rows.data_ptr(), plan.row_of.data_ptr(),
                                                 pts_out.data_ptr() if pts_out is not None else None, po_b, po_n, b, t, n,
                                                 plan.h, plan.w, cin, w1.shape[0], cout,
                                                 n_live.data_ptr() if n_live is not None else None, st)
    _lib.check(rc, "smos_pointnet_scatter_rows_live")
    return rows


def sparse_downsample(src, plan, wprep, bias, compact, out=None):
    """DownSample2D(stride 2) on the occupied cells only (csrc/stem.hip): per-class MFMA GEMM + per-pixel assembly.
    src: the compact row table of pointnet_scatter_rows (compact=True) or the dense channels-last grid [B,H,W,Cin]
    (compact=False).  Returns the channels-last [B,Cout,H/2,W/2] view.  No host synchronisation."""
    _require_cuda("sparse_downsample", src, bias, out, *wprep)
    b, h, w = plan.b, plan.h, plan.w
    cin = src.shape[-1]
    dev = src.device
    if src.dtype != torch.float32 or not src.is_contiguous() or (not compact and src.numel() != b * h * w * cin):
        raise RuntimeError("sparse_downsample: src must be contiguous float32 with B*H*W rows of Cin")
    if compact and src.dim() != 2:
        raise RuntimeError("sparse_downsample: the compact row table is [rows, Cin]")
    cout = bias.shape[0]
    per = b * (h // 2) * (w // 2)
    ys = [_stream_workspace("stem_y%d" % k, (per, (taps + 1) * cout), torch.float32, dev)                # worst-case capacity
          for k, taps in enumerate(STEM_TAPS)]
    if out is None:
        out = empty_cl(b, cout, (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1, dev)
    y_ptrs = (ctypes.c_void_p * 4)(*[y.data_ptr() for y in ys])
    w_ptrs = (ctypes.c_void_p * 4)(*[wt.data_ptr() for wt in wprep])
    lib = _lib.load()
    st = _stream(src)
    tag = "[%dx%dx%dx%d]" % (b, h, w, cin) if profiling.enabled() else ""
    with _on(dev):
        with profiling.span("stem_gemm" + tag):
            _lib.check(lib.smos_stem_gemm(src.data_ptr(), None if compact else plan.row_cell.data_ptr(), plan.meta.data_ptr(),
                                          w_ptrs, y_ptrs, cin, cout, st), "smos_stem_gemm")
        with profiling.span("stem_epilogue" + tag):
            _lib.check(lib.smos_stem_epilogue(y_ptrs, plan.meta.data_ptr(), plan.row_of.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                              _cl("sparse_downsample", out), b, h, w, cout, st), "smos_stem_epilogue")
    return out


def gather_scatter(grid, gcoord, gscale, scoord=None, sscale=None, out=None, pts_out=None):
    """grid [B,C,Hg,Wg] any strides; gcoord / scoord [B,N,K] contiguous; out [B,Ho,Wo,C] zero-filled
    channels-last (or None); pts_out [B,N,C] rows (or None)."""
    _require_cuda("gather_scatter", grid, gcoord, scoord, out, pts_out)
    b, c, hg, wg = grid.shape
    n, kg = gcoord.shape[1], gcoord.shape[2]
    ho = wo = ks = 0
    if out is not None:
        if not out.is_contiguous() or out.shape[3] != c:
            raise RuntimeError("gather_scatter: out must be contiguous channels-last [B,Ho,Wo,C]")
        ho, wo, ks = out.shape[1], out.shape[2], scoord.shape[2]
    po_b = po_n = 0
    if pts_out is not None:
        po_b, po_n = _rows("gather_scatter", pts_out, c)
    lib = _lib.load()
    label = "gather_scatter[%dx%dx%dx%d->%d->%dx%d]" % (b, c, hg, wg, n, ho, wo)
    with _on(grid.device), profiling.span(label):
        rc = lib.smos_gather_scatter(grid.data_ptr(), _lib.i64_array(grid.stride()), gcoord.data_ptr(), kg,
                                     _lib.f32_array(gscale), scoord.data_ptr() if out is not None else None, ks,
                                     _lib.f32_array(sscale) if out is not None else None,
                                     out.data_ptr() if out is not None else None,
                                     pts_out.data_ptr() if pts_out is not None else None, po_b, po_n, b, c, hg, wg, n, ho, wo,
                                     _stream(grid))
    _lib.check(rc, "smos_gather_scatter")


def nhwc_to_nchw(src, dst):
    """src [B,H,W,C] contiguous -> dst [B,C,H,W] view with contiguous planes (e.g. a channel slice)."""
    _require_cuda("nhwc_to_nchw", src, dst)
    b, h, w, c = src.shape
    db, dc, hw = _planes("nhwc_to_nchw", dst)
    if not src.is_contiguous() or tuple(dst.shape) != (b, c, h, w):
        raise RuntimeError("nhwc_to_nchw: shape mismatch %s -> %s" % (tuple(src.shape), tuple(dst.shape)))
    lib = _lib.load()
    with _on(src.device):
        rc = lib.smos_nhwc_to_nchw(src.data_ptr(), dst.data_ptr(), db, dc, b, c, hw, _stream(src))
    _lib.check(rc, "smos_nhwc_to_nchw")
    return dst


# ---------------------------------------------------------------------------------------------
# channels-last engine kernels: tensors are logical [B, C, H, W] with channels_last strides (possibly a channel
# slice of a wider channels-last buffer)
# ---------------------------------------------------------------------------------------------
_WINO1D_KERNELS = ((5, 3), (7, 3), (3, 5), (3, 7))


def conv_wino1d_ok(kernel, stride, cin, cout, residual=None, chan_sums=None):
    """Shapes smos_conv_wino1d_cl covers: 5x3 / 7x3 / 3x5 / 3x7, stride 1, "same" padding, Cin and Cout multiples of 16, no
    residual / channel sums (the Unbalance branches have neither)."""
    return (tuple(kernel) in _WINO1D_KERNELS and stride == 1 and cin % 16 == 0 and cout % 16 == 0 and residual is None and
            chan_sums is None)


def conv_wino1d_prepare(w, mb):
    """w [Cout, Cin, KH, KW] (BatchNorm folded; one extent 3, the other 5 or 7) -> the weight block of smos_conv_wino1d_cl:
    U = G g along the 3-tap axis per (cout, cin, long-axis tap) in float64, rounded once to float32, in MFMA operand order
    [cout tile][cin chunk of 16][k-step i][long tap][mb][lane][position], lane = q * 16 + m holding the four positions of
    w[ct*16*mb + mb_i*16 + m][chunk*16 + 4*q + i] (include/smos.h)."""
    cout, cin, kh, kw = w.shape
    if (kh, kw) not in _WINO1D_KERNELS or cin % 16 or mb not in (1, 2) or cout % (16 * mb):
        raise RuntimeError("conv_wino1d_prepare: 5x3 / 7x3 / 3x5 / 3x7 kernel, Cin %% 16 == 0 and Cout %% (16 * mb) == 0 required "
                           "(got %s, mb=%d)" % (tuple(w.shape), mb))
    g = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    wl = w.double() if kw == 3 else w.double().transpose(2, 3)      # [Cout, Cin, long tap, short tap]
    kl = wl.shape[2]
    u = (wl @ g.t()).float()                                        # [Cout, Cin, long tap, position]
    #          ct               mb_i m   chunk     q  i  kl  pos
    v = u.reshape(cout // (16 * mb), mb, 16, cin // 16, 4, 4, kl, 4)
    v = v.permute(0, 3, 5, 6, 1, 4, 2, 7)                           # -> [ct, chunk, i, kl, mb_i, q, m, pos]
    return v.reshape(-1).contiguous()


def conv_wino1d_cl(x, wprep, bias, act, cout, kernel, mb=2, out=None):
    """act(conv(x) + bias) (stride 1, "same" padding) for a 5x3 / 7x3 / 3x5 / 3x7 kernel on channels-last [B,C,H,W] views in
    one launch of the 1-D Winograd F(2, 3) kernel (csrc/conv_wino1d.hip).  wprep = conv_wino1d_prepare(w, mb)."""
    _require_cuda("conv_wino1d_cl", x, wprep, bias, out)
    b, cin, h, w = x.shape
    kh, kw = kernel
    if not conv_wino1d_ok(kernel, 1, cin, cout) or mb not in (1, 2) or cout % (16 * mb) or wprep.numel() != 4 * max(kh, kw) * cout * cin:
        raise RuntimeError("conv_wino1d_cl: unsupported shape %s -> %d k%dx%d (mb=%d)" % (tuple(x.shape), cout, kh, kw, mb))
    if out is None:
        out = empty_cl(b, cout, h, w, x.device)
    elif tuple(out.shape) != (b, cout, h, w):
        raise RuntimeError("conv_wino1d_cl: out has shape %s" % (tuple(out.shape),))
    lib = _lib.load()
    label = "conv_cl[%dx%dx%dx%d->%dx%dx%dk%dx%d]" % (b, cin, h, w, cout, h, w, kh, kw) if profiling.enabled() else None
    args = (x.data_ptr(), _cl("conv_wino1d_cl", x), wprep.data_ptr(), bias.data_ptr() if bias is not None else None,
            out.data_ptr(), _cl("conv_wino1d_cl", out), b, h, w, cin, cout, kh, kw, int(mb), int(act))
    fn = lib.smos_conv_wino1d_cl
    with _on(x.device), profiling.span(label, "conv_wino1d"):
        rc = fn(*args, _stream(x))
    _lib.check(rc, "smos_conv_wino1d_cl")
    if label is not None and profiling._replay_label == label:
        keep = (x, wprep, bias, out)

        def again(keep=keep):
            with _on(keep[0].device), profiling.span(label, "conv_wino1d"):
                _lib.check(fn(*args, _stream(keep[0])), "smos_conv_wino1d_cl")
        profiling.offer_replay(label, again)
    return out


def empty_cl(b, c, h, w, device, zero=False):
    """logical [B, C, H, W] with channels-last strides (one allocation call; the explicit strides keep C innermost also where a
    size-1 dimension would let torch's channels_last format choose others)"""
    if zero:
        return torch.zeros((b, h, w, c), dtype=torch.float32, device=device).permute(0, 3, 1, 2)
    return torch.empty_strided((b, c, h, w), (h * w * c, 1, w * c, c), dtype=torch.float32, device=device)


def _cl(name, t):
    """row pitch of a channels-last [B,C,H,W] view (C innermost, rows back to back over B*H*W)."""
    b, c, h, w = t.shape
    s0, s1, s2, pitch = t.stride()
    if s1 != 1 or s2 != w * pitch or (b > 1 and s0 != h * w * pitch) or pitch < c:
        raise RuntimeError("%s: expected a channels-last [B,C,H,W] view, got shape %s strides %s" % (name, tuple(t.shape), t.stride()))
    return pitch


def bias_act_cl(x, bias, act, out=None, residual=None):
    _require_cuda("bias_act_cl", x, bias, out, residual)
    b, c, h, w = x.shape
    if out is None:
        out = empty_cl(b, c, h, w, x.device)
    lib = _lib.load()
    with _on(x.device):
        rc = lib.smos_bias_act_cl(x.data_ptr(), _cl("bias_act_cl", x), bias.data_ptr() if bias is not None else None,
                                  residual.data_ptr() if residual is not None else None,
                                  _cl("bias_act_cl", residual) if residual is not None else 0, out.data_ptr(),
                                  _cl("bias_act_cl", out), b * h * w, c, act, _stream(x))
    _lib.check(rc, "smos_bias_act_cl")
    return out


def downsample_epilogue_cl(a, p, bias, stride, out=None):
    _require_cuda("downsample_epilogue_cl", a, p, bias, out)
    b, c, h, w = p.shape
    if out is None:
        out = empty_cl(b, c, a.shape[2], a.shape[3], a.device)
    lib = _lib.load()
    with _on(a.device):
        rc = lib.smos_downsample_epilogue_cl(a.data_ptr(), _cl("downsample_epilogue_cl", a), p.data_ptr(),
                                             _cl("downsample_epilogue_cl", p), bias.data_ptr(), out.data_ptr(),
                                             _cl("downsample_epilogue_cl", out), b, c, h, w, stride, _stream(a))
    _lib.check(rc, "smos_downsample_epilogue_cl")
    return out


def pool_branch_prepare(w):
    """1x1 pool-branch weights [Cout, Cin(, 1, 1)] (BatchNorm folded) -> the operand block of smos_downsample_pool_branch."""
    w2 = w.reshape(w.shape[0], -1)
    if w2.shape[0] != w2.shape[1] or w2.shape[0] not in (32, 64, 128):
        raise RuntimeError("pool_branch_prepare: Cin == Cout in {32, 64, 128} expected, got %s" % (tuple(w.shape),))
    return _tf_pairs(w2).reshape(-1).contiguous()


def pool_branch_ok(cin, cout, stride):
    """Shapes smos_downsample_pool_branch covers."""
    return cin == cout and cin in (32, 64, 128) and stride in (1, 2) and not (cin == 128 and stride == 1)


def downsample_pool_branch(x, wpairs, a, bias, stride, out=None):
    """relu(a + bias + maxpool3x3(conv1x1(x); stride, pad 1)) on channels-last views in one launch (csrc/downsample.hip): the
    DownSample2D tail with its pool branch computed on the fly; wpairs = pool_branch_prepare(w)."""
    _require_cuda("downsample_pool_branch", x, wpairs, a, bias, out)
    b, c, h, w = x.shape
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    if tuple(a.shape) != (b, c, ho, wo) or wpairs.numel() != c * c or not pool_branch_ok(c, c, stride):
        raise RuntimeError("downsample_pool_branch: unsupported shapes x %s a %s stride %d" % (tuple(x.shape), tuple(a.shape), stride))
    if out is None:
        out = empty_cl(b, c, ho, wo, x.device)
    elif tuple(out.shape) != (b, c, ho, wo):
        raise RuntimeError("downsample_pool_branch: out has shape %s" % (tuple(out.shape),))
    lib = _lib.load()
    with _on(x.device), profiling.span_f("downsample_pool_branch[%dx%dx%dx%d/s%d]", (b, c, h, w, stride)):
        rc = lib.smos_downsample_pool_branch(x.data_ptr(), _cl("downsample_pool_branch", x), wpairs.data_ptr(), a.data_ptr(),
                                             _cl("downsample_pool_branch", a), bias.data_ptr(), out.data_ptr(),
                                             _cl("downsample_pool_branch", out), b, h, w, c, c, int(stride), _stream(x))
    _lib.check(rc, "smos_downsample_pool_branch")
    return out


def channel_gate_residual_cl(y, bias, w1, b1, w2, b2, xres, ws, out=None):
    _require_cuda("channel_gate_residual_cl", y, bias, w1, b1, w2, b2, xres, ws, out)
    b, c, h, w = y.shape
    if out is None:
        out = empty_cl(b, c, h, w, y.device)
    lib = _lib.load()
    with _on(y.device):
        rc = lib.smos_channel_gate_residual_cl(y.data_ptr(), _cl("channel_gate_residual_cl", y), bias.data_ptr(), w1.data_ptr(),
                                               b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), xres.data_ptr(),
                                               _cl("channel_gate_residual_cl", xres), out.data_ptr(),
                                               _cl("channel_gate_residual_cl", out), ws.data_ptr(), ws.numel(), b, c, w1.shape[0],
                                               h * w, _stream(y))
    _lib.check(rc, "smos_channel_gate_residual_cl")
    return out


def channel_gate_apply_cl(y, bias, w1, b1, w2, b2, xres, chan_sums, gate_ws, out=None):
    """ChannelAtt + residual + ReLU from the channel sums conv_cl left in chan_sums [B, chunks, C]: gate MLP, then
    out = relu((y + bias) * gate + xres).  gate_ws: >= B*C floats of scratch."""
    _require_cuda("channel_gate_apply_cl", y, bias, w1, b1, w2, b2, xres, chan_sums, gate_ws, out)
    b, c, h, w = y.shape
    if chan_sums.dim() != 3 or chan_sums.shape[0] != b or chan_sums.shape[2] != c or not chan_sums.is_contiguous() or gate_ws.numel() < b * c:
        raise RuntimeError("channel_gate_apply_cl: chan_sums must be contiguous [B, chunks, C] and gate_ws hold B*C floats")
    if out is None:
        out = empty_cl(b, c, h, w, y.device)
    lib = _lib.load()
    with _on(y.device):
        rc = lib.smos_channel_gate_apply_cl(y.data_ptr(), _cl("channel_gate_apply_cl", y), bias.data_ptr(), w1.data_ptr(),
                                            b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), xres.data_ptr(),
                                            _cl("channel_gate_apply_cl", xres), out.data_ptr(), _cl("channel_gate_apply_cl", out),
                                            chan_sums.data_ptr(), chan_sums.shape[1], gate_ws.data_ptr(), b, c, w1.shape[0], h * w,
                                            _stream(y))
    _lib.check(rc, "smos_channel_gate_apply_cl")
    return out


def upsample_concat_cl(sources, size):
    import ctypes
    _require_cuda("upsample_concat_cl", *sources)
    b = sources[0].shape[0]
    ctot = sum(s.shape[1] for s in sources)
    out = empty_cl(b, ctot, size[0], size[1], sources[0].device)
    ptrs = (ctypes.c_void_p * len(sources))(*[s.data_ptr() for s in sources])
    lib = _lib.load()
    with _on(out.device):
        rc = lib.smos_upsample_concat_cl(ptrs, _lib.i64_array([s.shape[1] for s in sources]),
                                         _lib.i64_array([s.shape[2] for s in sources]),
                                         _lib.i64_array([s.shape[3] for s in sources]),
                                         _lib.i64_array([_cl("upsample_concat_cl", s) for s in sources]), len(sources),
                                         out.data_ptr(), b, size[0], size[1], _stream(out))
    _lib.check(rc, "smos_upsample_concat_cl")
    return out


def zero_views_cl(views):
    """Zero fill of up to four channels-last [B,C,H,W] views (dense maps or channel slices of wider ones) in one launch."""
    if not 1 <= len(views) <= 4:
        raise RuntimeError("zero_views_cl: 1 .. 4 views")
    _require_cuda("zero_views_cl", *views)
    k = len(views)
    ptrs, rows, rf, pit = (ctypes.c_void_p * k)(), (ctypes.c_int64 * k)(), (ctypes.c_int64 * k)(), (ctypes.c_int64 * k)()
    for i, v in enumerate(views):
        b, c, h, w = v.shape
        if v.dtype != torch.float32 or v.device != views[0].device:
            raise RuntimeError("zero_views_cl: float32 views on one device")
        ptrs[i], rows[i], rf[i], pit[i] = v.data_ptr(), b * h * w, c, _cl("zero_views_cl", v)
    with _on(views[0].device):
        rc = _lib.load().smos_zero_views_cl(k, ptrs, rows, rf, pit, _stream(views[0]))
    if rc:
        _lib.check(rc, "smos_zero_views_cl")


def _coord_view(name, t, b, n):
    """(floats per point, floats per sample) of a coordinate view [B, N, K >= 2] whose last dimension is dense -- contiguous or a
    slice of a wider tensor such as pcds_coord[:, 0, :, :, 0] of the reference's [B, T, N, 3, 1] layout."""
    if t.dtype != torch.float32 or t.dim() != 3 or t.shape[0] != b or (n is not None and t.shape[1] != n) or t.shape[2] < 2:
        raise RuntimeError("%s: coordinates must be float32 [B, N, K >= 2], got %s %s" % (name, tuple(t.shape), t.dtype))
    s0, s1, s2 = t.stride()
    if s2 != 1 or s1 < 2 or s0 < 0:
        raise RuntimeError("%s: coordinate view with strides %s (the last dimension must be dense)" % (name, t.stride()))
    return s1, s0


def gather_scatter_cl(grid, gcoord, gscale, scoord=None, sscale=None, out=None, pts_out=None, n_live=None):
    """grid: channels-last [B,C,Hg,Wg] view; out: channels-last [B,C,Ho,Wo] view, zero-filled (or None);
    pts_out: [B,N,C] rows (or None).  gcoord / scoord: [B,N,K>=2] float32 views with a dense last dimension (strided slices of
    the reference's coordinate tensors are taken as they lie).  n_live (device int32 tensor, optional): real points at the front
    of every sample; point rows of the padding tail are not written."""
    _require_cuda("gather_scatter_cl", grid, gcoord, scoord, out, pts_out, n_live)
    if n_live is not None and (n_live.dtype != torch.int32 or n_live.numel() < 1):
        raise RuntimeError("gather_scatter_cl: n_live must be a device int32 tensor")
    b, c, hg, wg = grid.shape
    n = gcoord.shape[1]
    kg, gbs = _coord_view("gather_scatter_cl", gcoord, b, None)
    ho = wo = ks = op = sbs = 0
    if out is not None:
        ho, wo, op = out.shape[2], out.shape[3], _cl("gather_scatter_cl", out)
        ks, sbs = _coord_view("gather_scatter_cl", scoord, b, n)
    po_b = po_n = 0
    if pts_out is not None:
        po_b, po_n = _rows("gather_scatter_cl", pts_out, c)
    lib = _lib.load()
    label = ("gather_scatter_cl[%dx%dx%dx%d->%d->%dx%d%s]" % (b, c, hg, wg, n, ho, wo,
                                                           "+pts" if pts_out is not None and out is not None else "")
             if profiling.enabled() else None)
    with _on(grid.device), profiling.span(label):
        rc = lib.smos_gather_scatter_cl_view(grid.data_ptr(), _cl("gather_scatter_cl", grid), gcoord.data_ptr(), kg, gbs,
                                             _lib.f32_array(gscale), scoord.data_ptr() if out is not None else None, ks, sbs,
                                             _lib.f32_array(sscale) if out is not None else None,
                                             out.data_ptr() if out is not None else None, op,
                                             pts_out.data_ptr() if pts_out is not None else None, po_b, po_n, b, c, hg, wg, n, ho, wo,
                                             n_live.data_ptr() if n_live is not None else None, _stream(grid))
    _lib.check(rc, "smos_gather_scatter_cl_view")
