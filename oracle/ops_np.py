"""numpy restatements of the four native primitives of the path -- TEST INFRASTRUCTURE.

Each function cites the reference code it follows.  Pinned by tests/golden/ops_*.npz (outputs of the
real reference, see tests/golden/make_golden.py) in tests/test_oracle_golden.py.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------
# deep_point.VoxelMaxPool
# ---------------------------------------------------------------------------------------------
def voxel_cell_index(ind, out_size, scale):
    """Flat spatial cell per point, -1 when dropped.

    Follows deep_point/src/point_deep_cuda_kernel.cu:36-52 (== point_deep.cpp:35-45):
    ``int64(float(coord_d) * scale_d)`` truncates toward zero, a point is valid iff
    0 <= cell_d < out_size_d for every d.  ind: (BS, N, D) float, scale: float32 per dim.
    Returns int64 (BS, N) flat offset over the spatial dims only (row-major D1..Dn).
    """
    ind = np.asarray(ind)
    bs, n, d = ind.shape
    flat = np.zeros((bs, n), dtype=np.int64)
    valid = np.ones((bs, n), dtype=bool)
    stride = 1
    strides = []
    for s in reversed(out_size):
        strides.append(stride)
        stride *= int(s)
    strides = strides[::-1]
    for k in range(d):
        prod = ind[:, :, k].astype(np.float32) * np.float32(scale[k])
        with np.errstate(invalid="ignore"):
            cell = np.trunc(prod.astype(np.float64))
        ok = np.isfinite(prod) & (cell >= 0) & (cell < out_size[k])
        valid &= ok
        flat += np.where(ok, cell, 0).astype(np.int64) * strides[k]
    return np.where(valid, flat, -1)


def voxel_maxpool_fwd(feat, ind, out_size, scale):
    """feat (BS,C,N) , ind (BS,N,D) -> out (BS,C,*out_size), voxel_max_idx (BS,N) int64.

    point_deep.cpp:19-88 / point_deep_cuda_kernel.cu:56-99: an occupied cell ends up with the maximum
    of its members (the Init pass stores one member first so that negative maxima survive the
    zero-initialised output, then a `<` / atomMax pass); an empty cell keeps the caller's zero.
    voxel_max_idx holds bs*stride0 + spatial offset (channel 0) or -1 (cuda_kernel.cu:34-51).
    """
    feat = np.asarray(feat)
    bs, c, n = feat.shape
    cells = int(np.prod(out_size))
    flat = voxel_cell_index(ind, out_size, scale)
    out = np.zeros((bs, c, cells), dtype=feat.dtype)
    for b in range(bs):
        keep = flat[b] >= 0
        tgt = flat[b][keep]
        if tgt.size == 0:
            continue
        buf = np.full((c, cells), -np.inf, dtype=feat.dtype)
        for ch in range(c):
            np.maximum.at(buf[ch], tgt, feat[b, ch][keep])
        occ = np.zeros(cells, dtype=bool)
        occ[tgt] = True
        out[b][:, occ] = buf[:, occ]
    idx = np.where(flat >= 0, flat + np.arange(bs, dtype=np.int64)[:, None] * (c * cells), -1)
    return out.reshape((bs, c) + tuple(int(s) for s in out_size)), idx


def voxel_maxpool_bwd(feat, ind, out, grad_out, out_size, scale):
    """point_deep_cuda_kernel.cu:109-132: every point whose value equals its cell's max gets the
    cell's gradient (ties all receive it), everything else stays zero."""
    feat = np.asarray(feat)
    bs, c, n = feat.shape
    cells = int(np.prod(out_size))
    flat = voxel_cell_index(ind, out_size, scale)
    out = np.asarray(out).reshape(bs, c, cells)
    grad_out = np.asarray(grad_out).reshape(bs, c, cells)
    grad = np.zeros_like(feat)
    for b in range(bs):
        keep = flat[b] >= 0
        tgt = flat[b][keep]
        hit = out[b][:, tgt] == feat[b][:, keep]
        g = np.where(hit, grad_out[b][:, tgt], 0)
        grad[b][:, keep] = g
    return grad


# ---------------------------------------------------------------------------------------------
# networks/backbone.py::BilinearSample  (F.grid_sample, bilinear, zeros, align_corners=True)
# ---------------------------------------------------------------------------------------------
def bilinear_pixel_coords(coord, scale, h, w):
    """The float32 op sequence of backbone.py:467-468 followed by PyTorch's align_corners=True
    un-normalisation ((g + 1) / 2) * (size - 1): returns (iy, ix) float32 pixel positions.
    Row (y) comes from coord[..., 0], column (x) from coord[..., 1]."""
    f = np.float32
    gx = (f(2) * coord[..., 1].astype(f) * f(scale[1]) / f(w - 1)) - f(1)
    gy = (f(2) * coord[..., 0].astype(f) * f(scale[0]) / f(h - 1)) - f(1)
    ix = ((gx + f(1)) / f(2)) * f(w - 1)
    iy = ((gy + f(1)) / f(2)) * f(h - 1)
    return iy.astype(f), ix.astype(f)


def bilinear_sample(grid_feat, coord, scale):
    """grid_feat (B,C,H,W), coord (B,N,2) -> (B,C,N); taps outside the map contribute zero."""
    grid_feat = np.asarray(grid_feat, dtype=np.float32)
    b, c, h, w = grid_feat.shape
    iy, ix = bilinear_pixel_coords(np.asarray(coord), scale, h, w)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    out = np.zeros((b, c, coord.shape[1]), dtype=np.float32)
    for bi in range(b):
        acc = np.zeros((c, coord.shape[1]), dtype=np.float32)
        for yy, xx, ww in ((y0, x0, w_nw), (y0, x1, w_ne), (y1, x0, w_sw), (y1, x1, w_se)):
            yb, xb, wb = yy[bi], xx[bi], ww[bi]
            ok = (yb >= 0) & (yb <= h - 1) & (xb >= 0) & (xb <= w - 1) & np.isfinite(yb) & np.isfinite(xb)
            yi = np.where(ok, yb, 0).astype(np.int64)
            xi = np.where(ok, xb, 0).astype(np.int64)
            tap = grid_feat[bi][:, yi, xi]
            acc += np.where(ok[None, :], tap * wb[None, :].astype(np.float32), np.float32(0))
        out[bi] = acc
    return out


# ---------------------------------------------------------------------------------------------
# deformattn: ms_deformable_im2col (forward)
# ---------------------------------------------------------------------------------------------
def msda_forward(value, spatial_shapes, level_start_index, loc, attn, dtype=None):
    """value (N,S,M,D), loc (N,Lq,M,L,P,2) (x,y in [0,1]), attn (N,Lq,M,L,P) -> (N,Lq,M*D).

    deformattn/src/cuda/ms_deform_im2col_cuda.cuh:237-299 and :33-84: h = loc_y*H - 0.5,
    w = loc_x*W - 0.5, a sample counts only if -1 < h < H and -1 < w < W, four-tap bilinear with
    zero outside, value laid out (S, M, D) row-major.
    """
    value = np.asarray(value)
    dt = dtype or value.dtype
    n, s, m, d = value.shape
    _, lq, _, l, p, _ = loc.shape
    out = np.zeros((n, lq, m, d), dtype=dt)
    one = dt.type(1) if hasattr(dt, "type") else np.dtype(dt).type(1)
    half = one / 2
    for lv in range(l):
        hh, ww = int(spatial_shapes[lv][0]), int(spatial_shapes[lv][1])
        start = int(level_start_index[lv])
        val_l = value[:, start:start + hh * ww].reshape(n, hh, ww, m, d)
        for pt in range(p):
            w_im = loc[:, :, :, lv, pt, 0].astype(dt) * one * ww - half
            h_im = loc[:, :, :, lv, pt, 1].astype(dt) * one * hh - half
            inside = (h_im > -1) & (w_im > -1) & (h_im < hh) & (w_im < ww)
            h0 = np.floor(h_im)
            w0 = np.floor(w_im)
            lh, lw = h_im - h0, w_im - w0
            hh_, hw_ = one - lh, one - lw
            acc = np.zeros((n, lq, m, d), dtype=dt)
            for dy, dx, wt in ((0, 0, hh_ * hw_), (0, 1, hh_ * lw), (1, 0, lh * hw_), (1, 1, lh * lw)):
                yy, xx = h0 + dy, w0 + dx
                ok = inside & (yy >= 0) & (yy <= hh - 1) & (xx >= 0) & (xx <= ww - 1)
                yi = np.where(ok, yy, 0).astype(np.int64)
                xi = np.where(ok, xx, 0).astype(np.int64)
                bi = np.arange(n)[:, None, None]
                mi = np.arange(m)[None, None, :]
                tap = val_l[bi, yi, xi, mi]                      # (n, lq, m, d)
                acc = acc + np.where(ok[..., None], wt[..., None] * tap, 0).astype(dt)
            out = out + acc * attn[:, :, :, lv, pt][..., None].astype(dt)
    return out.reshape(n, lq, m * d)


# ---------------------------------------------------------------------------------------------
# voxel_voting.py
# ---------------------------------------------------------------------------------------------
VOTE_FOV = ((-50.0, -50.0, -4.0), (50.0, 50.0, 2.0))    # voxel_voting.py:138
VOTE_SIZE = (512, 512, 30)                                # voxel_voting.py:229
CROP_EPS = 1e-4                                           # utils/transforms.py:140


def vote_crop_mask(points, fov=VOTE_FOV, eps=CROP_EPS):
    """utils/transforms.py:151-161: open interval lo+eps < p < hi-eps on x, y, z.  Points are float32;
    the bounds lo+eps / hi-eps are Python doubles, and torch compares a float32 tensor with a Python
    scalar after casting the scalar to float32."""
    points = np.asarray(points, dtype=np.float32)
    keep = np.ones(points.shape[0], dtype=bool)
    for d in range(3):
        lo = np.float32(fov[0][d] + eps)
        hi = np.float32(fov[1][d] - eps)
        keep &= (points[:, d] > lo) & (points[:, d] < hi)
    return keep


def vote_quantize(points, fov=VOTE_FOV, size=VOTE_SIZE):
    """voxel_voting.py:77-91 then `.to(torch.int64)` (:234): float32 subtract, float32 true divide by the
    cell size (a Python double rounded to float32), truncation toward zero."""
    points = np.asarray(points, dtype=np.float32)
    cols = []
    for d in range(3):
        cell = np.float32((fov[1][d] - fov[0][d]) / size[d])
        q = (points[:, d] - np.float32(fov[0][d])) / cell
        cols.append(np.trunc(q).astype(np.int64))
    return np.stack(cols, axis=-1)


def vote_voxel_labels(voxel_coords, labels, size=VOTE_SIZE):
    """voxel_voting.py:55-75: dense per-voxel class histogram, argmax with ties to the lowest class."""
    num_classes = int(labels.max()) + 1
    lin = voxel_coords[:, 0] * (size[1] * size[2]) + voxel_coords[:, 1] * size[2] + voxel_coords[:, 2]
    votes = np.zeros((size[0] * size[1] * size[2], num_classes), dtype=np.int64)
    np.add.at(votes, (lin, labels.astype(np.int64)), 1)
    return votes.argmax(axis=-1).reshape(size)


def vote_point_labels(cur_coords, voxel_labels, size=VOTE_SIZE):
    """voxel_voting.py:38-53: gather the voxel label for in-grid points, 0 for the others."""
    ok = (cur_coords >= 0).all(axis=1) & (cur_coords[:, 0] < size[0]) & (cur_coords[:, 1] < size[1]) \
        & (cur_coords[:, 2] < size[2])
    out = np.zeros(cur_coords.shape[0], dtype=np.int64)
    c = cur_coords[ok]
    out[ok] = voxel_labels.reshape(-1)[c[:, 0] * size[1] * size[2] + c[:, 1] * size[2] + c[:, 2]]
    return out


def vote_frame(cur_points, cur_pred, hist_points, hist_pred):
    """One frame of the voxel_voting.py:214-242 pipeline.  cur_points (n,>=3) float32 in the current
    sensor frame with predictions cur_pred (n,) in {0,1,2}; hist_points already pose-aligned into the
    current frame (concatenated history window).  Returns the refined (n,) labels."""
    hk = vote_crop_mask(hist_points)
    ck = vote_crop_mask(cur_points)
    pts = np.concatenate((np.asarray(hist_points)[hk][:, :3], np.asarray(cur_points)[ck][:, :3]), axis=0)
    lab = np.concatenate((np.asarray(hist_pred)[hk], np.asarray(cur_pred)[ck]), axis=0).astype(np.int64)
    coords = vote_quantize(pts)
    voxel_labels = vote_voxel_labels(coords, lab)
    new = vote_point_labels(coords[int(hk.sum()):], voxel_labels)
    out = np.asarray(cur_pred).astype(np.int64).copy()
    out[ck] = new
    return out


# ---------------------------------------------------------------------------------------------------
# Instance-level voting (voxel_instance_voting.py:144-193, 195-272).  PINNED: the script cannot be imported (argparse,
# yaml.load without Loader, a dataset walk and a process pool at module level), but its functions -- post_processing()
# included -- are extracted from the source text and run on a synthetic sequence by tests/golden/make_golden.py; the label
# files they write are tests/golden/instance.npz, and tests/test_oracle_golden.py::test_instance_voting_matches_reference
# holds this restatement to them bit for bit.  It calls the same third-party routines as the reference (scikit-learn
# DBSCAN, scipy ConvexHull / Delaunay; unpinned in requirements.txt:3,9 -- here scikit-learn 1.7, scipy 1.15).  Not
# covered by the fixture: a degenerate (coplanar) cluster, where the reference's except branch itself fails on current
# numpy / scipy (np.bool, scipy.spatial.qhull).
# ---------------------------------------------------------------------------------------------------
def instance_box_corners(cluster_points):
    """min_bounding_box_3d (:43-60): the axis-aligned box of the convex-hull vertices, i.e. of the points.  The
    reference lets scipy raise on a degenerate (coplanar) cluster; here the box is taken from the points directly."""
    from scipy.spatial import ConvexHull, QhullError
    pts = np.array(cluster_points)
    try:
        pts = pts[ConvexHull(pts).vertices]
    except QhullError:
        pass
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    return np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                     [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])


def instance_in_hull(p, corners):
    """in_hull (:62-76): Delaunay triangulation of the corners, find_simplex >= 0; a degenerate box contains nothing."""
    from scipy.spatial import Delaunay, QhullError
    try:
        return Delaunay(corners).find_simplex(p) >= 0
    except QhullError:
        return np.zeros(p.shape[0], dtype=bool)


def instance_cluster(cur_points, cur_pred, cur_bf, local_points, local_pred, eps=0.3, min_samples=5, min_points=30):
    """cluster() (:144-193).  cur_points (n,>=3) float32 raw scan, cur_pred (n,) labels after the voxel vote, cur_bf
    (n,) movable-object prediction (2 = foreground), local_points / local_pred: the cropped local map (history + current)
    with its PRE-vote predictions.  Returns the (n,) labels with every kept cluster overwritten by its majority class."""
    from sklearn.cluster import DBSCAN
    out = np.asarray(cur_pred).copy()
    fg = np.where(np.asarray(cur_bf) == 2)[0]
    if len(fg) == 0:
        return out
    fpts = np.asarray(cur_points)[fg][:, :3]
    lab = DBSCAN(eps=eps, min_samples=min_samples).fit_predict(fpts)
    local_xyz = np.asarray(local_points)[:, :3]
    local_pred = np.asarray(local_pred)
    for c in np.unique(lab):
        if c == -1:
            continue
        member = lab == c
        if int(member.sum()) <= min_points:
            continue
        corners = instance_box_corners(fpts[member])
        z_min = np.min(corners[:, -1])
        corners[np.where(corners[:, -1] == z_min), -1] += 0.2          # :171-173, in the corners' float32
        inside = instance_in_hull(local_xyz, corners)
        pred_in = local_pred[inside]
        static_num = int(np.sum(pred_in[pred_in == 1]))                  # :178  = count of 1s
        dynamic_num = int(np.sum(pred_in[pred_in == 2]))                 # :179  = 2 x count of 2s (a SUM of labels)
        out[fg[member]] = 2 if dynamic_num > static_num else 1
    return out


def instance_vote_frame(cur_points, cur_pred, cur_bf, hist_points, hist_pred):
    """post_processing (:195-272) for one frame: the voxel vote of vote_frame, then cluster() on the result, with the
    local map = cropped history (already pose-aligned) + cropped current scan and their pre-vote predictions."""
    hk = vote_crop_mask(hist_points)
    ck = vote_crop_mask(cur_points)
    local_points = np.concatenate((np.asarray(hist_points)[hk], np.asarray(cur_points)[ck]), axis=0)
    local_pred = np.concatenate((np.asarray(hist_pred)[hk], np.asarray(cur_pred)[ck]), axis=0).astype(np.int64)
    voted = vote_frame(cur_points, cur_pred, hist_points, hist_pred)
    return instance_cluster(cur_points, voted, cur_bf, local_points, local_pred)


def instance_cluster_stats(cur_points, cur_pred, cur_bf, eps=0.3, min_samples=5):
    """DBSCAN clusters of the foreground of one frame: [{points}] (used by the fixture generator to assert which cases a
    synthetic frame exercises)."""
    from sklearn.cluster import DBSCAN
    fg = np.where(np.asarray(cur_bf) == 2)[0]
    if len(fg) == 0:
        return []
    lab = DBSCAN(eps=eps, min_samples=min_samples).fit_predict(np.asarray(cur_points)[fg][:, :3])
    return [{"points": int((lab == c).sum())} for c in np.unique(lab) if c != -1]
