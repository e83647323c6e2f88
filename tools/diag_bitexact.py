"""Counts (instead of tolerating) the elements that differ between the device kernels and the host restatement on the
two "bit-exact" rows: the pose-aligned voting map and the device preprocessing.  Prints per-quantity mismatch counts and
a few offending values.  Run on the GPU box: python tools/diag_bitexact.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ops_np  # noqa: E402
from streammos_amd import device_preprocess, ops, preprocess, synth  # noqa: E402

DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def vote():
    rng = np.random.Generator(np.random.PCG64(77))
    cur_id = 12
    scans = {k: synth.synthetic_scan(k) for k in range(cur_id - 8, cur_id + 1)}
    preds = {k: rng.integers(0, 3, scans[k].shape[0]).astype(np.uint8) for k in scans}
    poses = {k: synth.synthetic_pose(k) for k in scans}
    inv_cur = np.linalg.inv(poses[cur_id])
    table = torch.empty(512 * 512 * 30, dtype=torch.int64, device=DEV)
    ops.vote_clear(table)
    hist_pts, hist_lab = [], []
    for k in range(cur_id - 1, cur_id - 9, -1):
        ops.vote_accumulate(_t(scans[k]), _t(preds[k]), table, pose_diff=inv_cur.dot(poses[k]))
        hist_pts.append(preprocess.pose_align(scans[k], inv_cur.dot(poses[k])))
        hist_lab.append(preds[k])
    ops.vote_accumulate(_t(scans[cur_id]), _t(preds[cur_id]), table)
    got = ops.vote_resolve(_t(scans[cur_id]), _t(preds[cur_id]), table).cpu().numpy()
    want = ops_np.vote_frame(scans[cur_id], preds[cur_id], np.concatenate(hist_pts, 0), np.concatenate(hist_lab, 0))
    bad = np.nonzero(got != want)[0]
    print("vote: %d of %d refined labels differ" % (bad.size, got.size), bad[:10])


def prep(beams, azimuth, npad):
    spec = preprocess.VoxelSpec()
    scans = [synth.synthetic_scan(k, beams, azimuth) for k in (5, 4, 3)]
    poses = [synth.synthetic_pose(k) for k in (5, 4, 3)]
    host = preprocess.build_sample(scans, poses, npad, spec, tta=True)
    pre = device_preprocess.DevicePreprocessor(DEV, spec, npad, tta=True)
    inv_cur = np.linalg.inv(poses[0])
    built = pre.build([torch.from_numpy(s).to(DEV) for s in scans], [inv_cur.dot(p) for p in poses])
    xyzi, coord, sph = (built[k].cpu().numpy() for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord"))
    print("prep %dx%d -> %d:" % (beams, azimuth, npad))
    names = ("x", "y", "z", "intensity", "dist", "frac_x", "frac_y")
    for c in range(7):
        d = xyzi[:, :, c] != host["pcds_xyzi"][:, :, c]
        print("  xyzi[%s]: %d of %d differ, max abs %.3g" % (names[c], int(d.sum()), d.size,
                                                            float(np.abs(xyzi[:, :, c] - host["pcds_xyzi"][:, :, c]).max())))
        if d.any():
            i = np.argwhere(d)[:3]
            for j in i:
                a, b = xyzi[:, :, c][tuple(j)], host["pcds_xyzi"][:, :, c][tuple(j)]
                print("     at", tuple(j), "dev %r host %r" % (float(a), float(b)), "t=%d" % j[1])
    for c in range(3):
        d = coord[:, :, :, c] != host["pcds_coord"][:, :, :, c]
        print("  coord[%d]: %d of %d differ" % (c, int(d.sum()), d.size))
    for c in range(2):
        d = sph[:, :, :, c] != host["pcds_sphere_coord"][:, :, :, c]
        print("  sphere[%d]: %d of %d differ, max abs %.3g" % (c, int(d.sum()), d.size,
                                                              float(np.abs(sph[:, :, :, c] - host["pcds_sphere_coord"][:, :, :, c]).max())))
    # which stage: the pose-aligned points themselves
    moved_dev = torch.empty((scans[1].shape[0], 4), dtype=torch.float32, device=DEV)
    from streammos_amd import _lib
    lib = _lib.load()
    for t in (1, 2):
        s = torch.from_numpy(scans[t]).to(DEV)
        moved = torch.empty_like(s)
        mask = torch.empty(s.shape[0], dtype=torch.int32, device=DEV)
        pd = _lib.f64_array(np.asarray(inv_cur.dot(poses[t]), dtype=np.float64).reshape(-1)[:16])
        _lib.check(lib.smos_prep_transform_mask(s.data_ptr(), s.shape[0], pd, pre._range6, moved.data_ptr(), mask.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), "tm")
        want = preprocess.pose_align(scans[t], inv_cur.dot(poses[t]))
        d = moved.cpu().numpy() != want
        print("  pose-aligned scan t=%d: %d of %d float32 values differ" % (t, int(d.sum()), d.size))


if __name__ == "__main__":
    vote()
    prep(16, 120, 2048)
    prep(64, 1875, 160000)
