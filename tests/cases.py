"""Seeded test cases shared by tests/golden/make_golden.py (reference side) and the parity tests.

Every generator is a pure function of its literal seeds (numpy PCG64), so the build container and the
GPU box regenerate byte-identical inputs; the golden files store the SHA-256 of the inputs they were
made from and the tests check it before comparing outputs.
"""
import numpy as np

from streammos_amd import preprocess, synth


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def grad_like(shape, name):
    return _rng(abs(hash_name(name)) + 7).normal(0, 1, shape).astype(np.float32)


def hash_name(name):
    import zlib
    return zlib.crc32(name.encode())


# ---------------------------------------------------------------------------------------------
def voxel_maxpool_cases():
    """name -> (feat (BS,C,N) f32, ind (BS,N,D) f32, out_size, scale).  Cover: negative maxima,
    coordinates in (-1,0) (truncate into cell 0), exactly -1.0 (dropped), exact integers, the far
    edge, everything out of range, duplicates/ties, D=3, N=1."""
    cases = {}
    r = _rng(101)
    ind = r.uniform(-2.0, 12.0, (2, 200, 2)).astype(np.float32)
    ind[0, :8, 0] = [-0.5, -1.0, 0.0, 7.0, 8.0, 7.9999, -0.9999, 3.0]
    ind[0, :8, 1] = [-0.5, 2.0, 0.0, 9.0, 3.0, 9.9999, 1.0, 10.0]
    feat = r.normal(0, 1, (2, 3, 200)).astype(np.float32)
    feat[1, :, 50:60] = feat[1, :, 40:50]            # exact ties
    ind[1, 50:60] = ind[1, 40:50]
    cases["basic"] = (feat, ind, (8, 10), (1.0, 1.0))

    ind = r.uniform(-3.0, 30.0, (1, 300, 2)).astype(np.float32)
    feat = -np.abs(r.normal(0, 1, (1, 4, 300))).astype(np.float32)        # all-negative cells
    cases["scaled_neg"] = (feat, ind, (4, 6), (0.5, 0.25))

    ind = r.uniform(-1.0, 6.0, (2, 150, 3)).astype(np.float32)
    feat = r.normal(0, 1, (2, 2, 150)).astype(np.float32)
    cases["dim3"] = (feat, ind, (4, 5, 3), (1.0, 0.5, 1.0))

    ind = np.full((1, 20, 2), -1000.0, dtype=np.float32)
    ind[:, 10:] = 5000.0
    feat = r.normal(0, 1, (1, 2, 20)).astype(np.float32)
    cases["all_out"] = (feat, ind, (8, 8), (1.0, 1.0))

    cases["single"] = (np.array([[[-2.5], [3.0]]], dtype=np.float32), np.array([[[1.5, 2.5]]], dtype=np.float32),
                       (4, 4), (1.0, 1.0))

    # model-like: non-negative features, BEV-style coordinates incl. the reference's pad value
    ind = r.uniform(-5.0, 70.0, (2, 600, 2)).astype(np.float32)
    ind[:, -50:] = -4864.0
    feat = np.maximum(r.normal(0, 1, (2, 8, 600)), 0).astype(np.float32)
    cases["relu_like"] = (feat, ind, (32, 32), (0.5, 0.5))
    return cases


def bilinear_cases():
    """name -> (grid (B,C,H,W) f32, coord (B,N,2) f32, scale)."""
    cases = {}
    r = _rng(202)
    grid = r.normal(0, 1, (2, 3, 5, 7)).astype(np.float32)
    coord = np.stack((r.uniform(-1.5, 6.0, (2, 80)), r.uniform(-1.5, 8.0, (2, 80))), -1).astype(np.float32)
    coord[0, :6] = [[0, 0], [4, 6], [4.0001, 6.0001], [-0.0001, 3], [2, 6.5], [-1, -1]]
    cases["unit"] = (grid, coord, (1.0, 1.0))
    grid = r.normal(0, 1, (1, 4, 16, 12)).astype(np.float32)
    coord = np.stack((r.uniform(-4, 36, (1, 300)), r.uniform(-4, 28, (1, 300))), -1).astype(np.float32)
    cases["half"] = (grid, coord, (0.5, 0.5))
    grid = r.normal(0, 1, (2, 2, 8, 32)).astype(np.float32)
    coord = np.stack((r.uniform(-8, 40, (2, 200)), r.uniform(-8, 140, (2, 200))), -1).astype(np.float32)
    coord[:, -20:] = -4864.0
    cases["quarter_rv"] = (grid, coord, (0.25, 0.25))
    return cases


def msda_cases():
    """name -> (value (N,S,M,D), shapes (L,2) i64, level_start (L,) i64, loc (N,Lq,M,L,P,2), attn (N,Lq,M,L,P)).
    'reftest' has the shape of the reference's own check (deformattn/test.py:21-28: N=1, M=2, D=2, Lq=2,
    L=2, P=2, levels (6,4),(3,2)); 'model' is the model's configuration scaled down (L=1, M=4, D=32, P=4)."""
    cases = {}
    r = _rng(303)

    def make(n, m, d, lq, shapes, p, lo, hi):
        shapes = np.asarray(shapes, dtype=np.int64)
        lsi = np.concatenate(([0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1])).astype(np.int64)
        s = int((shapes[:, 0] * shapes[:, 1]).sum())
        value = (r.random((n, s, m, d)) * 0.01).astype(np.float32)
        loc = r.uniform(lo, hi, (n, lq, m, len(shapes), p, 2)).astype(np.float32)
        attn = (r.random((n, lq, m, len(shapes), p)) + 1e-5).astype(np.float32)
        attn /= attn.sum(-1, keepdims=True).sum(-2, keepdims=True)
        return value, shapes, lsi, loc, attn.astype(np.float32)

    cases["reftest"] = make(1, 2, 2, 2, [(6, 4), (3, 2)], 2, 0.0, 1.0)
    cases["model"] = make(2, 4, 32, 64, [(8, 8)], 4, -0.3, 1.3)
    cases["ragged"] = make(1, 3, 5, 7, [(5, 3), (2, 4), (1, 1)], 3, -0.2, 1.2)
    return cases


def msda_module_case():
    r = _rng(404)
    hh = ww = 8
    q = r.normal(0, 1, (2, hh * ww, 128)).astype(np.float32)
    src = r.normal(0, 1, (2, hh * ww, 128)).astype(np.float32)
    ys = (np.arange(hh, dtype=np.float32) + 0.5) / hh
    xs = (np.arange(ww, dtype=np.float32) + 0.5) / ww
    ref = np.stack(np.broadcast_arrays(xs[None, :], ys[:, None]), -1).reshape(1, hh * ww, 1, 2)
    ref = np.ascontiguousarray(np.broadcast_to(ref, (2, hh * ww, 1, 2))).astype(np.float32)
    return q, ref, src, np.array([[hh, ww]], dtype=np.int64), np.array([0], dtype=np.int64)


def voting_cases():
    """name -> (cur (n,4) f32, cur_pred (n,) i64, hist (m,4) f32, hist_pred (m,) i64)."""
    cases = {}
    r = _rng(505)

    def cloud(n, spread):
        pts = np.concatenate((r.uniform(-spread, spread, (n, 2)), r.uniform(-5.0, 3.0, (n, 1)),
                              r.random((n, 1))), axis=1).astype(np.float32)
        return pts

    cur = cloud(4000, 60.0)
    # pile points into a few voxels so that votes really compete, incl. exact ties
    cur[:600, :3] = (r.integers(-20, 20, (600, 3)) * np.array([0.19, 0.19, 0.05]) + 0.01).astype(np.float32)
    hist = cloud(20000, 60.0)
    hist[:6000, :3] = (r.integers(-20, 20, (6000, 3)) * np.array([0.19, 0.19, 0.05]) + 0.01).astype(np.float32)
    # boundary values of the open crop interval and of the voxel grid
    cur[600:606, 0] = [-50.0, -49.9999, -49.99989, 49.9999, 49.99989, 50.0]
    cur[606:610, 2] = [-4.0, -3.9999, 1.9999, 2.0]
    cases["dense"] = (cur, r.integers(0, 3, 4000), hist, r.integers(0, 3, 20000))
    cur = cloud(500, 10.0)
    hist = cloud(1500, 10.0)
    cases["two_class"] = (cur, r.integers(0, 2, 500), hist, r.integers(0, 2, 1500))
    cur = cloud(300, 10.0)
    cases["no_history_inside"] = (cur, r.integers(1, 3, 300), cloud(50, 5.0) + np.float32(500.0), r.integers(1, 3, 50))
    return cases


def preprocess_case():
    scan = synth.synthetic_scan(3, n_beams=16, n_azimuth=50)
    scan[:5, :3] = [[-50.0, 0, 0], [50.0, 0, 0], [49.99999, 1, -3.99], [0, -50.0, 1.99], [10, 10, 2.0]]
    pose_diff = np.linalg.inv(synth.synthetic_pose(3)).dot(synth.synthetic_pose(1))
    return scan, pose_diff


E2E_POINTS = 2048
E2E_BEAMS, E2E_AZIMUTH = 16, 120


def e2e_frames(n_frames=3, tta_rows=(0, 3)):
    """Yields the ``infer`` batch (numpy, without the DataLoader dim) of frames 0..n_frames-1 of a small
    synthetic sequence: B=2 (TTA variants 0 and 3), T=3, N=2048."""
    spec = preprocess.VoxelSpec()
    total = n_frames + 2
    scans = [synth.synthetic_scan(k, E2E_BEAMS, E2E_AZIMUTH) for k in range(total)]
    poses = [synth.synthetic_pose(k) for k in range(total)]
    for i in range(n_frames):
        idx = preprocess.window_indices(i, total, 3)
        s = preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], E2E_POINTS, spec, tta=True)
        rows = list(tta_rows)
        yield {k: np.ascontiguousarray(s[k][rows]) for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}


def instance_sequence(n_frames=10):
    """Synthetic sequence for the instance-voting fixture (voxel_instance_voting.py:195-272): per frame a raw scan (n,4)
    float32, the network's per-point prediction in {0,1,2}, the movable-object (`_bf`) prediction (2 = foreground) and a
    4x4 pose.  Foreground objects are 1 m tall columns that travel with the sensor (so a box mostly holds the current
    frame's own points): A (60 points), B (exactly 30: NOT processed, the reference wants > 30), C (exactly 31), D and E
    with hand-set predictions inside their lifted boxes (D: four 1s and two 2s -- the reference compares SUMS of labels,
    4 vs 4, a tie -> static; E: three 1s and two 2s, 3 vs 4 -> moving although 2s are the minority), F far outside the
    voting crop, a sparse 4-point group (DBSCAN noise).  Every cluster's extreme points lie exactly on its box faces."""
    frames = []
    for k in range(n_frames):
        r = _rng(9000 + k)
        parts, preds, bfs = [], [], []

        def add(xyz, pred, bf):
            parts.append(np.asarray(xyz, dtype=np.float32))
            preds.append(np.asarray(pred, dtype=np.uint8))
            bfs.append(np.full(len(xyz), bf, dtype=np.uint8))

        g = np.stack((r.uniform(-30, 30, 500), r.uniform(-30, 30, 500), r.normal(-1.75, 0.01, 500)), -1)
        add(g, np.where(r.random(500) < 0.03, 2, 1), 1)                                     # ground, a few false movers

        def column(cx, cy, n, z0=-1.5, z1=-0.5):
            return np.stack((r.normal(cx, 0.04, n), r.normal(cy, 0.04, n), np.sort(r.uniform(z0, z1, n))), -1)

        add(column(8.0, 3.0, 60), np.where(r.random(60) < 0.8, 2, 1), 2)                      # A: mostly moving
        add(column(-6.0, 5.0, 30), np.full(30, 2), 2)                                        # B: 30 points
        add(column(-6.0, -7.0, 31), np.where(r.random(31) < 0.3, 2, 1), 2)                   # C: 31 points
        for (cx, cy, ones, twos) in ((12.0, -9.0, 4, 2), (15.0, 9.0, 3, 2)):               # D (tie), E (minority wins)
            col = column(cx, cy, 40)
            # points are sorted by z: the box floor is lifted by 0.2, so give labels by height -- everything below
            # z_min + 0.25 is 0 (not counted wherever the face falls), the next `ones` points 1, then `twos` points 2, rest 0
            pr = np.zeros(40, dtype=np.uint8)
            above = np.nonzero(col[:, 2] > col[0, 2] + 0.25)[0]
            pr[above[:ones]] = 1
            pr[above[ones:ones + twos]] = 2
            add(col, pr, 2)
        add(column(70.0, 0.0, 45), np.full(45, 2), 2)                                        # F: outside the +-50 m crop
        add(np.stack((r.uniform(20, 24, 4), r.uniform(-20, -16, 4), r.uniform(-1, 0, 4)), -1), np.full(4, 2), 2)   # noise
        xyz = np.concatenate(parts, 0)
        scan = np.concatenate((xyz, r.random((len(xyz), 1)).astype(np.float32)), 1).astype(np.float32)
        perm = r.permutation(len(scan))                                                    # file order is not object order
        frames.append((np.ascontiguousarray(scan[perm]), np.concatenate(preds)[perm], np.concatenate(bfs)[perm],
                       synth.synthetic_pose(k)))
    return frames


def training_batch():
    """The `batch` dict of AttNet.forward (models/StreamMOS.py:155-179; _seg: models/StreamMOS_seg.py:173-196): three
    chained samples `_0.._2` of the small e2e sequence (B=2, T=3, N=2048) with seeded per-point targets, BEV targets and
    movable-object (`_bf`) targets in {0, 1, 2} (0 = ignored by the losses)."""
    r = _rng(606)
    batch = {}
    for i, f in enumerate(e2e_frames(3)):
        for k, v in f.items():
            batch["%s_%d" % (k, i)] = v
        batch["pcds_target_%d" % i] = r.integers(0, 3, (2, E2E_POINTS, 1)).astype(np.int64)
        batch["pcds_bev_target_%d" % i] = r.integers(0, 3, (2, 256, 256, 1)).astype(np.int64)
        batch["pcds_bf_target_%d" % i] = r.integers(0, 3, (2, E2E_POINTS, 1)).astype(np.int64)
    return batch


# parameters whose gradients the training fixture stores (stage 1): the first point MLP conv (reached through the
# VoxelMaxPool backward), a conv of every BEV stage, the deformable-attention projections (through the sampler's backward),
# the learned memory embedding (through the 3-frame chain), the point head
TRAINING_GRAD_KEYS = (
    "point_pre.layer.0.layer.1.weight", "bev_net.header_bev.0.conv_branch.0.weight", "bev_net.res1_bev.2.layer.0.weight",
    "bev_net.res2.1.layer.3.weight", "bev_net.header_rv.1.layer.0.weight",
    "bev_net.deformattn_module.deformattn_layers.0.cross_attn.sampling_offsets.weight",
    "bev_net.deformattn_module.deformattn_layers.1.cross_attn.value_proj.weight",
    "bev_net.deformattn_module.deformattn_layers.0.linear1.weight", "bev_net.query_embed.weight",
    "bev_net.conv_1.conv.weight", "bev_net.aux_head2.weight", "point_post.merge_layer.0.weight", "pred_layer.pred_layer.0.weight",
    "pred_layer.pred_layer.0.bias")
