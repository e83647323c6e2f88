"""Instance-level voting on the GPU against the CPU restatement in oracle/ops_np.py, which calls the
same scikit-learn / scipy routines as voxel_instance_voting.py.  Integer / index work: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import ops_np
from streammos_amd import ops, preprocess, streaming, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rank(names):
    """cluster names (index of the lowest core point, -1 noise) -> scikit-learn's 0,1,2.. numbering"""
    names = np.asarray(names)
    out = np.full(names.shape, -1, dtype=np.int64)
    for r, v in enumerate(np.unique(names[names >= 0])):
        out[names == v] = r
    return out


def _cloud(seed, n_blobs, per, n_noise, sigma):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-20, 20, (n_blobs, 3)) * np.array([1, 1, 0.1])
    pts = [rng.normal(c, sigma, (per, 3)) for c in centres]
    pts.append(rng.uniform(-25, 25, (n_noise, 3)) * np.array([1, 1, 0.1]))
    # a thin chain of points 0.25 apart (every link is a core-to-core edge only through its neighbours)
    chain = np.stack((np.arange(60) * 0.25 - 7.0, np.full(60, 22.0), np.zeros(60)), axis=1)
    pts.append(np.repeat(chain, 2, axis=0))                      # duplicates: distance exactly 0
    pts = np.concatenate(pts).astype(np.float32)
    return pts[rng.permutation(len(pts))]


@pytest.mark.parametrize("seed,n_blobs,per,n_noise,sigma,eps,min_samples", [
    (0, 6, 120, 400, 0.25, 0.3, 5), (1, 12, 60, 1500, 0.4, 0.3, 5), (2, 3, 700, 50, 0.6, 0.3, 5),
    (3, 8, 40, 300, 0.2, 0.5, 3), (4, 5, 90, 0, 0.3, 0.3, 12), (5, 0, 0, 300, 0.3, 0.3, 5)])
def test_dbscan_matches_sklearn(seed, n_blobs, per, n_noise, sigma, eps, min_samples):
    from sklearn.cluster import DBSCAN
    pts = _cloud(seed, n_blobs, max(per, 1), n_noise, sigma)
    want = DBSCAN(eps=eps, min_samples=min_samples).fit_predict(pts)
    rows = torch.from_numpy(np.concatenate((pts, np.zeros((len(pts), 1), np.float32)), axis=1)).to(DEV)   # stride 4 rows
    got = ops.dbscan(rows, eps, min_samples).cpu().numpy()
    assert np.array_equal(_rank(got), want)
    # the name of a cluster is the index of its lowest-index core point
    core = np.zeros(len(pts), dtype=bool)
    core[DBSCAN(eps=eps, min_samples=min_samples).fit(pts).core_sample_indices_] = True
    for name in np.unique(got[got >= 0]):
        assert core[name] and name == np.where(core & (got == name))[0].min()


def test_dbscan_empty_and_tiny_inputs():
    assert ops.dbscan(torch.zeros((0, 3), device=DEV), 0.3, 5).numel() == 0
    one = ops.dbscan(torch.zeros((1, 3), device=DEV), 0.3, 5)
    assert one.tolist() == [-1]
    same = ops.dbscan(torch.ones((7, 4), device=DEV), 0.3, 5)           # 7 coincident points: one cluster named 0
    assert same.tolist() == [0] * 7


def test_box_vote_counts_match_numpy():
    rng = np.random.default_rng(3)
    n, k = 50000, 37
    pts = (rng.uniform(-60, 60, (n, 4)) * np.array([1, 1, 0.08, 1])).astype(np.float32)
    lab = rng.integers(0, 3, n).astype(np.uint8)
    lo = (rng.uniform(-45, 35, (k, 3)) * np.array([1, 1, 0.05])).astype(np.float32)
    hi = lo + rng.uniform(0.5, 12, (k, 3)).astype(np.float32)
    lo[:5] = pts[:5, :3]                                                   # faces that pass exactly through points
    hi[5:10] = pts[5:10, :3]
    boxes = np.concatenate((lo, hi), axis=1).astype(np.float32)
    pose = synth.synthetic_pose(3)
    diff = np.linalg.inv(synth.synthetic_pose(5)).dot(pose)
    for pd in (None, diff):
        moved = pts if pd is None else preprocess.pose_align(pts, pd)
        keep = ops_np.vote_crop_mask(moved)
        want = np.zeros((k, 3), dtype=np.int64)
        for b in range(k):
            inside = keep & np.all((moved[:, :3] >= boxes[b, :3]) & (moved[:, :3] <= boxes[b, 3:]), axis=1)
            for c in (1, 2):
                want[b, c] = int((inside & (lab == c)).sum())
        counts = torch.zeros((k, 3), dtype=torch.int32, device=DEV)
        ops.box_vote(torch.from_numpy(pts).to(DEV), torch.from_numpy(lab).to(DEV), torch.from_numpy(boxes).to(DEV), counts,
                     pose_diff=pd)
        assert np.array_equal(counts.cpu().numpy(), want)


def _sequence(n_frames):
    scans, preds, bfs, poses = [], [], [], []
    for k in range(n_frames):
        scan, lab = synth.synthetic_scan(k, 32, 400, with_labels=True)
        rng = np.random.default_rng(100 + k)
        pred = lab.copy()
        flip = rng.random(len(lab)) < 0.08                      # a noisy network: 8 % of the points get the other class
        pred[flip] = 3 - pred[flip]
        if k % 3 == 0:                                          # and some frames call whole objects static
            pred[:800] = 1
        bf = np.where(lab == 2, 2, 1)
        bf[rng.random(len(lab)) < 0.002] = 2                   # isolated false foreground: DBSCAN noise
        scans.append(scan); preds.append(pred.astype(np.uint8)); bfs.append(bf.astype(np.uint8)); poses.append(synth.synthetic_pose(k))
    return scans, preds, bfs, poses


def test_instance_voter_matches_oracle():
    n_frames, window = 7, 4
    scans, preds, bfs, poses = _sequence(n_frames)
    voter = streaming.InstanceVoter(DEV, window=window)
    plain = streaming.VoxelVoter(DEV, window=window)
    got, got_plain = {}, {}
    for k in range(n_frames):
        pts, pr = torch.from_numpy(scans[k]).to(DEV), torch.from_numpy(preds[k]).to(DEV)
        for fid, lab in voter.push(pts, pr, poses[k], torch.from_numpy(bfs[k]).to(DEV)):
            got[fid] = lab.cpu().numpy()
        for fid, lab in plain.push(pts, pr, poses[k]):
            got_plain[fid] = lab.cpu().numpy()
    assert sorted(got) == list(range(n_frames))
    lut = np.zeros(256, dtype=np.int32)
    lut[1], lut[2] = 9, 251
    changed = 0
    for fid in range(n_frames):
        hist_ids = streaming.vote_history_ids(fid, window)
        inv_cur = np.linalg.inv(poses[fid])
        hp = np.concatenate([preprocess.pose_align(scans[h], inv_cur.dot(poses[h])) for h in hist_ids], 0)
        hl = np.concatenate([preds[h] for h in hist_ids], 0)
        want = ops_np.instance_vote_frame(scans[fid], preds[fid], bfs[fid], hp, hl)
        assert np.array_equal(got[fid], lut[want]), fid
        changed += int((got[fid] != got_plain[fid]).sum())
    assert changed > 0          # the instance stage did overrule the voxel vote somewhere


def test_instance_voter_matches_reference_golden(golden):
    """The device path (DBSCAN, box vote, voxel vote: csrc/instance.hip, csrc/vote.hip) against the label files the
    REFERENCE's post_processing() wrote for the 10-frame fixture sequence (tests/golden/instance.npz): bit-exact."""
    from tests import cases
    from tests.util import check_inputs
    g = golden("instance")
    frames = cases.instance_sequence()
    voter = streaming.InstanceVoter(DEV)            # window 8, LUT {0: 0, 1: 9, 2: 251}
    got = {}
    for fid, (scan, pred, bf, pose) in enumerate(frames):
        check_inputs(g, "inst_f%d_in_sha" % fid, scan, pred, bf, pose)
        for k, lab in voter.push(torch.from_numpy(scan).to(DEV), torch.from_numpy(pred).to(DEV), pose, torch.from_numpy(bf).to(DEV)):
            got[k] = lab.cpu().numpy()
    assert sorted(got) == list(range(len(frames)))
    for fid in range(len(frames)):
        want = g["inst_f%d_refined" % fid]
        assert np.array_equal(got[fid], want), (fid, int((got[fid] != want).sum()))


def test_instance_voter_needs_bf_labels():
    voter = streaming.InstanceVoter(DEV)
    with pytest.raises(RuntimeError, match="_bf"):
        voter.push(torch.zeros((4, 4), device=DEV), torch.zeros(4, dtype=torch.uint8, device=DEV), np.eye(4))


def test_run_sequence_with_instance_voting(tmp_path):
    """StreamMOS_seg through run_sequence with vote="instance": the refined files exist for every scan and hold LUT
    words; the `_bf` files hold raw 0/1/2 (val_StreamMOS_seg.py:141)."""
    from streammos_amd import kitti, run_sequence
    seq = tmp_path / "sequences" / "08"
    (seq / "velodyne").mkdir(parents=True)
    n = 10
    for k in range(n):
        synth.synthetic_scan(k, 16, 120).tofile(seq / "velodyne" / ("%06d.bin" % k))
    kitti.write_poses(seq / "poses.txt", [synth.synthetic_pose(k) for k in range(n)])
    kitti.write_calibration(seq / "calib.txt")
    model = run_sequence.load_model(None, DEV, seg=True)
    out = tmp_path / "out"
    res = run_sequence.run_sequence(model, str(seq), str(out), DEV, vote="instance", frame_point_num=2048)
    assert res["scans"] == n
    for k in range(n):
        npts = kitti.read_scan(seq / "velodyne" / ("%06d.bin" % k)).shape[0]
        words = np.fromfile(out / "refined" / ("%06d.label" % k), dtype=np.uint32)
        assert words.shape[0] == npts and set(np.unique(words)) <= {0, 9, 251}
        bf = np.fromfile(out / "predictions_bf" / ("%06d.label" % k), dtype=np.uint32)
        assert bf.shape[0] == npts and set(np.unique(bf)) <= {0, 1, 2}
    # the same run with the validation preprocessing on the device (only raw scans uploaded): same files
    out2 = tmp_path / "out_dev"
    run_sequence.run_sequence(model, str(seq), str(out2), DEV, vote="instance", frame_point_num=2048, device_preprocess=True)
    same = [np.mean(np.fromfile(out / "refined" / ("%06d.label" % k), dtype=np.uint32) ==
                    np.fromfile(out2 / "refined" / ("%06d.label" % k), dtype=np.uint32)) for k in range(n)]
    assert min(same) >= 0.99          # asinf / atan2f may differ in the last ulp between numpy and the device
