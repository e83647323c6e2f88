"""Device-side preprocessing (row f1) against the host numpy restatement, which is bit-exact against the reference."""
import numpy as np
import pytest
import torch

from streammos_amd import device_preprocess, preprocess, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("beams,azimuth,npad", [(16, 120, 2048), (64, 1875, 160000)])
def test_device_preprocess_matches_host(beams, azimuth, npad):
    spec = preprocess.VoxelSpec()
    scans = [synth.synthetic_scan(k, beams, azimuth) for k in (5, 4, 3)]
    poses = [synth.synthetic_pose(k) for k in (5, 4, 3)]
    # push a few points exactly onto the half-open range boundaries
    scans[0][:4, :3] = [[-50.0, 0, 0], [50.0, 0, 0], [49.999996, 1, -4.0], [0, -50.0, 1.9999999]]
    host = preprocess.build_sample(scans, poses, npad, spec, tta=True)
    pre = device_preprocess.DevicePreprocessor(DEV, spec, npad, tta=True)
    inv_cur = np.linalg.inv(poses[0])
    built = pre.build([torch.from_numpy(s).to(DEV) for s in scans], [inv_cur.dot(p) for p in poses])
    assert np.array_equal(built["mask"].cpu().numpy().astype(bool), host["valid_mask"])
    n_valid = int(host["valid_mask"].sum())
    assert int(built["prefix"][-1]) == n_valid == npad - host["pad_length"]
    xyzi, coord, sph = (built[k].cpu().numpy() for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord"))
    assert xyzi.shape == host["pcds_xyzi"].shape and coord.shape == host["pcds_coord"].shape
    # pose alignment (float64 product in dgemm order, rounded to float32 once), range mask, compaction, padding, TTA
    # flips, BEV quantisation and the 7-channel point feature are bit-exact against the host restatement, which is
    # bit-exact against the reference (tests/test_oracle_golden.py::test_preprocess_matches_reference)
    assert np.array_equal(coord, host["pcds_coord"]), int((coord != host["pcds_coord"]).sum())
    assert np.array_equal(xyzi, host["pcds_xyzi"]), int((xyzi != host["pcds_xyzi"]).sum())
    # the range-view coordinates go through arcsin / arctan2: numpy evaluates them with its own float32 SIMD routines
    # (whose last bit depends on the host CPU's dispatch: the reference is not bit-reproducible across hosts here), the
    # device with asinf / atan2f.  Measured: 3-5 % of the values differ by <= 1 ulp of the angle (3e-5 / 2.4e-4 of a
    # range-image cell); the same range-image cell (after the model's 0.5 scale) for all but a vanishing fraction
    ds = np.abs(sph - host["pcds_sphere_coord"])
    assert ds.max() <= 2e-3
    same_cell = np.floor(sph * 0.5) == np.floor(host["pcds_sphere_coord"] * 0.5)
    assert same_cell.mean() >= 0.9999
    # labels back to the raw scan
    lab = torch.randint(0, 3, (npad,), dtype=torch.uint8, device=DEV)
    raw = pre.unpad_labels(lab, built).cpu().numpy()
    want = np.zeros(scans[0].shape[0], dtype=np.uint8)
    want[host["valid_mask"]] = lab.cpu().numpy()[:n_valid]
    assert np.array_equal(raw, want)


def test_step_raw_equals_host_preprocessed_step():
    """StreamRunner.step_raw (device preprocessing) against step() on host-preprocessed inputs: same labels for the
    raw scan except where a 1-ulp angle difference moves a point across a range-image cell (observed on MI355X: logits
    4e-6 .. 1.6e-5 of the range, no label flipped; bars ~10x that, one flipped label of 2048 = 4.9e-4)."""
    from streammos_amd import streaming
    from streammos_amd.refapi.config import StreamMOS as cfg
    from streammos_amd.refapi.models import StreamMOS
    model = StreamMOS.AttNet(cfg.get_config()[2])
    model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    spec = preprocess.VoxelSpec()
    scans = [synth.synthetic_scan(k, 16, 120) for k in range(6)]
    poses = [synth.synthetic_pose(k) for k in range(6)]
    a = streaming.StreamRunner(model, DEV, vote=True)
    b = streaming.StreamRunner(model, DEV, vote=True)
    a.voter.window = b.voter.window = 3
    for i in range(4):
        idx = preprocess.window_indices(i, 6, 3)
        sample = preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], 2048, spec, tta=True)
        oa = a.step(a.upload(sample, scans[i]), poses[i])
        ob = b.step_raw([scans[j] for j in idx], [poses[j] for j in idx], frame_point_num=2048)
        err = (oa["pred_cls"] - ob["pred_cls"]).abs().max().item() / oa["pred_cls"].abs().max().item()
        same = (oa["raw_labels"] == ob["raw_labels"]).float().mean().item()
        print("step_raw vs host-preprocessed step, frame %d: logits %.2e of range, raw labels %.6f" % (i, err, same))
        assert err <= 2e-4 and same >= 0.9995, (i, err, same)
        assert [f for f, _ in oa["voted"]] == [f for f, _ in ob["voted"]]
        for (_, la), (_, lb) in zip(oa["voted"], ob["voted"]):
            assert (la == lb).float().mean().item() >= 0.9995


def test_step_raw_with_look_ahead_equals_plain_step_raw():
    """step_raw(next_scans=...) uploads, preprocesses and encodes the following frame on the side stream beside the current
    frame's decoder; labels, logits and voted frames must be those of the serial raw path, bit for bit (host arrays and
    device-resident scans; the last frame has no successor)."""
    from streammos_amd import streaming
    from streammos_amd.refapi.config import StreamMOS as cfg
    from streammos_amd.refapi.models import StreamMOS
    model = StreamMOS.AttNet(cfg.get_config()[2])
    model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    scans = [synth.synthetic_scan(k, 16, 120) for k in range(7)]
    poses = [synth.synthetic_pose(k) for k in range(7)]
    for on_device in (False, True):
        src = [torch.from_numpy(s).to(DEV) for s in scans] if on_device else scans
        plain = streaming.StreamRunner(model, DEV, vote=True)
        ahead = streaming.StreamRunner(model, DEV, vote=True, pipeline=True)
        plain.voter.window = ahead.voter.window = 3
        for i in range(6):
            idx = preprocess.window_indices(i, 7, 3)
            nxt = preprocess.window_indices(i + 1, 7, 3) if i < 5 else None
            oa = plain.step_raw([src[j] for j in idx], [poses[j] for j in idx], frame_point_num=2048)
            ob = ahead.step_raw([src[j] for j in idx], [poses[j] for j in idx], frame_point_num=2048,
                                next_scans=[src[j] for j in nxt] if nxt else None,
                                next_poses=[poses[j] for j in nxt] if nxt else None)
            assert torch.equal(oa["pred_cls"], ob["pred_cls"]) and torch.equal(oa["raw_labels"], ob["raw_labels"])
            assert [f for f, _ in oa["voted"]] == [f for f, _ in ob["voted"]]
            for (_, la), (_, lb) in zip(oa["voted"], ob["voted"]):
                assert torch.equal(la, lb)


def test_look_ahead_orders_the_side_stream_behind_uploads_pending_on_main():
    """The window of frame 0 is uploaded on the main stream; the look-ahead window of frame 1 re-uses two of those scans
    on the side stream through the upload cache.  With main's copies held back by a long spin kernel queued in front of
    them, the side stream reads stale memory unless the cache hit waits for the copy's event (the cache
    kept no record of the uploading stream).  Results must equal the serial raw path bit for bit."""
    from streammos_amd import streaming
    from streammos_amd.refapi.config import StreamMOS as cfg
    from streammos_amd.refapi.models import StreamMOS
    model = StreamMOS.AttNet(cfg.get_config()[2])
    model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    scans = [synth.synthetic_scan(k, 16, 120) for k in range(5)]
    poses = [synth.synthetic_pose(k) for k in range(5)]
    plain = streaming.StreamRunner(model, DEV, vote=False)
    ahead = streaming.StreamRunner(model, DEV, vote=False, pipeline=True)
    want = []
    for i in range(3):
        idx = preprocess.window_indices(i, 5, 3)
        want.append(plain.step_raw([scans[j] for j in idx], [poses[j] for j in idx], frame_point_num=2048)["pred_cls"].clone())
    torch.cuda.synchronize()
    # poison the allocator's free blocks so that a read before the copy lands sees garbage, not an old copy of the scan
    junk = [torch.full((s.size,), float("nan"), device=DEV) for s in scans for _ in range(2)]
    del junk
    for i in range(3):
        idx = preprocess.window_indices(i, 5, 3)
        nxt = preprocess.window_indices(i + 1, 5, 3)
        torch.cuda._sleep(200_000_000)          # ~0.1 s in front of this step's H2D copies on the main stream
        got = ahead.step_raw([scans[j] for j in idx], [poses[j] for j in idx], frame_point_num=2048,
                             next_scans=[scans[j] for j in nxt], next_poses=[poses[j] for j in nxt])["pred_cls"]
        assert torch.equal(got, want[i]), i
