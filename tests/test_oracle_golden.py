"""The CPU oracle against the golden vectors produced by the real reference (CPU-only, no GPU)."""
import numpy as np
import pytest
import torch

from oracle import net_torch, ops_np
from streammos_amd import preprocess, synth
from tests import cases
from tests.util import check_inputs


@pytest.mark.parametrize("name", sorted(cases.voxel_maxpool_cases()))
def test_voxel_maxpool_oracle_is_bit_exact(golden, name):
    g = golden("ops_voxel_maxpool")
    feat, ind, out_size, scale = cases.voxel_maxpool_cases()[name]
    check_inputs(g, "vmp_%s_in_sha" % name, feat, ind)
    out, idx = ops_np.voxel_maxpool_fwd(feat, ind, out_size, scale)
    assert np.array_equal(out, g["vmp_%s_out" % name])
    grad = ops_np.voxel_maxpool_bwd(feat, ind, out, cases.grad_like(out.shape, name), out_size, scale)
    assert np.array_equal(grad, g["vmp_%s_grad" % name])
    assert ((idx >= 0) == (ops_np.voxel_cell_index(ind, out_size, scale) >= 0)).all()


@pytest.mark.parametrize("name", sorted(cases.bilinear_cases()))
def test_bilinear_oracle(golden, name):
    g = golden("ops_bilinear")
    grid, coord, scale = cases.bilinear_cases()[name]
    check_inputs(g, "bil_%s_in_sha" % name, grid, coord)
    out = ops_np.bilinear_sample(grid, coord, scale)
    # tolerance: float32 blend, four products of O(1) values -> a few ulp
    np.testing.assert_allclose(out, g["bil_%s_out" % name], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", sorted(cases.msda_cases()))
def test_msda_oracle(golden, name):
    g = golden("ops_msda")
    value, shapes, lsi, loc, attn = cases.msda_cases()[name]
    check_inputs(g, "msda_%s_in_sha" % name, value, shapes, lsi, loc, attn)
    out64 = ops_np.msda_forward(value.astype(np.float64), shapes, lsi, loc, attn, dtype=np.dtype(np.float64))
    # deformattn/test.py:31-44 uses torch.allclose defaults (rtol 1e-5, atol 1e-8) in double
    np.testing.assert_allclose(out64, g["msda_%s_out64" % name], rtol=1e-5, atol=1e-8)
    out32 = ops_np.msda_forward(value, shapes, lsi, loc, attn, dtype=np.dtype(np.float32))
    # deformattn/test.py:47-60 float tolerance is rtol 1e-2 / atol 1e-3; we hold 1e-5 / 1e-7
    np.testing.assert_allclose(out32, g["msda_%s_out32" % name], rtol=1e-5, atol=1e-7)
    tv, tl, ta = torch.from_numpy(value), torch.from_numpy(loc), torch.from_numpy(attn)
    if shapes.shape[0] == 1:
        t = net_torch.OracleNet.msda_core(tv, (int(shapes[0, 0]), int(shapes[0, 1])), tl, ta).numpy()
        np.testing.assert_allclose(t, g["msda_%s_out32" % name], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", sorted(cases.voting_cases()))
def test_voting_oracle_is_bit_exact(golden, name):
    g = golden("ops_voting")
    cur, cur_pred, hist, hist_pred = cases.voting_cases()[name]
    check_inputs(g, "vote_%s_in_sha" % name, cur, cur_pred, hist, hist_pred)
    ck, hk = ops_np.vote_crop_mask(cur), ops_np.vote_crop_mask(hist)
    assert np.array_equal(ck, g["vote_%s_cur_mask" % name])
    assert np.array_equal(hk, g["vote_%s_hist_mask" % name])
    coords = ops_np.vote_quantize(np.concatenate((hist[hk], cur[ck]), 0))
    assert np.array_equal(coords, g["vote_%s_coords" % name])
    labels = np.concatenate((hist_pred[hk], cur_pred[ck]), 0)
    vox = ops_np.vote_voxel_labels(coords, labels).reshape(-1)
    nz = np.nonzero(vox)[0]
    assert np.array_equal(nz, g["vote_%s_voxel_nz_idx" % name])
    assert np.array_equal(vox[nz], g["vote_%s_voxel_nz_val" % name])
    refined = ops_np.vote_frame(cur, cur_pred, hist, hist_pred)
    assert np.array_equal(refined, g["vote_%s_refined" % name])


def test_preprocess_matches_reference(golden):
    g = golden("preprocess")
    scan, pose_diff = cases.preprocess_case()
    check_inputs(g, "pre_in_sha", scan, pose_diff)
    spec = preprocess.VoxelSpec()
    moved = preprocess.pose_align(scan, pose_diff)
    assert np.array_equal(moved, g["pre_moved"])
    mask = preprocess.range_mask(moved, spec)
    assert np.array_equal(mask, g["pre_mask"])
    kept = moved[mask]
    coord = preprocess.quantize_bev(kept, spec)
    sph = preprocess.quantize_sphere(kept, spec)
    feat = preprocess.point_features(kept, coord)
    assert [str(coord.dtype), str(sph.dtype), str(feat.dtype)] == list(g["pre_dtypes"])
    assert np.array_equal(coord.astype(np.float32), g["pre_coord"])
    assert np.array_equal(sph.astype(np.float32), g["pre_sphere"])
    assert np.array_equal(feat.astype(np.float32), g["pre_feat"])


def test_e2e_oracle_matches_reference(golden):
    """OracleNet (oracle/net_torch.py) vs the real AttNet.infer over 3 chained frames."""
    g = golden("e2e")
    import json, os
    layout = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_layout.json")))["stage1"]
    net = net_torch.OracleNet(synth.seeded_state_dict([(k, s) for k, s, _ in layout]))
    memory = None
    for i, batch in enumerate(cases.e2e_frames()):
        check_inputs(g, "e2e_f%d_in_sha" % i, batch["pcds_xyzi"], batch["pcds_coord"], batch["pcds_sphere_coord"])
        pred, a0, a1, a2, memory = net.stage_forward(*(torch.from_numpy(batch[k]) for k in
                                                        ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")), memory)
        ref = g["e2e_f%d_pred" % i]
        scale = np.abs(ref).max()
        # same library (torch CPU) on both sides, different graph expression: 1e-4 of the logit range
        assert np.abs(pred.numpy() - ref).max() <= 1e-4 * scale
        assert (pred.numpy().argmax(1) == ref.argmax(1)).mean() >= 0.999
        np.testing.assert_allclose(memory[:, ::8, ::4, ::4].numpy(), g["e2e_f%d_mem_sub" % i], rtol=0, atol=2e-4)
        aux = torch.stack((a0, a1, a2))[:, :, :, ::8, ::8].numpy()
        np.testing.assert_allclose(aux, g["e2e_f%d_aux_sub" % i], rtol=0, atol=1e-4 * np.abs(g["e2e_f%d_aux_sub" % i]).max())


def test_instance_voting_matches_reference(golden):
    """Row f3 pinned: the oracle's instance_vote_frame against the label files the reference's own post_processing()
    (voxel_instance_voting.py:195-270, extracted and run by tests/golden/make_golden.py::gen_instance) wrote for a 10-frame
    synthetic sequence -- both history branches (frames < 8 look at frames 0..7 except themselves), a 30- and a 31-point
    cluster, the sum-of-labels tie (4 x 1 vs 2 x 2 -> static), the minority win (3 x 1 vs 2 x 2 -> moving), a cluster
    outside the voting crop (empty box -> static), cluster points exactly on the box faces.  Bit-exact."""
    from streammos_amd import preprocess, streaming
    g = golden("instance")
    frames = cases.instance_sequence()
    lut = np.zeros(256, dtype=np.int32)
    lut[1], lut[2] = 9, 251
    for fid, (scan, pred, bf, pose) in enumerate(frames):
        check_inputs(g, "inst_f%d_in_sha" % fid, scan, pred, bf, pose)
        inv = np.linalg.inv(pose)
        hist = streaming.vote_history_ids(fid, 8)
        hp = np.concatenate([preprocess.pose_align(frames[h][0], inv.dot(frames[h][3])) for h in hist], 0)
        hl = np.concatenate([frames[h][1] for h in hist], 0)
        got = lut[ops_np.instance_vote_frame(scan, pred, bf, hp, hl)]
        assert np.array_equal(got, g["inst_f%d_refined" % fid]), fid
