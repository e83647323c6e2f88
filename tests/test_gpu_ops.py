"""Parity of the HIP kernels (through the C ABI) against the golden vectors and the CPU oracle.

Bars: bit-exact for the scatter (max is order independent) and for every voting index / label;
for the floating-point gathers and the deformable sampler the tolerance is written at the assert.
"""
import numpy as np
import pytest
import torch

from oracle import ops_np
from streammos_amd import ops
from streammos_amd.refapi import deep_point
from streammos_amd.refapi.deformattn.functions import MSDeformAttnFunction
from tests import cases
from tests.util import check_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ------------------------------------------------------------------------------------------
# VoxelMaxPool
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.voxel_maxpool_cases()))
def test_voxel_maxpool_golden(golden, name):
    g = golden("ops_voxel_maxpool")
    feat, ind, out_size, scale = cases.voxel_maxpool_cases()[name]
    check_inputs(g, "vmp_%s_in_sha" % name, feat, ind)
    f = _t(feat).unsqueeze(-1).requires_grad_(True)
    y = deep_point.VoxelMaxPool(f, _t(ind).unsqueeze(-1), out_size, scale)
    assert np.array_equal(y.detach().cpu().numpy(), g["vmp_%s_out" % name])
    y.backward(_t(cases.grad_like(y.shape, name)))
    assert np.array_equal(f.grad[..., 0].cpu().numpy(), g["vmp_%s_grad" % name])


@pytest.mark.parametrize("name", sorted(cases.voxel_maxpool_cases()))
def test_voxel_maxpool_channels_last_rows_kernel(golden, name):
    """point-major features + channels-last grid take the "rows" lane mapping; same numbers."""
    g = golden("ops_voxel_maxpool")
    feat, ind, out_size, scale = cases.voxel_maxpool_cases()[name]
    bs, c, n = feat.shape
    reps = max(1, 8 // c + 1)                          # the rows kernel needs C >= 8: tile the channels
    feat_w = np.concatenate([feat] * reps, axis=1)
    f = _t(feat_w.transpose(0, 2, 1)).permute(0, 2, 1)                 # [BS,C,N] view of a [BS,N,C] buffer
    out = torch.zeros((bs,) + tuple(out_size) + (c * reps,), device=DEV)
    out_v = out.movedim(-1, 1)                                           # [BS,C,*size] view, channel stride 1
    idx = torch.full((bs, n), -1, dtype=torch.int64, device=DEV)
    ops.voxel_maxpool_fwd(f, _t(ind), out_v, out_size, scale, voxel_max_idx=idx)
    want = np.concatenate([g["vmp_%s_out" % name]] * reps, axis=1)
    assert np.array_equal(out_v.cpu().numpy(), want)
    assert np.array_equal((idx >= 0).cpu().numpy(), ops_np.voxel_cell_index(ind, out_size, scale) >= 0)


def test_voxel_maxpool_voxel_max_idx_matches_reference_definition():
    feat, ind, out_size, scale = cases.voxel_maxpool_cases()["basic"]
    f, i = _t(feat), _t(ind)
    out = torch.zeros((feat.shape[0], feat.shape[1]) + tuple(out_size), device=DEV)
    idx = torch.full(ind.shape[:2], -1, dtype=torch.int64, device=DEV)
    ops.voxel_maxpool_fwd(f, i, out, out_size, scale, voxel_max_idx=idx)
    _, want = ops_np.voxel_maxpool_fwd(feat, ind, out_size, scale)
    assert np.array_equal(idx.cpu().numpy(), want)


def test_voxel_maxpool_full_size_properties():
    """BASELINE shape of the input scatter (12 x 64 x 160000 -> 512 x 512): compared bit-exactly with an
    independent formulation (torch scatter_reduce amax on the GPU), plus permutation invariance and
    idempotence (pooling the pooled grid's own occupied cells changes nothing)."""
    gen = torch.Generator(device="cpu").manual_seed(11)
    bs, c, n = 12, 64, 160000
    feat = torch.relu(torch.randn((bs, c, n), generator=gen)).to(DEV)
    ind = (torch.rand((bs, n, 2), generator=gen) * 560.0 - 24.0)
    ind[:, -30000:] = -4864.0                                   # the reference's padding rows
    ind = ind.to(DEV)
    out = torch.zeros((bs, c, 512, 512), device=DEV)
    ops.voxel_maxpool_fwd(feat, ind, out, (512, 512), (1.0, 1.0))
    cy, cx = ind[..., 0].double().trunc(), ind[..., 1].double().trunc()
    ok = (ind[..., 0] > -1) & (cy < 512) & (ind[..., 1] > -1) & (cx < 512)
    flat = torch.where(ok, cy * 512 + cx, torch.full_like(cy, 512 * 512)).long()
    want = torch.zeros((bs, c, 512 * 512 + 1), device=DEV)
    want.scatter_reduce_(2, flat[:, None, :].expand(bs, c, n), feat, "amax", include_self=False)
    assert torch.equal(out.view(bs, c, -1), want[:, :, :-1])
    perm = torch.randperm(n, generator=gen).to(DEV)
    out2 = torch.zeros_like(out)
    ops.voxel_maxpool_fwd(feat[:, :, perm].contiguous(), ind[:, perm].contiguous(), out2, (512, 512), (1.0, 1.0))
    assert torch.equal(out, out2)
    ops.voxel_maxpool_fwd(feat, ind, out2, (512, 512), (1.0, 1.0))      # idempotent on a filled grid
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16])
@pytest.mark.parametrize("name", sorted(cases.voxel_maxpool_cases()))
def test_voxel_maxpool_half_and_double(name, dtype):
    """The reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF (point_deep_cuda_kernel.cu:147,168): float16 and float64
    through the same entry points, forward and backward, bit for bit against the numpy restatement run on the SAME rounded
    inputs (a max picks one of its inputs: exact in any dtype) and, for float64, against the CPU twin."""
    feat, ind, out_size, scale = cases.voxel_maxpool_cases()[name]
    npdt = np.float64 if dtype == torch.float64 else np.float16
    feat_t, ind_t = feat.astype(npdt), ind.astype(npdt)
    want, want_idx = ops_np.voxel_maxpool_fwd(feat_t, ind_t.astype(np.float32), out_size, scale)
    f = torch.from_numpy(feat_t).to(DEV).unsqueeze(-1).requires_grad_(True)
    i = torch.from_numpy(ind_t).to(DEV).unsqueeze(-1)
    y = deep_point.VoxelMaxPool(f, i, out_size, scale)
    assert y.dtype == dtype and np.array_equal(y.detach().cpu().numpy(), want.reshape(y.shape))
    go = cases.grad_like(y.shape, name).astype(npdt)
    y.backward(torch.from_numpy(go).to(DEV))
    want_grad = ops_np.voxel_maxpool_bwd(feat_t, ind_t.astype(np.float32), want, go, out_size, scale)
    assert np.array_equal(f.grad[..., 0].cpu().numpy(), want_grad)
    idx = torch.full(ind.shape[:2], -1, dtype=torch.int64, device=DEV)
    out = torch.zeros(y.shape, dtype=dtype, device=DEV)
    ops.voxel_maxpool_fwd(f.detach()[..., 0], i[..., 0], out, out_size, scale, voxel_max_idx=idx)
    assert torch.equal(out, y.detach()) and np.array_equal(idx.cpu().numpy(), want_idx)
    if dtype == torch.float64:                       # and the DataLoader-side CPU twin (point_deep.cpu_kernel)
        yc = deep_point.VoxelMaxPool(torch.from_numpy(feat_t).unsqueeze(-1), torch.from_numpy(ind_t).unsqueeze(-1), out_size, scale)
        assert np.array_equal(yc.numpy(), y.detach().cpu().numpy())


def test_voxel_maxpool_errors_are_loud():
    f = torch.zeros((1, 2, 4, 1), device=DEV, dtype=torch.bfloat16)          # not a dtype the reference dispatches
    i = torch.zeros((1, 4, 2, 1), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        deep_point.VoxelMaxPool(f, i, (4, 4), (1.0, 1.0))
    with pytest.raises(RuntimeError):
        ops.voxel_maxpool_fwd(torch.zeros(1, 2, 4), torch.zeros(1, 4, 2), torch.zeros(1, 2, 4, 4), (4, 4), (1.0, 1.0))


# ------------------------------------------------------------------------------------------
# BilinearSample
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.bilinear_cases()))
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_bilinear_gather(golden, name, layout):
    g = golden("ops_bilinear")
    grid, coord, scale = cases.bilinear_cases()[name]
    check_inputs(g, "bil_%s_in_sha" % name, grid, coord)
    reps = 1 if layout == "nchw" else 4
    grid_w = np.concatenate([grid] * reps, axis=1)
    gt = _t(grid_w)
    if layout == "nhwc":
        gt = gt.contiguous(memory_format=torch.channels_last)
    out = ops.bilinear_gather(gt, _t(coord), scale, point_major=(layout == "nhwc"))
    want = np.concatenate([g["bil_%s_out" % name]] * reps, axis=1)
    # float32 four-tap blend of O(1) values; FMA contraction vs ATen's separate mul/add: <= 1e-6 abs
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6)
    np.testing.assert_allclose(out.cpu().numpy(), np.concatenate([ops_np.bilinear_sample(grid, coord, scale)] * reps, 1),
                               rtol=0, atol=2e-6)


def test_bilinear_gather_model_shape_against_grid_sample():
    """Model shape, against torch's own GPU grid_sample.  The normalised grid is built on the CPU (true
    float32 division, the formulation the golden vectors pin): on the GPU torch divides by a Python scalar
    as a multiply by the reciprocal, which moves the sampling position by ~3e-5 px and a randn map by up to
    ~2e-4 -- the reference's own CPU/GPU spread, not a property of this kernel."""
    gen = torch.Generator(device="cpu").manual_seed(5)
    grid = torch.randn((4, 32, 256, 256), generator=gen)
    coord = (torch.rand((4, 160000, 2), generator=gen) * 560 - 24)
    coord[:, -20000:] = -4864.0
    gx = (2 * coord[:, :, 1] * 0.5 / 255) - 1
    gy = (2 * coord[:, :, 0] * 0.5 / 255) - 1
    out = ops.bilinear_gather(grid.to(DEV), coord.to(DEV), (0.5, 0.5))
    want = torch.nn.functional.grid_sample(grid.to(DEV), torch.stack((gx, gy), -1)[:, :, None].to(DEV), mode="bilinear",
                                           padding_mode="zeros", align_corners=True)[..., 0]
    assert (out - want).abs().max().item() <= 5e-6


# ------------------------------------------------------------------------------------------
# deformable attention sampler
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.msda_cases()))
def test_msda_forward(golden, name):
    g = golden("ops_msda")
    value, shapes, lsi, loc, attn = cases.msda_cases()[name]
    check_inputs(g, "msda_%s_in_sha" % name, value, shapes, lsi, loc, attn)
    out32 = MSDeformAttnFunction.apply(_t(value), _t(shapes), _t(lsi), _t(loc), _t(attn), 256)
    # the reference's own float check allows rtol 1e-2 / atol 1e-3 (deformattn/test.py:58); we hold 1e-5 / 1e-7
    np.testing.assert_allclose(out32.cpu().numpy(), g["msda_%s_out32" % name], rtol=1e-5, atol=1e-7)
    out64 = MSDeformAttnFunction.apply(_t(value).double(), _t(shapes), _t(lsi), _t(loc).double(), _t(attn).double(), 256)
    # deformattn/test.py:41: torch.allclose defaults in double
    np.testing.assert_allclose(out64.cpu().numpy(), g["msda_%s_out64" % name], rtol=1e-5, atol=1e-8)


def test_msda_model_shape_fast_path_equals_generic_path():
    """D=32, one 64x64 level, 4 heads x 4 points (the model's shape): the shuffle kernel must agree with
    the generic kernel (forced by viewing the same data as float64) to float32 rounding."""
    gen = torch.Generator(device="cpu").manual_seed(9)
    n, s, m, d, lq, p = 4, 4096, 4, 32, 4096, 4
    value = torch.randn((n, s, m, d), generator=gen).to(DEV)
    loc = (torch.rand((n, lq, m, 1, p, 2), generator=gen) * 1.2 - 0.1).to(DEV)
    attn = torch.softmax(torch.randn((n, lq, m, p), generator=gen), -1).view(n, lq, m, 1, p).to(DEV)
    shapes = torch.tensor([[64, 64]], device=DEV)
    lsi = torch.zeros(1, dtype=torch.long, device=DEV)
    fast = ops.msda_fwd(value, shapes, lsi, loc, attn)
    slow = ops.msda_fwd(value.double(), shapes, lsi, loc.double(), attn.double())
    assert (fast.double() - slow).abs().max().item() <= 2e-5
    want = ops_np.msda_forward(value[:1, :, :, :].cpu().numpy(), [[64, 64]], [0], loc[:1, :256].cpu().numpy(),
                               attn[:1, :256].cpu().numpy())
    np.testing.assert_allclose(fast[:1, :256].cpu().numpy(), want, rtol=1e-4, atol=1e-5)


def test_msda_rejects_cpu_and_noncontiguous():
    value, shapes, lsi, loc, attn = cases.msda_cases()["model"]
    with pytest.raises(RuntimeError):
        MSDeformAttnFunction.apply(torch.from_numpy(value), torch.from_numpy(shapes), torch.from_numpy(lsi),
                                   torch.from_numpy(loc), torch.from_numpy(attn), 256)
    with pytest.raises(RuntimeError):
        ops.msda_fwd(_t(value).transpose(2, 3), _t(shapes), _t(lsi), _t(loc), _t(attn))


# ------------------------------------------------------------------------------------------
# voting + TTA reduce
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(cases.voting_cases()))
def test_vote_kernels_bit_exact(golden, name):
    g = golden("ops_voting")
    cur, cur_pred, hist, hist_pred = cases.voting_cases()[name]
    check_inputs(g, "vote_%s_in_sha" % name, cur, cur_pred, hist, hist_pred)
    table = torch.empty(512 * 512 * 30, dtype=torch.int64, device=DEV)
    ops.vote_clear(table)
    ops.vote_accumulate(_t(hist), _t(hist_pred.astype(np.uint8)), table)
    ops.vote_accumulate(_t(cur), _t(cur_pred.astype(np.uint8)), table)
    refined = ops.vote_resolve(_t(cur), _t(cur_pred.astype(np.uint8)), table)
    assert np.array_equal(refined.cpu().numpy(), g["vote_%s_refined" % name])
    lut = torch.zeros(256, dtype=torch.int32, device=DEV)
    lut[1], lut[2] = 9, 251
    mapped = ops.vote_resolve(_t(cur), _t(cur_pred.astype(np.uint8)), table, lut=lut)
    assert np.array_equal(mapped.cpu().numpy(), g["vote_%s_lut" % name])
    # the packed table against the reference's dense argmax, at every voxel that received a vote
    t = table.cpu().numpy().astype(np.uint64)
    c = np.stack(((t >> np.uint64(0)) & np.uint64(0x1FFFFF), (t >> np.uint64(21)) & np.uint64(0x1FFFFF),
                  (t >> np.uint64(42)) & np.uint64(0x1FFFFF)), -1)
    lab = c.argmax(-1)
    nz = np.nonzero(lab)[0]
    assert np.array_equal(nz, g["vote_%s_voxel_nz_idx" % name])
    assert np.array_equal(lab[nz], g["vote_%s_voxel_nz_val" % name])


def test_vote_pose_transform_matches_float64_host_path():
    from streammos_amd import preprocess, synth
    scan = synth.synthetic_scan(5, 32, 400)
    pose = np.linalg.inv(synth.synthetic_pose(9)).dot(synth.synthetic_pose(5))
    pred = np.random.Generator(np.random.PCG64(1)).integers(0, 3, scan.shape[0]).astype(np.uint8)
    moved = preprocess.pose_align(scan, pose)
    t_dev = torch.zeros(512 * 512 * 30, dtype=torch.int64, device=DEV)
    t_host = torch.zeros_like(t_dev)
    ops.vote_accumulate(_t(scan), _t(pred), t_dev, pose_diff=pose)
    ops.vote_accumulate(_t(moved), _t(pred), t_host)
    assert torch.equal(t_dev, t_host)


def test_tta_argmax_matches_torch():
    gen = torch.Generator(device="cpu").manual_seed(3)
    pred = (torch.randn((4, 3, 50000, 1), generator=gen) * 3).to(DEV)
    labels, prob = ops.tta_argmax(pred, want_prob=True)
    want = torch.softmax(pred, 1).mean(0).permute(2, 1, 0).squeeze(0)
    assert (prob - want).abs().max().item() <= 1e-6
    agree = (labels.long() == want.argmax(1)).float().mean().item()
    assert agree == 1.0 or ((prob.sort(1)[0][:, -1] - prob.sort(1)[0][:, -2])[labels.long() != want.argmax(1)].max() < 1e-6)


# ------------------------------------------------------------------------------------------
# fused point-side kernels of the inference engine
# ------------------------------------------------------------------------------------------
def _model_like_coords(gen, b, n, hi_y, hi_x, pad=0.1):
    c = torch.stack((torch.rand((b, n), generator=gen) * (hi_y * 1.1) - 0.05 * hi_y,
                     torch.rand((b, n), generator=gen) * (hi_x * 1.1) - 0.05 * hi_x), -1)
    c[:, -int(n * pad):] = -4864.0
    return c


@pytest.mark.parametrize("n", [1000, 4096 + 37])
def test_pointnet_scatter_against_unfused_ops(n):
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(21)
    b, t, h, w = 2, 3, 64, 48
    xyzi = torch.randn((b, t, 7, n, 1), generator=gen).to(DEV)
    coord = torch.cat((_model_like_coords(gen, b * t, n, h, w), torch.rand((b * t, n, 1), generator=gen)), -1)
    coord = coord.view(b, t, n, 3, 1).to(DEV)
    w1, b1 = (torch.randn((64, 7, 1, 1), generator=gen) * 0.4).to(DEV), torch.randn(64, generator=gen).to(DEV) * 0.1
    w2, b2 = (torch.randn((64, 64, 1, 1), generator=gen) * 0.15).to(DEV), torch.randn(64, generator=gen).to(DEV) * 0.1
    bev = torch.zeros((b, h, w, t * 64), device=DEV)
    rows = torch.full((b, n, 192), -7.0, device=DEV)
    ops.pointnet_scatter(xyzi, coord, w1, b1, w2, b2, bev, pts_out=rows[:, :, 64:128])
    x = xyzi.view(b * t, 7, n, 1).double()
    pts = F.relu(F.conv2d(F.relu(F.conv2d(x, w1.double(), b1.double())), w2.double(), b2.double())).float()
    want = torch.zeros((b * t, 64, h, w), device=DEV)
    ops.voxel_maxpool_fwd(pts, coord.view(b * t, n, 3)[:, :, :2].contiguous(), want, (h, w), (1.0, 1.0))
    got = bev.permute(0, 3, 1, 2).reshape(b, t, 64, h, w).reshape(b * t, 64, h, w)
    # same max-scatter, features from an fp32 fma chain vs an fp64 reference: 1e-5 relative
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    assert ((got > 0) == (want > 0)).float().mean().item() > 0.9999
    pts0 = pts.view(b, t, 64, n)[:, 0].permute(0, 2, 1)
    assert (rows[:, :, 64:128] - pts0).abs().max().item() <= 1e-5 * pts0.abs().max().item()
    assert (rows[:, :, :64] == -7.0).all() and (rows[:, :, 128:] == -7.0).all()


@pytest.mark.parametrize("b,n,m3", [(2, 5000 + 13, 3), (1, 64, 3), (3, 1000, 20)])
def test_point_head_against_float64_reference(b, n, m3):
    """CatFusion + PredBranch as one kernel (MFMA chain, intermediates in registers) vs the three 1x1 layers in float64.
    fp32 fma chains in a different order than any GEMM library: 2e-5 of the output range."""
    gen = torch.Generator(device="cpu").manual_seed(41)
    rows = torch.randn((b, n, 200), generator=gen).to(DEV)[:, :, :192]                       # pitch 200 floats
    l1 = ((torch.randn((96, 192, 1, 1), generator=gen) * 0.1).to(DEV), (torch.randn(96, generator=gen) * 0.2).to(DEV))
    l2 = ((torch.randn((64, 96, 1, 1), generator=gen) * 0.15).to(DEV), (torch.randn(64, generator=gen) * 0.2).to(DEV))
    l3 = ((torch.randn((m3, 64, 1, 1), generator=gen) * 0.2).to(DEV), torch.randn(m3, generator=gen).to(DEV))
    wprep, got_m3 = ops.point_head_prepare(l1, l2, l3)
    assert got_m3 == m3
    out = ops.point_head(rows, wprep, m3)
    x = rows.double().reshape(b * n, 192)
    z = torch.relu(x @ l1[0].double().view(96, 192).t() + l1[1].double())
    z = torch.relu(z @ l2[0].double().view(64, 96).t() + l2[1].double())
    want = (z @ l3[0].double().view(m3, 64).t() + l3[1].double()).view(b, n, m3).permute(0, 2, 1)
    assert out.shape == want.shape
    assert (out.double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("ho,wo,sizes,cout", [(64, 48, ((32, 24), (16, 12)), 32), (40, 40, ((20, 20),), 32), (33, 47, ((9, 13), (5, 6)), 32),
                                              (20, 24, ((15, 12), (5, 6)), 32),        # a source taller than half the output: two launches
                                              (256, 256, ((128, 128), (64, 64)), 128),   # the network's geometry: 32-row strips
                                              (1, 9, ((1, 4),), 32), (70, 8, ((3, 3), (35, 4)), 32)])
def test_upconv3x3_equals_conv_of_upsampled_concat(ho, wo, sizes, cout, monkeypatch):
    """conv3x3(cat(x0, up(x1), up(x2))) + bias + LeakyReLU computed as direct conv on x0 + tap GEMMs at source resolution
    + separable interpolation (one launch, or x pass + y pass), against the direct form in float64.  Odd sizes exercise
    the align_corners ratios and the image borders (taps leaving the upsampled image must be dropped, not clamped).  The
    one-launch form runs the same operations in the same order as the pair: equal bits."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(43)
    b, c0, cs = 2, 8, 16
    x0 = torch.randn((b, c0, ho, wo), generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    xs = [torch.randn((b, cs) + hw, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last) for hw in sizes]
    cin = c0 + cs * len(xs)
    w = (torch.randn((cout, cin, 3, 3), generator=gen) * 0.1).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    ups = [F.interpolate(x.double(), size=(ho, wo), mode="bilinear", align_corners=True) for x in xs]
    want = F.leaky_relu(F.conv2d(torch.cat([x0.double()] + ups, 1), w.double(), bias.double(), 1, 1), 0.01)
    conv_a = F.conv2d(x0, w[:, :c0].contiguous(memory_format=torch.channels_last), None, 1, 1)
    assert conv_a.is_contiguous(memory_format=torch.channels_last)
    srcs = [(x, ops.upconv_tap_weights(w, c0 + i * cs, c0 + (i + 1) * cs)) for i, x in enumerate(xs)]
    got = ops.upconv3x3(conv_a.clone(), bias, srcs, 2)
    assert (got.double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    monkeypatch.setattr(ops, "_UPCONV_XY", False)
    assert torch.equal(got, ops.upconv3x3(conv_a.clone(), bias, srcs, 2))
    monkeypatch.setattr(ops, "_UPCONV_XY", True)
    # other weights at (very likely) the same addresses: nothing derived from the first set may survive in a cache
    # keyed by a weight's address (a freed model's memory is handed to the next model by the caching allocator)
    ptr = w.data_ptr()
    del srcs, w
    w = (torch.randn((cout, cin, 3, 3), generator=gen) * 0.1).to(DEV)
    want2 = F.leaky_relu(F.conv2d(torch.cat([x0.double()] + ups, 1), w.double(), bias.double(), 1, 1), 0.01)
    conv_a = F.conv2d(x0, w[:, :c0].contiguous(memory_format=torch.channels_last), None, 1, 1)
    srcs = [(x, ops.upconv_tap_weights(w, c0 + i * cs, c0 + (i + 1) * cs)) for i, x in enumerate(xs)]
    got2 = ops.upconv3x3(conv_a, bias, srcs, 2)
    assert (got2.double() - want2).abs().max().item() <= 2e-5 * want2.abs().max().item(), (ptr == w.data_ptr())


@pytest.mark.parametrize("hw,m,pnt", [((64, 64), 4, 4), ((20, 36), 2, 8), ((7, 5), 4, 3)])
def test_msda_fwd_qp_equals_module_formulation(hw, m, pnt):
    """The sampler fed by the raw query projection (softmax, offset normalisation and reference points folded in) against
    the module's own sequence softmax -> ref + off / (W, H) -> smos_msda_fwd.  Same arithmetic except the order of the
    P-term softmax sum: 1e-5 of the output range."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(53)
    h, w = hw
    n, lq, d = 2, h * w, 32
    value = torch.randn((n, lq, m, d), generator=gen).to(DEV)
    qp = torch.cat((torch.randn((n, lq, m * pnt * 2), generator=gen) * 3.0, torch.randn((n, lq, m * pnt), generator=gen)), -1).to(DEV)
    got = ops.msda_fwd_qp(value, qp, h, w, pnt)
    shapes = torch.tensor([[h, w]], dtype=torch.long, device=DEV)
    lsi = torch.zeros((1,), dtype=torch.long, device=DEV)
    ys = (torch.arange(h, dtype=torch.float32, device=DEV) + 0.5) / h
    xs = (torch.arange(w, dtype=torch.float32, device=DEV) + 0.5) / w
    ref = torch.stack((xs[None, :].expand(h, w), ys[:, None].expand(h, w)), -1).reshape(1, lq, 1, 1, 1, 2)
    norm = torch.tensor([w, h], dtype=torch.float32, device=DEV)
    off = qp[..., :m * pnt * 2].reshape(n, lq, m, 1, pnt, 2)
    attn = F.softmax(qp[..., m * pnt * 2:].reshape(n, lq, m, pnt), -1).view(n, lq, m, 1, pnt)
    want = ops.msda_fwd(value, shapes, lsi, (ref + off / norm).contiguous(), attn.contiguous())
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("c", [64, 128, 512])
def test_add_layer_norm_against_torch(c):
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(47)
    x = (torch.randn((3, 1000, c), generator=gen) * 3 + 1).to(DEV)
    r = torch.randn((3, 1000, c), generator=gen).to(DEV)
    g, b = torch.randn(c, generator=gen).to(DEV), torch.randn(c, generator=gen).to(DEV)
    want = F.layer_norm((x + r).double(), (c,), g.double(), b.double(), 1e-5)
    got = ops.add_layer_norm(x, r, g, b, 1e-5)
    assert (got.double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    got1 = ops.add_layer_norm(x, None, g, b, 1e-5)
    want1 = F.layer_norm(x.double(), (c,), g.double(), b.double(), 1e-5)
    assert (got1.double() - want1).abs().max().item() <= 1e-5 * want1.abs().max().item()


def test_pointnet_scatter_run_boundaries_at_cell_zero():
    """Regression: cell 0 is a legal cell.  Run ends are found with a lane shuffle; evaluated under a partial exec
    mask a lane reading a masked-off neighbour gets 0, which made a point of cell 0 at position 30 of a 32-point tile
    merge into the following point's run (seen at full size only: one point in 1.9 M)."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(29)
    b, t, h, w, n = 1, 3, 32, 32, 32 * 40
    xyzi = torch.randn((b, t, 7, n, 1), generator=gen).to(DEV)
    coord = torch.cat((_model_like_coords(gen, b * t, n, h, w), torch.rand((b * t, n, 1), generator=gen)), -1)
    for pos in (0, 1, 15, 29, 30, 31):                      # lone cell-0 points at several tile positions
        coord[:, pos + 64 * (pos % 5)::32 * 7, 0] = 0.41
        coord[:, pos + 64 * (pos % 5)::32 * 7, 1] = 0.19
    coord[:, 32 * 20:32 * 22, :2] = 0.5                     # and a run of cell 0 spanning two tiles
    coord = coord.view(b, t, n, 3, 1).to(DEV)
    w1, b1 = (torch.randn((64, 7, 1, 1), generator=gen) * 0.4).to(DEV), torch.randn(64, generator=gen).to(DEV) * 0.1
    w2, b2 = (torch.randn((64, 64, 1, 1), generator=gen) * 0.15).to(DEV), torch.randn(64, generator=gen).to(DEV) * 0.1
    bev = torch.zeros((b, h, w, t * 64), device=DEV)
    ops.pointnet_scatter(xyzi, coord, w1, b1, w2, b2, bev)
    x = xyzi.view(b * t, 7, n, 1).double()
    pts = F.relu(F.conv2d(F.relu(F.conv2d(x, w1.double(), b1.double())), w2.double(), b2.double())).float()
    want = torch.zeros((b * t, 64, h, w), device=DEV)
    ops.voxel_maxpool_fwd(pts, coord.view(b * t, n, 3)[:, :, :2].contiguous(), want, (h, w), (1.0, 1.0))
    got = bev.permute(0, 3, 1, 2).reshape(b, t, 64, h, w).reshape(b * t, 64, h, w)
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    assert want[:, :, 0, 0].max().item() > 0


@pytest.mark.parametrize("c,hw_g,hw_o,sg,ss", [(32, (64, 64), (8, 256), (0.5, 0.5), (0.5, 0.5)),
                                              (64, (16, 128), (32, 32), (0.25, 0.25), (0.25, 0.25))])
def test_gather_scatter_against_unfused_ops(c, hw_g, hw_o, sg, ss):
    gen = torch.Generator(device="cpu").manual_seed(23)
    b, n = 2, 5000 + 13
    grid = torch.relu(torch.randn((b, 2 * c) + hw_g, generator=gen)).to(DEV)[:, c:]          # a channel slice (strided)
    gcoord = _model_like_coords(gen, b, n, hw_g[0] / sg[0], hw_g[1] / sg[1]).to(DEV)
    scoord = _model_like_coords(gen, b, n, hw_o[0] / ss[0], hw_o[1] / ss[1]).to(DEV)
    out = torch.zeros((b,) + hw_o + (c,), device=DEV)
    rows = torch.zeros((b, n, c + 8), device=DEV)
    ops.gather_scatter(grid, gcoord, sg, scoord, ss, out=out, pts_out=rows[:, :, 8:])
    pts = ops.bilinear_gather(grid, gcoord, sg)
    want = torch.zeros((b, c) + hw_o, device=DEV)
    ops.voxel_maxpool_fwd(pts, scoord, want, hw_o, ss)
    assert torch.equal(rows[:, :, 8:], pts.permute(0, 2, 1))
    assert torch.equal(out.permute(0, 3, 1, 2), want)
    nchw = torch.full((b, c + 5) + hw_o, 3.0, device=DEV)
    ops.nhwc_to_nchw(out, nchw[:, 5:])
    assert torch.equal(nchw[:, 5:], want) and (nchw[:, :5] == 3.0).all()
    rows2 = torch.zeros((b, n, c), device=DEV)
    ops.gather_scatter(grid, gcoord, sg, pts_out=rows2)                                       # gather only
    assert torch.equal(rows2, pts.permute(0, 2, 1))


def test_downsample_epilogue_channels_last_kernel():
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(29)
    for stride, hw in ((2, (64, 200)), (1, (16, 70))):
        p = torch.randn((2, 64) + hw, generator=gen).to(DEV)
        a = torch.randn((2, 64, (hw[0] - 1) // stride + 1, (hw[1] - 1) // stride + 1), generator=gen).to(DEV)
        bias = torch.randn(64, generator=gen).to(DEV)
        want = torch.relu(a + bias[None, :, None, None] + F.max_pool2d(p, 3, stride, 1))
        big = torch.zeros((2, 80) + tuple(a.shape[2:]), device=DEV)
        ops.downsample_epilogue(a.contiguous(memory_format=torch.channels_last), p.contiguous(memory_format=torch.channels_last),
                                bias, stride, out=big[:, 16:])
        assert (big[:, 16:] - want).abs().max().item() < 1e-6 and big[:, :16].abs().max().item() == 0


def test_vote_full_size_local_map_bit_exact():
    """A realistic local map (8 history scans + the current one, 120k points each, ~1 M points) through the packed
    vote table against the numpy restatement of the reference's dense histogram + argmax."""
    from streammos_amd import preprocess, synth
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
    # bit-exact, as north_star demands for the voting indices.  The float64 pose product is written in the dgemm
    # micro-kernel's order with explicit fma (csrc/smos_common.h::pose_row_f64); a different float64 summation order
    # could only show after the rounding to float32 with probability ~1e-9 per coordinate (measured: 0 of 1.44 M
    # pose-aligned coordinates, 0 of 120 000 refined labels differ; tools/diag_bitexact.py).
    assert np.array_equal(got, want), int((got != want).sum())
    # the one-launch form of the same window (what the streaming runner issues): identical table, word for word;
    # 14 frames exercise the split into launches of at most 12, an empty frame in the middle is skipped
    window = [(_t(scans[k]), _t(preds[k]), inv_cur.dot(poses[k])) for k in range(cur_id - 1, cur_id - 9, -1)]
    window.append((_t(scans[cur_id]), _t(preds[cur_id]), None))
    batched = torch.empty_like(table)
    ops.vote_clear(batched)
    ops.vote_accumulate_frames(window, batched)
    assert torch.equal(batched, table)
    empty = (torch.zeros((0, 4), device=DEV), torch.zeros(0, dtype=torch.uint8, device=DEV), None)
    small = [(w[0][:9000], w[1][:9000], w[2]) for w in (window + window[:5])]
    ops.vote_clear(batched)
    ops.vote_accumulate_frames(small[:7] + [empty] + small[7:], batched)
    ops.vote_clear(table)
    for p, l, d in small:
        ops.vote_accumulate(p, l, table, pose_diff=d)
    assert torch.equal(batched, table) and int(table.sum().item()) > 0


def test_zero_sized_inputs_are_no_ops():
    out = torch.zeros((2, 4, 8, 8), device=DEV)
    ops.voxel_maxpool_fwd(torch.zeros((2, 4, 0), device=DEV), torch.zeros((2, 0, 2), device=DEV), out, (8, 8), (1.0, 1.0))
    assert out.abs().max().item() == 0
    assert ops.bilinear_gather(torch.randn(2, 4, 8, 8, device=DEV), torch.zeros((2, 0, 2), device=DEV), (1.0, 1.0)).shape == (2, 4, 0)
    table = torch.zeros(512 * 512 * 30, dtype=torch.int64, device=DEV)
    ops.vote_accumulate(torch.zeros((0, 4), device=DEV), torch.zeros(0, dtype=torch.uint8, device=DEV), table)
    assert ops.vote_resolve(torch.zeros((0, 4), device=DEV), torch.zeros(0, dtype=torch.uint8, device=DEV), table).numel() == 0
    assert table.abs().max().item() == 0


# ------------------------------------------------------------------------------------------
# channels-last engine kernels
# ------------------------------------------------------------------------------------------
def _to_cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def test_channels_last_elementwise_kernels_against_torch():
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(31)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    # bias_act_cl into a channel slice of a wider buffer, with a residual that is itself a slice
    x, res, bias = _to_cl(r(3, 32, 10, 12)), _to_cl(r(3, 64, 10, 12)), r(32)
    big = ops.empty_cl(3, 96, 10, 12, DEV, zero=True)
    for act, fn in ((0, lambda t: t), (1, torch.relu), (2, lambda t: F.leaky_relu(t, 0.01))):
        ops.bias_act_cl(x, bias, act, out=big[:, 32:64], residual=res[:, 16:48])
        want = fn(x + bias[None, :, None, None] + res[:, 16:48])
        assert (big[:, 32:64] - want).abs().max().item() < 1e-6
        assert big[:, :32].abs().max().item() == 0 and big[:, 64:].abs().max().item() == 0
    # downsample epilogue
    for stride, hw in ((2, (32, 48)), (1, (8, 64)), (2, (31, 17))):
        p, bias = _to_cl(r(2, 32, *hw)), r(32)
        a = _to_cl(r(2, 32, (hw[0] - 1) // stride + 1, (hw[1] - 1) // stride + 1))
        want = torch.relu(a + bias[None, :, None, None] + F.max_pool2d(p, 3, stride, 1))
        got = ops.downsample_epilogue_cl(a, p, bias, stride)
        assert (got - want).abs().max().item() < 1e-6
    # channel gate + residual
    for c, hw in ((32, (16, 40)), (128, (64, 64))):
        y, xr, bias = _to_cl(r(2, c, *hw)), _to_cl(r(2, c, *hw)), r(c)
        w1, b1, w2, b2 = r(c // 4, c), r(c // 4), r(c, c // 4), r(c)
        z = y + bias[None, :, None, None]
        g = torch.sigmoid(F.linear(torch.relu(F.linear(z.mean((2, 3)), w1, b1)), w2, b2))
        want = torch.relu(z * g[:, :, None, None] + xr)
        got = ops.channel_gate_residual_cl(y, bias, w1, b1, w2, b2, xr, torch.zeros((hw[0] * hw[1] // 512 + 2) * 2 * c, device=DEV))
        # N(0,1) gate weights over up to 128 channels: the pre-sigmoid sums are O(30); float32 summation order
        assert (got - want).abs().max().item() < 1e-4 * want.abs().max().item()
    # upsample + concat
    a, b, c = _to_cl(r(2, 8, 16, 24)), _to_cl(r(2, 12, 8, 12)), _to_cl(r(2, 4, 4, 6))
    want = torch.cat([F.interpolate(t, size=(16, 24), mode="bilinear", align_corners=True) for t in (a, b, c)], 1)
    got = ops.upsample_concat_cl([a, b, c], (16, 24))
    assert torch.equal(got[:, :8], a) and (got - want).abs().max().item() < 1e-5


@pytest.mark.parametrize("c,hw_g,hw_o,sg,ss", [(32, (64, 64), (8, 256), (0.5, 0.5), (0.5, 0.5)),
                                              (64, (16, 128), (32, 32), (0.25, 0.25), (0.25, 0.25))])
def test_gather_scatter_channels_last_against_unfused_ops(c, hw_g, hw_o, sg, ss):
    gen = torch.Generator(device="cpu").manual_seed(37)
    b, n = 2, 5000 + 13
    wide = _to_cl(torch.relu(torch.randn((b, 2 * c) + hw_g, generator=gen)).to(DEV))
    grid = wide[:, c:]                                                        # channel slice of a wider cl buffer
    gcoord = _model_like_coords(gen, b, n, hw_g[0] / sg[0], hw_g[1] / sg[1]).to(DEV)
    scoord = _model_like_coords(gen, b, n, hw_o[0] / ss[0], hw_o[1] / ss[1]).to(DEV)
    target = ops.empty_cl(b, 2 * c, hw_o[0], hw_o[1], DEV, zero=True)
    rows = torch.zeros((b, n, c + 8), device=DEV)
    ops.gather_scatter_cl(grid, gcoord, sg, scoord, ss, out=target[:, c:], pts_out=rows[:, :, 8:])
    pts = ops.bilinear_gather(grid.contiguous(), gcoord, sg)
    want = torch.zeros((b, c) + hw_o, device=DEV)
    ops.voxel_maxpool_fwd(pts, scoord, want, hw_o, ss)
    assert (rows[:, :, 8:] - pts.permute(0, 2, 1)).abs().max().item() <= 1e-6
    assert (target[:, c:] - want).abs().max().item() <= 1e-6 and target[:, :c].abs().max().item() == 0
    rows2 = torch.zeros((b, n, c), device=DEV)
    ops.gather_scatter_cl(grid, gcoord, sg, pts_out=rows2)
    assert torch.equal(rows2, rows[:, :, 8:])


@pytest.mark.parametrize("c,hw_g,hw_o,scale", [(64, (32, 128), (16, 64), 0.25), (32, (64, 64), (8, 256), 0.5)])
def test_gather_scatter_cl_float4_lanes_equal_the_one_lane_per_channel_kernel(c, hw_g, hw_o, scale):
    """csrc/cl_kernels.hip: gather_scatter_cl4 (a row read by C / 4 lanes as float4, 256 / C points per wave instruction; what
    the engine's 16-byte aligned maps get) against gather_scatter_cl (one lane per channel; what rows that are not 16-byte
    aligned still get -- selected here by an odd pitch): the same expression per point and a maximum per cell, so the scatter
    target and the point rows are equal bit for bit.  Ragged point count, padding tail, points outside both maps."""
    gen = torch.Generator(device="cpu").manual_seed(43)
    b, n = 3, 64 * 37 + 21
    grid = _to_cl(torch.relu(torch.randn((b, c) + hw_g, generator=gen)).to(DEV))
    gcoord = _model_like_coords(gen, b, n, hw_g[0] / scale, hw_g[1] / scale).to(DEV)
    scoord = _model_like_coords(gen, b, n, hw_o[0] / scale, hw_o[1] / scale).to(DEV)
    gcoord[:, -300:] = -4864.0
    scoord[:, -300:] = -4864.0
    res = []
    for pad in (0, 1):                                  # pad = 1: rows of C + 1 floats -> not 16-byte aligned -> the scalar-lane kernel
        tgt_w = torch.zeros((b, hw_o[0], hw_o[1], c + pad), device=DEV)
        rows_w = torch.full((b, n, c + pad), 5.0, device=DEV)
        ops.gather_scatter_cl(grid, gcoord, (scale, scale), scoord, (scale, scale), out=tgt_w[..., :c].permute(0, 3, 1, 2),
                              pts_out=rows_w[:, :, :c])
        res.append((tgt_w[..., :c].clone(), rows_w[:, :, :c].clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][0].abs().max().item() > 0 and bool((res[0][1][:, -300:] == 0).all())


def test_gather_scatter_cl_takes_strided_coordinate_views_and_zero_views_fills_slices():
    """smos_gather_scatter_cl_view: the coordinates as slices of the reference's [B, T, N, 3, 1] / [B, T, N, 2, 1] tensors (pitch 3,
    batch stride T * N * 3) give what their compacted copies give, bit for bit.  smos_zero_views_cl: a dense map and a channel slice
    of a wider one zeroed in one launch, the other half of the wide map untouched."""
    gen = torch.Generator(device="cpu").manual_seed(47)
    b, t, n, c, scale = 3, 3, 64 * 11 + 5, 64, 0.25
    hw_g, hw_o = (32, 128), (16, 64)
    grid = _to_cl(torch.relu(torch.randn((b, c) + hw_g, generator=gen)).to(DEV))
    coord5 = torch.randn((b, t, n, 3, 1), generator=gen).to(DEV)
    sphere5 = torch.randn((b, t, n, 2, 1), generator=gen).to(DEV)
    coord5[:, 0, :, :2, 0] = _model_like_coords(gen, b, n, hw_g[0] / scale, hw_g[1] / scale).to(DEV)
    sphere5[:, 0, :, :, 0] = _model_like_coords(gen, b, n, hw_o[0] / scale, hw_o[1] / scale).to(DEV)
    gview, sview = coord5[:, 0, :, :2, 0], sphere5[:, 0, :, :, 0]
    assert not gview.is_contiguous() and gview.stride() == (t * n * 3, 3, 1)
    res = []
    for g, s_ in ((gview, sview), (gview.contiguous(), sview.contiguous())):
        wide = torch.full((b, hw_o[0], hw_o[1], 2 * c), 3.0, device=DEV).permute(0, 3, 1, 2)
        dense = torch.full((b, hw_o[0], hw_o[1], c), 7.0, device=DEV).permute(0, 3, 1, 2)
        ops.zero_views_cl([dense, wide[:, c:]])
        assert float(dense.abs().max()) == 0.0 and float(wide[:, c:].abs().max()) == 0.0 and float((wide[:, :c] - 3.0).abs().max()) == 0.0
        rows = torch.empty((b, n, c), device=DEV)
        ops.gather_scatter_cl(grid, g, (scale, scale), s_, (scale, scale), out=wide[:, c:], pts_out=rows)
        res.append((wide[:, c:].clone(), rows))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][0].abs().max().item() > 0
    with pytest.raises(RuntimeError):
        ops.gather_scatter_cl(grid, coord5[:, 0, :, :, 0].transpose(1, 2), (scale, scale), pts_out=torch.empty((b, n, c), device=DEV))
    with pytest.raises(RuntimeError):
        ops.zero_views_cl([])


@pytest.mark.parametrize("c,hw_g,hw_o,scale", [(64, (16, 512), (128, 128), 0.25), (32, (256, 256), (32, 1024), 0.5)])
def test_gather_scatter_channels_last_full_size(c, hw_g, hw_o, scale):
    """The engine's cross-view transfers at the validation shape (4 x 160 000 points incl. the padding tail at -1000):
    fused gather + max scatter against bilinear gather + VoxelMaxPool, bit for bit; point rows for every point."""
    gen = torch.Generator(device="cpu").manual_seed(41)
    b, n = 4, 160000
    grid = _to_cl(torch.relu(torch.randn((b, c) + hw_g, generator=gen)).to(DEV))
    gcoord = _model_like_coords(gen, b, n, hw_g[0] / scale, hw_g[1] / scale).to(DEV)
    scoord = _model_like_coords(gen, b, n, hw_o[0] / scale, hw_o[1] / scale).to(DEV)
    gcoord[:, -40000:] = -4864.0
    scoord[:, -40000:] = -4864.0
    target = ops.empty_cl(b, c, hw_o[0], hw_o[1], DEV, zero=True)
    rows = torch.empty((b, n, c), device=DEV)
    ops.gather_scatter_cl(grid, gcoord, (scale, scale), scoord, (scale, scale), out=target, pts_out=rows)
    pts = ops.bilinear_gather(grid.contiguous(), gcoord, (scale, scale))
    want = torch.zeros((b, c) + hw_o, device=DEV)
    ops.voxel_maxpool_fwd(pts, scoord, want, hw_o, (scale, scale))
    assert torch.equal(rows, pts.permute(0, 2, 1)) and torch.equal(target, want)


def test_msda_module_forward_matches_reference_golden(golden):
    """MSDeformAttn.forward (deformattn/modules/ms_deform_attn.py:78-116) as a module: value / offset / weight
    projections, softmax over the points, offset normalisation, the HIP sampler, output projection -- against the
    reference module's output on the same seeded weights (tests/golden/make_golden.py::gen_msda)."""
    from streammos_amd import synth
    from streammos_amd.refapi.deformattn.modules import MSDeformAttn
    g = golden("ops_msda")
    q, ref, src, shapes, lsi = cases.msda_module_case()
    check_inputs(g, "msda_module_in_sha", q, ref, src)
    mod = MSDeformAttn(d_model=128, n_levels=1, n_heads=4, n_points=4).eval()
    sd = synth.seeded_state_dict({("cross_attn." + k): v for k, v in mod.state_dict().items()})
    mod.load_state_dict({k[len("cross_attn."):]: v for k, v in sd.items()})
    mod = mod.to(DEV)
    with torch.no_grad():
        y = mod(_t(q), _t(ref), _t(src), _t(shapes), _t(lsi))
    want = g["msda_module_out"]
    err = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    print("MSDeformAttn.forward vs reference: %.2e of the output range" % err)
    assert y.shape == want.shape and err <= 1e-5


# every convolution shape of the network (networks/multi_view_encoder.py:344-375,478-497; backbone.py:14-34,136-159) in small,
# plus odd sizes (partial 32-pixel tiles, a row count that is not a multiple of 4) and every mt
_CONV_CASES = [
    # cin, cout, (kh, kw), stride, (h, w), mt, act, residual
    (32, 32, (3, 3), 1, (40, 64), 1, 1, False),
    (32, 32, (7, 3), 1, (20, 64), 1, 1, False),
    (32, 32, (3, 7), 1, (20, 64), 1, 1, False),
    (64, 32, (3, 3), 1, (24, 96), 1, 1, True),
    (64, 64, (3, 3), 2, (48, 64), 1, 0, False),
    (64, 64, (3, 3), 2, (48, 64), 2, 0, False),
    (64, 64, (1, 1), 1, (24, 64), 2, 0, False),
    (64, 64, (5, 3), 1, (24, 32), 2, 1, False),
    (64, 64, (3, 5), 1, (24, 32), 1, 1, False),
    (128, 64, (3, 3), 1, (16, 32), 2, 1, True),
    (128, 128, (3, 3), 1, (16, 32), 1, 1, True),
    (128, 128, (3, 3), 2, (32, 64), 2, 0, False),
    (128, 128, (1, 1), 1, (16, 32), 4, 0, False),
    (64, 128, (3, 3), 1, (16, 64), 4, 2, False),
    (128, 64, (3, 3), 1, (16, 64), 2, 2, False),
    (32, 32, (3, 3), 1, (13, 45), 1, 2, True),          # partial tiles in both directions
    (64, 64, (3, 3), 2, (27, 51), 2, 1, False),         # odd input under stride 2
    (32, 64, (3, 3), 1, (3, 8), 1, 1, False),           # smaller than one tile
    (128, 1152, (1, 1), 1, (8, 32), 4, 0, False),       # the decoder's nine tap products as one 1x1 convolution (Cout > 1024)
]


@pytest.mark.parametrize("cin,cout,kernel,stride,hw,mt,act,with_res", _CONV_CASES)
def test_conv_cl_against_float64(cin, cout, kernel, stride, hw, mt, act, with_res):
    """csrc/conv_igemm.hip (streamed-weight MFMA implicit GEMM, bias + activation + residual fused) against conv2d in
    float64; input / residual / output are channel slices of wider channels-last buffers.  fp32 MFMA = k-ordered fmaf
    chain: <= 2e-5 of the output range for K up to 1152."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(61)
    b, (h, w) = 2, hw
    kh, kw = kernel
    wide = torch.randn((b, h, w, cin + 32), generator=gen).to(DEV)
    x = wide[..., 32:].permute(0, 3, 1, 2)                                   # channel slice, pitch cin + 32
    ho, wo = (h + 2 * (kh // 2) - kh) // stride + 1, (w + 2 * (kw // 2) - kw) // stride + 1
    res_wide = torch.randn((b, ho, wo, cout + 16), generator=gen).to(DEV)
    res = res_wide[..., :cout].permute(0, 3, 1, 2) if with_res else None
    wt = (torch.randn((cout, cin, kh, kw), generator=gen) * (2.0 / (cin * kh * kw)) ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    out_wide = torch.full((b, ho, wo, cout + 8), 7.0, device=DEV)
    out = out_wide[..., 4:4 + cout].permute(0, 3, 1, 2)
    got = ops.conv_cl(x, ops.conv_prepare(wt, mt), bias, act, cout, kernel, stride=stride, mt=mt, residual=res, out=out)
    want = F.conv2d(x.double(), wt.double(), bias.double(), stride, (kh // 2, kw // 2))
    if with_res:
        want = want + res.double()
    want = F.relu(want) if act == 1 else (F.leaky_relu(want, 0.01) if act == 2 else want)
    assert got.shape == want.shape
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    assert err <= 2e-5, err
    assert bool((out_wide[..., :4] == 7.0).all()) and bool((out_wide[..., 4 + cout:] == 7.0).all())   # neighbours untouched
    # no bias, no residual, fresh output
    got2 = ops.conv_cl(x, ops.conv_prepare(wt, mt), None, 0, cout, kernel, stride=stride, mt=mt)
    want2 = F.conv2d(x.double(), wt.double(), None, stride, (kh // 2, kw // 2))
    assert (got2.double() - want2).abs().max().item() <= 2e-5 * want2.abs().max().item()


def test_conv_cl_is_deterministic_and_batch_independent():
    """Same input twice -> same bits (no atomics, fixed summation order); a sample's result does not depend on what else
    is in the batch (what lets 8 concurrent streams equal their solo runs bit for bit)."""
    gen = torch.Generator(device="cpu").manual_seed(67)
    x = torch.randn((3, 64, 64, 64), generator=gen).to(DEV).permute(0, 3, 1, 2)
    wt = (torch.randn((64, 64, 3, 3), generator=gen) * 0.05).to(DEV)
    wp = ops.conv_prepare(wt, 2)
    a = ops.conv_cl(x, wp, None, 1, 64, (3, 3), mt=2)
    b = ops.conv_cl(x, wp, None, 1, 64, (3, 3), mt=2)
    c = ops.conv_cl(x[1:2], wp, None, 1, 64, (3, 3), mt=2)
    assert torch.equal(a, b) and torch.equal(a[1:2], c)


@pytest.mark.parametrize("cin,cout,hw,mt", [(32, 32, (40, 64), 1), (64, 64, (13, 45), 2), (128, 128, (16, 32), 4),
                                            (64, 64, (13, 45), 1), (32, 32, (3, 8), 1)])
def test_conv_cl_channel_sums_and_the_gate_built_on_them(cin, cout, hw, mt):
    """conv_cl(chan_sums=...): the epilogue also leaves the channel sums of every 32-pixel output row segment (DPP reduction
    in a fixed order; segments and rows outside the image hold 0), and channel_gate_apply_cl builds ChannelAtt + residual
    (networks/backbone.py:57-73, 87-102) on them -- against float64 and against the three-pass kernel it replaces."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(71)
    b, (h, w) = 3, hw
    x = torch.randn((b, h, w, cin), generator=gen).to(DEV).permute(0, 3, 1, 2)
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (cin * 9)) ** 0.5).to(DEV)
    chunks = ops.conv_sum_chunks(h, w)
    sums = torch.full((b, chunks, cout), 7.0, device=DEV)
    y = ops.conv_cl(x, ops.conv_prepare(wt, mt), None, 0, cout, (3, 3), mt=mt, chan_sums=sums)
    y_plain = ops.conv_cl(x, ops.conv_prepare(wt, mt), None, 0, cout, (3, 3), mt=mt)
    assert torch.equal(y, y_plain)                                        # the output itself is untouched
    # segment sums: chunk = ((y // 4) * ceil(W / 32) + x // 32) * 4 + y % 4
    xt = (w + 31) // 32
    want = torch.zeros((b, chunks, cout), dtype=torch.float64, device=DEV)
    yd = y.double()
    for yy in range(h):
        for t in range(xt):
            want[:, ((yy // 4) * xt + t) * 4 + yy % 4] = yd[:, :, yy, 32 * t:32 * t + 32].sum(-1)
    scale = yd.abs().sum((2, 3)).max().item() / (h * xt)
    assert (sums.double() - want).abs().max().item() <= 1e-5 * scale
    sums2 = torch.empty_like(sums)
    ops.conv_cl(x, ops.conv_prepare(wt, mt), None, 0, cout, (3, 3), mt=mt, chan_sums=sums2)
    assert torch.equal(sums, sums2)                                       # fixed summation order
    assert (sums.sum(1).double() - yd.sum((2, 3))).abs().max().item() <= 1e-5 * yd.abs().sum((2, 3)).max().item()
    # the gate on top of them
    cr = max(cout // 4, 8)
    bias, b1, b2 = (torch.randn(n, generator=gen).to(DEV) for n in (cout, cr, cout))
    w1 = (torch.randn((cr, cout), generator=gen) * 0.3).to(DEV)
    w2 = (torch.randn((cout, cr), generator=gen) * 0.3).to(DEV)
    xres = torch.randn((b, h, w, cout), generator=gen).to(DEV).permute(0, 3, 1, 2)
    got = ops.channel_gate_apply_cl(y, bias, w1, b1, w2, b2, xres, sums, torch.empty(b * cout, device=DEV))
    z = yd + bias.double()[None, :, None, None]
    g = torch.sigmoid(F.linear(F.relu(F.linear(z.mean((2, 3)), w1.double(), b1.double())), w2.double(), b2.double()))
    ref = F.relu(z * g[:, :, None, None] + xres.double())
    assert (got.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    ws = torch.zeros(b * cout * (h * w // 512 + 2), device=DEV)
    old = ops.channel_gate_residual_cl(y, bias, w1, b1, w2, b2, xres, ws)
    assert (got - old).abs().max().item() <= 2e-6 * ref.abs().max().item()


_ROWS_CASES = [
    # cin, cout, (kh, kw), (h, w), act, residual, sums, mt
    (32, 32, (3, 3), (40, 64), 1, False, False, 1),
    (32, 32, (7, 3), (20, 64), 1, False, False, 1),
    (32, 32, (3, 7), (20, 64), 1, False, False, 1),
    (64, 32, (3, 3), (24, 96), 1, True, False, 1),
    (64, 64, (5, 3), (24, 32), 1, False, False, 1),
    (64, 64, (3, 5), (24, 32), 2, False, False, 1),
    (128, 128, (3, 3), (16, 32), 1, True, False, 1),
    (128, 64, (3, 3), (16, 64), 0, False, True, 1),
    (32, 32, (3, 3), (13, 45), 2, True, False, 1),          # partial tiles in both directions
    (32, 64, (3, 3), (3, 8), 1, False, True, 1),            # smaller than one tile
    (32, 32, (1, 3), (9, 70), 0, False, False, 1),
    (64, 64, (3, 3), (24, 40), 1, False, False, 2),
    (128, 64, (3, 3), (16, 64), 2, True, False, 2),
    (64, 128, (5, 3), (12, 33), 0, False, True, 2),
    (64, 64, (3, 7), (8, 32), 1, True, False, 2),
]


@pytest.mark.parametrize("cin,cout,kernel,hw,act,with_res,with_sums,mt", _ROWS_CASES)
def test_conv_rows_cl_equals_conv_cl(cin, cout, kernel, hw, act, with_res, with_sums, mt):
    """csrc/conv_rows.hip (input rows staged through LDS once per kernel row) against float64 and against conv_cl: the
    same fmaf chains in a different stage order (ky, chunk, kx) -- <= 2e-5 of the output range vs float64; channel slices,
    residual, channel sums as for conv_cl."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(67)
    b, (h, w) = 2, hw
    kh, kw = kernel
    wide = torch.randn((b, h, w, cin + 32), generator=gen).to(DEV)
    x = wide[..., 32:].permute(0, 3, 1, 2)
    res_wide = torch.randn((b, h, w, cout + 16), generator=gen).to(DEV)
    res = res_wide[..., :cout].permute(0, 3, 1, 2) if with_res else None
    wt = (torch.randn((cout, cin, kh, kw), generator=gen) * (2.0 / (cin * kh * kw)) ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    out_wide = torch.full((b, h, w, cout + 8), 7.0, device=DEV)
    out = out_wide[..., 4:4 + cout].permute(0, 3, 1, 2)
    sums = torch.full((b, ops.conv_sum_chunks(h, w), cout), 3.0, device=DEV) if with_sums else None
    got = ops.conv_rows_cl(x, ops.conv_prepare(wt, mt, order="rows"), bias, act, cout, kernel, mt=mt, residual=res, out=out, chan_sums=sums)
    want = F.conv2d(x.double(), wt.double(), bias.double(), 1, (kh // 2, kw // 2))
    if with_res:
        want = want + res.double()
    want = F.relu(want) if act == 1 else (F.leaky_relu(want, 0.01) if act == 2 else want)
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    print("conv_rows_cl %d->%d k%dx%d %dx%d: rel err %.2e" % (cin, cout, kh, kw, h, w, err))
    assert err <= 2e-5
    assert (out_wide[..., :4] == 7.0).all() and (out_wide[..., 4 + cout:] == 7.0).all()
    sums2 = torch.empty_like(sums) if with_sums else None
    ref = ops.conv_cl(x, ops.conv_prepare(wt, mt), bias, act, cout, kernel, mt=mt, residual=res, chan_sums=sums2)
    assert (got - ref).abs().max().item() <= 1e-5 * want.abs().max().item()
    if with_sums:
        assert (sums - sums2).abs().max().item() <= 1e-5 * sums2.abs().max().item()
    again = ops.conv_rows_cl(x, ops.conv_prepare(wt, mt, order="rows"), bias, act, cout, kernel, mt=mt, residual=res)
    assert torch.equal(again, got.contiguous(memory_format=torch.channels_last)) or torch.equal(again, got)


def _binding(kind, name):
    """The ctypes-backed Python module refapi.install() publishes under the reference's pybind name, or the COMPILED
    pybind11 twin (csrc/shim/pybind_shims.cpp = INTEGRATION.md section 3, built by __graft_entry__.build())."""
    if kind == "pybind":
        from streammos_amd.refapi import compiled
        return compiled.load(name)
    if name == "point_deep_cuda_kernel":
        from streammos_amd.refapi.point_deep import cuda_kernel
        return cuda_kernel
    from streammos_amd.refapi import MultiScaleDeformableAttention
    return MultiScaleDeformableAttention


@pytest.mark.parametrize("binding", ["ctypes", "pybind"])
@pytest.mark.parametrize("name", ["basic", "scaled_neg", "dim3", "relu_like"])
def test_pybind_name_point_deep_cuda_kernel_with_the_references_argument_lists(golden, name, binding):
    """`point_deep.cuda_kernel.voxel_maxpooling_forward / _backward` called exactly as the reference's autograd Function
    calls them (deep_point/__init__.py:25-44, :48-61): caller-allocated zero / -1 filled buffers and the four small META
    TENSORS ON THE DEVICE (int64 sizes / strides / output size, float32 scale).  Bit-exact against the reference's output
    and gradient, through the Python binding and through the compiled pybind11 module."""
    cuda_kernel = _binding(binding, "point_deep_cuda_kernel")
    g = golden("ops_voxel_maxpool")
    feat, ind, output_size, scale_rate = cases.voxel_maxpool_cases()[name]
    pcds_feat, pcds_ind = _t(feat).unsqueeze(-1), _t(ind).unsqueeze(-1)
    voxel_out_shape = [pcds_feat.size(0), pcds_feat.size(1)] + list(output_size)
    voxel_out = torch.zeros(voxel_out_shape, dtype=pcds_feat.dtype, device=pcds_feat.device)
    voxel_max_idx = torch.full([pcds_ind.size(0), pcds_ind.size(1)], -1, dtype=torch.int64, device=pcds_feat.device)
    voxel_out_size_pt = torch.LongTensor(voxel_out_shape).to(pcds_feat.device)
    voxel_out_stride_pt = torch.LongTensor(voxel_out.stride()).to(pcds_feat.device)
    output_size_pt = voxel_out_size_pt[2:]
    scale_rate_pt = torch.FloatTensor(list(scale_rate)).to(pcds_feat.device)
    cuda_kernel.voxel_maxpooling_forward(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, voxel_out_size_pt, voxel_out_stride_pt,
                                         output_size_pt, scale_rate_pt)
    assert np.array_equal(voxel_out.cpu().numpy(), g["vmp_%s_out" % name])
    assert np.array_equal((voxel_max_idx >= 0).cpu().numpy(), ops_np.voxel_cell_index(ind, output_size, scale_rate) >= 0)
    grad_voxel_out = _t(cases.grad_like(tuple(voxel_out.shape), name)).contiguous()
    grad_pcds_feat = torch.zeros(pcds_feat.shape, dtype=grad_voxel_out.dtype, device=grad_voxel_out.device)
    cuda_kernel.voxel_maxpooling_backward(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, grad_pcds_feat, grad_voxel_out,
                                          voxel_out_size_pt, voxel_out_stride_pt, output_size_pt, scale_rate_pt)
    assert np.array_equal(grad_pcds_feat[..., 0].cpu().numpy(), g["vmp_%s_grad" % name])
    with pytest.raises(RuntimeError):                                                   # CHECK_INPUT: CUDA + contiguous
        cuda_kernel.voxel_maxpooling_forward(pcds_feat.cpu(), pcds_ind, voxel_out, voxel_max_idx, voxel_out_size_pt,
                                             voxel_out_stride_pt, output_size_pt, scale_rate_pt)


@pytest.mark.parametrize("binding", ["ctypes", "pybind"])
def test_pybind_name_msda_module_with_the_references_argument_lists(golden, binding):
    """`MultiScaleDeformableAttention.ms_deform_attn_forward / _backward` called as MSDeformAttnFunction calls them
    (deformattn/functions/ms_deform_attn_func.py:21-38): int64 shape / level-start tensors on the device, im2col_step as
    an int; forward against the reference's golden output, backward against autograd of the torch formulation; through
    the Python binding and through the compiled pybind11 module."""
    MSDA = _binding(binding, "MultiScaleDeformableAttention")
    from streammos_amd.refapi.deformattn.functions import ms_deform_attn_core_pytorch
    g = golden("ops_msda")
    value, shapes, lsi, loc, attn = cases.msda_cases()["reftest"]
    tv, ts, ti, tl, ta = _t(value), _t(shapes), _t(lsi), _t(loc), _t(attn)
    out = MSDA.ms_deform_attn_forward(tv, ts, ti, tl, ta, 2)
    np.testing.assert_allclose(out.cpu().numpy(), g["msda_reftest_out32"], rtol=1e-5, atol=1e-7)
    grad_output = torch.randn(out.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    gv, gl, ga = MSDA.ms_deform_attn_backward(tv, ts, ti, tl, ta, grad_output, 2)
    rv, rl, ra = (t.clone().requires_grad_(True) for t in (tv, tl, ta))
    ms_deform_attn_core_pytorch(rv, ts, rl, ra).backward(grad_output.view(out.shape[0], out.shape[1], -1))
    for got, want in ((gv, rv.grad), (gl, rl.grad), (ga, ra.grad)):
        assert (got - want).abs().max().item() <= 1e-4 * max(want.abs().max().item(), 1e-6)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_forward(tv.cpu(), ts, ti, tl, ta, 2)
    with pytest.raises(RuntimeError):                                                   # batch % im2col_step
        MSDA.ms_deform_attn_forward(torch.cat((tv, tv, tv)), ts, ti, torch.cat((tl, tl, tl)), torch.cat((ta, ta, ta)), 2)


_WINO_CASES = [
    # cin, cout, (h, w), mb, act, residual
    (32, 32, (40, 64), 2, 1, False),
    (32, 32, (40, 64), 1, 1, True),
    (64, 32, (24, 96), 2, 1, True),
    (64, 64, (16, 32), 2, 1, False),
    (128, 64, (16, 32), 2, 1, True),
    (128, 128, (16, 64), 2, 1, True),
    (128, 128, (16, 64), 1, 0, False),
    (64, 128, (16, 64), 2, 2, False),
    (128, 64, (16, 64), 2, 2, False),
    (16, 16, (8, 32), 1, 1, False),             # one chunk, one item
    (48, 48, (9, 33), 1, 2, True),              # Cin not a multiple of 32; one row / column past a block
    (32, 32, (13, 45), 2, 2, True),             # odd sizes: half tiles at the right and bottom edges
    (32, 64, (3, 8), 2, 1, False),              # smaller than one block
    (32, 32, (1, 1), 2, 0, False),              # a single pixel
    (32, 32, (70, 130), 2, 1, False),           # several blocks per image in both directions (items > resident blocks: no)
]


@pytest.mark.parametrize("cin,cout,hw,mb,act,with_res", _WINO_CASES)
def test_conv_wino_cl_against_float64(cin, cout, hw, mb, act, with_res):
    """csrc/conv_wino.hip (Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32, host-transformed weights, fused epilogue)
    against conv2d in float64 and against the direct own conv; channel slices of wider buffers as operands.  Same bar as the
    direct kernel (2e-5 of the output range; tools/winograd_numerics.py: 2e-7 .. 6e-7 expected)."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(83)
    b, (h, w) = 3, hw
    wide = torch.randn((b, h, w, cin + 32), generator=gen).to(DEV)
    x = wide[..., 32:].permute(0, 3, 1, 2)
    res_wide = torch.randn((b, h, w, cout + 16), generator=gen).to(DEV)
    res = res_wide[..., :cout].permute(0, 3, 1, 2) if with_res else None
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (cin * 9)) ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    out_wide = torch.full((b, h, w, cout + 8), 7.0, device=DEV)
    out = out_wide[..., 4:4 + cout].permute(0, 3, 1, 2)
    got = ops.conv_wino_cl(x, ops.conv_wino_prepare(wt, mb), bias, act, cout, mb=mb, residual=res, out=out)
    want = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1)
    if with_res:
        want = want + res.double()
    want = F.relu(want) if act == 1 else (F.leaky_relu(want, 0.01) if act == 2 else want)
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    print("wino %d->%d @%s mb %d: %.2e of range" % (cin, cout, hw, mb, err))
    assert err <= 2e-5, err
    assert bool((out_wide[..., :4] == 7.0).all()) and bool((out_wide[..., 4 + cout:] == 7.0).all())   # neighbours untouched
    got2 = ops.conv_wino_cl(x, ops.conv_wino_prepare(wt, mb), None, 0, cout, mb=mb)
    want2 = F.conv2d(x.double(), wt.double(), None, 1, 1)
    assert (got2.double() - want2).abs().max().item() <= 2e-5 * want2.abs().max().item()
    if cin % 32 == 0 and cout % 32 == 0:                                  # and the direct kernel next to it
        direct = ops.conv_cl(x, ops.conv_prepare(wt, 1), None, 0, cout, (3, 3), mt=1)
        assert (got2 - direct).abs().max().item() <= 4e-6 * want2.abs().max().item()


def test_conv_wino_cl_is_deterministic_and_batch_independent():
    gen = torch.Generator(device="cpu").manual_seed(89)
    x = torch.randn((3, 64, 64, 64), generator=gen).to(DEV).permute(0, 3, 1, 2)
    wt = (torch.randn((64, 64, 3, 3), generator=gen) * 0.05).to(DEV)
    wp = ops.conv_wino_prepare(wt, 2)
    a = ops.conv_wino_cl(x, wp, None, 1, 64, mb=2)
    b = ops.conv_wino_cl(x, wp, None, 1, 64, mb=2)
    c = ops.conv_wino_cl(x[1:2], wp, None, 1, 64, mb=2)
    assert torch.equal(a, b) and torch.equal(a[1:2], c)


@pytest.mark.parametrize("cin,cout,hw,kernel,mb,act", [
    (32, 32, (64, 64), (7, 3), 2, 1), (32, 32, (64, 64), (3, 7), 2, 1), (64, 64, (32, 48), (5, 3), 2, 1), (64, 64, (32, 48), (3, 5), 2, 1),
    (32, 32, (37, 45), (7, 3), 2, 2), (32, 32, (37, 45), (3, 7), 1, 0), (16, 48, (21, 70), (5, 3), 1, 1), (48, 16, (70, 21), (3, 5), 1, 2),
    (32, 32, (3, 2), (7, 3), 2, 1), (32, 32, (2, 3), (3, 7), 2, 1), (64, 64, (128, 128), (5, 3), 2, 1), (32, 32, (256, 256), (3, 7), 2, 1)])
def test_conv_wino1d_cl_against_float64(cin, cout, hw, kernel, mb, act):
    """The 1-D Winograd F(2, 3) kernel (csrc/conv_wino1d.hip) on the k x 3 / 3 x k shapes of the Unbalance blocks, ragged
    sizes (items cut by every border, images smaller than a tile) and both orientations, against conv2d in float64; the
    output may be a channel slice of a wider map (as engine._block_cl writes it).  Bar: 2e-5 of the output range, as for
    the direct kernels (observed ~3e-7)."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(77)
    b = 2
    kh, kw = kernel
    x = torch.randn((b, cin) + hw, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, kh, kw), generator=gen) / (kh * kw * cin) ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    want = F.conv2d(x.double(), w.double(), bias.double(), 1, (kh // 2, kw // 2))
    want = want if act == 0 else (torch.relu(want) if act == 1 else F.leaky_relu(want, 0.01))
    both = ops.empty_cl(b, 2 * cout, hw[0], hw[1], x.device)
    both.fill_(7.0)
    got = ops.conv_wino1d_cl(x, ops.conv_wino1d_prepare(w, mb), bias, act, cout, kernel, mb=mb, out=both[:, cout:])
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    print("wino1d %d->%d @%s k%dx%d mb %d: %.2e of range" % (cin, cout, hw, kh, kw, mb, err))
    assert err <= 2e-5
    assert torch.equal(both[:, :cout], torch.full_like(both[:, :cout], 7.0))          # the neighbouring channels are untouched
    got2 = ops.conv_wino1d_cl(x, ops.conv_wino1d_prepare(w, mb), bias, act, cout, kernel, mb=mb)
    assert torch.equal(got2, got)                                                     # run-to-run identical, pitch-independent


@pytest.fixture
def conv_grid_cap():
    """smos_debug_set_conv_grid_cap for the duration of a test: the persistent conv kernels launch at most `blocks` blocks, so a
    block walks several work items on shapes that are still small enough for a float64 reference."""
    from streammos_amd import _lib
    lib = _lib.load()

    def set_cap(blocks):
        _lib.check(lib.smos_debug_set_conv_grid_cap(int(blocks)), "smos_debug_set_conv_grid_cap")
    yield set_cap
    lib.smos_debug_set_conv_grid_cap(0)


_WINO_WALK_CASES = [
    # cin, cout, (h, w), mb, act, residual, batch, grid cap (0 = the device's own resident grid)
    (64, 64, (256, 256), 2, 1, False, 4, 0),        # 2048 items on 512 resident blocks (conv_2-like walk at full grid)
    (64, 64, (40, 96), 2, 2, True, 3, 7),           # 5 x 3 x 2 x 3 = 90 items on 7 blocks: 13 per block, nct = 2, x / y / sample wraps inside a block
    (32, 96, (24, 64), 2, 1, False, 2, 5),          # nct = 3 (odd): blocks start at every cout tile; weight-slice counter wraps mid-block
    (48, 32, (17, 70), 1, 0, True, 2, 4),           # mb = 1 (the early-request schedule), 3 chunks, ragged edges, 54 items on 4 blocks
    (16, 32, (16, 64), 1, 1, False, 3, 3),          # one chunk per item: every k-step group ends an item (mb = 1: epilogue before the region request)
    (16, 32, (16, 64), 2, 1, True, 3, 3),           # the same at mb = 2
]


@pytest.mark.parametrize("cin,cout,hw,mb,act,with_res,b,cap", _WINO_WALK_CASES)
def test_conv_wino_cl_persistent_walk_over_several_items(cin, cout, hw, mb, act, with_res, b, cap, conv_grid_cap):
    """The persistent multi-item walk of csrc/conv_wino.hip (next_item, the region prefetch across an item boundary, the
    epilogue between items, the cyclic weight-slice counter, conv_wino.hip:72-82,178-191,404-447): shapes with MORE work items
    than blocks, with the cout-tile / x / y / sample wraps all inside one block -- against conv2d in float64 (2e-5 of the range)
    and against the direct kernel (4e-6), whose walk is independent of this one.  Layers: networks/backbone.py:136-159,
    networks/multi_view_encoder.py:478-497."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(101)
    h, w = hw
    x = torch.randn((b, h, w, cin), generator=gen).to(DEV).permute(0, 3, 1, 2)
    res = torch.randn((b, h, w, cout), generator=gen).to(DEV).permute(0, 3, 1, 2) if with_res else None
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (cin * 9)) ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    items = b * ((h + 7) // 8) * ((w + 31) // 32) * (cout // (16 * mb))
    conv_grid_cap(cap)
    got = ops.conv_wino_cl(x, ops.conv_wino_prepare(wt, mb), bias, act, cout, mb=mb, residual=res)
    conv_grid_cap(0)
    want = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1)
    if with_res:
        want = want + res.double()
    want = F.relu(want) if act == 1 else (F.leaky_relu(want, 0.01) if act == 2 else want)
    scale = want.abs().max().item()
    err = (got.double() - want).abs().max().item() / scale
    print("wino walk %d->%d @%s mb %d: %d items, cap %d: %.2e of range" % (cin, cout, hw, mb, items, cap, err))
    assert err <= 2e-5, err
    if cin % 32 == 0 and cout % 32 == 0:
        direct = ops.conv_cl(x, ops.conv_prepare(wt, 1), bias, act, cout, (3, 3), mt=1, residual=res)
        assert (got - direct).abs().max().item() <= 4e-6 * scale
    # the uncapped launch of the same layer is the same function: bit-identical (what the engine relies on across devices)
    again = ops.conv_wino_cl(x, ops.conv_wino_prepare(wt, mb), bias, act, cout, mb=mb, residual=res)
    assert torch.equal(again, got)


@pytest.mark.parametrize("cin,cout,hw,kernel,mb,act,b,cap", [
    (64, 64, (256, 256), (5, 3), 2, 1, 3, 0),       # 768 items on 256 resident blocks, nct = 2
    (32, 32, (256, 256), (3, 7), 2, 1, 4, 0),       # 512 items, nct = 1, the x-long orientation
    (32, 64, (50, 70), (7, 3), 2, 2, 2, 5),         # 4 x 3 x 2 x 2 = 48 items on 5 blocks: every wrap inside a block, ragged
    (48, 48, (70, 50), (3, 5), 1, 0, 2, 4),         # mb = 1, nct = 3, 3 chunks
    (16, 32, (33, 64), (5, 3), 2, 1, 2, 3),         # one chunk per item
])
def test_conv_wino1d_cl_persistent_walk_over_several_items(cin, cout, hw, kernel, mb, act, b, cap, conv_grid_cap):
    """The same for csrc/conv_wino1d.hip (one block per CU; 16 L-rows x 32 S-columns x 16 mb couts per item)."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(103)
    kh, kw = kernel
    x = torch.randn((b, cin) + hw, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, kh, kw), generator=gen) / (kh * kw * cin) ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=gen).to(DEV)
    conv_grid_cap(cap)
    got = ops.conv_wino1d_cl(x, ops.conv_wino1d_prepare(w, mb), bias, act, cout, kernel, mb=mb)
    conv_grid_cap(0)
    want = F.conv2d(x.double(), w.double(), bias.double(), 1, (kh // 2, kw // 2))
    want = want if act == 0 else (torch.relu(want) if act == 1 else F.leaky_relu(want, 0.01))
    scale = want.abs().max().item()
    err = (got.double() - want).abs().max().item() / scale
    print("wino1d walk %d->%d @%s k%dx%d mb %d cap %d: %.2e of range" % (cin, cout, hw, kh, kw, mb, cap, err))
    assert err <= 2e-5
    if cin % 32 == 0 and cout % 32 == 0:
        direct = ops.conv_cl(x, ops.conv_prepare(w, 1), bias, act, cout, kernel, mt=1)
        assert (got - direct).abs().max().item() <= 4e-6 * scale
    assert torch.equal(ops.conv_wino1d_cl(x, ops.conv_wino1d_prepare(w, mb), bias, act, cout, kernel, mb=mb), got)


@pytest.mark.parametrize("cin,cout,hw,mb", [(32, 32, (40, 64), 2), (64, 64, (13, 45), 2), (128, 128, (16, 32), 1), (32, 32, (3, 8), 1),
                                            (32, 32, (24, 40), 2)])
def test_conv_wino_cl_channel_sums_and_the_gate_built_on_them(cin, cout, hw, mb):
    """conv_wino_cl(chan_sums=...): channel sums per (8-row x 32-column block, tile row) in a fixed order; the ChannelAtt gate
    (networks/backbone.py:57-73, 87-102) built on them equals float64."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(97)
    b, (h, w) = 3, hw
    x = torch.randn((b, h, w, cin), generator=gen).to(DEV).permute(0, 3, 1, 2)
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (cin * 9)) ** 0.5).to(DEV)
    chunks = ops.conv_wino_sum_chunks(h, w)
    sums = torch.full((b, chunks, cout), 7.0, device=DEV)
    wp = ops.conv_wino_prepare(wt, mb)
    y = ops.conv_wino_cl(x, wp, None, 0, cout, mb=mb, chan_sums=sums)
    assert torch.equal(y, ops.conv_wino_cl(x, wp, None, 0, cout, mb=mb))            # the output itself is untouched
    xb = (w + 31) // 32
    want = torch.zeros((b, chunks, cout), dtype=torch.float64, device=DEV)
    yd = y.double()
    for yy in range(0, h, 2):
        for t in range(xb):
            want[:, ((yy // 8) * xb + t) * 4 + (yy % 8) // 2] = yd[:, :, yy:yy + 2, 32 * t:32 * t + 32].sum((2, 3))
    scale = yd.abs().sum((2, 3)).max().item() / max(h // 2 * xb, 1)
    assert (sums.double() - want).abs().max().item() <= 1e-5 * scale
    sums2 = torch.empty_like(sums)
    ops.conv_wino_cl(x, wp, None, 0, cout, mb=mb, chan_sums=sums2)
    assert torch.equal(sums, sums2)                                                  # fixed summation order
    cr = max(cout // 4, 8)
    bias, b1, b2 = (torch.randn(n, generator=gen).to(DEV) for n in (cout, cr, cout))
    w1 = (torch.randn((cr, cout), generator=gen) * 0.3).to(DEV)
    w2 = (torch.randn((cout, cr), generator=gen) * 0.3).to(DEV)
    xres = torch.randn((b, h, w, cout), generator=gen).to(DEV).permute(0, 3, 1, 2)
    got = ops.channel_gate_apply_cl(y, bias, w1, b1, w2, b2, xres, sums, torch.empty(b * cout, device=DEV))
    z = yd + bias.double()[None, :, None, None]
    g = torch.sigmoid(F.linear(F.relu(F.linear(z.mean((2, 3)), w1.double(), b1.double())), w2.double(), b2.double()))
    ref = F.relu(z * g[:, :, None, None] + xres.double())
    assert (got.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("c,hw,gated,sliced", [(32, (24, 40), False, False), (64, (13, 45), True, False), (128, (16, 32), True, True),
                                               (16, (9, 33), False, True)])
def test_basic_block_cl_equals_the_separate_launches(c, hw, gated, sliced):
    """smos_basic_block_cl (csrc/blocks.hip) = conv_wino_cl, conv_wino_cl [, channel_gate_apply_cl] behind one foreign call:
    bit-identical to the separate calls, also with x / out as channel slices of wider maps; aliasing y with out is refused."""
    gen = torch.Generator(device="cpu").manual_seed(211)
    b, (h, w) = 2, hw
    wide = torch.randn((b, h, w, 2 * c), generator=gen).to(DEV).permute(0, 3, 1, 2)
    x = wide[:, c:] if sliced else wide[:, :c].contiguous(memory_format=torch.channels_last)
    w1, w2 = ((torch.randn((c, c, 3, 3), generator=gen) * (2.0 / (c * 9)) ** 0.5).to(DEV) for _ in range(2))
    b1, b2 = (torch.randn(c, generator=gen).to(DEV) * 0.2 for _ in range(2))
    gate = None
    if gated:
        cr = c // 4
        gate = ((torch.randn((cr, c), generator=gen) * 0.3).to(DEV), torch.randn(cr, generator=gen).to(DEV),
                (torch.randn((c, cr), generator=gen) * 0.3).to(DEV), torch.randn(c, generator=gen).to(DEV))
    assert ops.basic_block_ok(c, gated)
    plan = ops.BasicBlockPlan(w1, b1, w2, b2, gate)
    mb = plan.mb
    u1, u2 = ops.conv_wino_prepare(w1, mb), ops.conv_wino_prepare(w2, mb)
    y = ops.conv_wino_cl(x, u1, b1, ops.ACT_RELU, c, mb=mb)
    if gated:
        chunks = ops.conv_wino_sum_chunks(h, w)
        sums = torch.empty((b, chunks, c), device=DEV)
        t = ops.conv_wino_cl(y, u2, None, ops.ACT_NONE, c, mb=mb, chan_sums=sums)
        want = ops.channel_gate_apply_cl(t, b2, *gate, x, sums, torch.empty(b * c, device=DEV))
    else:
        want = ops.conv_wino_cl(y, u2, b2, ops.ACT_RELU, c, mb=mb, residual=x)
    ws = torch.empty(b * c * (ops.conv_wino_sum_chunks(h, w) + 1), device=DEV) if gated else None
    dst = torch.full((b, h, w, 3 * c), 5.0, device=DEV).permute(0, 3, 1, 2)
    out = dst[:, c:2 * c] if sliced else None
    got = ops.basic_block_cl(x, plan, out=out, ws=ws)
    assert torch.equal(got, want)
    if sliced:
        assert float((dst[:, :c] - 5.0).abs().max()) == 0.0 and float((dst[:, 2 * c:] - 5.0).abs().max()) == 0.0
    scratch = ops.empty_cl(b, c, h, w, DEV)
    with pytest.raises(RuntimeError):
        ops.basic_block_cl(x, plan, y=scratch, out=scratch, ws=ws)
    if gated:
        with pytest.raises(RuntimeError):
            ops.basic_block_cl(x, plan)                      # no scratch for the channel sums


@pytest.mark.parametrize("c,hw,k", [(32, (24, 40), 7), (64, (13, 45), 5)])
def test_unbalance_block_cl_equals_the_separate_launches(c, hw, k):
    """smos_unbalance_block_cl = conv_wino1d_cl (k x 3), conv_wino1d_cl (3 x k), conv_wino_cl (2C -> C, + x) behind one foreign
    call: bit-identical to the separate calls."""
    gen = torch.Generator(device="cpu").manual_seed(223)
    b, (h, w) = 2, hw
    x = torch.randn((b, h, w, c), generator=gen).to(DEV).permute(0, 3, 1, 2)
    wa = (torch.randn((c, c, k, 3), generator=gen) * (2.0 / (c * 3 * k)) ** 0.5).to(DEV)
    wb = (torch.randn((c, c, 3, k), generator=gen) * (2.0 / (c * 3 * k)) ** 0.5).to(DEV)
    wc = (torch.randn((c, 2 * c, 3, 3), generator=gen) * (1.0 / (c * 9)) ** 0.5).to(DEV)
    ba, bb, bc = (torch.randn(c, generator=gen).to(DEV) * 0.2 for _ in range(3))
    plan = ops.UnbalanceBlockPlan(wa, ba, wb, bb, wc, bc)
    mb = plan.mb
    both = ops.empty_cl(b, 2 * c, h, w, DEV)
    ops.conv_wino1d_cl(x, ops.conv_wino1d_prepare(wa, mb), ba, ops.ACT_RELU, c, (k, 3), mb=mb, out=both[:, :c])
    ops.conv_wino1d_cl(x, ops.conv_wino1d_prepare(wb, mb), bb, ops.ACT_RELU, c, (3, k), mb=mb, out=both[:, c:])
    want = ops.conv_wino_cl(both, ops.conv_wino_prepare(wc, mb), bc, ops.ACT_RELU, c, mb=mb, residual=x)
    got = ops.unbalance_block_cl(x, plan)
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        ops.unbalance_block_cl(x, plan, both=ops.empty_cl(b, c, h, w, DEV))      # `both` too narrow


@pytest.mark.parametrize("c,hw,b,n_blocks,gated", [(128, (64, 64), 4, 5, True), (64, (16, 64), 2, 2, False), (32, (8, 32), 1, 3, True)])
def test_conv_wino_chain_cl_equals_the_separate_launches(c, hw, b, n_blocks, gated):
    """EXPERIMENTAL smos_conv_wino_chain_cl: the 2 k convolutions of k BasicBlocks in one launch with per-region dataflow waits,
    against the same convolutions launch by launch (conv_wino_cl): bit for bit, on three launches in a row that share the
    workspace (its counters are monotonic), with the last block's conv leaving its channel sums; the workspace never reports
    a wait that gave up.  (128 ch @64^2 x 4 = the third BEV stage: 256 work items, one per CU.)"""
    gen = torch.Generator(device="cpu").manual_seed(307)
    h, w = hw
    ws = ops.WinoChainWorkspace(2 * n_blocks, b, h, w, DEV)
    wts = [((torch.randn((c, c, 3, 3), generator=gen) * (1.0 / (c * 9)) ** 0.5).to(DEV), (torch.randn(c, generator=gen) * 0.1).to(DEV))
           for _ in range(2 * n_blocks)]
    preps = [ops.conv_wino_prepare(wt, 2) for wt, _ in wts]
    chunks = ops.conv_wino_sum_chunks(h, w)
    for trial in range(3):
        x = torch.randn((b, h, w, c), generator=gen).to(DEV).permute(0, 3, 1, 2)
        # launch by launch
        cur, want_sums = x, None
        for k in range(n_blocks):
            y = ops.conv_wino_cl(cur, preps[2 * k], wts[2 * k][1], ops.ACT_RELU, c, mb=2)
            if gated and k == n_blocks - 1:
                want_sums = torch.empty((b, chunks, c), device=DEV)
                cur = ops.conv_wino_cl(y, preps[2 * k + 1], None, ops.ACT_NONE, c, mb=2, chan_sums=want_sums)
            else:
                cur = ops.conv_wino_cl(y, preps[2 * k + 1], wts[2 * k + 1][1], ops.ACT_RELU, c, mb=2, residual=cur)
        want = cur
        # one launch
        layers, sums = [], None
        for k in range(n_blocks):
            layers.append((preps[2 * k], wts[2 * k][1], -1, ops.empty_cl(b, c, h, w, DEV), ops.ACT_RELU))
            if gated and k == n_blocks - 1:
                sums = torch.full((b, chunks, c), 3.0, device=DEV)
                layers.append((preps[2 * k + 1], None, -1, ops.empty_cl(b, c, h, w, DEV), ops.ACT_NONE))
            else:
                layers.append((preps[2 * k + 1], wts[2 * k + 1][1], 2 * k, ops.empty_cl(b, c, h, w, DEV), ops.ACT_RELU))
        got = ops.conv_wino_chain_cl(x, layers, ws, chan_sums=sums)
        assert torch.equal(got, want), "trial %d" % trial
        if gated:
            assert torch.equal(sums, want_sums)
    assert ws.launch_no == 3 and not ws.gave_up()
    with pytest.raises(RuntimeError):                                   # two layers may not share an output map
        shared = ops.empty_cl(b, c, h, w, DEV)
        ops.conv_wino_chain_cl(x, [(preps[0], None, -1, shared, 0), (preps[1], None, -1, shared, 0)] + layers[2:], ws)


def _tf_layer_weights(gen, ffn, nq):
    def lin(o, i):
        return ((torch.randn((o, i), generator=gen) / i ** 0.5).to(DEV), (torch.randn(o, generator=gen) * 0.3).to(DEV))

    def norm():
        return ((1.0 + 0.3 * torch.randn(128, generator=gen)).to(DEV), (0.2 * torch.randn(128, generator=gen)).to(DEV), 1e-5)
    return {"out": lin(128, 128), "norm1": norm(), "lin1": lin(ffn, 128), "lin2": lin(128, ffn), "norm2": norm(),
            "next": lin(nq, 128) if nq else None}


@pytest.mark.parametrize("tokens,ffn,nq,pitched", [(4 * 64 * 64, 512, 48, False), (777, 512, 48, True), (64, 512, 0, False),
                                                    (1000, 64, 16, False), (5, 1024, 64, True)])
def test_tfusion_layer_against_float64(tokens, ffn, nq, pitched):
    """csrc/tfusion.hip::tfusion_layer -- output_proj, + query, LayerNorm, linear1, ReLU, linear2, +, LayerNorm and the next
    layer's offset / logit projection in one launch (networks/multi_view_encoder.py:314-320,
    deformattn/modules/ms_deform_attn.py:94-115) -- against the same chain in float64: ragged token counts (the last block and
    the last wave partly empty), a pitched query (a channel slice of a wider map), with and without the next projection."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(131)
    w = _tf_layer_weights(gen, ffn, nq)
    sampled = torch.randn((tokens, 128), generator=gen).to(DEV)
    wide = torch.randn((tokens, 160), generator=gen).to(DEV)
    query = wide[:, 16:144] if pitched else wide[:, 16:144].contiguous()
    prep = ops.TfusionLayer(w["out"], w["norm1"], w["lin1"], w["lin2"], w["norm2"], next_qproj=w["next"])
    out, nxt = ops.tfusion_layer(sampled, query, prep)
    d = lambda t: t.double()
    q1 = F.layer_norm(d(query) + F.linear(d(sampled), d(w["out"][0]), d(w["out"][1])), (128,), d(w["norm1"][0]), d(w["norm1"][1]), 1e-5)
    ffn_out = F.linear(F.relu(F.linear(q1, d(w["lin1"][0]), d(w["lin1"][1]))), d(w["lin2"][0]), d(w["lin2"][1]))
    want = F.layer_norm(q1 + ffn_out, (128,), d(w["norm2"][0]), d(w["norm2"][1]), 1e-5)
    err = (out.double() - want).abs().max().item() / want.abs().max().item()
    print("tfusion_layer %d tokens, ffn %d: %.2e of range" % (tokens, ffn, err))
    assert out.shape == (tokens, 128) and err <= 2e-6, err
    if nq:
        want_q = F.linear(want, d(w["next"][0]), d(w["next"][1]))
        errq = (nxt.double() - want_q).abs().max().item() / want_q.abs().max().item()
        assert nxt.shape == (tokens, nq) and errq <= 2e-6, errq
    else:
        assert nxt is None
    out2, nxt2 = ops.tfusion_layer(sampled, query, prep)
    assert torch.equal(out, out2) and (nxt is None or torch.equal(nxt, nxt2))          # fixed order: run-to-run identical
    # against the unfused chain the engine used before (library GEMMs + add_layer_norm): same function in fp32
    q1f = ops.add_layer_norm(query.contiguous(), F.linear(sampled, *w["out"]), w["norm1"][0], w["norm1"][1], 1e-5)
    wantf = ops.add_layer_norm(q1f, F.linear(F.relu(F.linear(q1f, *w["lin1"])), *w["lin2"]), w["norm2"][0], w["norm2"][1], 1e-5)
    assert (out - wantf).abs().max().item() <= 4e-6 * want.abs().max().item()


@pytest.mark.parametrize("tokens", [4 * 64 * 64, 1000, 3])
def test_tfusion_project_against_float64(tokens):
    """csrc/tfusion.hip::tfusion_project: the value projections of two layers (one shared input) and an offset / logit
    projection of 48 channels from another, pitched input as three jobs of one launch, against F.linear in float64."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(137)
    src = torch.randn((tokens, 128), generator=gen).to(DEV)
    wide = torch.randn((tokens, 192), generator=gen).to(DEV)
    query = wide[:, 32:160]
    ws = [((torch.randn((o, 128), generator=gen) / 128 ** 0.5).to(DEV), torch.randn(o, generator=gen).to(DEV)) for o in (128, 128, 48, 256)]
    outs = ops.tfusion_project([(src, ops.tfusion_pack_linear(ws[0][0]), ws[0][1]), (src, ops.tfusion_pack_linear(ws[1][0]), ws[1][1]),
                                (query, ops.tfusion_pack_linear(ws[2][0]), ws[2][1]), (query, ops.tfusion_pack_linear(ws[3][0]), ws[3][1])])
    for got, x, (w, b) in zip(outs, (src, src, query, query), ws):
        want = F.linear(x.double(), w.double(), b.double())
        assert got.shape == want.shape
        err = (got.double() - want).abs().max().item() / want.abs().max().item()
        assert err <= 2e-6, err
    with pytest.raises(RuntimeError):
        ops.tfusion_project([(src.cpu(), ops.tfusion_pack_linear(ws[0][0]), ws[0][1])])
    # jobs of different length, 1152 outputs without a bias: the decoder's tap products (csrc/upconv.hip) as jobs of one launch
    short = torch.randn((max(tokens // 4, 1), 128), generator=gen).to(DEV)
    wt = (torch.randn((1152, 128), generator=gen) / 128 ** 0.5).to(DEV)
    a, b = ops.tfusion_project([(src, ops.tfusion_pack_linear(wt), 1152), (short, ops.tfusion_pack_linear(wt), 1152)])
    for got, x in ((a, src), (b, short)):
        want = F.linear(x.double(), wt.double())
        assert got.shape == want.shape and (got.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()


@pytest.mark.parametrize("c,hw,stride,b", [(64, (256, 256), 2, 4), (128, (128, 128), 2, 4), (32, (32, 1024), 1, 4), (64, (16, 512), 1, 4),
                                           (64, (41, 71), 2, 2), (128, (9, 37), 2, 3), (32, (13, 45), 1, 2), (64, (5, 3), 1, 1),
                                           (32, (6, 100), 2, 2), (64, (1, 1), 2, 1)])
def test_downsample_pool_branch_against_float64_and_the_two_launch_form(c, hw, stride, b):
    """csrc/downsample.hip: relu(a + bias + maxpool3x3(conv1x1(x); stride, pad 1)) -- the DownSample2D tail with its pool branch
    computed on the fly (networks/backbone.py:105-134) -- at the network's four shapes and on ragged ones (tiles cut by every
    border, maps smaller than a tile, all-negative windows at the border: a tap outside the image must not count as 0), inputs
    and output as channel slices, output written over `a` as the engine does; against float64 and against the two launches it
    replaces (smos_conv_cl 1x1 + smos_downsample_epilogue_cl)."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(149)
    h, w = hw
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    xw = (torch.randn((b, h, w, c + 32), generator=gen) - 0.5).to(DEV)          # mostly negative pool inputs near the borders too
    x = xw[..., 16:16 + c].permute(0, 3, 1, 2)
    wt = (torch.randn((c, c, 1, 1), generator=gen) / c ** 0.5).to(DEV)
    aw = torch.randn((b, ho, wo, c + 8), generator=gen).to(DEV)
    a = aw[..., :c].permute(0, 3, 1, 2)
    bias = (torch.randn(c, generator=gen) - 1.0).to(DEV)
    want = F.relu(a.double() + bias.double()[None, :, None, None] + F.max_pool2d(F.conv2d(x.double(), wt.double()), 3, stride, 1))
    got = ops.downsample_pool_branch(x, ops.pool_branch_prepare(wt), a, bias, stride)
    err = (got.double() - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
    print("pool branch %d ch @%s /%d: %.2e of range" % (c, hw, stride, err))
    assert got.shape == want.shape and err <= 2e-6, err
    if c % 32 == 0:
        qf = ops.conv_cl(x, ops.conv_prepare(wt, 1), None, 0, c, (1, 1), mt=1)
        two = ops.downsample_epilogue_cl(a.clone(memory_format=torch.preserve_format), qf, bias, stride)
        assert (got - two).abs().max().item() <= 4e-6 * max(want.abs().max().item(), 1e-6)
    # in place over `a` (engine._block_cl), neighbours of the slice untouched
    keep = aw[..., c:].clone()
    inplace = ops.downsample_pool_branch(x, ops.pool_branch_prepare(wt), a, bias, stride, out=a)
    assert torch.equal(inplace, got) and torch.equal(aw[..., c:], keep)


@pytest.mark.parametrize("n_live", [0, 1, 31, 32, 1000, 2047, 2048, 5000])
def test_point_head_and_gathers_leave_the_padding_tail_out(n_live):
    """The `_live` entry points (smos_point_head_live, smos_gather_scatter_cl_live; include/smos.h): given a DEVICE-side count of
    the real points at the front of every sample they produce the same values for those points, bit for bit, and zeros / nothing
    for the tail [n_live, N) -- for counts of 0, inside a tile, at tile borders, N and beyond N (clamped)."""
    gen = torch.Generator(device="cpu").manual_seed(151)
    b, n = 3, 2048
    rows = torch.randn((b, n, 192), generator=gen).to(DEV)
    l1 = ((torch.randn((96, 192, 1, 1), generator=gen) * 0.1).to(DEV), (torch.randn(96, generator=gen) * 0.2).to(DEV))
    l2 = ((torch.randn((64, 96, 1, 1), generator=gen) * 0.15).to(DEV), (torch.randn(64, generator=gen) * 0.2).to(DEV))
    l3 = ((torch.randn((3, 64, 1, 1), generator=gen) * 0.2).to(DEV), torch.randn(3, generator=gen).to(DEV))
    wprep, m3 = ops.point_head_prepare(l1, l2, l3)
    full = ops.point_head(rows, wprep, m3)
    cnt = torch.tensor([n_live], dtype=torch.int32, device=DEV)
    live = ops.point_head(rows, wprep, m3, n_live=cnt)
    k = min(n_live, n)
    assert torch.equal(live[:, :, :k], full[:, :, :k]) and float(live[:, :, k:].abs().sum()) == 0.0
    # the gather of point rows: real points identical; tail rows that lie outside the source map stay untouched
    c, hw = 64, (32, 64)
    grid = _to_cl(torch.relu(torch.randn((b, c) + hw, generator=gen)).to(DEV))
    coord = _model_like_coords(gen, b, n, hw[0] / 0.5, hw[1] / 0.5).to(DEV)
    coord[:, k:] = -4864.0                                        # what the reference's padding looks like after quantisation
    want = torch.full((b, n, c), 3.0, device=DEV)
    ops.gather_scatter_cl(grid, coord, (0.5, 0.5), pts_out=want)
    got = torch.full((b, n, c), 3.0, device=DEV)
    ops.gather_scatter_cl(grid, coord, (0.5, 0.5), pts_out=got, n_live=cnt)
    assert torch.equal(got[:, :k], want[:, :k])
    tail = got[:, (k + 63) // 64 * 64:]                           # whole 64-point runs inside the tail are skipped
    assert tail.numel() == 0 or bool((tail == 3.0).all())
