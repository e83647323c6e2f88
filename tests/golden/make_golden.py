"""Generate the golden vectors in tests/golden/*.npz by running the REAL reference on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The fixtures are data: seeded inputs (or their SHA-256 when they are regenerated deterministically by
``streammos_amd.synth`` / ``streammos_amd.preprocess``) and the reference's outputs.  No reference
source or bytecode is stored.  How the reference is made importable on CPU: oracle/ref_import.py.
"""
import hashlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from tests import cases  # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def gen_voxel_maxpool(ns, out):
    for name, (feat, ind, out_size, scale) in cases.voxel_maxpool_cases().items():
        f = torch.from_numpy(feat).unsqueeze(-1).requires_grad_(True)
        i = torch.from_numpy(ind).unsqueeze(-1)
        y = ns.deep_point.VoxelMaxPool(f, i, tuple(out_size), tuple(scale))
        g = torch.from_numpy(cases.grad_like(y.shape, name))
        y.backward(g)
        out["vmp_%s_in_sha" % name] = np.array(sha(feat, ind))
        out["vmp_%s_out" % name] = y.detach().numpy()
        out["vmp_%s_grad" % name] = f.grad[..., 0].numpy()


def gen_bilinear(ns, out):
    for name, (grid, coord, scale) in cases.bilinear_cases().items():
        mod = ns.backbone.BilinearSample(in_dim=grid.shape[1], scale_rate=scale)
        y = mod(torch.from_numpy(grid), torch.from_numpy(coord).unsqueeze(-1))
        out["bil_%s_in_sha" % name] = np.array(sha(grid, coord))
        out["bil_%s_out" % name] = y[..., 0].numpy()


def gen_msda(ns, out):
    core = ns.msda_func.ms_deform_attn_core_pytorch
    for name, (value, shapes, lsi, loc, attn) in cases.msda_cases().items():
        y64 = core(torch.from_numpy(value).double(), torch.from_numpy(shapes),
                   torch.from_numpy(loc).double(), torch.from_numpy(attn).double())
        y32 = core(torch.from_numpy(value), torch.from_numpy(shapes),
                   torch.from_numpy(loc), torch.from_numpy(attn))
        out["msda_%s_in_sha" % name] = np.array(sha(value, shapes, lsi, loc, attn))
        out["msda_%s_out64" % name] = y64.numpy()
        out["msda_%s_out32" % name] = y32.numpy()
    # module-level: MSDeformAttn.forward with seeded weights (deformattn/modules/ms_deform_attn.py:78-116)
    from streammos_amd import synth
    mod = ns.msda_modules.MSDeformAttn(d_model=128, n_levels=1, n_heads=4, n_points=4).eval()
    sd = synth.seeded_state_dict({("cross_attn." + k): v for k, v in mod.state_dict().items()})
    mod.load_state_dict({k[len("cross_attn."):]: v for k, v in sd.items()})
    q, ref, src, shapes, lsi = cases.msda_module_case()
    with torch.no_grad():
        y = mod(torch.from_numpy(q), torch.from_numpy(ref), torch.from_numpy(src),
                torch.from_numpy(shapes), torch.from_numpy(lsi))
    out["msda_module_in_sha"] = np.array(sha(q, ref, src))
    out["msda_module_out"] = y.numpy()


def gen_voting(ns, out):
    crop = ns.transforms.Crop(dims=(0, 1, 2), fov=[[-50, -50, -4], [50, 50, 2]])
    v = ns.voting
    size = (512, 512, 30)
    for name, (cur, cur_pred, hist, hist_pred) in cases.voting_cases().items():
        h_pts, h_lab, h_mask = crop(torch.tensor(hist), torch.tensor(hist_pred.astype("uint8")))
        c_pts, c_lab, c_mask = crop(torch.tensor(cur), torch.tensor(cur_pred.astype("uint8")))
        n_hist = len(h_pts)
        pts = torch.cat((h_pts, c_pts), 0)
        lab = torch.cat((h_lab, c_lab), 0)
        quan = v.Quantize(pts, range_x=(-50.0, 50.0), range_y=(-50.0, 50.0), range_z=(-4.0, 2.0), size=size)
        vox = v.determine_voxel_labels(quan.to(torch.int64), lab.to(torch.int64), size)
        new = v.get_point_labels_from_voxel_labels(quan[n_hist:].to(torch.int64), vox, size)
        refined = cur_pred.copy()
        refined[c_mask.numpy()] = new.numpy()
        nz = torch.nonzero(vox.reshape(-1)).reshape(-1)
        out["vote_%s_in_sha" % name] = np.array(sha(cur, cur_pred, hist, hist_pred))
        out["vote_%s_cur_mask" % name] = c_mask.numpy()
        out["vote_%s_hist_mask" % name] = h_mask.numpy()
        out["vote_%s_coords" % name] = quan.to(torch.int64).numpy().astype(np.int32)
        out["vote_%s_voxel_nz_idx" % name] = nz.numpy()
        out["vote_%s_voxel_nz_val" % name] = vox.reshape(-1)[nz].numpy().astype(np.int8)
        out["vote_%s_refined" % name] = refined.astype(np.int8)
        lut = v.map(refined.astype(np.int64), {0: 0, 1: 9, 2: 251})
        out["vote_%s_lut" % name] = lut.astype(np.int32)


def gen_preprocess(out):
    """datasets/utils.py is loaded by path (the package __init__ drags in the whole dataset module);
    make_point_feat (datasets/data_StreamMOS.py:25-50) is pulled out of the source text."""
    import ast
    import types
    spec = importlib.util.spec_from_file_location(
        "smos_ref_dataset_utils", os.path.join(ref_import.REF_ROOT, "datasets", "utils.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    path = os.path.join(ref_import.REF_ROOT, "datasets", "data_StreamMOS.py")
    tree = ast.parse(open(path).read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "make_point_feat"]
    mod = types.ModuleType("smos_ref_make_point_feat")
    mod.__dict__.update(np=np)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), mod.__dict__)

    class Voxel:
        RV_theta = (-25.0, 3.0)
        range_x = (-50.0, 50.0)
        range_y = (-50.0, 50.0)
        range_z = (-4.0, 2.0)
        bev_shape = (512, 512, 30)
        rv_shape = (64, 2048)

    scan, pose_diff = cases.preprocess_case()
    moved = du.Trans(scan, pose_diff)
    mask = du.filter_pcds_mask(moved, range_x=Voxel.range_x, range_y=Voxel.range_y, range_z=Voxel.range_z)
    kept = moved[mask]
    coord = du.Quantize(kept, range_x=Voxel.range_x, range_y=Voxel.range_y, range_z=Voxel.range_z,
                        size=Voxel.bev_shape)
    sph = du.SphereQuantize(kept, phi_range=(-180.0, 180.0), theta_range=Voxel.RV_theta, size=Voxel.rv_shape)
    feat = mod.make_point_feat(kept, coord, sph, Voxel)
    out["pre_in_sha"] = np.array(sha(scan, pose_diff))
    out["pre_moved"] = moved
    out["pre_mask"] = mask
    out["pre_coord"] = coord.astype(np.float32)
    out["pre_sphere"] = sph.astype(np.float32)
    out["pre_feat"] = feat.astype(np.float32)
    out["pre_dtypes"] = np.array([str(coord.dtype), str(sph.dtype), str(feat.dtype)])


def gen_e2e(ns, out):
    from streammos_amd import synth
    model = ns.StreamMOS.AttNet(ns.config.get_config()[2]).eval()
    model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    memory = None
    for i, batch in enumerate(cases.e2e_frames()):
        tb = {k: torch.from_numpy(v).unsqueeze(0) for k, v in batch.items()}
        with torch.no_grad():
            pred, a0, a1, a2, memory = model.infer(tb, i, memory)
        out["e2e_f%d_in_sha" % i] = np.array(sha(batch["pcds_xyzi"], batch["pcds_coord"], batch["pcds_sphere_coord"]))
        out["e2e_f%d_pred" % i] = pred.numpy()
        out["e2e_f%d_mem_sub" % i] = memory[:, ::8, ::4, ::4].contiguous().numpy()
        out["e2e_f%d_mem_stats" % i] = np.array([memory.double().sum().item(), memory.double().abs().sum().item()])
        out["e2e_f%d_aux_sub" % i] = torch.stack((a0, a1, a2))[:, :, :, ::8, ::8].contiguous().numpy()


def gen_seg(ns, out):
    """Stage-2 model (models/StreamMOS_seg.py): state-dict layout and two chained frames incl. the refine head."""
    import contextlib
    import io
    import json
    import models.StreamMOS_seg as seg
    import config.StreamMOS_seg as seg_cfg
    from streammos_amd import synth
    model = seg.AttNet(seg_cfg.get_config()[2]).eval()
    layout_path = os.path.join(HERE, "state_dict_layout.json")
    layout = json.load(open(layout_path))
    layout["stage2_seg"] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]
    json.dump(layout, open(layout_path, "w"), indent=0)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    memory = None
    for i, batch in enumerate(cases.e2e_frames(2)):
        tb = {k: torch.from_numpy(v).unsqueeze(0) for k, v in batch.items()}
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):     # the reference prints the input shape
            pred, bf, a0, a1, a2, memory = model.infer(tb, i, memory)
        out["seg_f%d_in_sha" % i] = np.array(sha(batch["pcds_xyzi"], batch["pcds_coord"], batch["pcds_sphere_coord"]))
        out["seg_f%d_pred" % i] = pred.numpy()
        out["seg_f%d_bf_pred" % i] = bf.numpy()


def gen_losses(ns, out):
    """Training losses (row f2): the reference's CE_OHEM (utils/criterion.py:10-28) and lovasz_softmax
    (utils/lovasz_losses.py:147-222) on seeded logits, with the gradient of ce + 3 * lovasz."""
    import utils.criterion as rc
    import utils.lovasz_losses as rl
    g = torch.Generator().manual_seed(0)
    pred = torch.randn(2, 3, 500, 1, generator=g, requires_grad=True)
    gt = torch.randint(0, 3, (2, 500, 1), generator=g)
    a = rc.CE_OHEM(top_ratio=0.2, top_weight=4.0, ignore_index=0)(pred, gt)
    b = rl.lovasz_softmax(pred, gt, ignore=0)
    out["pred"], out["gt"] = pred.detach().numpy(), gt.numpy()
    out["ohem"], out["lovasz"] = np.array(a.item()), np.array(b.item())
    out["grad"] = torch.autograd.grad(a + 3 * b, pred)[0].numpy()


def gen_instance(ns, out):
    """Row f3: the reference's own post_processing() (voxel_instance_voting.py:195-270: voxel vote, then cluster() = DBSCAN +
    hull box + in-box majority) run on a synthetic sequence written to a temporary SemanticKITTI-style tree; the fixture
    holds the refined label words it writes for every frame.  The script's module level is replaced by the globals it
    would have defined; `.cuda()` is the identity here (no GPU in the build container)."""
    import tempfile
    import yaml
    spec = importlib.util.spec_from_file_location(
        "smos_ref_dataset_utils", os.path.join(ref_import.REF_ROOT, "datasets", "utils.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    task_cfg = yaml.safe_load(open(os.path.join(ref_import.REF_ROOT, "datasets", "semantic-kitti.yaml")))
    frames = cases.instance_sequence()
    lut = np.zeros(256, dtype=np.uint32)
    lut[1], lut[2] = 9, 251
    with tempfile.TemporaryDirectory() as tmp:
        dirs = {k: os.path.join(tmp, k) + "/" for k in ("velodyne", "pred", "pred_bf", "refined")}
        for d in dirs.values():
            os.makedirs(d)
        files = []
        for k, (scan, pred, bf, pose) in enumerate(frames):
            name = "%06d.bin" % k
            files.append(name)
            scan.tofile(dirs["velodyne"] + name)
            lut[pred].tofile(dirs["pred"] + name[:-4] + ".label")                      # val_StreamMOS_seg.py:139-140: LUT words
            bf.astype(np.uint32).tofile(dirs["pred_bf"] + name[:-4] + ".label")         # :141: raw 0 / 1 / 2
        mod = ref_import.extract_instance_voting(dict(
            utils=du, task_cfg=task_cfg, files=files, poses_list=[f[3] for f in frames], frames_num_max=8,
            data_path=dirs["velodyne"], pred_path=dirs["pred"], pred_bf_path=dirs["pred_bf"], save_path=dirs["refined"],
            crop_to_fov=ns.transforms.Crop(dims=(0, 1, 2), fov=[[-50, -50, -4], [50, 50, 2]])))
        saved = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            for k in range(len(frames)):
                mod.post_processing(k)
        finally:
            torch.Tensor.cuda = saved
        for k, (scan, pred, bf, pose) in enumerate(frames):
            out["inst_f%d_in_sha" % k] = np.array(sha(scan, pred, bf, pose))
            out["inst_f%d_refined" % k] = np.fromfile(dirs["refined"] + "%06d.label" % k, dtype=np.int32)
    # what the fixture exercises (checked here so that a change of the generator cannot silently lose a case)
    from oracle import ops_np
    from streammos_amd import preprocess
    stats = ops_np.instance_cluster_stats(frames[9][0], frames[9][1], frames[9][2])
    sizes = sorted(s["points"] for s in stats)
    assert 30 in sizes and 31 in sizes, sizes
    del preprocess


def _grad_summary(out, prefix, model, keys):
    named = dict(model.named_parameters())
    for k in keys:
        g = named[k].grad
        flat = g.detach().reshape(-1)
        out["%s_grad_norm_%s" % (prefix, k)] = np.array(flat.double().norm().item())
        out["%s_grad_head_%s" % (prefix, k)] = flat[:: max(1, flat.numel() // 512)][:512].numpy().copy()


def gen_training(ns, out):
    """Row f2: loss and gradients of one training step of the reference on CPU.
    stage 1: AttNet.forward (models/StreamMOS.py:155-179; three chained samples, OHEM-CE + 3 x Lovasz on the points and the
             three BEV heads), (a) in eval mode (BatchNorm running statistics, no dropout: fully deterministic), (b) in train
             mode (BatchNorm batch statistics + running-statistics update) with the dropout masks removed
             (torch.nn.functional.dropout patched to the identity for the call: the masks depend on the device's generator).
    stage 2: StreamMOS_seg.AttNet.forward with everything but `refine.*` frozen as train_StreamMOS_seg.py:165-174 does;
             loss only on bf_pred_cls (models/StreamMOS_seg.py:164-171)."""
    import contextlib
    import io
    import torch.nn.functional as F
    import models.StreamMOS_seg as seg
    import config.StreamMOS_seg as seg_cfg
    from streammos_amd import synth
    batch = {k: torch.from_numpy(v) for k, v in cases.training_batch().items()}
    out["train_in_sha"] = np.array(sha(*[batch[k].numpy() for k in sorted(batch)]))

    def identity_dropout(x, p=0.5, training=True, inplace=False):
        return x

    for mode in ("eval", "train"):
        model = ns.StreamMOS.AttNet(ns.config.get_config()[2])
        model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
        model.train(mode == "train")
        saved = F.dropout
        F.dropout = identity_dropout if mode == "train" else saved
        try:
            loss = model(batch)
            loss.backward()
        finally:
            F.dropout = saved
        out["stage1_%s_loss" % mode] = np.array(loss.item())
        _grad_summary(out, "stage1_%s" % mode, model, cases.TRAINING_GRAD_KEYS)
        if mode == "train":        # BatchNorm running statistics after the three chained forwards
            sd = model.state_dict()
            for k in ("point_pre.layer.0.layer.2.running_mean", "bev_net.res2.1.layer.1.running_var", "bev_net.conv_1.bn.running_mean"):
                out["stage1_train_bn_%s" % k] = sd[k].numpy().copy()

    model = seg.AttNet(seg_cfg.get_config()[2])
    model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    for p_ in model.parameters():                        # train_StreamMOS_seg.py:165-174
        p_.requires_grad = False
    for p_ in model.refine.parameters():
        p_.requires_grad = True
    model.eval()
    with contextlib.redirect_stdout(io.StringIO()):      # the reference prints the input shape
        loss = model(batch)
    loss.backward()
    out["stage2_eval_loss"] = np.array(loss.item())
    with_grad = [k for k, p_ in model.named_parameters() if p_.grad is not None]
    out["stage2_params_with_grad"] = np.array(with_grad)
    _grad_summary(out, "stage2_eval", model, with_grad)


def main():
    ns = ref_import.import_reference()
    torch.set_num_threads(8)
    for name, fn, needs_ns in (("ops_voxel_maxpool", gen_voxel_maxpool, True), ("ops_bilinear", gen_bilinear, True),
                               ("ops_msda", gen_msda, True), ("ops_voting", gen_voting, True),
                               ("preprocess", gen_preprocess, False), ("e2e", gen_e2e, True),
                               ("losses", gen_losses, True), ("seg", gen_seg, True), ("instance", gen_instance, True), ("training", gen_training, True)):
        out = {}
        fn(ns, out) if needs_ns else fn(out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-20s %4d arrays %8.1f KiB" % (name, len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
