"""CPU-only checks: the C ABI exports what include/*.h declares, the host-side mirror of the reference
interface, the CPU twin of VoxelMaxPool, sequence sharding (incl. a world_size-2 gloo run)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from streammos_amd import _lib, preprocess, streaming, synth
from tests import cases
from tests.util import check_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smos_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    names = _declared("smos.h")
    assert len(names) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)          # loads without a GPU; no compute call is made
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SIGNATURES) | {"smos_last_error"}
    assert lib.smos_abi_version() == 1


def test_cpu_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libsmos_cpu.so"))
    for n in _declared("smos_cpu.h"):
        assert hasattr(lib, n), n


def test_ops_refuse_cpu_tensors_loudly():
    from streammos_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.bilinear_gather(torch.zeros(1, 2, 4, 4), torch.zeros(1, 3, 2), (1.0, 1.0))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.tta_argmax(torch.zeros(2, 3, 5))
    from streammos_amd.refapi import MultiScaleDeformableAttention as msda
    with pytest.raises(RuntimeError, match="CPU"):
        msda.ms_deform_attn_forward(torch.zeros(1, 4, 1, 2), torch.tensor([[2, 2]]), torch.tensor([0]),
                                    torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1), 64)


@pytest.mark.parametrize("name", sorted(cases.voxel_maxpool_cases()))
def test_cpu_twin_voxel_maxpool_bit_exact(golden, name):
    """deep_point.VoxelMaxPool on CPU tensors (the DataLoader-side use) -> libsmos_cpu.so."""
    from streammos_amd.refapi import deep_point
    g = golden("ops_voxel_maxpool")
    feat, ind, out_size, scale = cases.voxel_maxpool_cases()[name]
    f = torch.from_numpy(feat).unsqueeze(-1).requires_grad_(True)
    y = deep_point.VoxelMaxPool(f, torch.from_numpy(ind).unsqueeze(-1), out_size, scale)
    assert np.array_equal(y.detach().numpy(), g["vmp_%s_out" % name])
    y.backward(torch.from_numpy(cases.grad_like(y.shape, name)))
    assert np.array_equal(f.grad[..., 0].numpy(), g["vmp_%s_grad" % name])
    y64 = deep_point.VoxelMaxPool(torch.from_numpy(feat).double().unsqueeze(-1), torch.from_numpy(ind).double().unsqueeze(-1),
                                  out_size, scale)
    assert np.array_equal(y64.numpy(), g["vmp_%s_out" % name].astype(np.float64))


def test_pybind_named_shims_keep_the_reference_argument_lists():
    from streammos_amd.refapi.point_deep import cpu_kernel
    feat, ind, out_size, scale = cases.voxel_maxpool_cases()["basic"]
    f, i = torch.from_numpy(feat).unsqueeze(-1), torch.from_numpy(ind).unsqueeze(-1)
    out = torch.zeros((2, 3) + tuple(out_size))
    idx = torch.full((2, 200), -1, dtype=torch.int64)
    cpu_kernel.voxel_maxpooling_cpu_forward(f, i, out, idx, torch.tensor(out.shape), torch.tensor(out.stride()),
                                            torch.tensor(out_size), torch.tensor(scale))
    from oracle import ops_np
    want, want_idx = ops_np.voxel_maxpool_fwd(feat, ind, out_size, scale)
    assert np.array_equal(out.numpy(), want) and np.array_equal(idx.numpy(), want_idx)


def test_compiled_pybind_shims_load_and_refuse_cpu_tensors():
    """csrc/shim/pybind_shims.cpp compiles against torch + include/smos.h, links libsmos_hip.so and
    exports the reference's four function names; without a GPU the only thing to call is the CUDA-tensor check."""
    from streammos_amd import build
    from streammos_amd.refapi import compiled
    build.build_pybind_shims()
    pd = compiled.load("point_deep_cuda_kernel")
    ms = compiled.load("MultiScaleDeformableAttention")
    assert callable(pd.voxel_maxpooling_forward) and callable(pd.voxel_maxpooling_backward)
    assert callable(ms.ms_deform_attn_forward) and callable(ms.ms_deform_attn_backward)
    z = torch.zeros(1, 2, 4, 1)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        pd.voxel_maxpooling_forward(z, z, z, z.long(), z.long(), z.long(), z.long(), z)
    v = torch.zeros(1, 4, 1, 8)
    with pytest.raises(RuntimeError, match="CPU"):
        ms.ms_deform_attn_forward(v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 4, 1, 1, 1, 2), torch.zeros(1, 4, 1, 1, 1), 1)


def test_attnet_state_dict_layout_is_the_reference_layout():
    from streammos_amd.refapi.config import StreamMOS as cfg
    from streammos_amd.refapi.models import StreamMOS
    m = StreamMOS.AttNet(cfg.get_config()[2])
    layout = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))["stage1"]
    mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert mine == layout
    assert sum(p.numel() for p in m.parameters()) == 4367726
    # the twice-registered Unbalance blocks share storage (reference quirk, multi_view_encoder.py:344-354)
    sd = m.state_dict()
    assert sd["bev_net.header_unbalance_conv.layer7x3.0.weight"].data_ptr() == sd["bev_net.header_bev.1.layer7x3.0.weight"].data_ptr()
    m.load_state_dict(synth.seeded_state_dict(sd), strict=True)


def test_attnet_module_graph_on_cpu_matches_reference_golden(golden, monkeypatch):
    """The host-side graph (module wiring, CPU twin scatter, grid_sample gather) against the reference's
    outputs; the GPU-only sampler is swapped for the debug torch formulation, as deformattn/test.py does."""
    from streammos_amd.refapi import MultiScaleDeformableAttention as msda
    from streammos_amd.refapi.config import StreamMOS as cfg
    from streammos_amd.refapi.deformattn.functions import ms_deform_attn_core_pytorch
    from streammos_amd.refapi.models import StreamMOS
    monkeypatch.setattr(msda, "ms_deform_attn_forward",
                        lambda v, s, l, loc, w, step: ms_deform_attn_core_pytorch(v, s, loc, w))
    g = golden("e2e")
    m = StreamMOS.AttNet(cfg.get_config()[2]).eval()
    m.load_state_dict(synth.seeded_state_dict(m.state_dict()), strict=True)
    memory = None
    with torch.no_grad():
        for i, batch in enumerate(cases.e2e_frames(2)):
            check_inputs(g, "e2e_f%d_in_sha" % i, batch["pcds_xyzi"], batch["pcds_coord"], batch["pcds_sphere_coord"])
            tb = {k: torch.from_numpy(v).unsqueeze(0) for k, v in batch.items()}
            pred, _, _, _, memory = m.infer(tb, i, memory)
            ref = g["e2e_f%d_pred" % i]
            assert np.abs(pred.numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


def test_refapi_install_publishes_reference_names():
    code = ("import sys; sys.path.insert(0, %r); import streammos_amd.refapi as r; r.install(); "
            "import deep_point, point_deep.cuda_kernel, point_deep.cpu_kernel, MultiScaleDeformableAttention; "
            "from models import StreamMOS; from networks import backbone; from deformattn.modules import MSDeformAttn; "
            "import config.StreamMOS as c; m = eval('StreamMOS.AttNet')(c.get_config()[2]); "
            "print(len(m.state_dict()), deep_point.VoxelMaxPool.__module__)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "474"


def test_refapi_runs_an_unchanged_entry_script(tmp_path):
    """python -m streammos_amd.refapi <script> [args]: the script sees the reference's import names, its own argv and its
    own directory on sys.path -- the way val_StreamMOS.py starts (`from models import *`, eval of the config's prefix)."""
    (tmp_path / "helper_next_to_script.py").write_text("VALUE = 7\n")
    script = tmp_path / "val_like.py"
    script.write_text(
        "import sys, argparse\n"
        "import deep_point\n"
        "from models import *\n"
        "import helper_next_to_script\n"
        "import importlib\n"
        "ap = argparse.ArgumentParser(); ap.add_argument('--config'); a = ap.parse_args()\n"
        "cfg = importlib.import_module(a.config.replace('.py', '').replace('/', '.'))\n"
        "g, d, m, o = cfg.get_config()\n"
        "model = eval(m.prefix)(m)\n"
        "print(__name__, len(model.state_dict()), helper_next_to_script.VALUE, sys.argv[1:])\n")
    out = subprocess.run([sys.executable, "-m", "streammos_amd.refapi", str(script), "--config", "config/StreamMOS.py"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[:3] == ["__main__", "474", "7"] and "--config" in out.stdout


def test_vote_history_window_follows_the_reference():
    assert streaming.vote_history_ids(8) == [7, 6, 5, 4, 3, 2, 1, 0]       # voxel_voting.py:182
    assert streaming.vote_history_ids(20) == list(range(19, 11, -1))
    assert streaming.vote_history_ids(0) == [1, 2, 3, 4, 5, 6, 7]           # :199-200 (future frames)
    assert streaming.vote_history_ids(5) == [0, 1, 2, 3, 4, 6, 7]


def test_window_indices_follow_dataloadval():
    assert preprocess.window_indices(0, 100, 3) == [0, 1, 2]
    assert preprocess.window_indices(1, 100, 3) == [1, 2, 3]
    assert preprocess.window_indices(2, 100, 3) == [2, 1, 0]
    assert preprocess.window_indices(50, 100, 3) == [50, 49, 48]


def test_shard_sequences_is_balanced_and_complete():
    # SemanticKITTI test sequences 11-21 (scan counts), SURVEY.md section 8d config 4
    lengths = {11: 921, 12: 1061, 13: 3281, 14: 631, 15: 1901, 16: 1731, 17: 491, 18: 1801, 19: 4981, 20: 831, 21: 2721}
    shards = streaming.shard_sequences(lengths, 8)
    assert sorted(s for sh in shards for s in sh) == sorted(lengths)
    loads = [sum(lengths[s] for s in sh) for sh in shards]
    assert max(loads) == 4981                      # bounded by the longest sequence
    assert streaming.shard_sequences(lengths, 1) == [sorted(lengths, key=lambda k: -lengths[k])]


_GLOO_WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from streammos_amd import streaming
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lengths = {11: 921, 12: 1061, 13: 3281, 14: 631, 15: 1901}
mine = streaming.shard_sequences(lengths, world)[rank]
# every rank streams only its own sequences; the only exchange is the final gather of per-sequence counts
done = torch.zeros(32, dtype=torch.int64)
for s in mine:
    done[s] = lengths[s]
dist.all_reduce(done)
elapsed = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"total": int(done.sum()), "max_elapsed": float(elapsed), "world": world}))
dist.destroy_process_group()
"""


def test_sequence_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"total": 921 + 1061 + 3281 + 631 + 1901, "max_elapsed": 2.0, "world": 2}


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must create the two ranks itself (the flag was
    parsed and ignored).  --dry-launch walks the same spawn path with gloo ranks that only join the group and time a
    barrier, so it runs without a GPU; the JSON line's n_gpus is the size of the group that really formed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "3"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE line, from rank 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["dry_launch"] and lines[0]["config"]["parallelism"] == "sequence-shard x2"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-launch"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_gpus_2_dry_launch_trains_stage2_on_both_ranks():
    """`bench.py --gpus 2 --dry-launch --train-steps 1`: the ranks the program started itself form one group and run the
    stage-2 DDP step through bench.train_bench (CPU tensors over gloo at the rehearsal shape); rank 0 reports the size of the
    group that trained and the collectives it counted on the wire (configs[4] had no multi-rank entry)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--train-steps", "1", "--steps", "2"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    tr = lines[0]["stage2_training"]
    assert lines[0]["n_gpus"] == 2 and tr["world"] == 2 and tr["backend"] == "gloo" and tr["steps"] == 1
    assert np.isfinite(tr["loss"]) and tr["trainable_tensors"] == 8 and len(tr["grad_norms"]) == 8
    wire = tr["collectives_per_step_on_the_wire"]
    assert wire["ddp_gradient_bucket_all_reduce"] >= 1
    assert wire["process_group_sequence_numbers"] is None or wire["process_group_sequence_numbers"] >= wire["ddp_gradient_bucket_all_reduce"]


def test_self_launch_reports_a_failing_rank(tmp_path):
    """A rank that dies must fail the whole launch (non-zero exit of the parent)."""
    from streammos_amd import launch
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    assert launch.self_launch(2, [], script=str(bad), timeout=120) != 0
    good = tmp_path / "good.py"
    good.write_text("import os\nassert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n")
    assert launch.self_launch(2, [], script=str(good), timeout=120) == 0


def test_kitti_formats_round_trip(tmp_path):
    from streammos_amd import kitti
    poses = [synth.synthetic_pose(k) for k in range(4)]
    tr = np.eye(4)
    tr[:3, 3] = [0.1, -0.2, 0.3]
    kitti.write_poses(tmp_path / "poses.txt", [tr.dot(p).dot(np.linalg.inv(tr)) for p in poses])
    kitti.write_calibration(tmp_path / "calib.txt", tr)
    back = kitti.read_poses(tmp_path / "poses.txt", kitti.read_calibration(tmp_path / "calib.txt"))
    assert np.abs(np.array(back) - np.array(poses)).max() < 1e-12          # inv(Tr) * pose * Tr (datasets/utils.py:52)
    scan = synth.synthetic_scan(0, 4, 10)
    scan.tofile(tmp_path / "000000.bin")
    assert np.array_equal(kitti.read_scan(tmp_path / "000000.bin"), scan)
    raw = np.array([0, 1, 40, 252, 9 | (7 << 16), 259 | (3 << 16)], dtype=np.uint32)     # instance ids in the high half
    raw.tofile(tmp_path / "gt.label")
    assert kitti.read_label(tmp_path / "gt.label").tolist() == [0, 0, 1, 2, 1, 2]
    kitti.write_prediction(str(tmp_path / "predictions" / "000000.label"), labels_012=np.array([0, 1, 2]))
    assert np.fromfile(tmp_path / "predictions" / "000000.label", dtype=np.uint32).tolist() == [0, 9, 251]
    m = kitti.MovingIoU()
    m.add(np.array([0, 1, 2, 2, 1]), np.array([2, 1, 2, 1, 1]))
    assert abs(m.result()["moving_iou"] - 0.5) < 1e-9 and abs(m.result()["static_iou"] - 2 / 3) < 1e-9


def test_training_losses_match_reference_golden(golden):
    from streammos_amd.refapi.models import losses
    g = golden("losses")
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    gt = torch.from_numpy(g["gt"])
    a = losses.ohem_cross_entropy(pred, gt, top_ratio=0.2, top_weight=4.0, ignore_index=0)
    b = losses.lovasz_softmax(pred, gt, ignore=0)
    assert abs(a.item() - float(g["ohem"])) <= 1e-6 * abs(float(g["ohem"]))
    assert abs(b.item() - float(g["lovasz"])) <= 1e-6
    grad = torch.autograd.grad(a + 3 * b, pred)[0].numpy()
    np.testing.assert_allclose(grad, g["grad"], rtol=1e-5, atol=1e-8)
    assert losses.lovasz_softmax(pred, torch.zeros_like(gt), ignore=0) == 0        # everything ignored


def test_training_forward_on_cpu_with_gloo_ddp(tmp_path):
    """Two-rank DDP (gloo) over the CPU module graph: gradients are all-reduced and both ranks end the step with
    identical weights.  The GPU-only sampler is swapped for the debug torch formulation (as deformattn/test.py does)."""
    script = tmp_path / "train_worker.py"
    script.write_text(_DDP_WORKER % ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="3"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["same_weights"] and line["loss_finite"] and line["changed"]


_DDP_WORKER = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
from streammos_amd import preprocess, synth
from streammos_amd.refapi import MultiScaleDeformableAttention as msda
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.deformattn.functions import ms_deform_attn_func as fn
from streammos_amd.refapi.models import StreamMOS

# CPU stand-in for the GPU-only sampler: the debug torch formulation, differentiable through autograd
class _Fn:
    @staticmethod
    def apply(value, shapes, lsi, loc, attn, step):
        return fn.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
import streammos_amd.refapi.deformattn._msda as mod
mod.MSDeformAttnFunction = _Fn

dist.init_process_group("gloo")
rank = dist.get_rank()
torch.manual_seed(0)
model = StreamMOS.AttNet(cfg.get_config()[2])
model.load_state_dict(synth.seeded_state_dict(model.state_dict()))
ddp = DDP(model.train(), find_unused_parameters=True)
opt = torch.optim.SGD(ddp.parameters(), lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-3)
spec = preprocess.VoxelSpec()
n = 512
scans = [synth.synthetic_scan(10 * rank + k, 8, 40) for k in range(5)]
poses = [synth.synthetic_pose(k) for k in range(5)]
gen = torch.Generator().manual_seed(rank)
batch = {}
for i in range(3):
    idx = preprocess.window_indices(i, 5, 3)
    s = preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], n, spec, tta=False)
    for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord"):
        batch["%%s_%%d" %% (k, i)] = torch.from_numpy(s[k])
    batch["pcds_target_%%d" %% i] = torch.randint(0, 3, (1, n, 1), generator=gen)
    batch["pcds_bev_target_%%d" %% i] = torch.randint(0, 3, (1, 256, 256, 1), generator=gen)
before = model.pred_layer.pred_layer[0].weight.detach().clone()
loss = ddp(batch)
loss.backward()
opt.step()
w = model.pred_layer.pred_layer[0].weight.detach().clone()
ws = [torch.zeros_like(w) for _ in range(2)]
dist.all_gather(ws, w)
if rank == 0:
    print(json.dumps({"same_weights": bool(torch.equal(ws[0], ws[1])), "loss_finite": bool(torch.isfinite(loss)),
                      "changed": bool((w - before).abs().max() > 0)}))
dist.destroy_process_group()
"""


def test_seg_variant_layout_and_cpu_graph_match_reference(golden, monkeypatch):
    """models.StreamMOS_seg.AttNet (stage 2, + refine head): 488-tensor layout and the module graph on CPU against the
    reference's outputs (GPU-only sampler swapped for the debug torch formulation)."""
    from streammos_amd.refapi import MultiScaleDeformableAttention as msda
    from streammos_amd.refapi.config import StreamMOS_seg as cfg
    from streammos_amd.refapi.deformattn.functions import ms_deform_attn_core_pytorch
    from streammos_amd.refapi.models import StreamMOS_seg
    monkeypatch.setattr(msda, "ms_deform_attn_forward",
                        lambda v, s, l, loc, w, step: ms_deform_attn_core_pytorch(v, s, loc, w))
    m = StreamMOS_seg.AttNet(cfg.get_config()[2]).eval()
    layout = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))["stage2_seg"]
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()] == layout
    assert len(layout) == 488 and cfg.get_config()[2].prefix == "StreamMOS_seg.AttNet"
    m.load_state_dict(synth.seeded_state_dict(m.state_dict()), strict=True)
    g = golden("seg")
    batch = next(iter(cases.e2e_frames(1)))
    check_inputs(g, "seg_f0_in_sha", batch["pcds_xyzi"], batch["pcds_coord"], batch["pcds_sphere_coord"])
    with torch.no_grad():
        out = m.infer({k: torch.from_numpy(v).unsqueeze(0) for k, v in batch.items()}, 0)
    assert len(out) == 6
    for got, key in ((out[0], "seg_f0_pred"), (out[1], "seg_f0_bf_pred")):
        assert np.abs(got.numpy() - g[key]).max() <= 1e-4 * np.abs(g[key]).max()


def test_conv_cl_supported_mirrors_the_kernel_limits():
    """The engine asks ops.conv_cl_supported before it routes a layer to the own conv kernels: channel
    multiples, Cout <= 2048, kernel <= 7, and the 2 GiB operand limit of the 32-bit buffer offsets.  Shapes only."""
    from streammos_amd import ops

    def cl(b, c, h, w):
        return torch.empty((b, h, w, c), device="meta").permute(0, 3, 1, 2)
    assert ops.conv_cl_supported(cl(4, 128, 256, 256), 64, (3, 3))
    assert ops.conv_cl_supported(cl(32, 128, 256, 256), 64, (3, 3))                 # 8 streams: 1.07 GB
    assert not ops.conv_cl_supported(cl(64, 128, 256, 256), 64, (3, 3))             # 16 streams: 2.1 GB input
    assert not ops.conv_cl_supported(cl(64, 64, 256, 256), 128, (3, 3))             # 2.1 GB output
    assert not ops.conv_cl_supported(cl(4, 48, 64, 64), 64, (3, 3))                 # Cin % 32
    assert not ops.conv_cl_supported(cl(4, 64, 64, 64), 4096, (1, 1))               # Cout > 2048
    assert not ops.conv_cl_supported(cl(4, 64, 64, 64), 64, (9, 9))
    assert not ops.conv_cl_supported(cl(4, 64, 64, 64), 64, (3, 3), stride=3)
    wide = torch.empty((44, 256, 256, 192), device="meta").permute(0, 3, 1, 2)[:, :64]   # a channel slice: the PITCH counts
    assert not ops.conv_cl_supported(wide, 64, (3, 3))


def test_conv_wino_prepare_operand_order_reproduces_the_convolution():
    """Host logic of the Winograd path without a GPU: ops.conv_wino_prepare packs U = G w G^T in the operand order
    include/smos.h documents; unpacking that block with the documented index formula and running F(2x2, 3x3) in numpy
    (B^T d B, 16 channel sums, A^T M A) must give conv2d.  Pins the packing independently of the kernel."""
    import torch.nn.functional as F
    from streammos_amd import ops
    gen = torch.Generator().manual_seed(5)
    for cin, cout, mb in ((16, 16, 1), (32, 64, 2), (48, 32, 1)):
        w = torch.randn((cout, cin, 3, 3), generator=gen, dtype=torch.float32)
        packed = ops.conv_wino_prepare(w, mb).numpy()
        assert packed.size == 16 * cout * cin
        # wprep[((((ct * (Cin/16) + cc) * 4 + i) * mb + m) * 4 + xi) * 64 + lane][nu] = U[xi][nu] of
        # w[ct*16*mb + m*16 + (lane & 15)][cc*16 + 4*(lane >> 4) + i]
        blk = packed.reshape(cout // (16 * mb), cin // 16, 4, mb, 4, 64, 4)
        u = np.zeros((cout, cin, 4, 4), dtype=np.float64)
        for lane in range(64):
            for i in range(4):
                co = np.arange(cout // (16 * mb))[:, None] * 16 * mb + np.arange(mb)[None, :] * 16 + (lane & 15)      # [ct, m]
                ci = np.arange(cin // 16) * 16 + 4 * (lane >> 4) + i                                                   # [cc]
                u[co[:, None, :], ci[None, :, None]] = blk[:, :, i, :, :, lane, :].transpose(0, 1, 2, 3, 4)
        g = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]])
        want_u = np.einsum("ij,ocjk,lk->ocil", g, w.numpy().astype(np.float64), g)
        assert np.abs(u - want_u).max() <= 1e-6 * np.abs(want_u).max()                 # float32 rounding of the float64 transform
        x = torch.randn((2, cin, 6, 8), generator=gen)
        bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
        at = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
        xp = np.pad(x.numpy().astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
        out = np.zeros((2, cout, 6, 8))
        for ty in range(3):
            for tx in range(4):
                d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
                v = np.einsum("ij,bcjk,lk->bcil", bt, d, bt)
                m = np.einsum("ocil,bcil->boil", u, v)
                out[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ij,bojk,lk->boil", at, m, at)
        want = F.conv2d(x.double(), w.double(), None, 1, 1).numpy()
        assert np.abs(out - want).max() <= 1e-5 * np.abs(want).max(), (cin, cout, mb)
    with pytest.raises(RuntimeError):
        ops.conv_wino_prepare(torch.zeros(32, 32, 5, 3), 2)          # not a 3x3 kernel
    with pytest.raises(RuntimeError):
        ops.conv_wino_prepare(torch.zeros(16, 32, 3, 3), 2)          # Cout not a multiple of 16 * mb


def test_conv_wino1d_prepare_operand_order_reproduces_the_convolution():
    """Host logic of the 1-D Winograd path without a GPU: ops.conv_wino1d_prepare packs U = G g (per long-axis tap) in the
    operand order include/smos.h documents; unpacking with the documented index formula and running F(2, 3) along the 3-tap
    axis in numpy (B^T d, sums over channels and long-axis taps, A^T m) must give conv2d -- for both orientations."""
    import torch.nn.functional as F
    from streammos_amd import ops
    gen = torch.Generator().manual_seed(6)
    g = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]])
    bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    at = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    for cin, cout, mb, (kh, kw) in ((16, 16, 1, (5, 3)), (32, 32, 2, (7, 3)), (16, 32, 2, (3, 5)), (32, 16, 1, (3, 7))):
        w = torch.randn((cout, cin, kh, kw), generator=gen, dtype=torch.float32)
        kl = max(kh, kw)
        packed = ops.conv_wino1d_prepare(w, mb).numpy()
        assert packed.size == 4 * kl * cout * cin
        # wprep[((((ct * (Cin/16) + cc) * 4 + i) * KL + kL) * mb + m) * 64 + lane][pos] = U[kL][pos] of
        # w[ct*16*mb + m*16 + (lane & 15)][cc*16 + 4*(lane >> 4) + i]
        blk = packed.reshape(cout // (16 * mb), cin // 16, 4, kl, mb, 64, 4)
        u = np.zeros((cout, cin, kl, 4), dtype=np.float64)
        for lane in range(64):
            for i in range(4):
                for ct in range(cout // (16 * mb)):
                    for m in range(mb):
                        co = ct * 16 * mb + m * 16 + (lane & 15)
                        for cc in range(cin // 16):
                            u[co, cc * 16 + 4 * (lane >> 4) + i] = blk[ct, cc, i, :, m, lane, :]
        wl = w.numpy().astype(np.float64) if kw == 3 else w.numpy().astype(np.float64).transpose(0, 1, 3, 2)     # [co, ci, long, short]
        want_u = np.einsum("pk,oclk->oclp", g, wl)
        assert np.abs(u - want_u).max() <= 1e-6 * np.abs(want_u).max()
        x = torch.randn((2, cin, 6, 8), generator=gen)
        want = F.conv2d(x.double(), w.double(), None, 1, (kh // 2, kw // 2)).numpy()
        xl = x.numpy().astype(np.float64) if kw == 3 else x.numpy().astype(np.float64).transpose(0, 1, 3, 2)     # [b, c, L, S]
        n_l, n_s = xl.shape[2], xl.shape[3]
        xp = np.pad(xl, ((0, 0), (0, 0), (kl // 2, kl // 2), (1, 1 + n_s % 2)))
        out = np.zeros((2, cout, n_l, n_s + n_s % 2))
        for l in range(n_l):
            for t in range((n_s + 1) // 2):
                m_acc = np.zeros((2, cout, 4))
                for k in range(kl):
                    v = np.einsum("pj,bcj->bcp", bt, xp[:, :, l + k, 2 * t:2 * t + 4])
                    m_acc += np.einsum("ocp,bcp->bop", u[:, :, k, :], v)
                out[:, :, l, 2 * t:2 * t + 2] = np.einsum("ep,bop->boe", at, m_acc)
        out = out[:, :, :, :n_s]
        out = out if kw == 3 else out.transpose(0, 1, 3, 2)
        assert np.abs(out - want).max() <= 1e-5 * np.abs(want).max(), (cin, cout, mb, kh, kw)
    with pytest.raises(RuntimeError):
        ops.conv_wino1d_prepare(torch.zeros(32, 32, 3, 3), 2)        # 3x3 is the 2-D kernel's
    with pytest.raises(RuntimeError):
        ops.conv_wino1d_prepare(torch.zeros(16, 32, 7, 3), 2)        # Cout not a multiple of 16 * mb
    assert ops.conv_wino1d_ok((7, 3), 1, 32, 32) and not ops.conv_wino1d_ok((7, 3), 2, 32, 32)
    assert not ops.conv_wino1d_ok((7, 3), 1, 32, 32, residual=object()) and not ops.conv_wino1d_ok((7, 7), 1, 32, 32)


def test_bench_roofline_picks_the_dominant_kernel_family_not_a_label():
    """bench.family_table / dominant_family: the 36 Winograd launches of a step carry 14 labels but are ONE
    kernel; summed per family they dominate although single labels of other kernels (point_head, upconv_xy) are larger than any
    one of theirs.  Executed FLOPs follow the kernel the label ran on (4/9 resp. 2/3 of the direct count)."""
    import bench
    summary = {
        "point_head[4x160000]": (8, 8 * 0.30, 0.30),
        "upconv_xy[4x256x256x128<-128x128+64x64]": (8, 8 * 0.32, 0.32),
        "conv_cl[4x128x64x64->128x64x64k3x3]": (48, 48 * 0.04, 0.04),
        "conv_cl[4x64x256x256->128x256x256k3x3]": (8, 8 * 0.19, 0.19),
        "conv_cl[4x32x256x256->32x256x256k7x3]": (8, 8 * 0.08, 0.08),
        "conv_cl[4x64x256x256->64x128x128k3x3]": (8, 8 * 0.05, 0.05),
    }
    family = {"conv_cl[4x128x64x64->128x64x64k3x3]": "conv_wino", "conv_cl[4x64x256x256->128x256x256k3x3]": "conv_wino",
              "conv_cl[4x32x256x256->32x256x256k7x3]": "conv_wino1d", "conv_cl[4x64x256x256->64x128x128k3x3]": "conv_igemm"}
    table = bench.family_table(summary, family, 8, 0.0, None)
    assert bench.dominant_family(table) == "conv_wino"
    w = table["conv_wino"]
    assert abs(w["launches"] - 7) < 1e-9 and abs(w["ms"] - (6 * 0.04 + 0.19)) < 1e-9
    direct = 6 * 2 * 4 * 64 * 64 * 128 * 128 * 9 + 2 * 4 * 256 * 256 * 128 * 64 * 9
    assert w["alg_flops"] == direct and w["exec_flops"] == 6 * (2 * 4 * 64 * 64 * 128 * 128 * 9 * 4 // 9) + 2 * 4 * 256 * 256 * 128 * 64 * 9 * 4 // 9
    assert table["conv_wino1d"]["exec_flops"] * 3 == table["conv_wino1d"]["alg_flops"] * 2
    assert table["conv_igemm"]["exec_flops"] == table["conv_igemm"]["alg_flops"]
    # the bracket's own time comes off every launch, never more than a tenth of it
    t2 = bench.family_table(summary, family, 8, 0.01, None)
    assert abs(t2["conv_wino"]["ms"] - (6 * (0.04 - 0.004) + 0.19 - 0.01)) < 1e-9
    # stem_gemm is priced by its FLOPs once the frames' occupancy is known (it was booked as HBM-bound with 0 FLOPs)
    ctx = {"stem_rows": 200000, "stem_class_rows": (50000, 50000, 50000, 50000)}
    assert bench.algorithmic_flops("stem_gemm[4x512x512x192]", ctx) == 2 * 50000 * 192 * 32 * (2 + 3 + 3 + 5)
    assert bench.algorithmic_flops("stem_gemm[4x512x512x192]") == 0


def test_release_stream_workspaces_filters_by_device_and_spares_graph_namespaces():
    """ops.release_stream_workspaces: a (device, stream) release drops that device's eager scratch only -- block
    tables of another device and entries baked into a runner's captured graphs (namespace ('graph', owner, group)) stay; the
    owner path drops exactly that runner's."""
    from streammos_amd import ops
    saved = (dict(ops._stem_ws), dict(ops._flag_ws))
    ops._stem_ws.clear()
    ops._flag_ws.clear()
    try:
        t0, t1 = ops.new_block_scratch("cuda:0"), ops.new_block_scratch("cuda:1")
        for t in (t0, t1):
            t[(111, 0)] = "eager"
            t[(111, ("graph", 7, 0))] = "graph7"
            t[(222, 0)] = "other stream"
        ops._stem_ws[("cuda:0", 111, "stem_rows", None, 0)] = 1
        ops._stem_ws[("cuda:1", 111, "stem_rows", None, 0)] = 2
        ops._stem_ws[("cuda:0", 111, "stem_rows", None, ("graph", 7, 1))] = 3
        ops.release_stream_workspaces("cuda:0", 111)
        assert sorted(map(str, t0)) == sorted(map(str, [(111, ("graph", 7, 0)), (222, 0)])) and len(t1) == 3
        assert sorted(ops._stem_ws.values()) == [2, 3]
        ops.release_stream_workspaces(owner=7)
        assert list(t0) == [(222, 0)] and sorted(map(str, t1)) == sorted(map(str, [(111, 0), (222, 0)]))
        assert list(ops._stem_ws.values()) == [2]
        ops.release_stream_workspaces()
        assert not t0 and not t1 and not ops._stem_ws
    finally:
        ops._stem_ws.update(saved[0])
        ops._flag_ws.update(saved[1])
