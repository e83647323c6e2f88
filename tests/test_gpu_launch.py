"""The multi-rank entry points as a user (or the driver) starts them: `python bench.py --gpus N` and
`python -m streammos_amd.run_sequence --gpus N` with no launcher around them create their ranks themselves
(streammos_amd/launch.py; the reference's launch line is README.md:97 / val_StreamMOS.py:205-218).

A one-GPU box cannot host two RCCL ranks, so the rehearsal knobs put both ranks on cuda:0 with a gloo group
(SMOS_BENCH_BACKEND=gloo, SMOS_BENCH_ONE_DEVICE=1): the spawn path, the barrier-bracketed timing, the MAX over ranks and
the per-rank sequence shard are the ones an 8-GPU run uses; only the backend name differs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from streammos_amd import kitti, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SMOS_BENCH_BACKEND="gloo", SMOS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_bench_gpus_2_spawns_two_ranks_on_the_device():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--frames", "3", "--cpu-scans", "0", "--no-raw"], capture_output=True, text=True, timeout=900, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = lines[0]
    print("2 ranks on one device: %.1f scans/s aggregate, %.2f ms/step" % (line["value"], line["ms_per_step"]))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["parallelism"] == "sequence-shard x2"
    assert line["steps"] == 3 and line["value"] > 0 and abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) < 0.01 * line["value"]


def _write_sequence(seq, n, first):
    (seq / "velodyne").mkdir(parents=True)
    for k in range(n):
        synth.synthetic_scan(first + k, 16, 120).tofile(seq / "velodyne" / ("%06d.bin" % k))
    kitti.write_poses(seq / "poses.txt", [synth.synthetic_pose(k) for k in range(n)])
    kitti.write_calibration(seq / "calib.txt")


def test_run_sequence_gpus_2_shards_whole_sequences(tmp_path):
    """Three sequences over two self-started ranks: longest-first assignment (streaming.shard_sequences), every scan of
    every sequence gets its prediction and refined file, and the sharded files equal those of a one-rank run."""
    lens = {"11": 9, "12": 5, "13": 4}
    for name, n in lens.items():
        _write_sequence(tmp_path / "sequences" / name, n, first=10 * int(name))
    dirs = [str(tmp_path / "sequences" / name) for name in lens]
    base = [sys.executable, "-m", "streammos_amd.run_sequence", "--frame-point-num", "2048", "--seq-dir"] + dirs
    two = subprocess.run(base + ["--out-dir", str(tmp_path / "two"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=900, env=_env(), cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    res = {r["sequence"]: r for r in (json.loads(l) for l in two.stdout.splitlines() if l.startswith("{"))}
    assert {k: v["scans"] for k, v in res.items()} == lens
    assert all(r["world"] == 2 for r in res.values())
    assert res["11"]["rank"] != res["12"]["rank"] and res["12"]["rank"] == res["13"]["rank"]     # 9 | 5 + 4
    one = subprocess.run(base + ["--out-dir", str(tmp_path / "one")], capture_output=True, text=True, timeout=900,
                         env=_env(), cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    for name, n in lens.items():
        for k in range(n):
            for sub in ("predictions", "refined"):
                a = np.fromfile(tmp_path / "two" / name / sub / ("%06d.label" % k), dtype=np.uint32)
                b = np.fromfile(tmp_path / "one" / name / sub / ("%06d.label" % k), dtype=np.uint32)
                assert a.shape == b.shape and np.array_equal(a, b), (name, k, sub)


def test_bench_gpus_2_trains_stage2_as_ddp_and_reproduces_the_pinned_loss(tmp_path, golden):
    """`bench.py --gpus 2 --train-steps 1`: after the inference timing BOTH ranks run the stage-2 step as real DDP on the group
    they already form (train_StreamMOS_seg.py:130,143,176-177), rank 0 reports `stage2_training` with the group's size, and an
    N > 1 line also carries `roofline` and `cpu_baseline`.  Fed the batch of tests/golden/training.npz (written by the
    reference) in the mode that fixture pins stage 2 in (eval), every rank must reproduce the reference's loss, and the
    gradients DDP leaves behind after its all-reduce must have the reference's norms.  (The batch is replicated, not sharded:
    OHEM top-k + Lovasz are set functions of the batch, so only a rank that sees the whole pinned batch can meet the pinned
    number; the average of identical gradients is what the all-reduce must return.)"""
    from tests import cases
    nb = cases.training_batch()
    np.savez(tmp_path / "batch.npz", **nb)
    env = dict(_env(), SMOS_BENCH_TRAIN_BATCH=str(tmp_path / "batch.npz"), SMOS_BENCH_TRAIN_EVAL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "3",
                          "--cpu-scans", "1", "--no-raw", "--train-steps", "1"], capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = lines[0]
    tr = line["stage2_training"]
    assert "error" not in tr, tr
    g = golden("training")
    want = float(g["stage2_eval_loss"])
    print("2-rank stage-2 step: loss %.6f (rank 0 %.6f) vs the reference's %.6f; %s ms/step; collectives %s"
          % (tr["loss"], tr["loss_rank0"], want, tr["ms_per_step"], {k: v for k, v in tr["collectives_per_step_on_the_wire"].items() if k != "note"}))
    assert line["n_gpus"] == 2 and tr["world"] == 2 and tr["mode"] == "eval"
    assert abs(tr["loss"] - want) <= 1e-4 * abs(want) and abs(tr["loss_rank0"] - want) <= 1e-4 * abs(want)
    assert sorted(tr["grad_norms"]) == sorted(str(k) for k in g["stage2_params_with_grad"])
    for k, v in tr["grad_norms"].items():
        ref = float(g["stage2_eval_grad_norm_%s" % k])
        assert abs(v - ref) <= 2e-3 * max(ref, 1e-12), (k, v, ref)
    assert tr["collectives_per_step_on_the_wire"]["ddp_gradient_bucket_all_reduce"] >= 1
    # the N > 1 line is complete: the dominant kernel's roofline and the CPU baseline ride along
    assert line["roofline"]["kernel"] and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_bench_gpus_2_train_mode_counts_syncbn_collectives_on_the_wire():
    """The same entry in TRAIN mode on per-rank synthetic samples at a small shape: SyncBatchNorm really synchronises (one
    all_gather per layer and forward, counted where torch.distributed issues it), both ranks finish with a finite loss."""
    env = dict(_env(), SMOS_BENCH_TRAIN_POINTS="4096", SMOS_BENCH_TRAIN_BATCH_PER_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "3",
                          "--cpu-scans", "0", "--no-raw", "--train-steps", "1"], capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    tr = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][-1]["stage2_training"]
    assert "error" not in tr, tr
    wire = tr["collectives_per_step_on_the_wire"]
    print("train mode, 2 ranks: %s" % {k: v for k, v in wire.items() if k != "note"})
    assert tr["world"] == 2 and tr["mode"] == "train" and np.isfinite(tr["loss"])
    gathers = wire.get("all_gather", 0) + wire.get("all_gather_into_tensor", 0)
    # three chained forwards per step x the 55 BatchNorm layers on the stage-2 forward path (61 in the model; the rest sit
    # in branches stage_forward does not run): the figure round 3 inferred from forward hooks, now read off the wire
    assert gathers == 3 * 55 and gathers <= 3 * tr["batchnorm_layers"]
    assert wire["ddp_gradient_bucket_all_reduce"] >= 1
