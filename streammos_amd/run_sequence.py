"""Stream one SemanticKITTI-format sequence through the MI355X path and write the reference's files.

    python -m streammos_amd.run_sequence --seq-dir .../sequences/08 --out-dir results/sequences/08 \
        [--checkpoint 40-model.pth] [--no-vote] [--device cuda:0]

Writes ``<out>/predictions/NNNNNN.label`` (network output, the files val_StreamMOS.py:121-126 writes) and, with
voting, ``<out>/refined/NNNNNN.label`` (the files voxel_voting.py:244-249 writes).  If the sequence has
``labels/``, the static / moving IoU of both is printed (utils/metric.py formula).  Several sequences are
sharded over ranks with ``streaming.shard_sequences``: under ``torch.distributed.run`` the ranks are the launcher's;
with ``--gpus N`` and no launcher around it the program starts its N ranks itself (``launch.self_launch``, the
reference's README.md:97 launch line folded in; the parent never touches the GPU).
"""
import argparse
import json
import os

import numpy as np
import torch

from . import kitti, launch, preprocess, streaming, synth


def load_model(checkpoint=None, device="cuda:0", seg=False):
    """seg=True: the stage-2 model (models/StreamMOS_seg.py) whose refine head also yields the `_bf` labels."""
    if seg:
        from .refapi.config import StreamMOS_seg as cfg
        from .refapi.models import StreamMOS_seg as StreamMOS
    else:
        from .refapi.config import StreamMOS as cfg
        from .refapi.models import StreamMOS
    model = StreamMOS.AttNet(cfg.get_config()[2])
    if checkpoint:
        state = torch.load(checkpoint, map_location="cpu", weights_only=True)
        state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}
        model.load_state_dict(state, strict=True)
    else:
        model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    return model.to(device).eval()


def run_sequence(model, seq_dir, out_dir, device="cuda:0", vote=True, frame_point_num=160000, limit=None, seq_num=3,
                 device_preprocess=False):
    """device_preprocess=True: only the raw scans are uploaded (each once) and the validation preprocessing runs on
    the GPU (identical to the host path except the last ulp of asinf / atan2f) -- the host then only
    reads files and writes labels."""
    spec = preprocess.VoxelSpec()
    files = sorted(f for f in os.listdir(os.path.join(seq_dir, "velodyne")) if f.endswith(".bin"))
    if limit:
        files = files[:limit]
    poses = kitti.read_poses(os.path.join(seq_dir, "poses.txt"), kitti.read_calibration(os.path.join(seq_dir, "calib.txt")))
    has_gt = os.path.isdir(os.path.join(seq_dir, "labels"))
    runner = streaming.StreamRunner(model, device, vote=vote)
    m_raw, m_ref = kitti.MovingIoU(), kitti.MovingIoU()
    cache = {}

    def scan(i):
        if i not in cache:
            cache[i] = kitti.read_scan(os.path.join(seq_dir, "velodyne", files[i]))
            for k in [k for k in cache if k < i - 12]:
                del cache[k]
        return cache[i]

    def gt(i):
        return kitti.read_label(os.path.join(seq_dir, "labels", files[i][:-4] + ".label"))

    def emit_refined(voted):
        for fid, lab in voted:
            lab = lab.cpu().numpy()
            kitti.write_prediction(os.path.join(out_dir, "refined", files[fid][:-4] + ".label"), lut_labels=lab)
            if has_gt:
                m_ref.add(gt(fid), np.where(lab == 251, 2, np.where(lab == 9, 1, 0)))

    dev_cache = {}

    def dev_scan(i):
        if i not in dev_cache:
            dev_cache[i] = torch.from_numpy(np.ascontiguousarray(scan(i))).to(device, non_blocking=True)
            for k in [k for k in dev_cache if k < i - 12]:
                del dev_cache[k]
        return dev_cache[i]

    for i in range(len(files)):
        idx = [min(j, len(files) - 1) for j in preprocess.window_indices(i, len(files), seq_num)]
        if device_preprocess:
            nxt = None
            if i + 1 < len(files):
                nxt = [min(j, len(files) - 1) for j in preprocess.window_indices(i + 1, len(files), seq_num)]
            out = runner.step_raw([dev_scan(j) for j in idx], [poses[j] for j in idx], frame_point_num,
                                  next_scans=[dev_scan(j) for j in nxt] if nxt else None,
                                  next_poses=[poses[j] for j in nxt] if nxt else None)
        else:
            sample = preprocess.build_sample([scan(j) for j in idx], [poses[j] for j in idx], frame_point_num, spec, tta=True)
            out = runner.step(runner.upload(sample, scan(i)), poses[i])
        raw = out["raw_labels"].cpu().numpy()
        if device_preprocess:
            runner.check_last_raw_sample()      # a scan that leaves no padding is an error, as on the host path
        kitti.write_prediction(os.path.join(out_dir, "predictions", files[i][:-4] + ".label"), labels_012=raw)
        if "bf_raw_labels" in out:          # val_StreamMOS_seg.py:141: raw 0/1/2 words, no LUT
            kitti.write_prediction(os.path.join(out_dir, "predictions_bf", files[i][:-4] + ".label"),
                                   lut_labels=out["bf_raw_labels"].cpu().numpy())
        if has_gt:
            m_raw.add(gt(i), raw)
        emit_refined(out["voted"])
    if runner.voter is not None:
        emit_refined(runner.voter.flush())
    res = {"sequence": os.path.basename(os.path.normpath(seq_dir)), "scans": len(files)}
    if has_gt:
        res["network"] = m_raw.result()
        if vote:
            res["voted"] = m_ref.result()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq-dir", nargs="+", required=True)
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--device", default=None)
    ap.add_argument("--no-vote", action="store_true")
    ap.add_argument("--seg", action="store_true", help="stage-2 model StreamMOS_seg (488-tensor checkpoint)")
    ap.add_argument("--instance-vote", action="store_true",
                    help="voxel_instance_voting.py instead of voxel_voting.py for the refined labels (needs --seg: the "
                         "clusters come from the `_bf` prediction)")
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks (one per GPU) the sequences are sharded over; > 1 without a launcher: started by this program")
    ap.add_argument("--frame-point-num", type=int, default=160000, help="Val.frame_point_num of config/StreamMOS.py:44")
    ap.add_argument("--device-preprocess", action="store_true",
                    help="range filter / pose alignment / TTA / quantisation on the GPU: only raw scans cross PCIe")
    args = ap.parse_args()
    if args.gpus > 1 and not launch.under_launcher():
        import sys
        sys.exit(launch.self_launch(args.gpus, sys.argv[1:], module="streammos_amd.run_sequence"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # SMOS_BENCH_ONE_DEVICE=1 (rehearsal on a one-GPU box, as in bench.py): every rank uses cuda:0
    local = 0 if os.environ.get("SMOS_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    device = args.device or "cuda:%d" % local
    lengths = {d: len(os.listdir(os.path.join(d, "velodyne"))) for d in args.seq_dir}
    mine = streaming.shard_sequences(lengths, world)[rank]
    if args.instance_vote and not args.seg:
        ap.error("--instance-vote needs --seg")
    model = load_model(args.checkpoint, device, seg=args.seg)
    vote = False if args.no_vote else ("instance" if args.instance_vote else True)
    for d in mine:
        out = os.path.join(args.out_dir, os.path.basename(os.path.normpath(d))) if len(args.seq_dir) > 1 else args.out_dir
        res = run_sequence(model, d, out, device, vote=vote, limit=args.limit, frame_point_num=args.frame_point_num,
                           device_preprocess=args.device_preprocess)
        print(json.dumps(dict(res, rank=rank, world=world)), flush=True)


if __name__ == "__main__":
    main()
