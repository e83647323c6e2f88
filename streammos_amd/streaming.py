"""Streaming runner: one LiDAR sequence = one stream (recurrent memory + voting window).

Does on the device what the reference spreads over ``val_StreamMOS.py:88-126`` (infer, softmax / TTA
mean / argmax, un-pad, scatter through the range mask) and ``voxel_voting.py:176-249`` (8-frame
voxel voting) -- the latter without its per-frame disk round trip: the last eight scans and their
predictions stay resident in HBM and are re-voxelised in the current frame's coordinates by the
voting kernels.

Sequences are independent streams: multi-GPU inference assigns whole sequences to
ranks (``shard_sequences``) and needs no collective.
"""
import collections

import numpy as np
import torch

from . import ops

VOTE_WINDOW = 8                      # frames_num_max, voxel_voting.py:140
LEARNING_MAP_INV = {0: 0, 1: 9, 2: 251}   # config/StreamMOS.py:58


def shard_sequences(lengths, world_size):
    """Longest-processing-time-first assignment of whole sequences to ranks.
    lengths: {sequence_id: n_scans}.  Returns a list (one entry per rank) of sequence-id lists."""
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for seq, n in sorted(lengths.items(), key=lambda kv: (-kv[1], str(kv[0]))):
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(seq)
        loads[r] += n
    return shards


def vote_history_ids(frame_id, window=VOTE_WINDOW):
    """voxel_voting.py:177-212: frames >= window use the previous `window` frames; earlier frames use
    frames 0..window-1 except themselves (i.e. also *future* frames)."""
    if frame_id >= window:
        return list(range(frame_id - 1, frame_id - window - 1, -1))
    return [k for k in range(window) if k != frame_id]


class VoxelVoter:
    """HBM-resident voting window.  ``push`` takes a frame (raw points in its own sensor frame, per-point
    predictions in {0,1,2}, 4x4 pose) and returns the frames whose refined labels became available:
    a list of (frame_id, int32 labels tensor)."""

    def __init__(self, device, window=VOTE_WINDOW, lut=LEARNING_MAP_INV, recip_quantize=False):
        self.device = torch.device(device)
        self.window = window
        self.recip = recip_quantize
        self.table = torch.zeros(512 * 512 * 30, dtype=torch.int64, device=self.device)
        self.lut = None
        if lut is not None:
            t = torch.zeros(256, dtype=torch.int32)
            for k, v in lut.items():
                t[k] = v
            self.lut = t.to(self.device)
        self.frames = collections.OrderedDict()     # frame_id -> (points, preds, pose)
        self.next_id = 0

    def reset(self):
        self.frames.clear()
        self.next_id = 0

    def _vote(self, fid):
        pts, pred, pose = self.frames[fid]
        inv_cur = np.linalg.inv(pose)
        ops.vote_clear(self.table)
        for h in vote_history_ids(fid, self.window):
            if h not in self.frames:
                continue
            hp, hl, hpose = self.frames[h]
            ops.vote_accumulate(hp, hl, self.table, pose_diff=inv_cur.dot(hpose), recip_quantize=self.recip)
        ops.vote_accumulate(pts, pred, self.table, recip_quantize=self.recip)
        return ops.vote_resolve(pts, pred, self.table, lut=self.lut, recip_quantize=self.recip)

    def push(self, points, preds, pose):
        fid = self.next_id
        self.next_id += 1
        self.frames[fid] = (points, preds, np.asarray(pose, dtype=np.float64))
        ready = []
        if fid == self.window - 1:
            ready = [(k, self._vote(k)) for k in range(self.window)]
        elif fid >= self.window:
            ready = [(fid, self._vote(fid))]
            self.frames.pop(fid - self.window, None)
        return ready

    def flush(self):
        """End of a sequence shorter than the window: vote with whatever frames exist."""
        if self.next_id < self.window:
            return [(k, self._vote(k)) for k in range(self.next_id)]
        return []


class StreamRunner:
    """infer -> TTA reduce -> labels for the raw scan -> voxel voting, all on one device, one stream."""

    def __init__(self, model, device="cuda:0", vote=True, recip_quantize=False, graph=False):
        """graph=True captures the network forward + TTA reduce of a frame into two hipGraphs (first frame: learned
        memory embedding; later frames: recurrent memory) that are replayed on static buffers -- one launch per scan
        instead of ~180.  The per-frame voting stays outside the graph (its pose matrices are launch arguments)."""
        self.device = torch.device(device)
        self.model = model.to(self.device).eval()
        self.voter = VoxelVoter(self.device, recip_quantize=recip_quantize) if vote else None
        self.use_graph = graph
        self._graphs = None
        self.reset()

    # ---- hipGraph capture -------------------------------------------------------------------
    def _forward(self, batch, first):
        pred_cls, _, _, _, mem = self.model.infer(batch, 0 if first else 1, None if first else self._g_mem)
        return pred_cls, ops.tta_argmax(pred_cls), mem

    def _capture(self, dev):
        keys = ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")
        self._g_in = {k: torch.empty_like(dev[k]) for k in keys}
        for k in keys:
            self._g_in[k].copy_(dev[k])
        batch = {k: self._g_in[k].unsqueeze(0) for k in keys}
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            # eager warm-up on the capture stream: MIOpen algorithm search, lazy engine build, allocator pools
            _, _, mem = self._warm(batch)
            self._g_mem = mem.clone()
            for _ in range(2):
                self._forward(batch, False)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._graphs = {}
        for first in (True, False):
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                pred, labels, mem = self._forward(batch, first)
                self._g_mem.copy_(mem)
            self._graphs[first] = (g, pred, labels)

    def _warm(self, batch):
        pred, labels, mem = None, None, None
        for _ in range(2):
            pred_cls, _, _, _, mem = self.model.infer(batch, 0, None)
            labels = ops.tta_argmax(pred_cls)
        return pred, labels, mem

    def reset(self):
        self.memory = None
        self.frame = 0
        if self.voter is not None:
            self.voter.reset()

    def upload(self, sample, raw_scan=None):
        """Host sample (streammos_amd.preprocess.build_sample) -> device-resident inputs."""
        dev = {k: torch.from_numpy(np.ascontiguousarray(sample[k])).to(self.device)
               for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
        dev["valid_index"] = torch.from_numpy(np.nonzero(sample["valid_mask"])[0]).to(self.device)
        dev["n_raw"] = int(sample["valid_mask"].shape[0])
        dev["n_valid"] = int(sample["valid_mask"].sum())
        if raw_scan is not None:
            dev["raw_scan"] = torch.from_numpy(np.ascontiguousarray(raw_scan)).to(self.device)
        return dev

    @torch.no_grad()
    def step_raw(self, scans, poses, frame_point_num=160000):
        """One scan from RAW data: scans = list of T host (numpy) or device [n,4] float32 scans, current first;
        poses = their 4x4 poses.  Preprocessing runs on the device (streammos_amd.device_preprocess, row f1):
        only the raw scans cross PCIe and nothing synchronises with the host."""
        from .device_preprocess import DevicePreprocessor
        if getattr(self, "_pre", None) is None or self._pre.N != frame_point_num:
            self._pre = DevicePreprocessor(self.device, frame_point_num=frame_point_num)
        dev_scans = [s if torch.is_tensor(s) else torch.from_numpy(np.ascontiguousarray(s)) for s in scans]
        dev_scans = [s.to(self.device, non_blocking=True) for s in dev_scans]
        inv_cur = np.linalg.inv(np.asarray(poses[0], dtype=np.float64))
        # the current scan also goes through inv(P_cur) * P_cur, which is the identity only up to rounding
        # (the reference does exactly that, datasets/data_StreamMOS.py:424-467)
        built = self._pre.build(dev_scans, [inv_cur.dot(np.asarray(p, dtype=np.float64)) for p in poses])
        built["raw_scan"] = dev_scans[0]
        return self.step(built, poses[0])

    @torch.no_grad()
    def step(self, dev, pose=None):
        """One scan.  Returns dict(pred_cls, labels (N_pad,) uint8, raw_labels (n_raw,) uint8,
        voted = [(frame_id, int32 LUT labels)])."""
        if self.use_graph:
            if self._graphs is None or any(self._g_in[k].shape != dev[k].shape for k in self._g_in):
                self._capture(dev)
            for k, buf in self._g_in.items():
                buf.copy_(dev[k])
            g, pred_cls, labels = self._graphs[self.frame == 0]
            g.replay()
            self.memory = self._g_mem       # static buffers: valid until the next step()
        else:
            batch = {k: dev[k].unsqueeze(0) for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
            pred_cls, _, _, _, self.memory = self.model.infer(batch, self.frame, self.memory)
            labels = ops.tta_argmax(pred_cls)
        out = {"pred_cls": pred_cls, "labels": labels, "voted": []}
        raw = None
        if "prefix" in dev:                                                       # device-preprocessed sample
            raw = self._pre.unpad_labels(labels, dev)
        elif "valid_index" in dev:
            raw = torch.zeros(dev["n_raw"], dtype=torch.uint8, device=self.device)
            raw.index_copy_(0, dev["valid_index"], labels[:dev["n_valid"]])      # val_StreamMOS.py:112-118
        if raw is not None:
            out["raw_labels"] = raw
            if self.voter is not None and "raw_scan" in dev:
                out["voted"] = self.voter.push(dev["raw_scan"], raw, pose if pose is not None else np.eye(4))
        self.frame += 1
        return out


class MultiStreamRunner:
    """S independent sequences advanced in lock step as ONE batch of S x V samples (BASELINE.json configs[2]:
    concurrent sequences on one GPU, every stream's recurrent memory and voting window resident in HBM).
    Streams never interact: the network is batch-independent, the TTA reduce and the voting run per stream."""

    def __init__(self, model, device="cuda:0", n_streams=8, vote=True):
        self.device = torch.device(device)
        self.model = model.to(self.device).eval()
        self.n = n_streams
        self.voters = [VoxelVoter(self.device) for _ in range(n_streams)] if vote else None
        self.memory = None
        self.frame = 0

    @staticmethod
    def batch_inputs(devs):
        """list of per-stream uploads (StreamRunner.upload) -> one batched dict (concatenated along the TTA dim)."""
        out = {k: torch.cat([d[k] for d in devs], 0) for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
        out["streams"] = [{k: d[k] for k in d if k not in out} for d in devs]
        return out

    @torch.no_grad()
    def step(self, batched, poses):
        batch = {k: batched[k].unsqueeze(0) for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
        pred_cls, _, _, _, self.memory = self.model.infer(batch, self.frame, self.memory)
        v = pred_cls.shape[0] // self.n
        outs = []
        for s, meta in enumerate(batched["streams"]):
            labels = ops.tta_argmax(pred_cls[s * v:(s + 1) * v])
            raw = torch.zeros(meta["n_raw"], dtype=torch.uint8, device=self.device)
            raw.index_copy_(0, meta["valid_index"], labels[:meta["n_valid"]])
            voted = []
            if self.voters is not None and "raw_scan" in meta:
                voted = self.voters[s].push(meta["raw_scan"], raw, poses[s])
            outs.append({"labels": labels, "raw_labels": raw, "voted": voted})
        self.frame += 1
        return pred_cls, outs
