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
import os

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
        window = []
        for h in vote_history_ids(fid, self.window):
            if h not in self.frames:
                continue
            hp, hl, hpose = self.frames[h]
            window.append((hp, hl, inv_cur.dot(hpose)))
        window.append((pts, pred, None))
        ops.vote_accumulate_frames(window, self.table, recip_quantize=self.recip)     # one launch for the whole window
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


class InstanceVoter(VoxelVoter):
    """Voxel voting followed by the instance-level vote of voxel_instance_voting.py:195-272.

    ``push`` additionally takes the frame's movable-object prediction (the `_bf` labels of StreamMOS_seg, 2 =
    foreground).  Per voted frame: the foreground points are clustered with DBSCAN(0.3, 5) on the device
    (ops.dbscan), clusters of more than 30 points get an axis-aligned box whose floor is lifted by 0.2 m, the class-1 /
    class-2 points of the local map inside each box are counted on the device (ops.box_vote; the local map is the
    pose-aligned, cropped window with its PRE-vote predictions) and every point of the cluster takes class 2 if
    2 * n2 > n1 (the reference sums label VALUES, :178-179) else class 1.  torch is used for the small bookkeeping in
    between (compaction, per-cluster min / max); the DBSCAN call synchronises the stream."""

    EPS, MIN_SAMPLES, MIN_POINTS, FLOOR_LIFT = 0.3, 5, 30, 0.2

    def __init__(self, device, window=VOTE_WINDOW, lut=LEARNING_MAP_INV, recip_quantize=False):
        super().__init__(device, window=window, lut=lut, recip_quantize=recip_quantize)
        self.bf = {}

    def reset(self):
        super().reset()
        self.bf.clear()

    def push(self, points, preds, pose, bf=None):
        if bf is None:
            raise RuntimeError("InstanceVoter.push needs the movable-object (`_bf`) labels of the frame")
        self.bf[self.next_id] = bf
        ready = super().push(points, preds, pose)
        for k in [k for k in self.bf if k not in self.frames]:
            del self.bf[k]
        return ready

    def cluster_boxes(self, fpts):
        """fpts [m,3] float32 foreground points -> (member [m] bool, slot [m] long into the kept clusters, boxes [K,6])."""
        names = ops.dbscan(fpts, self.EPS, self.MIN_SAMPLES)
        valid = names >= 0
        empty = (valid & False, torch.zeros_like(names, dtype=torch.long), fpts.new_zeros((0, 6)))
        if not bool(valid.any()):
            return empty
        uniq, inv, counts = torch.unique(names[valid], return_inverse=True, return_counts=True)
        keep = counts > self.MIN_POINTS                                   # :165
        if not bool(keep.any()):
            return empty
        k = uniq.numel()
        idx = inv[:, None].expand(-1, 3)
        lo = fpts.new_full((k, 3), float("inf")).scatter_reduce_(0, idx, fpts[valid], "amin")
        hi = fpts.new_full((k, 3), float("-inf")).scatter_reduce_(0, idx, fpts[valid], "amax")
        lo, hi = lo[keep], hi[keep]
        lifted = lo[:, 2] + self.FLOOR_LIFT                               # :171-173: corners at z_min move up, in float32
        flat = hi[:, 2] == lo[:, 2]                                       # every corner is at z_min: the whole box moves
        z0, z1 = torch.minimum(lifted, hi[:, 2]), torch.maximum(lifted, hi[:, 2])
        boxes = torch.stack((lo[:, 0], lo[:, 1], z0, hi[:, 0], hi[:, 1], z1), dim=1)
        # a box without volume is "not a hull" for the reference's Delaunay test (:72-74): it contains nothing
        degenerate = flat | (boxes[:, 3:] == boxes[:, :3]).any(dim=1)
        boxes[degenerate, :3] = float("inf")
        slot_of = torch.full((k,), -1, dtype=torch.long, device=fpts.device)
        slot_of[keep] = torch.arange(int(keep.sum()), device=fpts.device)
        slot = torch.full_like(names, -1, dtype=torch.long)
        slot[valid] = slot_of[inv]
        return slot >= 0, slot, boxes.contiguous()

    def _vote(self, fid):
        pts, pred, pose = self.frames[fid]
        inv_cur = np.linalg.inv(pose)
        history = [(h, inv_cur.dot(self.frames[h][2])) for h in vote_history_ids(fid, self.window) if h in self.frames]
        ops.vote_clear(self.table)
        ops.vote_accumulate_frames([(self.frames[h][0], self.frames[h][1], diff) for h, diff in history] + [(pts, pred, None)],
                                   self.table, recip_quantize=self.recip)
        labels = ops.vote_resolve(pts, pred, self.table, lut=None, recip_quantize=self.recip)
        fg = torch.nonzero(self.bf[fid] == 2).flatten()                   # :145
        if fg.numel() > 0:
            member, slot, boxes = self.cluster_boxes(pts[fg][:, :3].contiguous())
            if boxes.shape[0] > 0:
                counts = torch.zeros((boxes.shape[0], 3), dtype=torch.int32, device=self.device)
                for h, diff in history:
                    ops.box_vote(self.frames[h][0], self.frames[h][1], boxes, counts, pose_diff=diff)
                ops.box_vote(pts, pred, boxes, counts)
                verdict = torch.where(2 * counts[:, 2] > counts[:, 1], 2, 1).to(torch.int32)   # :178-183
                labels[fg[member]] = verdict[slot[member]]
        return self.lut[labels.long()] if self.lut is not None else labels


def concurrent_stream(device, candidates=8, spin_cycles=400000):
    """A HIP stream whose kernels really run beside those of the current stream.  The runtime multiplexes its streams
    onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams that share a queue execute their
    kernels one after the other -- measured on MI355X / ROCm 7.2: the first stream torch hands out shares its queue with the
    default stream, so a pipeline built on that pair overlaps nothing.  This probes up to `candidates` fresh streams with
    a one-block spin kernel (torch.cuda._sleep) next to the same kernel on the current stream and returns the first pair that
    takes the time of one kernel rather than of two; falls back to the best candidate.  ~10 ms, once per runner."""
    import time
    main = torch.cuda.current_stream(device)

    def run(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(spin_cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    with torch.cuda.device(device):
        run([main])                                  # lazy initialisation out of the way
        single = min(run([main]) for _ in range(3))
        best, best_t = None, float("inf")
        keep = []                                    # hold the rejected streams until the choice is made (no handle reuse)
        for _ in range(candidates):
            cand = torch.cuda.Stream(device, priority=int(os.environ.get("SMOS_SIDE_PRIORITY", "0")))
            keep.append(cand)
            t = min(run([main, cand]) for _ in range(2))
            if t < best_t:
                best, best_t = cand, t
            if t < 1.4 * single:
                break
    return best


class StreamRunner:
    """infer -> TTA reduce -> labels for the raw scan -> voxel voting, all on one device, one stream."""

    def __init__(self, model, device="cuda:0", vote=True, recip_quantize=False, graph=False, split=1, pipeline=False,
                 skip_padding=True):
        """graph=True captures the network forward of a frame into hipGraphs (first frame: learned memory embedding;
        later frames: recurrent memory) that are replayed on static buffers -- one launch per scan instead of ~180.
        split=k additionally cuts the TTA batch into k independent groups (TTA variants never interact inside the
        network) whose graphs are replayed on k HIP streams at once, so that the many small kernels of one group fill
        the CUs the other group leaves idle.  Voting stays outside the graphs (pose matrices are launch arguments)."""
        self.device = torch.device(device)
        self.model = model.to(self.device).eval()
        # vote: False / True (voxel voting) / "instance" (voxel + instance voting; needs the StreamMOS_seg model)
        if vote == "instance":
            self.voter = InstanceVoter(self.device, recip_quantize=recip_quantize)
        else:
            self.voter = VoxelVoter(self.device, recip_quantize=recip_quantize) if vote else None
        self.use_graph = graph
        self.split = max(1, int(split))
        self._graphs = None
        self._ws_owner = ops.new_workspace_owner()      # names this runner's scratch namespaces (never reused, unlike id())
        # pipeline=True: only the temporal fusion needs the previous frame, so the encoder of frame t+1 (point MLP,
        # scatters, the BEV / range-view stages: ~60 % of the work) is issued on a second HIP stream while frame t
        # is decoded on the main one.  step() must then be given the next frame's inputs (one frame of look-ahead).
        self.pipeline = pipeline
        # skip_padding: the reference pads every scan to frame_point_num with points at -1000 (datasets/data_StreamMOS.py:568-571)
        # and cuts their predictions off again (val_StreamMOS.py:113).  The runner tells the point head how many points are
        # real (a device-side count: no host sync); the padding tail's logits come back as zeros instead of being computed
        # (25 % of the points of a SemanticKITTI scan at frame_point_num = 160 000, 40 % of the bench's synthetic ones).
        # Labels, raw labels and votes of every real point are unchanged; AttNet.infer (the reference API) computes all N.
        self.skip_padding = bool(skip_padding)
        self._side = concurrent_stream(self.device) if pipeline else None
        self._pre_enc = None
        self.reset()

    # ---- hipGraph capture -------------------------------------------------------------------
    _KEYS = ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")

    def _capture(self, dev):
        # The channels-last engine is captured as it stands: every convolution is the library's own kernel
        # (csrc/conv_igemm.hip), so no MIOpen solver is replayed.  (Round 1 pinned the NCHW engine here because MIOpen's
        # composable-kernel grouped-conv solvers gave ~2 % wrong logits under replay; the cause was never established and
        # those solvers are no longer on the path.)
        with torch.no_grad():
            eng = self.model._engine_for(dev["pcds_xyzi"])
        if eng is None:
            raise RuntimeError("StreamRunner(graph=True) needs the fused GPU engine (eval mode, fast_inference)")
        self._g_engine = eng            # the captured kernels read this engine's folded weights: keep it alive
        v = dev["pcds_xyzi"].shape[0]
        k = self.split if v % self.split == 0 else 1
        per = v // k
        self._groups = []
        side = torch.cuda.Stream(self.device)
        try:
            self._capture_groups(dev, eng, k, per, side)
        finally:
            ops.set_workspace_namespace(0)
        self._g_shape = tuple(dev["pcds_xyzi"].shape)
        self._g_pred = torch.empty((v,) + tuple(self._groups[0]["graphs"][True][1].shape[1:]), dtype=torch.float32,
                                   device=self.device)
        self._graphs = True

    def _capture_groups(self, dev, eng, k, per, side):
        for gi in range(k):
            sl = slice(gi * per, (gi + 1) * per)
            ops.set_workspace_namespace(("graph", self._ws_owner, gi))    # per-group scratch: the groups' graphs replay concurrently
            g_in = {key: dev[key][sl].clone() for key in self._KEYS}
            batch = {key: g_in[key].unsqueeze(0) for key in self._KEYS}
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side), torch.no_grad():
                # eager warm-up on a side stream: MIOpen algorithm search, lazy engine build, allocator pools
                mem = None
                for _ in range(2):
                    mem = self.model.infer(batch, 0, None)[-1]
                g_mem = mem.clone()
                for _ in range(2):
                    self.model.infer(batch, 1, g_mem)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            graphs = {}
            for first in (True, False):
                g = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(g):
                    res = self.model.infer(batch, 0 if first else 1, None if first else g_mem)
                    pred, mem = res[0], res[-1]
                    g_mem.copy_(mem)
                graphs[first] = (g, pred)
            self._groups.append({"in": g_in, "mem": g_mem, "graphs": graphs, "slice": sl,
                                 "stream": torch.cuda.Stream(self.device)})

    def _replay(self, dev):
        main = torch.cuda.current_stream(self.device)
        first = self.frame == 0
        for grp in self._groups:
            for key in self._KEYS:
                grp["in"][key].copy_(dev[key][grp["slice"]])
        for grp in self._groups:
            grp["stream"].wait_stream(main)
            with torch.cuda.stream(grp["stream"]):
                g, pred = grp["graphs"][first]
                g.replay()
                self._g_pred[grp["slice"]].copy_(pred)
        for grp in self._groups:
            main.wait_stream(grp["stream"])
        return self._g_pred

    def reset(self):
        self.memory = None
        self.frame = 0
        self._pre_enc = None
        self._raw_ahead = None
        self.__dict__.pop("_raw_cache", None)
        if self.voter is not None:
            self.voter.reset()

    def close(self):
        """Frees the per-stream scratch this runner's HIP streams hold (ops._stream_workspace: ~1.2 GB per stream at the
        validation shape).  The runner stays usable; the scratch is re-allocated on the next frame."""
        main = torch.cuda.current_stream(self.device).cuda_stream
        ops.release_stream_workspaces(self.device, main)
        if self._side is not None:
            ops.release_stream_workspaces(self.device, self._side.cuda_stream)
        if self._graphs is not None:                 # the captured graphs go first: their kernels use that scratch
            self._graphs = self._groups = None
            ops.release_stream_workspaces(owner=self._ws_owner)

    def upload(self, sample, raw_scan=None):
        """Host sample (streammos_amd.preprocess.build_sample) -> device-resident inputs."""
        dev = {k: torch.from_numpy(np.ascontiguousarray(sample[k])).to(self.device)
               for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
        dev["valid_index"] = torch.from_numpy(np.nonzero(sample["valid_mask"])[0]).to(self.device)
        dev["n_raw"] = int(sample["valid_mask"].shape[0])
        dev["n_valid"] = int(sample["valid_mask"].sum())
        dev["n_live"] = torch.tensor([dev["n_valid"]], dtype=torch.int32).to(self.device)     # real points at the front of every sample
        if raw_scan is not None:
            dev["raw_scan"] = torch.from_numpy(np.ascontiguousarray(raw_scan)).to(self.device)
        dev["ready"] = torch.cuda.Event()            # consumers on other HIP streams wait for the copies above
        dev["ready"].record(torch.cuda.current_stream(self.device))
        return dev

    @torch.no_grad()
    def _upload_raw(self, scan):
        """Host scan -> device, once per scan object (a window re-uses the scans of the previous frames), through a small
        ring of pinned staging buffers: a copy from pageable memory would hold the host until the stream has drained, and
        pinning per call costs milliseconds."""
        cache = self.__dict__.setdefault("_raw_cache", collections.OrderedDict())
        cur = torch.cuda.current_stream(self.device)
        hit = cache.get(id(scan))
        if hit is not None and hit[0] is scan:
            # the copy may still be in flight on ANOTHER stream (the window of frame t is uploaded on main, the look-ahead
            # window of frame t+1 re-uses two of its scans on the side stream, and the other way round when the look-ahead
            # misses): order this stream behind the copy and tell the allocator about the second user
            _, dev, done, owner = hit
            if owner != cur.cuda_stream:
                cur.wait_event(done)
                dev.record_stream(cur)
            return dev
        host = scan if torch.is_tensor(scan) else torch.from_numpy(np.ascontiguousarray(scan, dtype=np.float32))
        if host.is_cuda:
            return host
        if not host.is_pinned():
            ring = self.__dict__.setdefault("_raw_ring", {"slots": [], "next": 0})
            if len(ring["slots"]) < 8:
                ring["slots"].append([torch.empty(max(host.numel(), 4 * 160000), dtype=torch.float32).pin_memory(), None])
                slot = ring["slots"][-1]
            else:
                slot = ring["slots"][ring["next"] % 8]
                ring["next"] += 1
            if slot[1] is not None:
                slot[1].synchronize()                       # the copy that last used this buffer (8 uploads ago)
            if slot[0].numel() < host.numel():
                slot[0] = torch.empty(host.numel(), dtype=torch.float32).pin_memory()
            staged = slot[0][:host.numel()].view(host.shape)
            staged.copy_(host)
            dev = staged.to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(cur)
            done = slot[1]
        else:
            dev = host.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(cur)
        cache[id(scan)] = (scan, dev, done, cur.cuda_stream)
        while len(cache) > 16:
            cache.popitem(last=False)
        return dev

    def _build_raw(self, scans, poses):
        """H2D of the raw scans not yet on the device + device preprocessing, on the CURRENT stream."""
        dev_scans = [self._upload_raw(s) for s in scans]
        inv_cur = np.linalg.inv(np.asarray(poses[0], dtype=np.float64))
        # the current scan also goes through inv(P_cur) * P_cur, which is the identity only up to rounding
        # (the reference does exactly that, datasets/data_StreamMOS.py:424-467)
        built = self._pre.build(dev_scans, [inv_cur.dot(np.asarray(p, dtype=np.float64)) for p in poses])
        built["raw_scan"] = dev_scans[0]
        return built

    def step_raw(self, scans, poses, frame_point_num=160000, next_scans=None, next_poses=None):
        """One scan from RAW data: scans = list of T host (numpy) or device [n,4] float32 scans, current first;
        poses = their 4x4 poses.  Preprocessing runs on the device (streammos_amd.device_preprocess, row f1):
        only the raw scans cross PCIe and nothing synchronises with the host.  next_scans / next_poses: the following
        frame's window (pipeline mode): its upload, preprocessing and encoder run on the side stream beside this frame's
        decoder, and the next call (whose ``scans[0]`` must be the same object as this ``next_scans[0]``) picks them up."""
        from .device_preprocess import DevicePreprocessor
        if getattr(self, "_pre", None) is None or self._pre.N != frame_point_num:
            self._pre = DevicePreprocessor(self.device, frame_point_num=frame_point_num)
            self._raw_ahead = None
        ahead, self._raw_ahead = getattr(self, "_raw_ahead", None), None
        built = ahead[1] if ahead is not None and ahead[0] is scans[0] else self._build_raw(scans, poses)
        nxt = None
        if next_scans is not None and self.pipeline and not self.use_graph:
            main = torch.cuda.current_stream(self.device)
            if any(torch.is_tensor(t) and t.is_cuda for t in next_scans):
                self._side.wait_stream(main)            # device-resident scans: produced by whatever main has queued so far
            with torch.cuda.stream(self._side):
                nxt = self._build_raw(next_scans, next_poses)
                nxt["ready"] = torch.cuda.Event()
                nxt["ready"].record(self._side)
            for t in nxt.values():                      # allocated on the side stream, consumed on the main one later
                if torch.is_tensor(t):
                    t.record_stream(main)
            self._raw_ahead = (next_scans[0], nxt)
        self._last_built = built        # run_sequence checks its in_range_counts when it reads the labels back
        return self.step(built, poses[0], next_dev=nxt)

    def check_last_raw_sample(self):
        """The device preprocessing drops points beyond frame_point_num instead of raising (no sync on the hot path);
        call this where the labels are read back anyway to get the host path's error (preprocess.pad_scan)."""
        built = getattr(self, "_last_built", None)
        if built is not None and getattr(self, "_pre", None) is not None:
            self._pre.check_capacity(built)

    def _pipelined(self, dev, next_dev):
        """Two HIP streams: encode(t+1) on the side stream while frame t is decoded on the main one.  (Putting the serial
        [third BEV stage -> temporal fusion] chain on a third stream of its own was measured too: 144 vs 146 scans/s, so
        the simpler two-stream schedule stays.)"""
        eng = self.model._engine_for(dev["pcds_xyzi"])
        if eng is None:
            raise RuntimeError("StreamRunner(pipeline=True) needs the fused GPU engine (eval mode, fast_inference)")
        main = torch.cuda.current_stream(self.device)
        if self._pre_enc is not None and self._pre_enc[0] is dev:
            enc = self._pre_enc[1]
            main.wait_stream(self._side)               # encoder of this frame, issued during the previous step
        else:
            enc = eng.encode(dev["pcds_xyzi"], dev["pcds_coord"], dev["pcds_sphere_coord"],
                             n_live=dev.get("n_live") if self.skip_padding else None)
        self._pre_enc = None
        if next_dev is not None:                        # encoder of the NEXT frame, concurrent with this decode
            # the side stream must start behind the producers of next_dev.  Inputs made by upload() carry the event
            # recorded behind their H2D copies; anything else (e.g. device preprocessing on main) is ordered behind
            # main's whole tail.  Two encodes of one engine may run concurrently (frame 0: encode(t) on main beside
            # encode(t+1) here): every piece of engine scratch is keyed by stream (engine._block_ws, ops._stream_workspace).
            ready = next_dev.get("ready")
            if ready is not None:
                self._side.wait_event(ready)
            else:
                self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                nxt = eng.encode(next_dev["pcds_xyzi"], next_dev["pcds_coord"], next_dev["pcds_sphere_coord"],
                                 n_live=next_dev.get("n_live") if self.skip_padding else None)
            for t in list(nxt.values()) + [next_dev[k] for k in self._KEYS]:
                if torch.is_tensor(t):
                    t.record_stream(main)
                    t.record_stream(self._side)
            self._pre_enc = (next_dev, nxt)
        return eng.decode(enc, self.memory if self.frame > 0 else None, want_aux=False,       # the runner uses pred_cls only
                          n_live=dev.get("n_live") if self.skip_padding else None)

    @torch.no_grad()
    def step(self, dev, pose=None, next_dev=None):
        """One scan.  Returns dict(pred_cls, labels (N_pad,) uint8, raw_labels (n_raw,) uint8,
        voted = [(frame_id, int32 LUT labels)]).  next_dev: the following frame's inputs (pipeline mode)."""
        bf_labels = None
        if self.use_graph:
            if self._graphs is None or self._g_shape != tuple(dev["pcds_xyzi"].shape):
                self._capture(dev)
            pred_cls = self._replay(dev)        # static buffer: valid until the next step()
            labels = ops.tta_argmax(pred_cls)
            self.memory = [grp["mem"] for grp in self._groups]
        else:
            if self.pipeline:
                res = self._pipelined(dev, next_dev)
            else:
                eng = self.model._engine_for(dev["pcds_xyzi"])
                if eng is not None:
                    # the same kernels as the pipelined form, one stream: encode, then decode without the three aux heads the
                    # runner never reads (model.infer computes them: 3 GEMMs + 2 resizes per frame that the two-stream step does
                    # not run, so serial traces would not be traces of the timed step)
                    enc = eng.encode(dev["pcds_xyzi"], dev["pcds_coord"], dev["pcds_sphere_coord"],
                                     n_live=dev.get("n_live") if self.skip_padding else None)
                    res = eng.decode(enc, self.memory if self.frame > 0 else None, want_aux=False,
                                     n_live=dev.get("n_live") if self.skip_padding else None)
                else:
                    batch = {k: dev[k].unsqueeze(0) for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
                    res = self.model.infer(batch, self.frame, self.memory)
            pred_cls, self.memory = res[0], res[-1]          # 5-tuple (stage 1) or 6-tuple (StreamMOS_seg)
            labels = ops.tta_argmax(pred_cls)
            if len(res) == 6:
                bf_labels = ops.tta_argmax(res[1])            # val_StreamMOS_seg.py:101-103
        out = {"pred_cls": pred_cls, "labels": labels, "voted": []}
        if bf_labels is not None:
            out["bf_pred_cls"], out["bf_labels"] = res[1], bf_labels
        raw = None
        if "prefix" in dev:                                                       # device-preprocessed sample
            raw = self._pre.unpad_labels(labels, dev)
        elif "valid_index" in dev:
            raw = torch.zeros(dev["n_raw"], dtype=torch.uint8, device=self.device)
            raw.index_copy_(0, dev["valid_index"], labels[:dev["n_valid"]])      # val_StreamMOS.py:112-118
        if bf_labels is not None and "prefix" in dev:
            out["bf_raw_labels"] = self._pre.unpad_labels(bf_labels, dev)
        elif bf_labels is not None and "valid_index" in dev:     # the `_bf` prediction file of val_StreamMOS_seg.py:128-131,141
            bf_raw = torch.zeros(dev["n_raw"], dtype=torch.uint8, device=self.device)
            bf_raw.index_copy_(0, dev["valid_index"], bf_labels[:dev["n_valid"]])
            out["bf_raw_labels"] = bf_raw
        if raw is not None:
            out["raw_labels"] = raw
            if self.voter is not None and "raw_scan" in dev:
                pose = pose if pose is not None else np.eye(4)
                if isinstance(self.voter, InstanceVoter):
                    if "bf_raw_labels" not in out:
                        raise RuntimeError("instance voting needs the `_bf` prediction of the StreamMOS_seg model")
                    out["voted"] = self.voter.push(dev["raw_scan"], raw, pose, out["bf_raw_labels"])
                else:
                    out["voted"] = self.voter.push(dev["raw_scan"], raw, pose)
        self.frame += 1
        return out


class MultiStreamRunner:
    """S independent sequences advanced in lock step as ONE batch of S x V samples (BASELINE.json configs[2]:
    concurrent sequences on one GPU, every stream's recurrent memory and voting window resident in HBM).
    Streams never interact: the network is batch-independent, the TTA reduce and the voting run per stream."""

    def __init__(self, model, device="cuda:0", n_streams=8, vote=True, pipeline=False):
        """pipeline=True: as StreamRunner(pipeline=True) -- step(batched, poses, next_batched=...) runs the encoder of the NEXT batch
        on a second HIP stream beside the decoder of the current one, and the three aux heads nobody reads are not computed.
        Same kernels, same results as pipeline=False."""
        self.device = torch.device(device)
        self.model = model.to(self.device).eval()
        self.n = n_streams
        self.voters = [VoxelVoter(self.device) for _ in range(n_streams)] if vote else None
        self.memory = None
        self.frame = 0
        self.pipeline = pipeline
        self._side = concurrent_stream(self.device) if pipeline else None
        self._pre_enc = None

    @staticmethod
    def batch_inputs(devs):
        """list of per-stream uploads (StreamRunner.upload) -> one batched dict (concatenated along the TTA dim)."""
        out = {k: torch.cat([d[k] for d in devs], 0) for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")}
        out["streams"] = [{k: d[k] for k in d if k not in out} for d in devs]
        return out

    _KEYS = ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")

    def _pipelined(self, batched, next_batched):
        eng = self.model._engine_for(batched["pcds_xyzi"])
        if eng is None:
            raise RuntimeError("MultiStreamRunner(pipeline=True) needs the fused GPU engine (eval mode, fast_inference)")
        main = torch.cuda.current_stream(self.device)
        if self._pre_enc is not None and self._pre_enc[0] is batched:
            enc = self._pre_enc[1]
            main.wait_stream(self._side)
        else:
            enc = eng.encode(*(batched[k] for k in self._KEYS))
        self._pre_enc = None
        if next_batched is not None:
            self._side.wait_stream(main)          # behind whatever produced next_batched (batch_inputs runs on the main stream)
            with torch.cuda.stream(self._side):
                nxt = eng.encode(*(next_batched[k] for k in self._KEYS))
            for t in list(nxt.values()) + [next_batched[k] for k in self._KEYS]:
                if torch.is_tensor(t):
                    t.record_stream(main)
                    t.record_stream(self._side)
            self._pre_enc = (next_batched, nxt)
        return eng.decode(enc, self.memory if self.frame > 0 else None, want_aux=False)

    @torch.no_grad()
    def step(self, batched, poses, next_batched=None):
        if self.pipeline:
            res = self._pipelined(batched, next_batched)
        else:
            batch = {k: batched[k].unsqueeze(0) for k in self._KEYS}
            res = self.model.infer(batch, self.frame, self.memory)
        pred_cls, self.memory = res[0], res[-1]
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
