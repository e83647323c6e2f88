"""bench.py -- LiDAR scans/s of the full StreamMOS streaming-inference step on MI355X.

One "step" = one streamed scan through the whole hot path with its inputs already resident in HBM:
AttNet.infer at the reference's validation shape (B = 4 TTA variants, T = 3 stacked scans, N = 160 000
padded points, fp32) -> TTA softmax/mean/argmax -> scatter to the raw scan -> 8-frame voxel voting.
Workload = BASELINE.json configs[1] on synthetic 120k-point scans (no SemanticKITTI offline).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU, each rank streams its OWN sequence (sequence sharding, no collective on the data path) ->
"scaling": "weak".  Under torch.distributed.run (WORLD_SIZE / RANK in the environment) the ranks are the launcher's; a
plain ``python bench.py --gpus N`` starts the N ranks ITSELF (streammos_amd.launch.self_launch: the parent never touches
the GPU, the ranks are children of a torch.distributed.run child) -- the reference's README.md:97 launch line folded into
the program.  ``--dry-launch`` walks the same spawn path with gloo ranks that only join the group and time a barrier
(no GPU needed: the CPU test of the launch path).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA peak (no xf32/TF32 on gfx950)
ALG_TFLOP_PER_SCAN = 0.53        # 133 GFLOP/sample x 4 TTA samples
ALG_GB_PER_SCAN = 11.2           # 2.806 GB/sample x 4 TTA samples
FRAME_POINT_NUM = 160000         # config/StreamMOS.py:44 (Val.frame_point_num)


def algorithmic_bytes(label, ctx=None):
    """Bytes a kernel launch must move if every operand is touched exactly once (float32).  ctx: per-run facts that
    the label does not carry (``stem_rows``: mean number of occupied input cells = compact rows per launch)."""
    ctx = ctx or {}
    name, dims = label.split("[", 1)
    dims = dims.rstrip("]")
    if name == "voxel_maxpool_fwd":
        src, dst = dims.split("->")
        bs, c, n = (int(v) for v in src.split("x"))
        cells = int(np.prod([int(v) for v in dst.split("x")]))
        d = len(dst.split("x"))
        return 4 * (bs * c * n + bs * n * d + bs * c * cells)
    if name == "pointnet_scatter":
        # fused point_pre + input scatter: reads the 7-channel point features and the 2 used coordinate columns, writes
        # the t = 0 point features [B, N, 64] and the scatter target.  The engine launches the COMPACT form (one
        # T*64-float row per occupied cell): stem_rows x 768 B; the dense 805 MB grid of the
        # reference is never built and is not charged.  The dense form (SMOS_SPARSE_STEM=0) is charged the grid.
        src, dst = dims.split("->")
        b, t, n = (int(v) for v in src.split("x"))
        cells = int(np.prod([int(v) for v in dst.split("x")]))
        rows = ctx.get("stem_rows")
        target = rows * t * 64 if rows is not None else b * cells * t * 64
        return int(4 * (b * t * 7 * n + b * t * n * 2 + target + b * n * 64))
    if name in ("gather_scatter", "gather_scatter_cl"):
        src, n, dst = dims.split("->")
        b, c, h, w = (int(v) for v in src.split("x"))
        rows = dst.endswith("+pts")                     # the gathered point rows [B, N, C] are an output as well
        cells = int(np.prod([int(v) for v in dst.replace("+pts", "").split("x")]))
        return 4 * (b * c * h * w + 4 * b * int(n) + b * c * cells + (b * c * int(n) if cells == 0 or rows else 0))
    if name == "bilinear_gather":
        src, n = dims.split("->")
        b, c, h, w = (int(v) for v in src.split("x"))
        n = int(n)
        return 4 * (b * c * h * w + b * n * 2 + b * c * n)
    if name == "msda_fwd":
        n, lq, m, d = (int(v) for v in dims.split("x"))
        return 4 * (n * lq * m * d * 2 + n * lq * m * 4 * 3)
    if name == "point_head":
        b, n = (int(v) for v in dims.split("x"))
        return 4 * b * n * (192 + 3)
    if name == "upconv_xpass":
        # tap products z [B, Hs, Ws, 9 C] in, x-interpolated rows t [B, 3, Hs, Wo, C] out (csrc/upconv.hip)
        src, wo = dims.split("->")
        b, hs, ws, c = (int(v) for v in src.split("x"))
        return 4 * (b * hs * ws * 9 * c + b * 3 * hs * int(wo) * c)
    if name == "upconv_ypass":
        # direct-conv part in, the two sources' x-pass rows in (the network's geometry: half and quarter height), map out
        b, ho, wo, c = (int(v) for v in dims.split("x"))
        return 4 * (2 * b * ho * wo * c + b * 3 * (ho // 2 + ho // 4) * wo * c)
    if name == "upconv_xy":
        # both passes in one launch: the sources' tap products z [B, Hs, Ws, 9 C] and the direct-conv part in, the map out
        dst, srcs = dims.split("<-")
        b, ho, wo, c = (int(v) for v in dst.split("x"))
        cells = sum(int(hw.split("x")[0]) * int(hw.split("x")[1]) for hw in srcs.split("+"))
        return 4 * (2 * b * ho * wo * c + b * cells * 9 * c)
    if name == "downsample_pool_branch":
        # x in (once), the conv-branch result a in, the block's output out: the 1x1 branch map never leaves the CU (csrc/downsample.hip)
        geo, stride = dims.split("/s")
        b, c, h, w = (int(v) for v in geo.split("x"))
        s_ = int(stride)
        return 4 * (b * c * h * w + 2 * b * c * ((h - 1) // s_ + 1) * ((w - 1) // s_ + 1) + c * c)
    if name == "tfusion_layer":
        # sampled + query in, the layer's output (and the next layer's projection) out, the weight stream once (csrc/tfusion.hip)
        tokens, c, ffn, nq = _tfusion_layer_dims(dims)
        return 4 * (3 * tokens * c + tokens * nq + c * c + 2 * c * ffn + c * nq)
    if name == "tfusion_project":
        jobs = _tfusion_jobs(dims)
        return 4 * sum(t * c + t * o + c * o for t, c, o in jobs)
    if name == "conv_cl":
        # own implicit-GEMM conv (csrc/conv_igemm.hip): label B x Cin x H x W -> Cout x Ho x Wo k KHxKW [+res]
        geo = _conv_geometry(dims)
        res = geo["b"] * geo["cout"] * geo["ho"] * geo["wo"] if geo["res"] else 0
        return 4 * (geo["b"] * geo["cin"] * geo["h"] * geo["w"] + geo["b"] * geo["cout"] * geo["ho"] * geo["wo"] + res +
                    geo["cout"] * geo["cin"] * geo["kh"] * geo["kw"])
    if name == "stem_mark+scan":
        # occupancy flags + row table of the input grid (4 B per cell, twice) + the zero fill of the compact rows it creates
        b, h, w = (int(v) for v in dims.split("x"))
        return int(8 * b * h * w + ctx.get("stem_rows", 0) * 768)
    if name in ("stem_gemm", "stem_epilogue"):
        # sparse DownSample2D 192 -> 32, stride 2 on the occupied cells.  gemm: the compact rows in, the tap products Y out (a
        # cell of parity class c yields (taps_c + 1) x 32 floats: its conv taps + the 1x1 pool branch); epilogue: Y in, the
        # half-resolution map out.  (Until round 4 Y was charged to neither launch, which made the epilogue look like it
        # moved 4.1 x its bytes.)
        b, h, w, cin = (int(v) for v in dims.split("x"))
        rows = ctx.get("stem_rows", b * h * w)
        cls = ctx.get("stem_class_rows")
        y_bytes = 4 * 32 * (sum(r * (taps + 1) for r, taps in zip(cls, (1, 2, 2, 4))) if cls is not None else rows * 3.25)
        if name == "stem_gemm":
            return int(4 * rows * cin + y_bytes)
        return int(4 * b * (h // 2) * (w // 2) * 32 + y_bytes)
    return 0


def _tfusion_layer_dims(dims):
    """tfusion_layer[tokens x 128 x ffn (+qN)] -> (tokens, d_model, ffn, channels of the next projection)"""
    core, _, q = dims.partition("+q")
    tokens, c, ffn = (int(v) for v in core.split("x"))
    return tokens, c, ffn, int(q) if q else 0


def _tfusion_jobs(dims):
    """tfusion_project[T x 128 -> Cout, ...] -> [(tokens, cin, cout)]"""
    jobs = []
    for job in dims.split(","):
        src, cout = job.split("->")
        t, c = (int(v) for v in src.split("x"))
        jobs.append((t, c, int(cout)))
    return jobs


def _conv_geometry(dims):
    src, rest = dims.split("->")
    dst, rest = rest.split("k", 1)
    b, cin, h, w = (int(v) for v in src.split("x"))
    cout, ho, wo = (int(v) for v in dst.split("x"))
    res = rest.endswith("+res")
    kh, kw = (int(v) for v in rest.replace("+res", "").split("x"))
    return {"b": b, "cin": cin, "h": h, "w": w, "cout": cout, "ho": ho, "wo": wo, "kh": kh, "kw": kw, "res": res}


def algorithmic_flops(label, ctx=None):
    """FLOPs of a launch that runs on the matrix cores (0 = not such a kernel)."""
    name, dims = label.split("[", 1)
    dims = dims.rstrip("]")
    if name == "point_head":
        b, n = (int(v) for v in dims.split("x"))
        return 2 * b * n * (192 * 96 + 96 * 64 + 64 * 3)
    if name == "pointnet_scatter":
        # 7 -> 64 (the bias rides on an 8th input feature = 1) -> 64 per point, on v_mfma_f32_32x32x2_f32
        src, _ = dims.split("->")
        b, t, n = (int(v) for v in src.split("x"))
        return 2 * b * t * n * (8 * 64 + 64 * 64)
    if name == "conv_cl":
        geo = _conv_geometry(dims)
        return 2 * geo["b"] * geo["ho"] * geo["wo"] * geo["cout"] * geo["cin"] * geo["kh"] * geo["kw"]
    if name == "downsample_pool_branch":
        geo, _ = dims.split("/s")
        b, c, h, w = (int(v) for v in geo.split("x"))
        return 2 * b * h * w * c * c                    # the 1x1 conv at full resolution (the halo recompute is not algorithmic)
    if name == "tfusion_layer":
        tokens, c, ffn, nq = _tfusion_layer_dims(dims)
        return 2 * tokens * (c * c + 2 * c * ffn + c * nq)
    if name == "tfusion_project":
        return 2 * sum(t * c * o for t, c, o in _tfusion_jobs(dims))
    if name == "stem_gemm":
        # sparse DownSample2D 192 -> 32 on the occupied cells: a cell of parity class c feeds STEM_TAPS[c] conv taps plus the
        # 1x1 pool branch (csrc/stem.hip).  Needs the frames' occupancy; without it the launch is priced by its bytes only.
        b, h, w, cin = (int(v) for v in dims.split("x"))
        rows = (ctx or {}).get("stem_class_rows")
        if rows is None:
            return 0
        return int(sum(2.0 * r * cin * 32 * (taps + 1) for r, taps in zip(rows, (1, 2, 2, 4))))
    return 0


def event_bracket_overhead(device, n=64):
    """ms an (event, launch, event) bracket adds to the launch it times, measured on a device copy of ~30 us."""
    src = torch.empty(48 << 20, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    for _ in range(4):
        dst.copy_(src)
    torch.cuda.synchronize()
    each = []
    for _ in range(n):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        dst.copy_(src)
        eb.record()
        each.append((ea, eb))
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for _ in range(n):
        dst.copy_(src)
    eb.record()
    torch.cuda.synchronize()
    per = sorted(a.elapsed_time(b) for a, b in each)[n // 2]
    return max(0.0, per - ea.elapsed_time(eb) / n)


def calibrated_bracket_ms(device, bracketed_launch_ms):
    """The bracket overhead to subtract from a bracketed launch time: the smallest of three calibrations (the estimate errs
    upwards when the back-to-back copies of the reference run overlap their tails: 2.5 - 3.7 us on most boxes, 6.5 us on one
    where rocprofv3 says 3.1), and never more than a tenth of the launch."""
    est = min(event_bracket_overhead(device) for _ in range(3))
    return min(est, 0.1 * bracketed_launch_ms)


def executed_launch_flops(label, wino=True, ctx=None, family=None):
    """FLOPs one labelled launch issues on the matrix cores: the algorithmic count, except that a stride-1 3x3 convolution
    in the Winograd F(2x2, 3x3) form (family conv_wino) issues 4 instead of 9 multiply-adds per output and channel pair, a
    k x 3 / 3 x k one in the 1-D F(2, 3) form (conv_wino1d) two thirds of the direct count.  family = the kernel ops.py ran
    the label on (profiling.KernelTimer.family); without it the form is inferred from the label's shape as the engine would."""
    fl = algorithmic_flops(label, ctx)
    live = (ctx or {}).get("live_fraction")
    if live is not None and label.startswith(("point_head[", "pointnet_scatter[")):
        # the runner leaves the scans' padding tails out (StreamRunner(skip_padding=True)): tiles of padding points are not computed
        return int(fl * live)
    if family is not None:
        return fl * 4 // 9 if family == "conv_wino" else (fl * 2 // 3 if family == "conv_wino1d" else fl)
    if fl and wino and label.startswith("conv_cl["):
        geo = _conv_geometry(label.split("[", 1)[1].rstrip("]"))
        if geo["ho"] == geo["h"] and geo["wo"] == geo["w"] and geo["cin"] % 16 == 0 and geo["cout"] % 16 == 0 and "+res" not in label[-6:]:
            if (geo["kh"], geo["kw"]) in ((5, 3), (7, 3), (3, 5), (3, 7)):
                return fl * 2 // 3              # 1-D F(2, 3) along the 3-tap axis
        if (geo["kh"], geo["kw"]) == (3, 3) and geo["ho"] == geo["h"] and geo["wo"] == geo["w"] and geo["cin"] % 16 == 0 and geo["cout"] % 16 == 0:
            return fl * 4 // 9
    return fl


def family_table(summary, family, n_steps, bracket_ms, ctx):
    """Per kernel FAMILY (one kernel template: conv_wino, conv_wino1d, conv_igemm, point_head, ...), over the labelled launches
    of `n_steps` steps: launches and summed ms per step (the event bracket's own time taken off every launch), algorithmic
    and executed FLOPs and algorithmic bytes per step.  summary: KernelTimer.summary(); family: KernelTimer.family."""
    table = {}
    for label, (calls, total_ms, _) in summary.items():
        fam = family.get(label, label.split("[", 1)[0])
        row = table.setdefault(fam, {"launches": 0.0, "ms": 0.0, "alg_flops": 0.0, "exec_flops": 0.0, "alg_bytes": 0.0, "labels": {}})
        ms = max(total_ms - calls * min(bracket_ms, 0.1 * total_ms / calls), 0.0) / n_steps
        per = calls / n_steps
        row["launches"] += per
        row["ms"] += ms
        row["alg_flops"] += per * algorithmic_flops(label, ctx)
        row["exec_flops"] += per * executed_launch_flops(label, ctx=ctx, family=fam)
        row["alg_bytes"] += per * algorithmic_bytes(label, ctx)
        row["labels"][label] = ms
    return table


def dominant_family(table):
    """The own kernel family with the most time per step (families without a byte / FLOP model are not candidates)."""
    cand = {k: v for k, v in table.items() if v["alg_bytes"] > 0 or v["alg_flops"] > 0}
    return max(cand, key=lambda k: cand[k]["ms"]) if cand else None


def executed_flops(engine, b, n, t, stem_class_rows, live_fraction=1.0):
    """FLOPs the engine really executes on the matrix cores for one scan (batch of b TTA samples, n padded points, t stacked
    scans): walks the engine's own folded weights.  Differs from the reference's dense count (0.53 TFLOP) by the
    sparse first stage (only occupied cells, only the taps their parity class feeds), by conv_1 running as a conv on the fine
    map + tap GEMMs of the coarse maps at their own resolution, and by the stride-1 3x3 layers running in the Winograd
    F(2x2, 3x3) form (4 instead of 9 multiply-adds per output and channel pair)."""
    hb, wb = engine.bev_hw
    total = 2.0 * b * t * n * (8 * 64 + 64 * 64) * live_fraction               # point MLP (padding-tail tiles are skipped)
    total += sum(2.0 * r * 192 * 32 * (taps + 1) for r, taps in zip(stem_class_rows, (1, 2, 2, 4)))   # sparse stem

    wino = getattr(engine, "wino", False)

    def conv(w, px, stride=1):
        # stride-1 3x3 layers run in the Winograd F(2x2, 3x3) form (csrc/conv_wino.hip): 16 multiply-adds per 2x2 outputs and
        # channel pair = 4 per output instead of 9 (the transforms are additions on the vector unit and are not counted)
        taps = w.shape[2] * w.shape[3]
        if stride == 1 and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0:
            if wino and tuple(w.shape[2:]) == (3, 3):
                taps = 4
            elif getattr(engine, "wino1d", False) and tuple(w.shape[2:]) in ((5, 3), (7, 3), (3, 5), (3, 7)):
                taps = taps * 2 // 3            # 1-D F(2, 3) along the 3-tap axis (csrc/conv_wino1d.hip): 4 instead of 6 per output pair
        return 2.0 * b * px * w.shape[0] * w.shape[1] * taps

    def stage(blocks, h, w, first_sparse=False):
        nonlocal total
        for i, p in enumerate(blocks):
            if p.kind == "down":
                if not (first_sparse and i == 0):
                    total += conv(p.wa, (h // p.stride) * (w // p.stride), p.stride) + conv(p.wp, h * w)
                h, w = h // p.stride, w // p.stride
            elif p.kind == "unbalance":
                total += conv(p.wa, h * w) + conv(p.wb, h * w) + conv(p.wc, h * w)
            else:
                total += conv(p.w1, h * w) + conv(p.w2, h * w)
        return h, w

    sparse = engine.sparse_stem and engine.stem_w is not None
    if not sparse:
        total -= sum(2.0 * r * 192 * 32 * (taps + 1) for r, taps in zip(stem_class_rows, (1, 2, 2, 4)))
    h0, w0 = stage(engine.header_bev, hb, wb, first_sparse=sparse)
    stage(engine.header_rv, 32, 1024)
    h1, w1 = stage(engine.res1_bev, h0, w0)
    stage(engine.res1_rv, 16, 512)
    h2, w2 = stage(engine.res2, h1, w1)
    c = engine.query_embed.shape[1]
    for L in engine.layers:                                                      # temporal fusion: five linears per layer
        lin = c * c + c * L.qproj[0].shape[0] + c * c + 2 * c * L.lin1[0].shape[0]
        total += 2.0 * b * h2 * w2 * lin
    c0, cc1, cc2 = engine.conv_1a.shape[1], engine.conv_1z[0].cin, engine.conv_1z[1].cin
    co = engine.conv_1a.shape[0]
    if engine.upconv:
        total += conv(engine.conv_1a, h0 * w0) + 2.0 * b * (h1 * w1 * 9 * cc1 * co + h2 * w2 * 9 * cc2 * co)
    else:
        total += 2.0 * b * h0 * w0 * 9 * (c0 + cc1 + cc2) * co
    total += conv(engine.conv_2[0], h0 * w0)
    k = engine.aux[2]
    total += 2.0 * b * k * (h0 * w0 * c0 + h1 * w1 * cc1 + h2 * w2 * cc2)
    heads = 2 if engine.refine is not None else 1
    total += heads * 2.0 * b * n * (192 * 96 + 96 * 64 + 64 * k) * live_fraction
    return total


def pmc_traffic(label):
    """HBM bytes per launch of `label` from the committed rocprofv3 --pmc passes (profiles/pmc_summary.py;
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950), or None if not collected."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["per_launch"]
        return table[label]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic_family(labels_per_step):
    """Mean HBM bytes per launch over a family's launches of one step ({label: launches per step}); None unless every label
    of the family has a PMC row."""
    total = n = 0.0
    for label, per in labels_per_step.items():
        t = pmc_traffic(label)
        if t is None:
            return None
        total += per * t
        n += per
    return round(total / n) if n else None


def make_frames(n_frames, seq_seed, tta=True):
    """Host preprocessing of a synthetic sequence -> list of (sample, raw_scan, pose)."""
    from streammos_amd import preprocess, synth
    spec = preprocess.VoxelSpec()
    total = n_frames + 2
    base = seq_seed * 1000
    scans = [synth.synthetic_scan(base + k) for k in range(total)]
    poses = [synth.synthetic_pose(k) for k in range(total)]
    out = []
    for i in range(n_frames):
        idx = preprocess.window_indices(i, total, 3)
        s = preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], FRAME_POINT_NUM, spec, tta=tta)
        out.append((s, scans[i], poses[i]))
    return out


def host_cores(share=16):
    """Cores this process may really use: cgroup quota if one is set, else the affinity mask, capped at the
    per-GPU CPU share of the box (16) -- oversubscribing the shared 256-thread host makes torch-CPU crawl."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, share))


def cpu_baseline(frames, state_dict, n_timed):
    """The CPU restatement of the same path (oracle/, bit-/tolerance-pinned to the reference) on the host
    cores of this box: 1 warm-up + n_timed scans at the same shape."""
    from oracle import net_torch
    cores = host_cores()
    torch.set_num_threads(cores)
    net = net_torch.OracleNet(state_dict)
    memory = None
    t0 = None
    for i in range(n_timed + 1):
        s = frames[i % len(frames)][0]
        if i == 1:
            t0 = time.perf_counter()
        pred, _, _, _, memory = net.stage_forward(*(torch.from_numpy(s[k]) for k in
                                                    ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")), memory)
        net_torch.tta_labels(pred)
    dt = time.perf_counter() - t0
    return {"value": n_timed / dt, "unit": "scans/s", "cores": cores, "kind": "port",
            "sample": "%d warm-up + %d timed scans, B=4 TTA x T=3 x N=%d, torch-CPU fp32 restatement "
                      "(forward + TTA argmax, voting excluded)" % (1, n_timed, FRAME_POINT_NUM)}


class _WireCounter:
    """Counts the collectives a training step puts on the wire, where they are issued: torch.distributed's Python entry points
    (SyncBatchNorm's forward all_gather and backward all_reduce go through them, torch/nn/modules/_functions.py:65-84), DDP's
    gradient buckets through a comm hook that wraps the default all-reduce hook, and -- for everything issued from C++ (the
    used-parameter bitmap of find_unused_parameters=True) -- the process group's own sequence number, which the backend
    advances once per collective."""
    NAMES = ("all_gather_into_tensor", "all_gather", "all_reduce", "reduce", "broadcast", "barrier", "reduce_scatter_tensor")

    def __init__(self):
        import collections
        self.counts = collections.Counter()
        self.tag = None
        self._saved = {}

    def __enter__(self):
        import torch.distributed as dist
        for name in self.NAMES:
            fn = getattr(dist, name, None)
            if fn is None:
                continue
            self._saved[name] = fn

            def wrapped(*a, _fn=fn, _name=name, **kw):
                self.counts[self.tag or _name] += 1
                return _fn(*a, **kw)
            setattr(dist, name, wrapped)
        return self

    def __exit__(self, *exc):
        import torch.distributed as dist
        for name, fn in self._saved.items():
            setattr(dist, name, fn)
        return False

    def ddp_hook(self, state, bucket):
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        self.tag = "ddp_gradient_bucket_all_reduce"
        try:
            return default_hooks.allreduce_hook(state, bucket)
        finally:
            self.tag = None

    @staticmethod
    def sequence_number():
        import torch.distributed as dist
        try:
            return int(dist.distributed_c10d._get_default_group()._get_sequence_number_for_group())
        except Exception:
            return None


def train_bench(device, steps, warmup=1, frame_point_num=130000, batch=4, rehearsal=False):
    """BASELINE configs[4] (reported beside `value`, never as it): the stage-2 training step of
    train_StreamMOS_seg.py:58,130,143,165-190 -- StreamMOS_seg.AttNet, everything but `refine.*` frozen, SyncBatchNorm
    conversion + DistributedDataParallel(find_unused_parameters=True), model.train(), three chained forwards per step
    (models/StreamMOS_seg.py:173-196), backward, SGD step -- at the reference's training shape (batch_size_per_gpu 4,
    Train.frame_point_num 130000, config/StreamMOS_seg.py:5,27) on synthetic scans and seeded targets.

    Runs on the process group that exists: under `bench.py --gpus N` all N ranks (one RCCL group, 'nccl') train as real DDP --
    every rank its own samples of the synthetic set (DistributedSampler-style: rank r of W takes samples r, r + W, ...),
    gradients all-reduced, statistics of every SyncBatchNorm layer all-gathered -- and the step is timed between barriers, MAX
    over ranks.  Without a group: a one-rank group is created (a one-GPU box).  The collectives are COUNTED ON THE WIRE
    (_WireCounter), not inferred.  rehearsal=True (bench.py --dry-launch): the same code path on CPU tensors over gloo at a
    reduced shape, with the GPU-only sampler swapped for the reference's debug torch formulation
    (deformattn/functions/ms_deform_attn_func.py:41-61, as deformattn/test.py does) -- a test of the launch and the group
    logic, not a measurement.

    Test knobs (tests/test_gpu_launch.py): SMOS_BENCH_TRAIN_BATCH=<npz> feeds every rank that file's batch dict (REPLICATED:
    the stage-2 loss -- OHEM top-k + Lovasz -- is a set function of the batch, not a per-sample mean, so only a rank that sees
    the whole pinned batch can reproduce the pinned loss; DDP then averages identical gradients), SMOS_BENCH_TRAIN_EVAL=1 runs
    the step in eval mode (the mode the reference fixture tests/golden/training.npz pins stage 2 in);
    SMOS_BENCH_TRAIN_POINTS / SMOS_BENCH_TRAIN_BATCH_PER_GPU shrink the synthetic shape."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from streammos_amd import preprocess, synth
    from streammos_amd.refapi.config import StreamMOS_seg as cfg
    from streammos_amd.refapi.models import StreamMOS_seg
    on_gpu = device.type == "cuda"
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        if on_gpu:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group("gloo", rank=0, world_size=1)
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    fixture = os.environ.get("SMOS_BENCH_TRAIN_BATCH")
    eval_mode = os.environ.get("SMOS_BENCH_TRAIN_EVAL") == "1"
    frame_point_num = int(os.environ.get("SMOS_BENCH_TRAIN_POINTS", frame_point_num))       # test knobs: a smaller shape
    batch = int(os.environ.get("SMOS_BENCH_TRAIN_BATCH_PER_GPU", batch))
    if rehearsal:
        frame_point_num, batch = 512, 1
        import streammos_amd.refapi.deformattn._msda as _msda_mod
        from streammos_amd.refapi.deformattn.functions import ms_deform_attn_func as _fn

        class _TorchSampler:
            @staticmethod
            def apply(value, shapes, lsi, loc, attn, step):
                return _fn.ms_deform_attn_core_pytorch(value, shapes, loc, attn)
        saved_sampler, _msda_mod.MSDeformAttnFunction = _msda_mod.MSDeformAttnFunction, _TorchSampler
    wire = _WireCounter()
    try:
        net = StreamMOS_seg.AttNet(cfg.get_config()[2])
        net.load_state_dict(synth.seeded_state_dict(net.state_dict()), strict=True)
        trainable = StreamMOS_seg.freeze_for_stage2(net)
        n_bn = sum(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in net.modules())
        if on_gpu:               # torch's SyncBatchNorm exists for GPU modules only: the CPU rehearsal keeps per-rank BatchNorm
            net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
        net = net.to(device)
        if on_gpu:
            model = DistributedDataParallel(net, device_ids=[device.index], output_device=device.index, find_unused_parameters=True)
        else:
            model = DistributedDataParallel(net, find_unused_parameters=True)
        model.register_comm_hook(None, wire.ddp_hook)
        opt = torch.optim.SGD(trainable, lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-3)
        data = {}
        if fixture:
            with np.load(fixture) as z:
                data = {k: torch.from_numpy(z[k]).to(device) for k in z.files}
            batch = int(data["pcds_xyzi_0"].shape[0])
            frame_point_num = int(data["pcds_xyzi_0"].shape[3])
            warmup = 0                          # the pinned numbers are those of the FIRST step from the seeded weights
        else:
            spec = preprocess.VoxelSpec()
            # sample g of the synthetic training set = three consecutive windows of a 5-scan sequence seeded 7000 + 10 g;
            # rank r takes sample r (+ W, + 2 W, ... for the following steps would be the sampler's job; the bench repeats
            # one batch per rank, what the reference's DataLoader hands over changes no shape and no collective)
            base = 7000 + 10 * rank
            if rehearsal:
                mk = lambda k: synth.synthetic_scan(base + k, 8, 40)
            elif frame_point_num < 100000:                 # test knob SMOS_BENCH_TRAIN_POINTS: scans that fit the small frame
                mk = lambda k: synth.synthetic_scan(base + k, 16, 120)
            else:
                mk = lambda k: synth.synthetic_scan(base + k)
            scans = [mk(k) for k in range(5)]
            poses = [synth.synthetic_pose(k) for k in range(5)]
            gen = torch.Generator(device="cpu").manual_seed(11 + rank)
            for i in range(3):                                     # three consecutive samples, chained through the memory
                idx = preprocess.window_indices(i, 5, 3)
                smp = preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], frame_point_num, spec, tta=True)
                for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord"):
                    data["%s_%d" % (k, i)] = torch.from_numpy(np.ascontiguousarray(smp[k][:batch])).to(device)
                data["pcds_target_%d" % i] = torch.randint(0, 3, (batch, frame_point_num, 1), generator=gen).to(device)
                data["pcds_bev_target_%d" % i] = torch.randint(0, 3, (batch, 256, 256, 1), generator=gen).to(device)
                data["pcds_bf_target_%d" % i] = torch.randint(0, 3, (batch, frame_point_num, 1), generator=gen).to(device)
        model.train(not eval_mode)

        def step():
            loss = model(data)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss

        def fence():
            if on_gpu:
                torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                if on_gpu:
                    torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        fence()
        seq0 = wire.sequence_number()
        with wire:
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            if on_gpu:
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        seq1 = wire.sequence_number()
        fence()
        grad_norms = {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}
        t = torch.tensor([dt, float(loss.detach())], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        tmax = t.clone()
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)               # what train_StreamMOS_seg.py:31-42,69 logs: the mean loss
        dt_max, loss_mean = float(tmax[0]), float(t[1]) / world
        py = {k: v / steps for k, v in sorted(wire.counts.items())}
        res = {"value": round(world * batch * steps / dt_max, 3),
               "unit": "samples/s aggregate over %d rank(s) (a sample = 3 chained forwards + backward)" % world,
               "world": world, "backend": "%s%s" % (backend, " (= RCCL)" if backend == "nccl" else ""),
               "ms_per_step": round(1e3 * dt_max / steps, 2), "steps": steps, "batch_per_gpu": batch, "frame_point_num": frame_point_num,
               "loss": round(loss_mean, 6), "loss_rank0": round(float(loss.detach()), 6), "batchnorm_layers": n_bn,
               "trainable_tensors": len(trainable), "mode": "eval" if eval_mode else "train",
               "collectives_per_step_on_the_wire": dict(
                   py, process_group_sequence_numbers=None if seq0 is None or seq1 is None else (seq1 - seq0) / steps,
                   note="counted where issued during the timed steps of rank 0: torch.distributed's Python entry points "
                        "(SyncBatchNorm: one all_gather of (mean, invstd, count) per layer and forward when world > 1, two-vector "
                        "all_reduce per backward whose input needs a gradient), DDP gradient buckets through a comm hook; "
                        "process_group_sequence_numbers = the backend's own collective counter over the same steps (adds what "
                        "C++ issues: DDP's used-parameter bitmap all-reduce)"),
               "grad_norms": {k: round(v, 8) for k, v in grad_norms.items()},
               "data": "fixture %s replicated on every rank" % os.path.basename(fixture) if fixture else
                       "synthetic, per-rank samples (rank r seeds 7000 + 10 r)",
               "note": "BASELINE configs[4]: stage-2 step (all but refine.* frozen) under SyncBatchNorm + DDP, module graph with torch "
                       "autograd (MIOpen convs; VoxelMaxPool / deformable-attention forward and backward on the HIP kernels); reported "
                       "beside `value`, never as it"}
        if on_gpu:
            res["hbm_allocated_gb"] = round(torch.cuda.max_memory_allocated(device) / 1e9, 2)
        if rehearsal:
            res["rehearsal"] = ("CPU tensors over gloo, N = %d, batch %d, torch formulation of the sampler, per-rank BatchNorm (torch's "
                                "SyncBatchNorm is GPU-only): launch-path test, not a measurement" % (frame_point_num, batch))
        return res
    finally:
        if rehearsal:
            _msda_mod.MSDeformAttnFunction = saved_sampler
        if own_group:
            dist.destroy_process_group()


def family_roofline(model, device, dev_frames, args, ctx, one_step):
    """The bench line's `roofline`: the dominant KERNEL of the step, chosen from SERIAL steady-state steps.

    Eight steps of StreamRunner(pipeline=False) (one stream, the two pipeline stages one after the other) run right after the
    timed region with a HIP-event bracket around every labelled launch, on the stream the launch goes to.  The launches are
    summed per kernel family (= kernel template: the 36 Winograd launches of a step carry 14 labels but are ONE kernel); the
    family with the most time per step is the dominant kernel.  A serial step is also what `rocprofv3 --kernel-trace` of
    `bench.py --no-pipeline` measures, so the figures can be checked against profiles/rNN_label_durations.csv family by family.
    (In the timed two-stream region a launch shares the CUs with the other stage's kernels, which stretches it by up to 2x -- a
    property of the pipelining, not of the kernel; that region carries no event brackets.)

    `frac`: an MFMA-bound family is priced on the FLOPs it EXECUTES on the matrix cores (a Winograd launch issues 4/9 of the
    direct count); `algorithmic_frac` beside it counts the direct-form FLOPs and may exceed 1."""
    from streammos_amd import profiling, streaming
    serial = streaming.StreamRunner(model, device, vote=not args.no_vote, pipeline=False, skip_padding=not args.no_skip_padding)
    n_ser = 8
    for i in range(2):
        serial.step(*dev_frames[i % len(dev_frames)])
    torch.cuda.synchronize()
    with profiling.kernel_timer() as kt_ser:
        for i in range(n_ser):
            serial.step(*dev_frames[(2 + i) % len(dev_frames)])
    ser = kt_ser.summary()
    serial.close()
    del serial
    if not ser:
        return None, None
    # what an (event, launch, event) bracket adds to a launch (the event packets' own processing; ~7 % of a 40 us launch): a
    # ~30 us device copy 64 times, every launch in its own bracket vs. all 64 in ONE bracket.  Subtracted (never more than a
    # tenth of a launch), and reported.
    bracket_ms = min(event_bracket_overhead(device) for _ in range(3))
    table = family_table(ser, kt_ser.family, n_ser, bracket_ms, ctx)
    fam = dominant_family(table)
    families = {k: {"launches_per_step": round(v["launches"], 2), "ms_per_step": round(v["ms"], 4)}
                for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"])}
    if fam is None:
        return None, families
    row = table[fam]
    sec = row["ms"] * 1e-3
    t_hbm, t_mfma = row["alg_bytes"] / (HBM_PEAK_GBS * 1e9), row["exec_flops"] / (FP32_PEAK_TFLOPS * 1e12)
    n = row["launches"]
    labels_per_step = {l: ser[l][0] / n_ser for l in row["labels"]}
    roof = {"kernel": fam, "launches_per_step": round(n, 2), "ms_per_step": round(row["ms"], 4),
            "avg_launch_ms": round(row["ms"] / n, 5), "labels": len(row["labels"]),
            "algorithmic_flops_per_launch": round(row["alg_flops"] / n), "executed_flops_per_launch": round(row["exec_flops"] / n),
            "algorithmic_bytes_per_launch": round(row["alg_bytes"] / n), "traffic": pmc_traffic_family(labels_per_step),
            "hbm_floor_ms_per_step": round(1e3 * t_hbm, 4), "mfma_floor_ms_per_step": round(1e3 * t_mfma, 4),
            "event_bracket_ms": round(bracket_ms, 5), "serial_steps": n_ser,
            "share_of_serial_labelled_ms": round(row["ms"] / max(sum(v["ms"] for v in table.values()), 1e-9), 4),
            "clock": "HIP events (torch.cuda.Event) on the launch stream around every labelled launch of %d serial steps "
                     "(StreamRunner(pipeline=False): one stream) right after the timed region, summed per kernel family; the "
                     "bracket's own time (event_bracket_ms) taken off every launch = what rocprofv3 --kernel-trace of `bench.py "
                     "--no-pipeline` reports (profiles/rNN_label_durations.csv, family rows)" % n_ser}
    if t_mfma > t_hbm:
        ach = row["exec_flops"] / sec / 1e12
        roof.update({"bound": "mfma", "achieved": round(ach, 1), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / FP32_PEAK_TFLOPS, 4),
                     "algorithmic_frac": round(row["alg_flops"] / sec / 1e12 / FP32_PEAK_TFLOPS, 4),
                     "frac_counts": "executed FLOPs (what the matrix cores issue); algorithmic_frac = direct-form FLOPs"})
    else:
        ach = row["alg_bytes"] / sec / 1e9
        roof.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)})
    if ctx.get("stem_rows") is not None:
        roof["stem_rows_per_launch"] = round(ctx["stem_rows"])
    # per label of the family (serial steps): where inside the family the time goes
    roof["by_label"] = {l: {"launches_per_step": round(labels_per_step[l], 2), "ms_per_launch": round(ms / labels_per_step[l], 5),
                            "frac": round((executed_launch_flops(l, ctx=ctx, family=fam) / FP32_PEAK_TFLOPS / 1e12 if roof["bound"] == "mfma"
                                           else algorithmic_bytes(l, ctx) / HBM_PEAK_GBS / 1e9) / max(ms / labels_per_step[l] * 1e-3, 1e-12), 4)}
                        for l, ms in sorted(row["labels"].items(), key=lambda kv: -kv[1])}
    # and the family's largest label re-run 30 times back to back with its own operands (inputs and weights hot in L2 /
    # Infinity Cache): an upper bound of what the kernel does, reported beside `frac`, never as it
    top = max(row["labels"], key=lambda l: row["labels"][l])
    if top.startswith("conv_cl"):
        profiling.request_replay(top)
        one_step(0)
        one_step(1)
        again = profiling.replay_of(top)
        if again is not None:
            torch.cuda.synchronize()
            with profiling.kernel_timer(only=top) as kt_iso:
                for _ in range(30):
                    again()
            hot = kt_iso.summary()[top][2]
            work = executed_launch_flops(top, ctx=ctx, family=fam) / FP32_PEAK_TFLOPS / 1e12 if roof["bound"] == "mfma" \
                else algorithmic_bytes(top, ctx) / HBM_PEAK_GBS / 1e9
            roof["hot_replay"] = {"label": top, "launch_ms": round(hot, 4), "frac": round(work / (hot * 1e-3), 4)}
    return roof, families


def dry_launch(args):
    """The launch path without the GPU: every rank joins a gloo group, the barrier-bracketed "timed region" is K
    barriers, the MAX over ranks is taken as in the real run and rank 0 prints the line with the group's real size.
    With --train-steps K the ranks also run K stage-2 DDP steps on CPU tensors through the same train_bench the GPU run uses
    (rehearsal shape), and rank 0 reports `stage2_training` with the world size of the group that trained."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    elapsed = 0.0
    training = None
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, world = float(t.item()), dist.get_world_size()
    if args.train_steps:
        training = train_bench(torch.device("cpu"), args.train_steps, warmup=0, rehearsal=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        line = {"metric": "dry launch (no GPU work)", "value": None, "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "dry_launch": True, "barrier_ms": round(1e3 * elapsed / max(args.steps, 1), 4),
                "config": {"parallelism": "sequence-shard x%d" % world}}
        if training is not None:
            line["stage2_training"] = training
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=10, help="distinct preprocessed scans cycled through")
    ap.add_argument("--no-vote", action="store_true")
    ap.add_argument("--layout", default="cl", choices=["cl", "nchw"], help="feature-map layout of the fused engine")
    ap.add_argument("--miopen-search", action="store_true",
                    help="torch.backends.cudnn.benchmark = True: MIOpen measures every solver once per conv shape")
    ap.add_argument("--graph", action="store_true", help="replay the forward as a captured hipGraph")
    ap.add_argument("--split", type=int, default=1, help="with --graph: TTA groups replayed concurrently on HIP streams")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap the encoder of frame t+1 with the decoder of frame t")
    ap.add_argument("--no-raw", action="store_true", help="skip the PCIe-inclusive raw-scan leg (profiling runs: keeps the trace to "
                                                         "the headline step)")
    ap.add_argument("--cpu-scans", type=int, default=4, help="timed scans of the CPU baseline (0 = skip); 4 scans = about 15 s")
    ap.add_argument("--streams", type=int, default=0,
                    help="also time S concurrent sequences batched on the GPU (BASELINE configs[2]); reported beside value")
    ap.add_argument("--train-steps", type=int, default=None,
                    help="timed stage-2 DDP training steps on all ranks (BASELINE configs[4]; 0 = skip; default 3, 0 with "
                         "--dry-launch); reported beside value")
    ap.add_argument("--no-skip-padding", action="store_true",
                    help="compute the point head on the padding tail of every scan too (A/B; the runner's default leaves those "
                         "logits at zero: val_StreamMOS.py:113 cuts them off)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="only rehearse the rank launch: gloo ranks join the group, time a barrier, rank 0 prints n_gpus")
    ap.add_argument("--label-log", default=None,
                    help="write the launch-ordered list of kernel labels of one step to this JSON file (used by "
                         "profiles/pmc_summary.py to tell the conv launches of a rocprofv3 --pmc pass apart; needs --no-pipeline)")
    args = ap.parse_args()

    from streammos_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # no launcher around us: become the launcher.  Nothing above this line touches the GPU (importing torch does not).
        sys.exit(launch.self_launch(args.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    if args.train_steps is None:
        args.train_steps = 0 if args.dry_launch else 3
    if args.dry_launch:
        return dry_launch(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (not used by the driver): SMOS_BENCH_BACKEND=gloo + SMOS_BENCH_ONE_DEVICE=1 let several ranks
    # share the single GPU of a development box to exercise the multi-rank code path
    backend = os.environ.get("SMOS_BENCH_BACKEND", "nccl")
    if os.environ.get("SMOS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()            # the ranks that really joined the group
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.miopen_search:
        torch.backends.cudnn.benchmark = True
    from streammos_amd import profiling, streaming, synth
    from streammos_amd.refapi.config import StreamMOS as cfg
    from streammos_amd.refapi.models import StreamMOS

    model = StreamMOS.AttNet(cfg.get_config()[2])
    state = synth.seeded_state_dict(model.state_dict())
    model.load_state_dict(state, strict=True)
    model.engine_layout = args.layout
    runner = streaming.StreamRunner(model, device, vote=not args.no_vote, graph=args.graph, split=args.split,
                                    pipeline=not (args.no_pipeline or args.graph), skip_padding=not args.no_skip_padding)

    frames = make_frames(args.frames, seq_seed=rank)
    dev_frames = [(runner.upload(s, raw), pose) for s, raw, pose in frames]

    def one_step(i):
        d, pose = dev_frames[i % len(dev_frames)]
        return runner.step(d, pose, next_dev=dev_frames[(i + 1) % len(dev_frames)][0])

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # first a few steps with HIP-event brackets around every labelled launch (reported as hip_kernel_ms_per_step_warmup).  The TIMED region
    # carries no instrumentation at all -- brackets around the 36 launches of the dominant family cost 0.3 ms of host time per
    # step and 3 % of `value` when they were tried; the roofline's kernel is chosen and timed afterwards, in serial steady-state
    # steps (family_roofline).
    n_look = min(max(args.warmup, 1), 5)
    with profiling.kernel_timer() as kt:
        for i in range(n_look):               # extra, instrumented steps: a look at the launches, not part of W
            one_step(i)
    warm = kt.summary()
    # the W untimed warm-up steps proper, exactly as the timed ones (no instrumentation): the timed region then starts on a GPU
    # that has been busy up to the synchronize in front of it (reading the brackets back leaves it idle for milliseconds)
    for i in range(args.warmup):
        one_step(i)

    if args.label_log and rank == 0:
        with profiling.kernel_timer() as kt_seq:
            one_step(args.warmup)
        torch.cuda.synchronize()
        json.dump(kt_seq.sequence, open(args.label_log, "w"))

    sync()
    t0 = time.perf_counter()
    half = args.steps // 2
    e_mid, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(args.steps):
        if i == half:
            e_mid.record()                 # on the runner's main stream, behind step `half - 1`
        one_step(args.warmup + i)
    e_end.record()
    enqueue = time.perf_counter() - t0     # host time to issue all steps (the GPU may still be running)
    sync()
    elapsed = time.perf_counter() - t0
    # the second half of the timed region alone: the shader clock is still ramping through the first steps of a short run
    # (2.09 -> 2.38 GHz, tools/conv_stamps.py); reported beside ms_per_step, which stays the whole region
    second_half_ms = e_mid.elapsed_time(e_end) / (args.steps - half) if args.steps - half > 0 and half > 0 else None

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = None
    if rank == 0:
        value = world * args.steps / elapsed
        eng = getattr(model, "_engine", None)
        ctx, stem_class_rows = {}, (0, 0, 0, 0)
        if eng is not None and eng.sparse_stem and eng.stem_w is not None:
            # occupancy of the frames the timed region cycled through (device-side counts of the stem plan, read here,
            # outside the timed region): rows per launch of the compact scatter target / the sparse first stage
            from streammos_amd import ops as _ops
            metas = [_ops.stem_plan(d["pcds_coord"], *eng.bev_hw).meta.cpu().numpy() for d, _ in dev_frames]
            stem_class_rows = tuple(float(np.mean([m[8 + c] - m[4 + c] for m in metas])) for c in range(4))
            ctx["stem_rows"] = float(np.mean([m[11] for m in metas]))
            ctx["stem_class_rows"] = stem_class_rows
        if not args.no_skip_padding:
            ctx["live_fraction"] = float(np.mean([d["n_valid"] for d, _ in dev_frames])) / FRAME_POINT_NUM
        roof, families = family_roofline(model, device, dev_frames, args, ctx, one_step)
        exec_tflop = None
        if eng is not None and eng.layout == "cl":
            d0 = dev_frames[0][0]
            bs, t, _, n = d0["pcds_xyzi"].shape[:4]
            exec_tflop = executed_flops(eng, bs, n, t, stem_class_rows, ctx.get("live_fraction", 1.0)) / 1e12
        line = {
            "metric": "LiDAR scans/sec (StreamMOS streaming inference + voxel voting)",
            "value": round(value, 3), "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: full StreamMOS streaming inference, 1 sequence per GPU, B=4 TTA x T=3 "
                                   "x N=160000 padded points (120k-point synthetic HDL-64E scans), 8-frame voxel "
                                   "voting %s" % ("off" if args.no_vote else "on"),
                       "tta": 4, "frame_point_num": FRAME_POINT_NUM, "parallelism": "sequence-shard x%d" % world,
                       "real_points_per_scan": round(float(np.mean([d["n_valid"] for d, _ in dev_frames]))),
                       "padding_tail": "point head skipped (logits zero)" if not args.no_skip_padding else "computed"},
            "roofline": roof,
            "path_roofline": {"bound": "hbm", "achieved": round(value / world * ALG_GB_PER_SCAN, 1), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(value / world * ALG_GB_PER_SCAN / HBM_PEAK_GBS, 4),
                              "note": "scans/s/GPU x 11.2 GB algorithmic bytes per scan (SURVEY.md 8d)"},
            "path_compute_roofline": (None if exec_tflop is None else
                                      {"bound": "mfma", "achieved": round(value / world * exec_tflop, 1),
                                       "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(value / world * exec_tflop / FP32_PEAK_TFLOPS, 4),
                                       "executed_tflop_per_scan": round(exec_tflop, 4),
                                       "dense_equivalent_tflops": round(value / world * ALG_TFLOP_PER_SCAN, 1),
                                       "note": "scans/s/GPU x the FLOPs the engine really executes per scan (bench.executed_flops "
                                               "walks the engine's layers and the frames' occupancy) against the fp32 MFMA = "
                                               "vector peak; dense_equivalent = the reference's 0.53 TFLOP/scan x scans/s, kept for "
                                               "comparison only (the sparse first stage and the restructured conv_1 skip work)"}),
            "host_enqueue_ms_per_step": round(1e3 * enqueue / args.steps, 3),
            "second_half_ms_per_step": None if second_half_ms is None else round(second_half_ms, 3),
            "stem_rows_per_launch": None if ctx.get("stem_rows") is None else round(ctx["stem_rows"]),
            "stem_class_rows": [round(r) for r in stem_class_rows],
            "live_fraction": None if ctx.get("live_fraction") is None else round(ctx["live_fraction"], 4),
            "kernel_families_serial": families,
            "hip_kernel_ms_per_step_warmup": {k: round(v[1] / n_look, 4) for k, v in sorted(warm.items())},
        }
        if world == 1 and not args.no_raw:
            # PCIe-inclusive variant (never `value`): raw scans uploaded every step, preprocessing on the device
            raw_runner = streaming.StreamRunner(model, device, vote=not args.no_vote, pipeline=not args.no_pipeline,
                                                skip_padding=not args.no_skip_padding)
            from streammos_amd import synth as _synth
            raw = [(_synth.synthetic_scan(k), _synth.synthetic_pose(k)) for k in range(6)]
            def raw_window(i):
                idx = [(i + 2) % 4 + 2 - j for j in range(3)]
                return [raw[j][0] for j in idx], [raw[j][1] for j in idx]

            def raw_step(i):
                (scans, poses), (nscans, nposes) = raw_window(i), raw_window(i + 1)
                raw_runner.step_raw(scans, poses, FRAME_POINT_NUM, next_scans=nscans, next_poses=nposes)
            for i in range(3):
                raw_step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(20):
                raw_step(i)
            torch.cuda.synchronize()
            line["raw_scan_pipeline"] = {"value": round(20 / (time.perf_counter() - t1), 3), "unit": "scans/s",
                                         "note": "H2D of 3 raw scans (5.8 MB) + device preprocessing (both on the side stream, one frame ahead) + the same step; "
                                                 "PCIe-inclusive, reported beside `value`, never as `value`"}
        if world == 1 and args.streams > 1:
            # configs[2]: S concurrent sequences advanced in lock step as one batch of 4*S samples
            S = args.streams
            ms = streaming.MultiStreamRunner(model, device, n_streams=S, vote=not args.no_vote, pipeline=not args.no_pipeline)
            per_stream = [make_frames(4, seq_seed=100 + q) for q in range(S)]
            batched = []
            for f in range(4):
                devs = [runner.upload(per_stream[q][f][0], per_stream[q][f][1]) for q in range(S)]
                batched.append((ms.batch_inputs(devs), [per_stream[q][f][2] for q in range(S)]))
            def ms_step(i):
                nxt = batched[(i + 1) % 4][0] if ms.pipeline else None      # the next batch's encoder beside this batch's decoder
                return ms.step(*batched[i % 4], next_batched=nxt)
            for i in range(3):
                ms_step(i)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            n_it = 10
            for i in range(n_it):
                ms_step(3 + i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t2
            line["batched_streams"] = {"streams": S, "value": round(S * n_it / dt, 3), "unit": "scans/s",
                                       "ms_per_batched_step": round(1e3 * dt / n_it, 3),
                                       "hbm_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
                                       "note": "BASELINE configs[2]; reported beside `value`, never as `value`"}

    if args.train_steps > 0:
        # configs[4]: every rank of the group trains (real DDP over RCCL when world > 1); rank 0 reports
        runner = dev_frames = None            # the inference buffers are not needed any more
        torch.cuda.empty_cache()
        guard = None
        if rank == 0 and world > 1:
            # never lose the headline line to the side measurement: an exception is caught below, but a rank that dies inside
            # the DDP step leaves the others waiting in a collective until the backend's own timeout (10 minutes and an abort).
            # Rank 0 then reports what it has and leaves.
            import threading
            limit = float(os.environ.get("SMOS_BENCH_TRAIN_LIMIT_S", "420"))

            def give_up():
                line["stage2_training"] = {"error": "no result within %.0f s (a rank stuck in a collective?); the inference "
                                                    "figures above were complete before the training step started" % limit}
                print(json.dumps(line), flush=True)
                os._exit(0)
            guard = threading.Timer(limit, give_up)
            guard.daemon = True
            guard.start()
        try:
            res = train_bench(device, args.train_steps)
        except Exception as e:
            res = {"error": repr(e)[:300]}
        if guard is not None:
            guard.cancel()
        if rank == 0:
            line["stage2_training"] = res
    if world > 1:
        import torch.distributed as dist
        if rank == 0 and "error" in (line.get("stage2_training") or {}):
            # a failed training step may have left other ranks inside a collective: tearing the group down could wait for them
            print(json.dumps(line), flush=True)
            os._exit(0)
        dist.destroy_process_group()
    if rank == 0:
        if args.cpu_scans > 0:
            # rank 0 alone, after the group is gone (the other ranks have left: the host cores are free)
            line["cpu_baseline"] = cpu_baseline(frames, state, args.cpu_scans)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
