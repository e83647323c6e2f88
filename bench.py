"""bench.py -- LiDAR scans/s of the full StreamMOS streaming-inference step on MI355X.

One "step" = one streamed scan through the whole hot path with its inputs already resident in HBM:
AttNet.infer at the reference's validation shape (B = 4 TTA variants, T = 3 stacked scans, N = 160 000
padded points, fp32) -> TTA softmax/mean/argmax -> scatter to the raw scan -> 8-frame voxel voting.
Workload = BASELINE.json configs[1] on synthetic 120k-point scans (no SemanticKITTI offline).

    python bench.py --gpus N --steps K --warmup W

N > 1: launched under torch.distributed.run, one rank per GPU, each rank streams its OWN sequence
(sequence sharding, no collective on the data path) -> "scaling": "weak".
Prints ONE JSON line on rank 0.
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


def algorithmic_bytes(label):
    """Bytes a kernel launch must move if every operand is touched exactly once (float32)."""
    name, dims = label.split("[", 1)
    dims = dims.rstrip("]")
    if name == "voxel_maxpool_fwd":
        src, dst = dims.split("->")
        bs, c, n = (int(v) for v in src.split("x"))
        cells = int(np.prod([int(v) for v in dst.split("x")]))
        d = len(dst.split("x"))
        return 4 * (bs * c * n + bs * n * d + bs * c * cells)
    if name == "pointnet_scatter":
        # fused point_pre + input scatter (zero fill of the target included in the timed span): reads the 7-channel
        # point features and the 2 used coordinate columns, produces the [B,512,512,T*64] grid and the t=0
        # point features; the 491 MB intermediate of the unfused form is not counted because it is never moved.
        # (With the sparse first stage the grid is produced as compact rows of its occupied cells: the figure below
        # stays the reference op's output size, `traffic` shows what actually moves.)
        src, dst = dims.split("->")
        b, t, n = (int(v) for v in src.split("x"))
        cells = int(np.prod([int(v) for v in dst.split("x")]))
        return 4 * (b * t * 7 * n + b * t * n * 2 + b * cells * t * 64 + b * n * 64)
    if name in ("gather_scatter", "gather_scatter_cl"):
        src, n, dst = dims.split("->")
        b, c, h, w = (int(v) for v in src.split("x"))
        cells = int(np.prod([int(v) for v in dst.split("x")]))
        return 4 * (b * c * h * w + 4 * b * int(n) + b * c * cells + (b * c * int(n) if cells == 0 else 0))
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
    if name in ("stem_gemm", "stem_epilogue", "stem_mark+compact"):
        # sparse DownSample2D 192 -> 32, stride 2: the dense op it replaces reads the grid once and writes the half-
        # resolution map; the three spans of the sparse form share that figure (gemm: grid in; epilogue: map out)
        b, h, w, cin = (int(v) for v in dims.split("x"))
        if name == "stem_gemm":
            return 4 * b * h * w * cin
        if name == "stem_epilogue":
            return 4 * b * (h // 2) * (w // 2) * 32
        return 4 * b * h * w
    return 0


def algorithmic_flops(label):
    """FLOPs of a launch whose time is set by the matrix cores rather than by HBM (0 = not such a kernel)."""
    name, dims = label.split("[", 1)
    dims = dims.rstrip("]")
    if name == "point_head":
        b, n = (int(v) for v in dims.split("x"))
        return 2 * b * n * (192 * 96 + 96 * 64 + 64 * 3)
    return 0


def pmc_traffic(label):
    """HBM bytes per launch of `label` from the committed rocprofv3 --pmc passes (profiles/pmc_summary.py;
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950), or None if not collected."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["per_launch"]
        return table[label]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


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
    ap.add_argument("--cpu-scans", type=int, default=2, help="timed scans of the CPU baseline (0 = skip)")
    ap.add_argument("--streams", type=int, default=0,
                    help="also time S concurrent sequences batched on the GPU (BASELINE configs[2]); reported beside value")
    args = ap.parse_args()

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
                                    pipeline=not (args.no_pipeline or args.graph))

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

    # warm-up: also finds the dominant hand-written kernel
    with profiling.kernel_timer() as kt:
        for i in range(args.warmup):
            one_step(i)
    warm = kt.summary()
    # dominant = the single-launch hand-written kernel with the most time (the stem_* spans cover several launches whose
    # bytes depend on the occupancy of the frame; they are listed in hip_kernel_ms_per_step_warmup)
    single = {k: v for k, v in warm.items() if not k.startswith("stem_")}
    dominant = max(single, key=lambda k: single[k][1]) if single else None

    sync()
    t0 = time.perf_counter()
    with profiling.kernel_timer(only=dominant) as kt:
        for i in range(args.steps):
            one_step(args.warmup + i)
        enqueue = time.perf_counter() - t0     # host time to issue all steps (the GPU may still be running)
        sync()
        elapsed = time.perf_counter() - t0
    timed = kt.summary()

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        value = world * args.steps / elapsed
        roof = None
        if dominant and dominant in timed:
            calls, total_ms, mean_ms = timed[dominant]
            ab = algorithmic_bytes(dominant)
            af = algorithmic_flops(dominant)
            if af and af / (FP32_PEAK_TFLOPS * 1e12) > ab / (HBM_PEAK_GBS * 1e9):
                # the matrix-core time of the launch exceeds its HBM time: an MFMA-bound kernel (the fused point head)
                achieved = af / (mean_ms * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": dominant, "achieved": round(achieved, 1), "peak": FP32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 4), "traffic": pmc_traffic(dominant),
                        "algorithmic_flops_per_launch": af, "algorithmic_bytes_per_launch": ab,
                        "avg_launch_ms": round(mean_ms, 4), "launches": calls}
            else:
                achieved = ab / (mean_ms * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dominant),
                        "algorithmic_bytes_per_launch": ab, "avg_launch_ms": round(mean_ms, 4), "launches": calls}
        if roof and dominant.startswith("point_head") and getattr(model, "_engine", None) is not None:
            # the same launch alone on the GPU (in the timed region it shares the CUs with the other pipeline stage)
            from streammos_amd import ops as _ops
            eng, d0 = model._engine, dev_frames[0][0]
            bs, n = d0["pcds_xyzi"].shape[0], d0["pcds_xyzi"].shape[3]
            rows = torch.randn((bs, n, 192), dtype=torch.float32, device=device)
            torch.cuda.synchronize()
            with profiling.kernel_timer(only=dominant) as kt_iso:
                for _ in range(20):
                    _ops.point_head(rows, eng.head_w[0], eng.head_w[1])
            iso = kt_iso.summary()[dominant][2]
            roof["isolated_launch_ms"] = round(iso, 4)
            roof["isolated_frac"] = round(roof["algorithmic_flops_per_launch"] / (iso * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
        if roof and dominant.startswith("pointnet_scatter") and getattr(model, "_engine", None) is not None:
            # the same launch alone on the GPU (in the timed region it shares the CUs with the other pipeline stage)
            from streammos_amd import ops as _ops
            eng, d0 = model._engine, dev_frames[0][0]
            bs, t, _, n = d0["pcds_xyzi"].shape[:4]
            rows = torch.empty((bs, n, 192), dtype=torch.float32, device=device)
            compact = eng.sparse_stem and eng.stem_w is not None        # the form the engine launches (engine._encode_cl)
            if compact:
                plan = _ops.stem_plan(d0["pcds_coord"], *eng.bev_hw)
            else:
                bev = torch.empty((bs,) + tuple(eng.bev_hw) + (t * 64,), dtype=torch.float32, device=device)
            torch.cuda.synchronize()
            with profiling.kernel_timer(only=dominant) as kt_iso:
                for _ in range(20):
                    if compact:
                        _ops.pointnet_scatter_rows(d0["pcds_xyzi"], d0["pcds_coord"], eng.pp1[0], eng.pp1[1], eng.pp2[0],
                                                   eng.pp2[1], plan, pts_out=rows[:, :, :64])
                    else:
                        _ops.pointnet_scatter(d0["pcds_xyzi"], d0["pcds_coord"], eng.pp1[0], eng.pp1[1], eng.pp2[0], eng.pp2[1],
                                              bev, pts_out=rows[:, :, :64], zero_fill=True)
            iso = kt_iso.summary()[dominant][2]
            roof["isolated_launch_ms"] = round(iso, 4)
            roof["isolated_frac"] = round(roof["algorithmic_bytes_per_launch"] / (iso * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        line = {
            "metric": "LiDAR scans/sec (StreamMOS streaming inference + voxel voting)",
            "value": round(value, 3), "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: full StreamMOS streaming inference, 1 sequence per GPU, B=4 TTA x T=3 "
                                   "x N=160000 padded points (120k-point synthetic HDL-64E scans), 8-frame voxel "
                                   "voting %s" % ("off" if args.no_vote else "on"),
                       "tta": 4, "frame_point_num": FRAME_POINT_NUM, "parallelism": "sequence-shard x%d" % world},
            "roofline": roof,
            "path_roofline": {"bound": "hbm", "achieved": round(value / world * ALG_GB_PER_SCAN, 1), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(value / world * ALG_GB_PER_SCAN / HBM_PEAK_GBS, 4),
                              "note": "scans/s/GPU x 11.2 GB algorithmic bytes per scan (SURVEY.md 8d)"},
            "path_compute_roofline": {"bound": "mfma", "achieved": round(value / world * ALG_TFLOP_PER_SCAN, 1),
                                      "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(value / world * ALG_TFLOP_PER_SCAN / FP32_PEAK_TFLOPS, 4),
                                      "note": "dense-equivalent rate: the reference's 0.53 TFLOP/scan (fp32) x scans/s against the fp32 "
                                              "MFMA/vector peak; the engine executes ~0.37 TFLOP of them (sparse first stage, "
                                              "conv_1 as tap GEMMs at source resolution), so this is not a hard ceiling"},
            "host_enqueue_ms_per_step": round(1e3 * enqueue / args.steps, 3),
            "hip_kernel_ms_per_step_warmup": {k: round(v[1] / max(args.warmup, 1), 4) for k, v in sorted(warm.items())},
        }
        if world == 1:
            # PCIe-inclusive variant (never `value`): raw scans uploaded every step, preprocessing on the device
            raw_runner = streaming.StreamRunner(model, device, vote=not args.no_vote)
            from streammos_amd import synth as _synth
            raw = [(_synth.synthetic_scan(k), _synth.synthetic_pose(k)) for k in range(6)]
            def raw_step(i):
                idx = [(i + 2) % 4 + 2 - j for j in range(3)]
                raw_runner.step_raw([raw[j][0] for j in idx], [raw[j][1] for j in idx], FRAME_POINT_NUM)
            for i in range(3):
                raw_step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(20):
                raw_step(i)
            torch.cuda.synchronize()
            line["raw_scan_pipeline"] = {"value": round(20 / (time.perf_counter() - t1), 3), "unit": "scans/s",
                                         "note": "H2D of 3 raw scans (5.8 MB) + device preprocessing + the same step; "
                                                 "PCIe-inclusive, reported beside `value`, never as `value`"}
        if world == 1 and args.streams > 1:
            # configs[2]: S concurrent sequences advanced in lock step as one batch of 4*S samples
            S = args.streams
            ms = streaming.MultiStreamRunner(model, device, n_streams=S, vote=not args.no_vote)
            per_stream = [make_frames(4, seq_seed=100 + q) for q in range(S)]
            batched = []
            for f in range(4):
                devs = [runner.upload(per_stream[q][f][0], per_stream[q][f][1]) for q in range(S)]
                batched.append((ms.batch_inputs(devs), [per_stream[q][f][2] for q in range(S)]))
            for i in range(3):
                ms.step(*batched[i % 4])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            n_it = 10
            for i in range(n_it):
                ms.step(*batched[(3 + i) % 4])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t2
            line["batched_streams"] = {"streams": S, "value": round(S * n_it / dt, 3), "unit": "scans/s",
                                       "ms_per_batched_step": round(1e3 * dt / n_it, 3),
                                       "hbm_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
                                       "note": "BASELINE configs[2]; reported beside `value`, never as `value`"}
        if world == 1 and args.cpu_scans > 0:
            line["cpu_baseline"] = cpu_baseline(frames, state, args.cpu_scans)
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
