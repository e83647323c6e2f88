"""Where the host time of one streaming step goes: cProfile over the enqueue of K steps (GPU work is asynchronous).
    python tools/host_profile.py [--steps 30] [--sort tottime]      (on the GPU box)"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from streammos_amd import streaming, synth  # noqa: E402
from streammos_amd.refapi.config import StreamMOS as cfg  # noqa: E402
from streammos_amd.refapi.models import StreamMOS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--sort", default="tottime")
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    model = StreamMOS.AttNet(cfg.get_config()[2])
    model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
    model.engine_layout = "cl"
    runner = streaming.StreamRunner(model, device, vote=True, pipeline=True)
    frames = bench.make_frames(6, seq_seed=0)
    dev_frames = [(runner.upload(s, raw), pose) for s, raw, pose in frames]

    def step(i):
        d, pose = dev_frames[i % len(dev_frames)]
        return runner.step(d, pose, next_dev=dev_frames[(i + 1) % len(dev_frames)][0])

    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("unprofiled: enqueue %.3f ms/step, wall %.3f ms/step" % (1e3 * t_enq / args.steps, 1e3 * t_all / args.steps))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats(args.sort).print_stats(args.top)


if __name__ == "__main__":
    main()
