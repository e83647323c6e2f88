"""Dev: the PCIe-inclusive raw-scan pipeline (bench.py's raw_scan_pipeline leg) alone, with A/B switches from the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streammos_amd import streaming, synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS
dev = torch.device("cuda:0")
model = StreamMOS.AttNet(cfg.get_config()[2])
model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
skip = os.environ.get("SKIP", "1") != "0"
r = streaming.StreamRunner(model, dev, vote=True, pipeline=True, skip_padding=skip)
raw = [(synth.synthetic_scan(k), synth.synthetic_pose(k)) for k in range(6)]
def window(i):
    idx = [(i + 2) % 4 + 2 - j for j in range(3)]
    return [raw[j][0] for j in idx], [raw[j][1] for j in idx]
def step(i):
    (s, p), (ns, np_) = window(i), window(i + 1)
    r.step_raw(s, p, 160000, next_scans=ns, next_poses=np_)
for i in range(5): step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(40): step(i)
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("skip_padding=%s: %.1f scans/s (%.3f ms/step, host enqueue %.3f ms/step)" % (skip, 40 / dt, 1e3 * dt / 40, 1e3 * enq / 40), flush=True)
