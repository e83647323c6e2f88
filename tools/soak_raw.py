"""Dev: 1500 raw-scan steps (upload ring, device preprocessing, look-ahead pipeline, voting) -- rate and allocated HBM every 300 steps; fails if the allocation grows."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from streammos_amd import streaming, synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS
dev = torch.device("cuda:0")
model = StreamMOS.AttNet(cfg.get_config()[2]); model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
runner = streaming.StreamRunner(model, dev, vote=True, pipeline=True)
raw = [(synth.synthetic_scan(k), synth.synthetic_pose(k)) for k in range(8)]
def win(i):
    idx = [(i + 2) % 6 + 2 - j for j in range(3)]
    return [raw[j][0] for j in idx], [raw[j][1] for j in idx]
mem = []
t0 = time.perf_counter()
for i in range(1500):
    (s, p), (ns, np_) = win(i), win(i + 1)
    out = runner.step_raw(s, p, 160000, next_scans=ns, next_poses=np_)
    if i % 300 == 299:
        torch.cuda.synchronize(); mem.append((i + 1, round(torch.cuda.memory_allocated() / 1e9, 3), round(torch.cuda.memory_reserved() / 1e9, 3), round((i + 1) / (time.perf_counter() - t0), 1)))
        print(mem[-1], flush=True)
assert mem[-1][1] <= mem[1][1] * 1.02, "allocated HBM grows"
print("soak ok", int(out["raw_labels"].sum()))
