"""Dev: where does the host spend its 2.8 ms per step?  cProfile over 30 pipelined steps (GPU work is async)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streammos_amd import streaming, synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS
DEV = "cuda:0"
model = StreamMOS.AttNet(cfg.get_config()[2]); model.load_state_dict(synth.seeded_state_dict(model.state_dict()))
model = model.to(DEV).eval()
runner = streaming.StreamRunner(model, DEV, vote=True, pipeline=True)
frames = bench.make_frames(6, 0)
devs = [(runner.upload(s, raw), pose) for s, raw, pose in frames]
def step(i):
    d, pose = devs[i % len(devs)]
    return runner.step(d, pose, next_dev=devs[(i + 1) % len(devs)][0])
for i in range(12): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(30): step(12 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
