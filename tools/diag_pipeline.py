import faulthandler, os, sys, time
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from streammos_amd import preprocess, streaming, synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS
def log(*a):
    print(*a, flush=True)
DEV = "cuda:0"
model = StreamMOS.AttNet(cfg.get_config()[2]); model.load_state_dict(synth.seeded_state_dict(model.state_dict()))
model = model.to(DEV).eval()
spec = preprocess.VoxelSpec()
scans = [synth.synthetic_scan(k, 16, 120) for k in range(8)]; poses = [synth.synthetic_pose(k) for k in range(8)]
runner = streaming.StreamRunner(model, DEV, vote=(len(sys.argv) > 1 and sys.argv[1] == "vote"), pipeline=True)
devs = []
for i in range(6):
    idx = preprocess.window_indices(i, 8, 3)
    devs.append(runner.upload(preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], 2048, spec, tta=True), scans[i]))
log("uploaded")
for i in range(6):
    t = time.time()
    o = runner.step(devs[i], poses[i], next_dev=devs[i + 1] if i < 5 else None)
    log("step", i, "enqueued", round(time.time() - t, 3))
    torch.cuda.synchronize()
    log("step", i, "synced", round(time.time() - t, 3), float(o["pred_cls"].abs().mean()))
log("done")
