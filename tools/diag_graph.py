"""Dev: bisect a hipGraph-replay discrepancy of the channels-last engine.  Captures encode / decode separately, replays each
twice and compares every output with the eager engine on the same static inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from streammos_amd import preprocess, streaming, synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS

DEV = "cuda:0"
model = StreamMOS.AttNet(cfg.get_config()[2])
model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
model = model.to(DEV).eval()
spec = preprocess.VoxelSpec()
scans = [synth.synthetic_scan(k, 16, 120) for k in range(6)]
poses = [synth.synthetic_pose(k) for k in range(6)]
samples = []
for i in range(3):
    idx = preprocess.window_indices(i, 6, 3)
    s = preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], 2048, spec, tta=True)
    samples.append({k: torch.from_numpy(np.ascontiguousarray(s[k])).to(DEV) for k in ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")})
with torch.no_grad():
    eng = model._engine_for(samples[0]["pcds_xyzi"])

def cmp(tag, a, b):
    if torch.is_tensor(a):
        d = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        print("   %-10s %s rel diff %.3e%s" % (tag, tuple(a.shape), d, "   <-- DIFFERS" if d > 1e-5 else ""))

static = {k: v.clone() for k, v in samples[0].items()}
side = torch.cuda.Stream(DEV)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(2):
        enc = eng.encode(static["pcds_xyzi"], static["pcds_coord"], static["pcds_sphere_coord"])
        out = eng.decode(enc, None)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()

g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    enc_g = eng.encode(static["pcds_xyzi"], static["pcds_coord"], static["pcds_sphere_coord"])
for rep, smp in enumerate((samples[1], samples[2], samples[1])):
    for k in static:
        static[k].copy_(smp[k])
    g.replay()
    torch.cuda.synchronize()
    got = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in enc_g.items()}
    with torch.no_grad():
        want = eng.encode(smp["pcds_xyzi"], smp["pcds_coord"], smp["pcds_sphere_coord"])
    print("encode replay %d:" % rep)
    for k in want:
        cmp(k, got[k], want[k])

# decode with a fixed encoder output
with torch.no_grad():
    enc_fix = eng.encode(samples[1]["pcds_xyzi"], samples[1]["pcds_coord"], samples[1]["pcds_sphere_coord"])
    enc_static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in enc_fix.items()}
    want = eng.decode(enc_fix, None)
    for _ in range(2):
        eng.decode(enc_static, None)
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g2):
    out_g = eng.decode(enc_static, None)
for rep in range(2):
    g2.replay()
    torch.cuda.synchronize()
    print("decode replay %d:" % rep)
    for i, (a, b) in enumerate(zip(out_g, want)):
        cmp("out[%d]" % i, a, b)

# ---- the runner's own capture / replay against the eager runner, frame by frame ----
import copy
spec = preprocess.VoxelSpec()
res = {}
for graph in (False, True):
    runner = streaming.StreamRunner(copy.deepcopy(model), DEV, vote=False, graph=graph)
    outs = []
    for i in range(4):
        idx = preprocess.window_indices(i, 6, 3)
        sample = preprocess.build_sample([scans[j] for j in idx], [poses[j] for j in idx], 2048, spec, tta=True)
        o = runner.step(runner.upload(sample, scans[i]), poses[i])
        mem = runner.memory[0] if isinstance(runner.memory, list) else runner.memory
        outs.append((o["pred_cls"].clone(), mem.clone()))
    res[graph] = outs
for i, ((p0, m0), (p1, m1)) in enumerate(zip(res[False], res[True])):
    print("runner frame %d:" % i)
    cmp("pred", p1, p0)
    cmp("memory", m1, m0)
