"""Dev: cost of the sparse first stage piece by piece vs the dense DownSample2D (bench frame, model weights)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streammos_amd import ops, synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS
DEV = "cuda:0"
model = StreamMOS.AttNet(cfg.get_config()[2]); model.load_state_dict(synth.seeded_state_dict(model.state_dict()))
model = model.to(DEV).eval()
s = bench.make_frames(1, 0)[0][0]
xyzi = torch.from_numpy(s["pcds_xyzi"]).to(DEV); coord = torch.from_numpy(s["pcds_coord"]).to(DEV)
with torch.no_grad():
    eng = model._engine_for(xyzi)
b, t, _, n = xyzi.shape[:4]
bev_cl = torch.zeros((b, 512, 512, 192), device=DEV)
ops.pointnet_scatter(xyzi, coord, eng.pp1[0], eng.pp1[1], eng.pp2[0], eng.pp2[1], bev_cl)
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
with torch.no_grad(), eng._conv_flags():
    print("dense block     %.3f ms" % timeit(lambda: eng._block_cl(bev_cl.permute(0, 3, 1, 2), eng.header_bev[0])))
    print("sparse (total)  %.3f ms" % timeit(lambda: eng._stem_sparse_cl(bev_cl, coord)))
    t = lambda fn: timeit(fn)
    print("dense scatter (fill + kernel)   %.3f ms" % t(lambda: ops.pointnet_scatter(xyzi, coord, eng.pp1[0], eng.pp1[1], eng.pp2[0], eng.pp2[1], bev_cl, zero_fill=True)))
    print("plan (mark + scan incl. zero fill) %.3f ms" % t(lambda: ops.stem_plan(coord, 512, 512, row_floats=192)))
    plan = ops.stem_plan(coord, 512, 512, row_floats=192)
    print("compact scatter (fill + kernel) %.3f ms" % t(lambda: ops.pointnet_scatter_rows(xyzi, coord, eng.pp1[0], eng.pp1[1], eng.pp2[0], eng.pp2[1], plan)))
    rows = ops.pointnet_scatter_rows(xyzi, coord, eng.pp1[0], eng.pp1[1], eng.pp2[0], eng.pp2[1], plan)
    print("sparse downsample, dense src    %.3f ms" % t(lambda: ops.sparse_downsample(bev_cl, plan, eng.stem_w, eng.header_bev[0].bias, compact=False)))
    print("sparse downsample, compact src  %.3f ms" % t(lambda: ops.sparse_downsample(rows, plan, eng.stem_w, eng.header_bev[0].bias, compact=True)))
    print("rows", plan.meta.tolist())
