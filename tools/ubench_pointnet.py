"""Dev: where does pointnet_scatter spend its time?  (normal / no valid cell / no point rows)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streammos_amd import ops
dev = "cuda:0"
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
s = bench.make_frames(1, 0)[0][0]
xyzi = torch.from_numpy(s["pcds_xyzi"]).to(dev); coord = torch.from_numpy(s["pcds_coord"]).to(dev)
g = torch.Generator(device="cpu").manual_seed(0)
w1 = (torch.randn(64, 7, 1, 1, generator=g) * 0.4).to(dev); b1 = torch.randn(64, generator=g).to(dev) * 0.1
w2 = (torch.randn(64, 64, 1, 1, generator=g) * 0.15).to(dev); b2 = torch.randn(64, generator=g).to(dev) * 0.1
bev = torch.zeros(4, 512, 512, 192, device=dev); rows = torch.empty(4, 160000, 192, device=dev)
print("fill only          %.3f ms" % timeit(lambda: bev.zero_()))
print("kernel, normal     %.3f ms" % timeit(lambda: ops.pointnet_scatter(xyzi, coord, w1, b1, w2, b2, bev, pts_out=rows[:, :, :64])))
print("kernel, no rows    %.3f ms" % timeit(lambda: ops.pointnet_scatter(xyzi, coord, w1, b1, w2, b2, bev)))
bad = torch.full_like(coord, -5000.0)
print("kernel, no cells   %.3f ms" % timeit(lambda: ops.pointnet_scatter(xyzi, bad, w1, b1, w2, b2, bev, pts_out=rows[:, :, :64])))
print("kernel, no cells, no rows %.3f ms" % timeit(lambda: ops.pointnet_scatter(xyzi, bad, w1, b1, w2, b2, bev)))
