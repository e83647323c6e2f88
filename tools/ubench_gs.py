"""Dev: the five gather_scatter_cl launches of one step, in isolation (bench-frame coordinates, post-ReLU random maps)."""
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
bev_xy = torch.from_numpy(s["pcds_coord"]).to(dev)[:, 0, :, :2, 0].contiguous()
sphere = torch.from_numpy(s["pcds_sphere_coord"]).to(dev)[:, 0, :, :, 0].contiguous()
B, N = bev_xy.shape[0], bev_xy.shape[1]
def cl(b, c, h, w):
    return torch.relu(torch.randn(b, h, w, c, device=dev)).permute(0, 3, 1, 2)
total = 0.0
check = []
for c, (hb, wb), (hr, wr), sc in ((32, (256, 256), (32, 1024), (0.5, 0.5)), (64, (128, 128), (16, 512), (0.25, 0.25))):
    bev, rv = cl(B, c, hb, wb), cl(B, c, hr, wr)
    out_rv = torch.zeros(B, hr, wr, c, device=dev).permute(0, 3, 1, 2)
    out_bev = torch.zeros(B, hb, wb, c, device=dev).permute(0, 3, 1, 2)
    rows = torch.empty(B, N, c, device=dev)
    t1 = timeit(lambda: ops.gather_scatter_cl(bev, bev_xy, sc, sphere, sc, out=out_rv))
    t2 = timeit(lambda: ops.gather_scatter_cl(rv, sphere, sc, bev_xy, sc, out=out_bev, pts_out=rows))
    print("C=%d  bev->rv %.3f ms   rv->bev(+rows) %.3f ms" % (c, t1, t2), flush=True)
    total += t1 + t2
    check += [out_rv.double().sum().item(), out_bev.double().sum().item(), rows.double().sum().item()]
bev = cl(B, 64, 256, 256)
rows = torch.empty(B, N, 64, device=dev)
t = timeit(lambda: ops.gather_scatter_cl(bev, bev_xy, (0.5, 0.5), pts_out=rows))
print("C=64 gather only 256x256 %.3f ms" % t)
check.append(rows.double().sum().item())
print("total %.3f ms" % (total + t))
print("checksums", ["%.6f" % v for v in check])
