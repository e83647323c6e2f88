"""Dev: the fused point head alone (B=4, N=160000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streammos_amd import ops
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(0)
rows = torch.randn((4, 160000, 192), generator=g).to(dev)
l1 = ((torch.randn((96, 192), generator=g) * 0.1).to(dev), torch.randn(96, generator=g).to(dev))
l2 = ((torch.randn((64, 96), generator=g) * 0.1).to(dev), torch.randn(64, generator=g).to(dev))
l3 = ((torch.randn((3, 64), generator=g) * 0.1).to(dev), torch.randn(3, generator=g).to(dev))
w, m3 = ops.point_head_prepare(l1, l2, l3)
for _ in range(5): ops.point_head(rows, w, m3)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): ops.point_head(rows, w, m3)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 30
print("point_head %.4f ms  %.1f TFLOP/s  (floor 0.2015 ms at 157.3 TFLOP/s)" % (ms, 2 * 640000 * (192 * 96 + 96 * 64 + 64 * 3) / ms / 1e9))

# the runner's form: the scan's padding tail left out (96 069 real points of 160 000 on the bench's synthetic scans)
n_live = torch.tensor([96069], dtype=torch.int32, device="cuda:0")
for _ in range(5): ops.point_head(rows, w, m3, n_live=n_live)
torch.cuda.synchronize()
a.record()
for _ in range(30): ops.point_head(rows, w, m3, n_live=n_live)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 30
print("point_head, 96 069 live points per sample: %.4f ms  (matrix floor %.4f ms)" % (ms, 0.2015 * 96069 / 160000))
