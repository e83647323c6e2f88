"""Dev: what would a K split buy on the small-map Winograd layers?  Proxy: the same layer with HALF the input channels at TWICE
the batch has the work items of a 2-way K split (two blocks per CU, half the chunk loop each) minus the reduction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streammos_amd import ops
dev = "cuda:0"
def timeit(fn, n=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def run(b, cin, cout, h, w, mb):
    x = torch.randn(b, h, w, cin, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    wq = ops.conv_wino_prepare(wt, mb)
    return timeit(lambda: ops.conv_wino_cl(x, wq, bias, 1, cout, mb=mb))
for name, cin, cout, h, w in (("res2 128->128 @64^2", 128, 128, 64, 64), ("res1_rv 64->64 @16x512", 64, 64, 16, 512),
                              ("hdr_rv 32->32 @32x1024", 32, 32, 32, 1024), ("res1 64->64 @128^2", 64, 64, 128, 128)):
    items = 4 * ((h + 7) // 8) * ((w + 31) // 32) * (cout // 32)
    t0 = run(4, cin, cout, h, w, 2)
    t1 = run(4, cin, cout, h, w, 1)
    tp = run(8, cin // 2, cout, h, w, 2) if cin >= 32 else float("nan")
    t2 = run(8, cin, cout, h, w, 2)
    print("%-24s items(mb2) %4d: mb2 %.4f ms | mb1 %.4f | proxy K/2 x 2B %.4f | 2B full K %.4f (per sample-batch %.4f)" %
          (name, items, t0, t1, tp, t2, t2 / 2), flush=True)
