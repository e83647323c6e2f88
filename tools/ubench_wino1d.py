"""Dev: csrc/conv_wino1d.hip (1-D Winograd F(2,3) along the 3-tap axis) against the direct own kernel (conv_rows) for the
k x 3 / 3 x k shapes of the network's Unbalance blocks at the validation batch (B = 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
LAYERS = [("hdr_bev 32->32 @256^2 k7x3", 32, 32, (256, 256), (7, 3)), ("hdr_bev 32->32 @256^2 k3x7", 32, 32, (256, 256), (3, 7)),
          ("res1 64->64 @128^2 k5x3", 64, 64, (128, 128), (5, 3)), ("res1 64->64 @128^2 k3x5", 64, 64, (128, 128), (3, 5))]
tot_d = tot_w = 0.0
for name, cin, cout, (h, w), (kh, kw) in LAYERS:
    x = torch.randn(4, h, w, cin, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(cout, cin, kh, kw, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    gf = 2.0 * 4 * h * w * cin * cout * kh * kw / 1e9
    mt = ops.conv_mt(cout, 4 * h * w)
    if ops.conv_rows_ok((kh, kw), 1, cin, cout) and mt <= 2:
        wr = ops.conv_prepare(wt, mt, order="rows")
        t_dir = timeit(lambda: ops.conv_rows_cl(x, wr, bias, 1, cout, (kh, kw), mt=mt))
    else:
        wp = ops.conv_prepare(wt, mt)
        t_dir = timeit(lambda: ops.conv_cl(x, wp, bias, 1, cout, (kh, kw), mt=mt))
    res = []
    for mb in (1, 2):
        wq = ops.conv_wino1d_prepare(wt, mb)
        res.append((timeit(lambda: ops.conv_wino1d_cl(x, wq, bias, 1, cout, (kh, kw), mb=mb)), mb))
    tot_d += t_dir; tot_w += min(res)[0]
    print("%-30s %6.2f GF  direct mt%d %.4f ms %5.1f TF | wino1d " % (name, gf, mt, t_dir, gf / t_dir) +
          "  ".join("mb%d %.4f ms %5.1f TF-eq (%.2f of the executed-FLOP peak) x%.2f" % (mb, t, gf / t, gf / 1.5 / t / 157.3, t_dir / t)
                    for t, mb in res), flush=True)
print("the four launches of a step: direct %.3f ms, wino1d (best mb) %.3f ms" % (tot_d, tot_w))
