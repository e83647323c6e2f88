"""Dev: csrc/conv_wino.hip (Winograd F(2x2,3x3)) against the direct own convs (conv_igemm / conv_rows) for every stride-1 3x3
shape of the network at the validation batch (B = 4).  Prints ms, direct-equivalent TFLOP/s and the speed-up."""
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
LAYERS = [  # name, cin, cout, (h, w), launches per step
    ("hdr_bev 32->32 @256^2", 32, 32, (256, 256), 4), ("hdr_bev 64->32 @256^2", 64, 32, (256, 256), 1),
    ("hdr_rv 32->32 @32x1024", 32, 32, (32, 1024), 5), ("res1 64->64 @128^2", 64, 64, (128, 128), 6),
    ("res1 128->64 @128^2", 128, 64, (128, 128), 1), ("res1_rv 64->64 @16x512", 64, 64, (16, 512), 7),
    ("res2 128->128 @64^2", 128, 128, (64, 64), 10), ("conv_1a 64->128 @256^2", 64, 128, (256, 256), 1),
    ("conv_2 128->64 @256^2", 128, 64, (256, 256), 1),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
tot_d = tot_w = 0.0
for name, cin, cout, (h, w), n_step in LAYERS:
    if only and only not in name:
        continue
    x = torch.randn(4, h, w, cin, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    gf = 2.0 * 4 * h * w * cin * cout * 9 / 1e9
    mt = ops.conv_mt(cout, 4 * h * w)
    if mt == 1:
        wr = ops.conv_prepare(wt, 1, order="rows")
        t_dir = timeit(lambda: ops.conv_rows_cl(x, wr, bias, 1, cout, (3, 3), mt=1))
    else:
        wp = ops.conv_prepare(wt, mt)
        t_dir = timeit(lambda: ops.conv_cl(x, wp, bias, 1, cout, (3, 3), mt=mt))
    res = []
    for mb in (1, 2):
        if cout % (16 * mb):
            continue
        wq = ops.conv_wino_prepare(wt, mb)
        res.append((timeit(lambda: ops.conv_wino_cl(x, wq, bias, 1, cout, mb=mb)), mb))
    best = min(res)
    tot_d += n_step * t_dir; tot_w += n_step * best[0]
    items16 = 4 * ((h + 7) // 8) * ((w + 31) // 32) * (cout // 16)
    print("%-24s %6.2f GF  direct mt%d %.4f ms %5.1f TF | wino " % (name, gf, mt, t_dir, gf / t_dir) +
          "  ".join("mb%d %.4f ms %5.1f TF-eq x%.2f" % (mb, t, gf / t, t_dir / t) for t, mb in res) +
          "   auto mb%d" % ops.conv_wino_mb(cout, items16), flush=True)
print("per step (launch counts of the network): direct %.3f ms, winograd (best mb) %.3f ms" % (tot_d, tot_w))
