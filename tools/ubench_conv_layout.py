"""Dev: per-layer NCHW vs channels_last conv time under MIOpen solver search, for the engine's exact layer list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda:0"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
layers = [  # name, count, cin, cout, (h,w), k, stride, pad
 ("hdr 7x3 32@256", 1, 32, 32, (256, 256), (7, 3), 1, (3, 1)), ("hdr 3x7 32@256", 1, 32, 32, (256, 256), (3, 7), 1, (1, 3)),
 ("hdr fuse 64->32@256", 1, 64, 32, (256, 256), (3, 3), 1, 1), ("bb 32@256", 4, 32, 32, (256, 256), (3, 3), 1, 1),
 ("rv down3 32@32x1024", 1, 32, 32, (32, 1024), (3, 3), 1, 1), ("rv down1 32@32x1024", 1, 32, 32, (32, 1024), (1, 1), 1, 0),
 ("rv bb 32@32x1024", 4, 32, 32, (32, 1024), (3, 3), 1, 1),
 ("r1 down3 64@256 s2", 1, 64, 64, (256, 256), (3, 3), 2, 1), ("r1 down1 64@256", 1, 64, 64, (256, 256), (1, 1), 1, 0),
 ("r1 5x3 64@128", 1, 64, 64, (128, 128), (5, 3), 1, (2, 1)), ("r1 3x5 64@128", 1, 64, 64, (128, 128), (3, 5), 1, (1, 2)),
 ("r1 fuse 128->64@128", 1, 128, 64, (128, 128), (3, 3), 1, 1), ("bb 64@128", 6, 64, 64, (128, 128), (3, 3), 1, 1),
 ("rv1 down3 64@16x512", 1, 64, 64, (16, 512), (3, 3), 1, 1), ("rv1 down1 64@16x512", 1, 64, 64, (16, 512), (1, 1), 1, 0),
 ("rv1 bb 64@16x512", 6, 64, 64, (16, 512), (3, 3), 1, 1),
 ("r2 down3 128@128 s2", 1, 128, 128, (128, 128), (3, 3), 2, 1), ("r2 down1 128@128", 1, 128, 128, (128, 128), (1, 1), 1, 0),
 ("bb 128@64", 10, 128, 128, (64, 64), (3, 3), 1, 1),
 ("conv_1 320->128@256", 1, 320, 128, (256, 256), (3, 3), 1, 1), ("conv_2 128->64@256", 1, 128, 64, (256, 256), (3, 3), 1, 1),
 ("aux 320->9@256", 1, 320, 9, (256, 256), (1, 1), 1, 0),
 ("s0 down3 192@512 s2", 1, 192, 32, (512, 512), (3, 3), 2, 1), ("s0 down1 192@512", 1, 192, 32, (512, 512), (1, 1), 1, 0),
]
tot_n = tot_c = tot_best = 0
for name, cnt, cin, cout, hw, k, st, pad in layers:
    x = torch.randn((4, cin) + hw, device=dev); w = torch.randn((cout, cin) + k, device=dev) * 0.05
    tn = timeit(lambda: F.conv2d(x, w, None, st, pad))
    xc = x.contiguous(memory_format=torch.channels_last); wc = w.contiguous(memory_format=torch.channels_last)
    tc = timeit(lambda: F.conv2d(xc, wc, None, st, pad))
    tot_n += cnt * tn; tot_c += cnt * tc; tot_best += cnt * min(tn, tc)
    ho, wo = (hw[0] + 2 * (pad if isinstance(pad, int) else pad[0]) - k[0]) // st + 1, (hw[1] + 2 * (pad if isinstance(pad, int) else pad[1]) - k[1]) // st + 1
    gf = 2.0 * 4 * ho * wo * cout * cin * k[0] * k[1] / 1e9
    print("%-24s x%2d  nchw %.3f  cl %.3f  %s  cl: %6.1f TF/s  layer total %.3f ms" % (name, cnt, tn, tc, "<-- cl" if tc < 0.9 * tn else "      ", gf / tc, cnt * tc), flush=True)
print("total nchw %.3f  cl %.3f  best-per-layer %.3f ms" % (tot_n, tot_c, tot_best))
