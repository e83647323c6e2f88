"""Dev: the own implicit-GEMM conv (csrc/conv_igemm.hip, epilogue fused) against MIOpen conv + bias_act_cl for every
convolution shape of the network at the validation batch (B = 4), for each admissible mt.  Prints ms and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from streammos_amd import ops
dev = "cuda:0"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
LAYERS = [  # name, cin, cout, (kh, kw), stride, (h, w) of the input
    ("hdr_bev 3x3 32", 32, 32, (3, 3), 1, (256, 256)), ("hdr_bev 7x3", 32, 32, (7, 3), 1, (256, 256)),
    ("hdr_bev 3x7", 32, 32, (3, 7), 1, (256, 256)), ("hdr_bev 64->32", 64, 32, (3, 3), 1, (256, 256)),
    ("hdr_rv 3x3 32", 32, 32, (3, 3), 1, (32, 1024)), ("hdr_rv 1x1", 32, 32, (1, 1), 1, (32, 1024)),
    ("res1 down 3x3s2", 64, 64, (3, 3), 2, (256, 256)), ("res1 down 1x1", 64, 64, (1, 1), 1, (256, 256)),
    ("res1 5x3", 64, 64, (5, 3), 1, (128, 128)), ("res1 3x5", 64, 64, (3, 5), 1, (128, 128)),
    ("res1 128->64", 128, 64, (3, 3), 1, (128, 128)), ("res1 3x3 64", 64, 64, (3, 3), 1, (128, 128)),
    ("res1_rv 3x3 64", 64, 64, (3, 3), 1, (16, 512)), ("res1_rv 1x1", 64, 64, (1, 1), 1, (16, 512)),
    ("res2 down 3x3s2", 128, 128, (3, 3), 2, (128, 128)), ("res2 down 1x1", 128, 128, (1, 1), 1, (128, 128)),
    ("res2 3x3 128", 128, 128, (3, 3), 1, (64, 64)),
    ("conv_1a 64->128", 64, 128, (3, 3), 1, (256, 256)), ("conv_2 128->64", 128, 64, (3, 3), 1, (256, 256)),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
tot_lib = tot_own = 0.0
for name, cin, cout, (kh, kw), stride, (h, w) in LAYERS:
    if only and only not in name:
        continue
    x = torch.randn(4, h, w, cin, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(cout, cin, kh, kw, device=dev) * 0.05
    wcl = wt.contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=dev)
    pad = (kh // 2, kw // 2)
    ho, wo = (h + 2 * pad[0] - kh) // stride + 1, (w + 2 * pad[1] - kw) // stride + 1
    gf = 2.0 * 4 * ho * wo * cin * cout * kh * kw / 1e9
    if os.environ.get("SMOS_UBENCH_NO_LIB"):      # own kernel only (knob sweeps)
        t_lib = t_conv = float("nan")
    else:
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):
            def lib():
                y = F.conv2d(x, wcl, None, stride, pad)
                return ops.bias_act_cl(y, bias, 1, out=y)
            t_lib = timeit(lib)
            t_conv = timeit(lambda: F.conv2d(x, wcl, None, stride, pad))
    res = []
    for mt in (1, 2, 4):
        if cout % (32 * mt):
            continue
        wp = ops.conv_prepare(wt, mt)
        t = timeit(lambda: ops.conv_cl(x, wp, bias, 1, cout, (kh, kw), stride=stride, mt=mt))
        res.append((t, mt))
    if ops.conv_rows_ok((kh, kw), stride, cin, cout):
        for rmt in (1, 2):
            if cout % (32 * rmt):
                continue
            wr = ops.conv_prepare(wt, rmt, order="rows")
            t = timeit(lambda: ops.conv_rows_cl(x, wr, bias, 1, cout, (kh, kw), mt=rmt))
            res.append((t, -rmt))                # printed as "mt-1" / "mt-2" = the row-staging kernel at 32 / 64 outputs per block
    best = min(res)
    tot_lib += t_lib; tot_own += best[0]
    print("%-18s %6.2f GF  MIOpen %.3f (+epi %.3f) ms %5.1f TF | own " % (name, gf, t_conv, t_lib, gf / t_conv) +
          "  ".join("mt%d %.3f ms %5.1f TF" % (mt, t, gf / t) for t, mt in res) + "   auto mt%d" % ops.conv_mt(cout, 4 * ho * wo),
          flush=True)
print("sum: MIOpen+epilogue %.3f ms, own (best mt) %.3f ms" % (tot_lib, tot_own))
