"""Dev: csrc/downsample.hip (pool branch + tail in one launch) against the two launches it replaces, at the network's four shapes."""
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
for c, (h, w), s in ((64, (256, 256), 2), (128, (128, 128), 2), (32, (32, 1024), 1), (64, (16, 512), 1)):
    x = torch.randn(4, h, w, c, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(c, c, 1, 1, device=dev) / c ** 0.5
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    a = torch.randn(4, ho, wo, c, device=dev).permute(0, 3, 1, 2)
    bias = torch.randn(c, device=dev)
    wq, wp = ops.pool_branch_prepare(wt), ops.conv_prepare(wt, ops.conv_mt(c, 4 * h * w))
    mt = ops.conv_mt(c, 4 * h * w)
    out = ops.empty_cl(4, c, ho, wo, dev)
    t_f = timeit(lambda: ops.downsample_pool_branch(x, wq, a, bias, s, out=out))
    t_c = timeit(lambda: ops.conv_cl(x, wp, None, 0, c, (1, 1), mt=mt))
    q = ops.conv_cl(x, wp, None, 0, c, (1, 1), mt=mt)
    t_e = timeit(lambda: ops.downsample_epilogue_cl(a, q, bias, s, out=out))
    mb = 4 * (4 * h * w * c + 2 * 4 * ho * wo * c) / 1e6
    print("%3d ch @%s /%d: fused %.4f ms (%.0f MB minimum: %.2f TB/s) | conv1x1 %.4f + epilogue %.4f = %.4f" %
          (c, (h, w), s, t_f, mb, mb / t_f / 1e3 / 1e3 * 1e3, t_c, t_e, t_c + t_e), flush=True)
