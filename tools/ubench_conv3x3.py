"""Dev: own fused conv3x3 (bias + ReLU in the epilogue) vs MIOpen conv + bias_act_cl for the network's BasicBlock shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
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
for c, (h, w) in ((32, (256, 256)), (64, (128, 128)), (32, (32, 1024)), (64, (16, 512))):
    x = torch.randn(4, h, w, c, device=dev).permute(0, 3, 1, 2)
    wt = (torch.randn(c, c, 3, 3, device=dev) * 0.05); wcl = wt.contiguous(memory_format=torch.channels_last)
    bias = torch.randn(c, device=dev); wp = ops.conv3x3_prepare(wt)
    with torch.backends.cudnn.flags(enabled=True, benchmark=True):
        def lib():
            y = F.conv2d(x, wcl, None, 1, 1)
            return ops.bias_act_cl(y, bias, 1, out=y)
        t_lib = timeit(lib)
        t_conv = timeit(lambda: F.conv2d(x, wcl, None, 1, 1))
    t_own = timeit(lambda: ops.conv3x3_cl(x, wp, bias, 1))
    gf = 2.0 * 4 * h * w * c * c * 9 / 1e9
    print("C=%d %dx%d  MIOpen conv %.3f (+epilogue %.3f) ms   own fused %.3f ms  %.1f TFLOP/s" % (c, h, w, t_conv, t_lib, t_own, gf / t_own), flush=True)
