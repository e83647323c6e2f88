"""Dev: does aten.miopen_convolution_relu (MIOpen fusion plan conv+bias+ReLU) beat conv + our bias_act_cl epilogue?"""
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
for c, hw in ((32, 256), (64, 128), (128, 64)):
    for cl in (True, False):
        x = torch.randn(4, c, hw, hw, device=dev); w = torch.randn(c, c, 3, 3, device=dev) * 0.05; bias = torch.randn(c, device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):
            t_conv = timeit(lambda: F.conv2d(x, w, None, 1, 1))
            def unfused():
                y = F.conv2d(x, w, None, 1, 1)
                return ops.bias_act_cl(y, bias, 1, out=y) if cl else ops.bias_act(y, bias, 1, out=y)
            t_unf = timeit(unfused)
            try:
                t_fus = timeit(lambda: torch.ops.aten.miopen_convolution_relu(x, w, bias, [1, 1], [1, 1], [1, 1], 1))
                y1 = torch.ops.aten.miopen_convolution_relu(x, w, bias, [1, 1], [1, 1], [1, 1], 1)
                err = (y1 - unfused()).abs().max().item()
            except Exception as e:
                t_fus, err = float("nan"), str(e)[:80]
        print("C=%d %dx%d %s  conv %.3f  conv+epilogue %.3f  miopen_convolution_relu %.3f  (diff %s)" %
              (c, hw, hw, "cl  " if cl else "nchw", t_conv, t_unf, t_fus, err), flush=True)
