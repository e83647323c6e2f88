"""Dev: the decoder's tap products ([65536 + 16384 tokens, 128] x [128, 1152]) on tfusion_project with 1 / 2 / 3 column ranges per
source, against the library GEMM pair."""
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
xa, xb = torch.randn(65536, 128, device=dev), torch.randn(16384, 128, device=dev)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.05
wt = ops.upconv_tap_weights(torch.cat((w, w), 1), 0, 128)
za, zb = torch.empty(65536, 1152, device=dev), torch.empty(16384, 1152, device=dev)
t = timeit(lambda: (torch.addmm(wt.zero, xa, wt.kn), torch.addmm(wt.zero, xb, wt.kn)))
print("library GEMMs: %.4f ms" % t)
fl = 2.0 * (65536 + 16384) * 128 * 1152
for parts in (1, 2, 3, 6):
    n = 1152 // parts
    jobs = []
    for x, z in ((xa, za), (xb, zb)):
        for k, ws in enumerate(wt.stream(parts)):
            jobs.append((x, ws, n, z[:, k * n:(k + 1) * n]))
    if len(jobs) > 8:
        t = timeit(lambda: (ops.tfusion_project(jobs[:parts]), ops.tfusion_project(jobs[parts:])))
    else:
        t = timeit(lambda: ops.tfusion_project(jobs))
    print("tfusion_project, %d column ranges per source: %.4f ms  %.1f TFLOP/s" % (parts, t, fl / t / 1e9), flush=True)
