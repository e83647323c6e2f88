"""Dev: the decoder's tap GEMMs ([B*Hs*Ws, 128] x [128, 1152]) through the library in both weight layouts, and the
attention-sized GEMMs ([16384, 128] x [128, 128 / 512]); ms and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda:0"
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for m, k, n in ((65536, 128, 1152), (16384, 128, 1152), (16384, 128, 128), (16384, 128, 512), (16384, 512, 128), (16384, 128, 48)):
    x = torch.randn(m, k, device=dev)
    w_nk = torch.randn(n, k, device=dev)           # nn.Linear layout
    w_kn = w_nk.t().contiguous()
    bias = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev)
    gf = 2.0 * m * k * n / 1e9
    res = {
        "mm(x, w_nk.t())": timeit(lambda: torch.mm(x, w_nk.t())),
        "mm(x, w_kn)": timeit(lambda: torch.mm(x, w_kn)),
        "mm(x, w_kn, out=)": timeit(lambda: torch.mm(x, w_kn, out=out)),
        "linear(x, w_nk, b)": timeit(lambda: torch.nn.functional.linear(x, w_nk, bias)),
        "addmm(b, x, w_kn)": timeit(lambda: torch.addmm(bias, x, w_kn)),
    }
    print("[%d x %d] x [%d x %d]  %.2f GF: " % (m, k, k, n, gf) + "  ".join("%s %.3f ms (%.0f TF)" % (a, t, gf / t) for a, t in res.items()), flush=True)
