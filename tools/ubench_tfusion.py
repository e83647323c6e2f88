"""Dev: csrc/tfusion.hip alone on the GPU at the model's shape (16 384 tokens, d_model 128, FFN 512, next projection 48)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streammos_amd import ops
dev = "cuda:0"
def timeit(fn, n=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g = torch.Generator(device="cpu").manual_seed(1)
def lin(o, i): return ((torch.randn((o, i), generator=g) / i ** 0.5).to(dev), torch.randn(o, generator=g).to(dev))
def norm(): return (torch.ones(128, device=dev), torch.zeros(128, device=dev), 1e-5)
for tokens in (16384, 32768, 65536):
    sampled, query = torch.randn(tokens, 128, device=dev), torch.randn(tokens, 128, device=dev)
    for ffn, nq in ((512, 48), (512, 0)):
        prep = ops.TfusionLayer(lin(128, 128), norm(), lin(ffn, 128), lin(128, ffn), norm(), next_qproj=lin(nq, 128) if nq else None)
        out = torch.empty(tokens, 128, device=dev)
        t = timeit(lambda: ops.tfusion_layer(sampled, query, prep, out=out))
        fl = 2.0 * tokens * (128 * 128 + 2 * 128 * ffn + 128 * nq)
        print("tfusion_layer tokens %6d ffn %d q%d: %.4f ms  %.1f TFLOP/s (%.2f of peak)" % (tokens, ffn, nq, t, fl / t / 1e9, fl / t / 1e9 / 157.3), flush=True)
    ws = [lin(128, 128), lin(128, 128), lin(48, 128)]
    jobs = [(sampled, ops.tfusion_pack_linear(ws[0][0]), ws[0][1]), (sampled, ops.tfusion_pack_linear(ws[1][0]), ws[1][1]),
            (query, ops.tfusion_pack_linear(ws[2][0]), ws[2][1])]
    t = timeit(lambda: ops.tfusion_project(jobs))
    print("tfusion_project tokens %6d 128+128+48: %.4f ms" % (tokens, t), flush=True)
