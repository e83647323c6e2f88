"""Dev: do two convolutions on two HIP streams overlap?  Times N launches of a layer on one stream against N/2 + N/2 on two
streams, for a few layer pairs (same layer twice; an MFMA-bound conv next to a bandwidth-bound gather)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streammos_amd import ops
dev = "cuda:0"
def layer(cin, cout, k, hw, mt):
    x = torch.randn(4, hw[0], hw[1], cin, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(cout, cin, *k, device=dev) * 0.05
    wp = ops.conv_prepare(wt, mt)
    return lambda: ops.conv_cl(x, wp, None, 1, cout, k, mt=mt)
def run(fa, fb, n=40):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res = []
    for two in (False, True):
        for _ in range(3): fa(); fb()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        if two:
            for i in range(n // 2):
                with torch.cuda.stream(s1): fa()
                with torch.cuda.stream(s2): fb()
        else:
            with torch.cuda.stream(s1):
                for i in range(n // 2): fa(); fb()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        e.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(e) / n)
    return res
pairs = {
    "conv_2 mt2 + conv_2 mt2": (layer(128, 64, (3, 3), (256, 256), 2), layer(128, 64, (3, 3), (256, 256), 2)),
    "conv_2 mt1 + conv_2 mt1": (layer(128, 64, (3, 3), (256, 256), 1), layer(128, 64, (3, 3), (256, 256), 1)),
    "hdr 3x3 32 + res2 3x3 128": (layer(32, 32, (3, 3), (256, 256), 1), layer(128, 128, (3, 3), (64, 64), 1)),
    "res1 3x3 64 mt2 + res2 3x3 128": (layer(64, 64, (3, 3), (128, 128), 2), layer(128, 128, (3, 3), (64, 64), 1)),
}
g = torch.randn(4, 256, 256, 64, device=dev).permute(0, 3, 1, 2)
co = torch.rand(4, 160000, 2, device=dev) * 500
rows = torch.empty(4, 160000, 64, device=dev)
gather = lambda: ops.gather_scatter_cl(g, co, (0.5, 0.5), pts_out=rows)
pairs["conv_2 mt2 + gather 64ch"] = (layer(128, 64, (3, 3), (256, 256), 2), gather)
for name, (fa, fb) in pairs.items():
    one, two = run(fa, fb)
    print("%-34s one stream %.3f ms / launch, two streams %.3f ms / launch  (x%.2f)" % (name, one, two, one / two), flush=True)
