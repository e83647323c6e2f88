import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from streammos_amd import ops, streaming
dev = torch.device("cuda:0")
def layer(cin, cout, k, hw, mt):
    x = torch.randn(4, hw[0], hw[1], cin, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(cout, cin, *k, device=dev) * 0.05
    wp = ops.conv_prepare(wt, mt)
    out = ops.empty_cl(4, cout, hw[0], hw[1], dev)
    return lambda: ops.conv_cl(x, wp, None, 1, cout, k, mt=mt, out=out)
g = torch.randn(4, 256, 256, 64, device=dev).permute(0, 3, 1, 2)
co = torch.rand(4, 160000, 2, device=dev) * 500
rows = torch.empty(4, 160000, 64, device=dev)
gather = lambda: ops.gather_scatter_cl(g, co, (0.5, 0.5), pts_out=rows)
conv = layer(128, 64, (3, 3), (256, 256), 2)
conv1 = layer(32, 32, (3, 3), (256, 256), 1)
main = torch.cuda.current_stream()
side = streaming.concurrent_stream(dev)
first = torch.cuda.Stream()
def timeit(fa, fb, sb, n=30):
    for _ in range(3): fa(); fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sb.wait_stream(main)
    for _ in range(n):
        fa()
        with torch.cuda.stream(sb): fb()
    main.wait_stream(sb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for name, fa, fb in (("conv_2 + gather", conv, gather), ("conv_2 + conv 32", conv, conv1), ("conv_2 + conv_2", conv, conv)):
    print("%-18s same stream %.3f ms/pair | probed side %.3f | a fresh pool stream %.3f" % (name, timeit(fa, fb, main), timeit(fa, fb, side), timeit(fa, fb, first)))
