"""Dev: do kernels on different HIP streams of this process run concurrently at all?  torch.cuda._sleep is a one-block spin
kernel: two of them on two streams take the time of one if the streams are on different hardware queues."""
import time, torch
dev = "cuda:0"
torch.cuda.init()
cyc = 2_000_000_0 // 10
def t(streams, n=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
d = torch.cuda.default_stream()
pool = [torch.cuda.Stream() for _ in range(8)]
base = t([d])
print("1 stream x4 sleeps: %.2f ms" % base)
print("default + pool[0]: %.2f ms (x8 sleeps; %.2f if serial)" % (t([d, pool[0]]), 2 * base))
for i in range(1, 8):
    print("pool[0] + pool[%d]: %.2f ms" % (i, t([pool[0], pool[i]])))
print("4 pool streams: %.2f ms (serial %.2f)" % (t(pool[:4]), 4 * base))
print("8 pool streams: %.2f ms (serial %.2f)" % (t(pool[:8]), 8 * base))
