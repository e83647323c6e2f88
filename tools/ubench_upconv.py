"""upconv3x3 at the network's decoder geometry: one-launch interpolation (smos_upconv_xy) vs the x pass + y pass pair.
    python tools/ubench_upconv.py        (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from streammos_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    b, c = 4, 128
    conv_a = torch.randn(b, c, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    x1 = torch.randn(b, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn(b, 128, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, 320, 3, 3, device=dev) * 0.02
    bias = torch.randn(c, device=dev)
    srcs = [(x1, ops.upconv_tap_weights(w, 64, 192)), (x2, ops.upconv_tap_weights(w, 192, 320))]
    out = torch.empty_like(conv_a)
    t_gemm = timeit(lambda: [torch.addmm(wt.zero, x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]), wt.kn) for x, wt in srcs])
    for flag in (True, False):
        ops._UPCONV_XY = flag
        t = timeit(lambda: ops.upconv3x3(conv_a, bias, srcs, 2, out=out))
        print("%-28s %.4f ms (tap GEMMs %.4f, interpolation %.4f)" % ("one launch" if flag else "x pass + y pass", t, t_gemm, t - t_gemm))


if __name__ == "__main__":
    main()
