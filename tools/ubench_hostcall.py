"""Host cost of one launch through the C-ABI: a no-op ctypes call, the smallest Winograd conv launch (ctypes marshalling +
hipLaunchKernel) and torch's allocator call, each over 2000 back-to-back calls with the GPU kept far from full.
    python tools/ubench_hostcall.py        (on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from streammos_amd import _lib, ops  # noqa: E402


def per_call(fn, n=2000):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / n


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    x = ops.empty_cl(1, 32, 8, 32, dev)
    x.normal_()
    w = torch.randn(32, 32, 3, 3, device=dev)
    wp = ops.conv_wino_prepare(w, 2)
    bias = torch.zeros(32, device=dev)
    out = ops.empty_cl(1, 32, 8, 32, dev)
    args = (x.data_ptr(), 32, wp.data_ptr(), bias.data_ptr(), None, 0, out.data_ptr(), 32, 1, 8, 32, 32, 32, 2, 0, None)
    st = ops._raw_stream(0)
    print("ctypes no-op (smos_abi_version)      %.2f us" % per_call(lib.smos_abi_version))
    print("smos_conv_wino_cl, raw ctypes call    %.2f us" % per_call(lambda: lib.smos_conv_wino_cl(*args, st)))
    print("ops.conv_wino_cl(out=...)             %.2f us" % per_call(lambda: ops.conv_wino_cl(x, wp, bias, 0, 32, mb=2, out=out)))
    print("ops.conv_wino_cl (allocating)         %.2f us" % per_call(lambda: ops.conv_wino_cl(x, wp, bias, 0, 32, mb=2)))
    print("ops.empty_cl                          %.2f us" % per_call(lambda: ops.empty_cl(1, 32, 8, 32, dev)))
    print("torch.zeros(4096)                     %.2f us" % per_call(lambda: torch.zeros(4096, device=dev)))
    print("tensor.zero_()                        %.2f us" % per_call(out.zero_))


if __name__ == "__main__":
    main()
