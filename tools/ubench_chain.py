"""Dev: the third BEV stage's ten 128 -> 128 @64x64 convolutions (B = 4) -- and the range-view stage's 64 -> 64 @16x512 ones -- as
ten conv_wino_cl launches vs ONE smos_conv_wino_chain_cl launch (EXPERIMENTAL, csrc/conv_wino_chain.hip).
    python tools/ubench_chain.py        (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from streammos_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


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


def case(c, h, w, b, n_blocks, mb=2):
    gen = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((b, h, w, c), generator=gen).to(DEV).permute(0, 3, 1, 2)
    wts = [((torch.randn((c, c, 3, 3), generator=gen) * (1.0 / (c * 9)) ** 0.5).to(DEV), (torch.randn(c, generator=gen) * 0.1).to(DEV))
           for _ in range(2 * n_blocks)]
    preps = [ops.conv_wino_prepare(wt, 2) for wt, _ in wts]
    preps_c = preps if mb == 2 else [ops.conv_wino_prepare(wt, mb) for wt, _ in wts]
    bufs = [ops.empty_cl(b, c, h, w, DEV) for _ in range(2 * n_blocks)]

    def separate():
        cur = x
        for k in range(n_blocks):
            y = ops.conv_wino_cl(cur, preps[2 * k], wts[2 * k][1], ops.ACT_RELU, c, mb=2, out=bufs[2 * k])
            cur = ops.conv_wino_cl(y, preps[2 * k + 1], wts[2 * k + 1][1], ops.ACT_RELU, c, mb=2, residual=cur, out=bufs[2 * k + 1])
        return cur

    ws = ops.WinoChainWorkspace(2 * n_blocks, b, h, w, DEV)
    outs = [ops.empty_cl(b, c, h, w, DEV) for _ in range(2 * n_blocks)]
    layers = []
    for k in range(n_blocks):
        layers.append((preps_c[2 * k], wts[2 * k][1], -1, outs[2 * k], ops.ACT_RELU))
        layers.append((preps_c[2 * k + 1], wts[2 * k + 1][1], 2 * k, outs[2 * k + 1], ops.ACT_RELU))

    def chained():
        return ops.conv_wino_chain_cl(x, layers, ws, mb=mb)

    want = separate().clone()
    got = chained()
    torch.cuda.synchronize()
    same = torch.equal(got, want)
    t_sep, t_chain = timeit(separate), timeit(chained)
    print("mb=%d " % mb + "%3d ch @%dx%d x %d, %2d layers: separate %.4f ms (%.1f us / layer), chained %.4f ms (%.1f us / layer)  equal=%s gave_up=%s"
          % (c, h, w, b, 2 * n_blocks, t_sep, 1e3 * t_sep / (2 * n_blocks), t_chain, 1e3 * t_chain / (2 * n_blocks), same, ws.gave_up()))


if __name__ == "__main__":
    case(128, 64, 64, 4, 5)
    case(128, 64, 64, 4, 5, mb=1)
    case(64, 16, 512, 4, 3)
    case(64, 16, 512, 4, 3, mb=1)
    case(32, 32, 1024, 4, 2)
    case(64, 128, 128, 4, 3)
