"""fp32 Winograd F(2x2, 3x3): does it keep the parity bar?  ("settle it with data first".)

CPU-only study, no GPU needed:
  1. per layer: every stride-1 3x3 shape of the network, random data as in tests/test_gpu_ops.py::_CONV_CASES, error of
     (a) the direct fp32 conv and (b) the fp32 Winograd form against a float64 direct conv;
  2. end to end: the CPU oracle (oracle/net_torch.py) with every stride-1 3x3 conv replaced by the Winograd form, against
     the reference's golden outputs (tests/golden/e2e.npz) at the 2e-5-of-range / 99.99 % bar of test_gpu_e2e.py.

The Winograd form here is what csrc/conv_wino.hip computes: weights transformed in float64 on the host (U = G g G^T,
rounded once to fp32), input transform V = B^T d B and output transform Y = A^T M A in fp32 (additions only), the 16
channel sums in fp32.  Summation ORDER differs from the kernel's; magnitudes are what this measures.

    python tools/winograd_numerics.py [--e2e]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3x3(x, w, bias=None):
    """conv2d(x, w, bias, stride 1, padding 1) as F(2x2, 3x3) in fp32.  x [B,C,H,W], w [Co,C,3,3]."""
    b, c, h, wd = x.shape
    co = w.shape[0]
    u = (G @ w.double() @ G.t()).float()                                   # [Co, C, 4, 4], one rounding
    th, tw = (h + 1) // 2, (wd + 1) // 2
    xp = F.pad(x, (1, 1 + 2 * tw - wd, 1, 1 + 2 * th - h))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                  # [B, C, th, tw, 4, 4]
    v = torch.einsum("ij,bcyxjk,lk->bcyxil", BT, d, BT)                     # B^T d B  (+-1: additions only)
    m = torch.einsum("ocil,bcyxil->boyxil", u, v)                           # 16 channel sums
    y = torch.einsum("ij,boyxjk,lk->boyxil", AT, m, AT)                     # [B, Co, th, tw, 2, 2]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(b, co, 2 * th, 2 * tw)[:, :, :h, :wd]
    return y if bias is None else y + bias[None, :, None, None]


def per_layer():
    torch.manual_seed(0)
    print("%-34s %12s %12s   (max |err| / max |ref|, vs float64 direct)" % ("layer", "direct fp32", "winograd fp32"))
    worst = 0.0
    for name, b, c, co, h, w in (("header_bev 32->32 @256^2", 1, 32, 32, 256, 256), ("header_bev 64->32 @256^2", 1, 64, 32, 256, 256),
                                 ("header_rv 32->32 @32x1024", 1, 32, 32, 32, 1024), ("res1_bev 64->64 @128^2", 1, 64, 64, 128, 128),
                                 ("res1_bev 128->64 @128^2", 1, 128, 64, 128, 128), ("res1_rv 64->64 @16x512", 1, 64, 64, 16, 512),
                                 ("res2 128->128 @64^2", 2, 128, 128, 64, 64), ("conv_1a 64->128 @256^2", 1, 64, 128, 256, 256),
                                 ("conv_2 128->64 @256^2", 1, 128, 64, 256, 256), ("ragged 32->32 @37x45", 2, 32, 32, 37, 45)):
        x = torch.randn(b, c, h, w)
        wt = torch.randn(co, c, 3, 3) / (3 * c ** 0.5)
        ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
        e_dir = ((F.conv2d(x, wt, None, 1, 1).double() - ref).abs().max() / ref.abs().max()).item()
        e_win = ((winograd_conv3x3(x, wt).double() - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, e_win)
        print("%-34s %12.2e %12.2e" % (name, e_dir, e_win))
    print("worst winograd layer error %.2e  (test_conv_cl_against_float64 bar: 2e-5... see tests/test_gpu_ops.py)" % worst)


def e2e():
    from oracle import net_torch
    from streammos_amd import synth
    from streammos_amd.refapi.config import StreamMOS as cfg
    from streammos_amd.refapi.models import StreamMOS
    from tests import cases

    class WinoNet(net_torch.OracleNet):
        n_wino = 0

        def conv(self, x, p, stride=1, padding=0):
            w = self.w[p + ".weight"]
            if tuple(w.shape[2:]) == (3, 3) and stride == 1 and padding in (1, (1, 1)):
                WinoNet.n_wino += 1
                return winograd_conv3x3(x, w, self.w.get(p + ".bias"))
            return super().conv(x, p, stride, padding)

    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e.npz"))
    model = StreamMOS.AttNet(cfg.get_config()[2])
    sd = synth.seeded_state_dict(model.state_dict())
    for label, net in (("direct  ", net_torch.OracleNet(sd)), ("winograd", WinoNet(sd))):
        memory = None
        for i, batch in enumerate(cases.e2e_frames()):
            pred, a0, a1, a2, memory = net.stage_forward(*(torch.from_numpy(batch[k]) for k in
                                                           ("pcds_xyzi", "pcds_coord", "pcds_sphere_coord")), memory)
            ref = g["e2e_f%d_pred" % i]
            err = np.abs(pred.numpy() - ref).max() / np.abs(ref).max()
            agree = (pred.numpy().argmax(1) == ref.argmax(1)).mean()
            print("%s frame %d vs the reference's golden logits: %.2e of range, labels %.6f" % (label, i, err, agree))
    print("stride-1 3x3 convs routed through the Winograd form per frame: %d" % (WinoNet.n_wino // 3))


if __name__ == "__main__":
    torch.set_num_threads(8)
    per_layer()
    if "--e2e" in sys.argv:
        e2e()
