"""Dev: where does a wave of csrc/conv_igemm.hip spend its cycles?  Builds a DIAGNOSTIC copy of the library with in-kernel
s_memtime stamps (-DSMOS_CONV_STAMPS; the shipped library has none), runs one layer and prints the share of each segment of
the stage body.  Shares, not lengths: the stamps' own waits forbid overlaps the real kernel has."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from streammos_amd import _lib, build, ops

diag = "/tmp/libsmos_stamps.so"
objs = []
for src in build.sources():
    obj = "/tmp/stamps_" + os.path.basename(src)[:-4] + ".o"
    subprocess.check_call([build.HIPCC] + build.FLAGS + ["-DSMOS_CONV_STAMPS", "-DSMOS_CONV_SCHED=0", "-c", src, "-o", obj])
    objs.append(obj)
subprocess.check_call([build.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", diag] + objs)
_lib.LIB_PATH = diag
dev = "cuda:0"
names = ["loop control", "G0", "M0 park+loadA+readA", "G1", "M1 loadB", "G2", "M2 adv+barrier+readA", "G3", "tail (rare branches)"]
order = [8, 0, 1, 2, 3, 4, 5, 6, 7]
for (cin, cout, k, hw, mt) in ((32, 32, (3, 3), (256, 256), 1), (128, 64, (3, 3), (256, 256), 1), (128, 64, (3, 3), (256, 256), 2)):
    x = torch.randn(4, hw[0], hw[1], cin, device=dev).permute(0, 3, 1, 2)
    wt = torch.randn(cout, cin, *k, device=dev) * 0.05
    wp = ops.conv_prepare(wt, mt)
    buf = torch.zeros(4 * 11 * 2048, dtype=torch.int64, device=dev)
    os.environ["SMOS_CONV_STAMP_PTR"] = str(buf.data_ptr())
    for _ in range(int(os.environ.get("STAMP_LAUNCHES", "3"))):
        ops.conv_cl(x, wp, None, 1, cout, k, mt=mt)
    torch.cuda.synchronize()
    s = buf.view(-1, 11).double()
    s = s[s.sum(1) > 0]
    clock = (s[:, 9] / s[:, 10]).median().item() * 100.0
    s = s[:, :9]
    tot = s.sum(1).mean().item()
    print("   in-kernel shader clock %.0f MHz (median over waves, last of the launches)" % clock)
    print("cin %d cout %d k%s %dx%d mt%d: %d waves, %.0f cycles per wave (last launch)" % (cin, cout, k, hw[0], hw[1], mt, s.shape[0], tot))
    for i in order:
        print("   %-24s %5.1f %%" % (names[order.index(i)], 100 * s[:, i].mean().item() / tot))
