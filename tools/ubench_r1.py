"""Dev micro-benchmarks (GPU): scatter / gather lane mappings and conv layout options."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from streammos_amd import ops
import bench

dev = "cuda:0"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

frames = bench.make_frames(1, 0)
s = frames[0][0]
coord = torch.from_numpy(s["pcds_coord"]).to(dev)      # (4,3,N,3,1)
sph = torch.from_numpy(s["pcds_sphere_coord"]).to(dev)
B, T, N = 4, 3, 160000
print("valid fraction", float((coord[:, :, :, 0, 0] > -1).float().mean()))

def scatter_case(name, bs, c, ind, hw, scale):
    feat = torch.relu(torch.randn(bs, c, N, device=dev))
    out = torch.zeros((bs, c) + hw, device=dev)
    t_fill = timeit(lambda: out.zero_())
    t = timeit(lambda: ops.voxel_maxpool_fwd(feat, ind, out, hw, scale))
    featz = torch.zeros_like(feat)
    tz = timeit(lambda: ops.voxel_maxpool_fwd(featz, ind, out, hw, scale))
    # rows: point-major feat, channels-last out
    feat_pm = feat.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    out_cl = torch.zeros((bs,) + hw + (c,), device=dev).permute(0, 3, 1, 2)
    tr = timeit(lambda: ops.voxel_maxpool_fwd(feat_pm, ind, out_cl, hw, scale))
    out.zero_(); ops.voxel_maxpool_fwd(feat, ind, out, hw, scale)
    out_cl.zero_(); ops.voxel_maxpool_fwd(feat_pm, ind, out_cl, hw, scale)
    same = torch.equal(out, out_cl)
    # mixed: channel-major feat -> channels-last out and point-major feat -> NCHW out
    out_cl.zero_()
    tm = timeit(lambda: ops.voxel_maxpool_fwd(feat, ind, out_cl, hw, scale))
    occ = float((out.abs().sum(1) > 0).float().mean())
    print("%-28s fill %.3f  points/NCHW %.3f (zero-feat %.3f)  rows/NHWC %.3f  chmajor->NHWC %.3f  same=%s occupancy=%.3f" % (name, t_fill, t, tz, tr, tm, same, occ))

ind_in = coord.view(B * T, N, 3, 1)[:, :, :2, 0].contiguous()
cur_xy = coord[:, 0, :, :2, 0].contiguous()
cur_sp = sph[:, 0, :, :, 0].contiguous()
scatter_case("P2B input 12x64->512^2", 12, 64, ind_in, (512, 512), (1.0, 1.0))
scatter_case("P2R 4x32->32x1024", 4, 32, cur_sp, (32, 1024), (0.5, 0.5))
scatter_case("P2B 4x32->256^2", 4, 32, cur_xy, (256, 256), (0.5, 0.5))
scatter_case("P2R 4x64->16x512", 4, 64, cur_sp, (16, 512), (0.25, 0.25))
scatter_case("P2B 4x64->128^2", 4, 64, cur_xy, (128, 128), (0.25, 0.25))

def gather_case(name, c, hw, ind, scale):
    grid = torch.randn((4, c) + hw, device=dev)
    t = timeit(lambda: ops.bilinear_gather(grid, ind, scale))
    gcl = grid.contiguous(memory_format=torch.channels_last)
    out = torch.empty((4, N, c), device=dev).permute(0, 2, 1)
    tr = timeit(lambda: ops.bilinear_gather(gcl, ind, scale, out=out))
    print("%-28s points/NCHW %.3f  rows/NHWC %.3f" % (name, t, tr))
gather_case("B2P 32@256^2", 32, (256, 256), cur_xy, (0.5, 0.5))
gather_case("R2P 32@32x1024", 32, (32, 1024), cur_sp, (0.5, 0.5))
gather_case("B2P 64@128^2", 64, (128, 128), cur_xy, (0.25, 0.25))
gather_case("R2P 64@16x512", 64, (16, 512), cur_sp, (0.25, 0.25))
gather_case("B2P 64@256^2", 64, (256, 256), cur_xy, (0.5, 0.5))

# conv layout options
def conv_case(name, cin, cout, hw, k, stride=1, pad=1, b=4):
    x = torch.randn((b, cin) + hw, device=dev); w = torch.randn(cout, cin, *k, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    t = timeit(lambda: F.conv2d(x, w, None, stride, pad), n=10)
    tb = timeit(lambda: F.conv2d(x, w, bias, stride, pad), n=10)
    xcl = x.contiguous(memory_format=torch.channels_last); wcl = w.contiguous(memory_format=torch.channels_last)
    tcl = timeit(lambda: F.conv2d(xcl, wcl, None, stride, pad), n=10)
    fl = 2 * b * cout * cin * k[0] * k[1] * (hw[0] // stride) * (hw[1] // stride)
    print("%-30s nchw %.3f ms (%.1f TF) +bias %.3f  channels_last %.3f" % (name, t, fl / t / 1e9, tb, tcl))
conv_case("conv_1 320->128 3x3 @256", 320, 128, (256, 256), (3, 3))
conv_case("conv_2 128->64 3x3 @256", 128, 64, (256, 256), (3, 3))
conv_case("hdr 192->32 3x3 s2 @512", 192, 32, (512, 512), (3, 3), 2)
conv_case("hdr 192->32 1x1 @512", 192, 32, (512, 512), (1, 1), 1, 0)
conv_case("bb 32->32 3x3 @256", 32, 32, (256, 256), (3, 3))
conv_case("unb 32->32 7x3 @256", 32, 32, (256, 256), (7, 3), 1, (3, 1))
conv_case("unb 64->64 5x3 @128", 64, 64, (128, 128), (5, 3), 1, (2, 1))
conv_case("bb 64->64 3x3 @128", 64, 64, (128, 128), (3, 3))
conv_case("bb 128->128 3x3 @64", 128, 128, (64, 64), (3, 3))
conv_case("rv 32->32 3x3 @32x1024", 32, 32, (32, 1024), (3, 3))
conv_case("pt 7->64 1x1 @Nx1 b12", 7, 64, (160000, 1), (1, 1), 1, 0, 12)
conv_case("pt 64->64 1x1 @Nx1 b12", 64, 64, (160000, 1), (1, 1), 1, 0, 12)
conv_case("pt 192->96 1x1 @Nx1", 192, 96, (160000, 1), (1, 1), 1, 0)
# point-major GEMM alternative
a = torch.randn(12 * 160000, 64, device=dev); wm = torch.randn(64, 64, device=dev)
print("gemm (1.92M x 64)@(64x64): %.3f ms" % timeit(lambda: a @ wm))
a2 = torch.randn(4 * 160000, 192, device=dev); wm2 = torch.randn(192, 96, device=dev)
print("gemm (640k x 192)@(192x96): %.3f ms" % timeit(lambda: a2 @ wm2))
x = torch.randn(4, 32, 256, 256, device=dev)
print("relu 4x32x256^2: %.4f ms; add %.4f" % (timeit(lambda: torch.relu(x)), timeit(lambda: x + x)))
