"""Dev: where the cycles of tfusion_layer go.  Builds libsmos_hip with -DSMOS_TF_STAMPS into a scratch library, runs the
kernel at the model's shape and prints, per phase, the median over blocks of the s_memtime differences of wave 0
(s_memtime counts at 100 MHz: 1 tick = 10 ns).  usage (GPU box): python tools/tf_stamps.py"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_dir = os.path.join(ROOT, "gpurun_out", "tf_stamps_lib")
os.makedirs(lib_dir, exist_ok=True)
pkg = os.path.join(ROOT, "streammos_amd")
obj = os.path.join(lib_dir, "tfusion.o")
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on"]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DSMOS_TF_STAMPS", "-c", os.path.join(pkg, "csrc", "tfusion.hip"), "-o", obj])
others = [o for o in glob.glob(os.path.join(pkg, "lib", "*.o")) if not o.endswith("tfusion.o")]
so = os.path.join(lib_dir, "libsmos_hip_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, obj] + others)
os.environ["SMOS_HIP_LIB"] = so
import torch
from streammos_amd import ops
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(1)
def lin(o, i): return ((torch.randn((o, i), generator=g) / i ** 0.5).to(dev), torch.randn(o, generator=g).to(dev))
def norm(): return (torch.ones(128, device=dev), torch.zeros(128, device=dev), 1e-5)
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sampled, query = torch.randn(tokens, 128, device=dev), torch.randn(tokens, 128, device=dev)
prep = ops.TfusionLayer(lin(128, 128), norm(), lin(512, 128), lin(128, 512), norm(), next_qproj=lin(48, 128))
nblk = (tokens + 63) // 64
stamps = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
os.environ["SMOS_TF_STAMP_PTR"] = str(stamps.data_ptr())
out = torch.empty(tokens, 128, device=dev)
for _ in range(20):
    ops.tfusion_layer(sampled, query, prep, out=out)
torch.cuda.synchronize()
s = stamps.view(nblk, 8).cpu().double()
names = ["prologue (requests .. first fragments)", "output_proj (8 slots)", "+query, LayerNorm 1", "FFN (64 slots)", "LayerNorm 2 + stores",
         "next projection (4 slots)"]
t0 = s[:, 0].min()
print("blocks %d; kernel span (first start .. last end) %.2f us" % (nblk, (s[:, 6].max() - t0) * 0.01))
print("block start spread: %.2f us; block duration median %.2f us (min %.2f, max %.2f)" %
      ((s[:, 0].max() - t0) * 0.01, (s[:, 6] - s[:, 0]).median() * 0.01, (s[:, 6] - s[:, 0]).min() * 0.01, (s[:, 6] - s[:, 0]).max() * 0.01))
for k, nm in enumerate(names):
    d = (s[:, k + 1] - s[:, k]) * 0.01
    print("  %-42s median %6.2f us   max %6.2f" % (nm, d.median(), d.max()))
print("FFN per slot: %.1f ns = %.0f cycles at 2.4 GHz (1024 = the 32 MFMAs alone)" % ((s[:, 4] - s[:, 3]).median() * 10 / 64, (s[:, 4] - s[:, 3]).median() * 10 / 64 * 2.4))
