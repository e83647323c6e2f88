"""Dev: full-size pointnet_scatter (model weights, bench frame) against an fp64 reference, and the e2e logit deviation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from streammos_amd import ops
from streammos_amd import synth
from streammos_amd.refapi.config import StreamMOS as cfg
from streammos_amd.refapi.models import StreamMOS
DEV = "cuda:0"
model = StreamMOS.AttNet(cfg.get_config()[2])
model.load_state_dict(synth.seeded_state_dict(model.state_dict()), strict=True)
model = model.to(DEV).eval()
sample = bench.make_frames(1, seq_seed=7)[0][0]
xyzi = torch.from_numpy(sample["pcds_xyzi"]).unsqueeze(0).to(DEV)[0]; coord = torch.from_numpy(sample["pcds_coord"]).to(DEV)
with torch.no_grad():
    eng = model._engine_for(xyzi)
(w1, b1), (w2, b2) = eng.pp1, eng.pp2
w1 = w1.reshape(64, 7).contiguous(); w2 = w2.reshape(64, 64).contiguous()
print("weights", w1.abs().max().item(), w2.abs().max().item(), "inputs", xyzi.abs().max().item())
b, t, _, n = xyzi.shape[:4]
bev = torch.zeros((b, 512, 512, t * 64), device=DEV)
rows = torch.zeros((b, n, 64), device=DEV)
ops.pointnet_scatter(xyzi, coord, w1, b1, w2, b2, bev, pts_out=rows)
x = xyzi.view(b * t, 7, n, 1).double()
pts = F.relu(F.conv2d(F.relu(F.conv2d(x, w1.double().view(64, 7, 1, 1), b1.double())), w2.double().view(64, 64, 1, 1), b2.double()))
pts0 = pts.view(b, t, 64, n)[:, 0].permute(0, 2, 1)
d = (rows.double() - pts0).abs()
print("rows: max abs diff %.4g  max ref %.4g  rel %.3g" % (d.max().item(), pts0.abs().max().item(), d.max().item() / pts0.abs().max().item()))
rel = d / pts0.abs().clamp_min(1e-3)
print("rows: max elementwise rel %.3g  (99.99%% quantile %.3g)" % (rel.max().item(), rel.flatten()[::97].quantile(0.9999).item()))
want = torch.zeros((b * t, 64, 512, 512), device=DEV)
ops.voxel_maxpool_fwd(pts.float(), coord.view(b * t, n, 3)[:, :, :2].contiguous(), want, (512, 512), (1.0, 1.0))
got = bev.permute(0, 3, 1, 2).reshape(b, t, 64, 512, 512).reshape(b * t, 64, 512, 512)
dd = (got - want).abs()
print("bev: max abs diff %.4g  max ref %.4g   cells differing >1e-4 rel: %d" % (dd.max().item(), want.abs().max().item(),
      int((dd > 1e-4 * want.abs().clamp_min(1.0)).sum())))
bad = (dd > 1e-4 * want.abs().clamp_min(1.0)).nonzero()
print("differing cells (sample, channel, y, x), got, want:")
for r in bad[:12].tolist():
    print(r, got[tuple(r)].item(), want[tuple(r)].item())
print("got<want:", int(((got < want) & (dd > 1e-3)).sum()), " got>want:", int(((got > want) & (dd > 1e-3)).sum()))
samples = torch.unique(bad[:, 0]).tolist(); print("samples", samples, "channels", torch.unique(bad[:, 1]).tolist()[:20])
# which points map to the first differing cell?
r = bad[0].tolist()
cc = coord.view(b * t, n, 3)[r[0]]
inside = ((cc[:, 0] > -1) & (cc[:, 0] < 512) & (cc[:, 1] > -1) & (cc[:, 1] < 512))
cellid = cc[:, 0].int() * 512 + cc[:, 1].int()
idx = ((cellid == r[2] * 512 + r[3]) & inside).nonzero().flatten()
print("points in that cell:", idx.tolist()[:20], "values", pts[r[0], r[1], idx, 0].tolist()[:20])
print("coords", cc[idx][:6].tolist())
lo = 3870 - 3870 % 32
print("tile cells:", cellid[lo:lo + 32].tolist())
print("tile inside:", inside[lo:lo + 32].int().tolist())
print("tile coords y:", [round(v, 3) for v in cc[lo:lo + 32, 0].tolist()])
print("tile coords x:", [round(v, 3) for v in cc[lo:lo + 32, 1].tolist()])
